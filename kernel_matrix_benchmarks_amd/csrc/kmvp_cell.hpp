// Gaussian pair loop with the kernel VALUE on the bf16 matrix cores: range reduction by grid cells.
//
// fast_kernel (kmvp_fast.hpp) takes the squared distance from the matrix pipe and still pays one
// v_exp_f32 per pair -- 70 % of its issue cycles, and the cap of every exp-per-pair kernel on this
// chip (~1.9e13 pairs/s).  Here the exponential itself is range-reduced the way a math library does
// it, with the cells of a regular grid as the reduction table.  Both clouds are binned into cells of
// side h (per axis: the bounding box divided into equal cells within the accuracy bound); a target i in a cell with centre c_T and a source j in a cell with centre c_S are
//     x_i = c_T + d_i ,   y_j = c_S + e_j ,   D = c_T - c_S ,   |d|, |e| <= h sqrt(dim) / 2
//     |x_i - y_j|^2 = |D + d_i|^2  +  (|e_j|^2 - 2 D.e_j)  -  2 d_i.e_j
//     exp(-|x_i - y_j|^2) = U_i(S) * W_j(T) * exp(t_ij) ,   t_ij = 2 d_i.e_j ,  |t| <= dim h^2 / 2
//         U_i(S) = exp(-|x_i - c_S|^2)          one exp per (target, source CELL)
//         W_j(T) = exp(e_j.(2 D - e_j))          one exp per (source, target CELL)
//         exp(t) = 1 + t + t^2/2 + O(t^3/6)      |t| <= 0.016  ->  truncation <= 6.8e-7 (corner-to-corner
//                                                pair of cells; ~1e-8 for a typical pair)
// and 1 + t + t^2/2 is BILINEAR in monomials of d_i and of e_j -- sixteen of them for dim = 3:
//     k = 0          1                         x  1
//     k = 1 + 3a..   d_h, d_h, d_m  (coord a)  x  (2e)_h, (2e)_m, (2e)_h     two-way bf16 split: 16 bits,
//                                                                          dropped terms <= 2^-17 |t| ~ 1.2e-7
//     k = 10..15     d_a d_b                   x  2 e_a^2 | 4 e_a e_b       (<= 1.3e-4: one bf16 term)
// i.e. ONE v_mfma_f32_32x32x16_bf16 per 32 sources x 32 targets hands the VALU the polynomial,
// to ~1e-7 relative for a typical pair and <= 1.3e-6 for the worst placed one (the class of the
// difference form, whose squared distance carries eps32 * s), random in sign.  What is left per pair is
// one FMA (p * W_j b_j into the tile sum); U multiplies the tile sum once.  Every pair is still
// evaluated; nothing is truncated in space.  Accuracy against the fp64 oracle on the headline cloud:
// see tests/test_gpu_parity.py::test_cell_kernel_*.
//
// Cells come from one radix sort per cloud (cell index = key); inside a cell points form tiles of 32,
// the last one padded (pad sources carry b = 0, pad targets are never read back).  Tiles of one
// cell share their centre, so U is recomputed only when the source tile's cell changes; the target
// tiles of a cell are padded to a multiple of TT, so the TT tiles of a wave always share one cell and
// W is computed (and read back from LDS) once per wave and source tile.  Differences are formed from the caller's
// coordinates first (d = x - c_T in fp32 is exact to an ulp of d; D is the difference of the stored
// fp32 centres), as on every other path.
//
// Mapping as fast_kernel: target on the lane (column), 16 sources in the lane's registers; a wave
// owns TT target tiles; a workgroup of 4 waves shares LDS stages of 4 source tiles copied by
// LDS-DMA, double buffered, one barrier per stage.  W_j b_j is computed with the SOURCE on the lane
// (both lane halves busy: two source tiles at once) and handed to the register layout through a
// wave-private LDS array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_fast.hpp"  // bf16x8, f32x16, f32x4, kexp2, block_to_work

namespace kmvp {

constexpr int CELL_TILE = 32;
constexpr int CELL_STAGE_TILES = 4;
constexpr int CELL_A_BYTES = CELL_TILE * 32;                       // 32 rows of 16 bf16
constexpr int CELL_TILE_BYTES = CELL_A_BYTES + CELL_TILE * 16;     // + (e_x, e_y, e_z, b) per source
constexpr int CELL_HDR_OFF = CELL_STAGE_TILES * CELL_TILE_BYTES;   // 4 x (c_x, c_y, c_z, cell key)
constexpr int CELL_STAGE_BYTES = 8192;
static_assert(CELL_HDR_OFF + CELL_STAGE_TILES * 16 <= CELL_STAGE_BYTES, "stage image too small");
constexpr float CELL_T_MAX = 0.016f;  // bound on |2 d.e|: truncation t^3/6 <= 6.8e-7, typically 10x less
constexpr int CELL_MAX_GRID = 1024;   // cells per axis (10 bits of the key each)

struct CellGrid {
  float lo[3];
  float h[3], inv_h[3];  // per axis: the box extent divided into equal cells of side <= the accuracy bound
  int g[3];
};

__host__ __device__ inline float cell_centre(unsigned key, int a, const CellGrid& grid) {
  return grid.lo[a] + ((float)((key >> (10 * a)) & 1023u) + 0.5f) * grid.h[a];
}

struct CellArgs {
  const float* xd;           // targets [n_slots][4]: d_x, d_y, d_z, 0 (cell-sorted, tiles padded)
  const float* tmeta;        // target tiles [n_slots / 32][4]: c_x, c_y, c_z, cell key (bits; bit 30: empty tile)
  const unsigned char* img;  // source stages [m_stages][CELL_STAGE_BYTES]
  double* part;              // partial sums of THIS launch's tiles [segments][NE][n_slots]
  int64_t n_slots;           // slots (tiles x 32) of this launch
  int64_t tile_base;         // first target tile of this launch (0: the list of whole groups; behind it: the leftover tiles)
  int64_t m_stages;
  int64_t seg_stages;
  int segments;
  int tile_blocks;
  int chunk_stages;
};

// SIG_PRODUCT (density = product with b = 1, set by the packer) or SIG_NORM
template <int SIG, int TT>
__global__ void __launch_bounds__(BLOCK_THREADS) cell_kernel(const CellArgs a) {
  constexpr int NE = (SIG == SIG_NORM) ? 2 : 1;
  constexpr int SB = CELL_STAGE_BYTES;
  constexpr int PIECES = SB / (16 * BLOCK_THREADS);
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SB];
  __shared__ __attribute__((aligned(16))) float wsc[WAVES_PER_BLOCK][2][NE][CELL_TILE];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t tile0 = a.tile_base + ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;

  // TT = 8 keeps the targets' offsets in LDS and re-reads them when U is recomputed (once per source
  // cell) instead of holding them: 24 registers decide between three and four waves per SIMD there
  constexpr bool HOLD_D = TT < 8;
  __shared__ __attribute__((aligned(16))) float dsh[HOLD_D ? 1 : WAVES_PER_BLOCK][HOLD_D ? 1 : TT][CELL_TILE][4];
  float dl[HOLD_D ? TT : 1][3], cT[3], U[TT];
  bf16x8 xb[TT];
  bool live[TT];  // false: one of the empty tiles that pad a cell to a multiple of TT tiles (wave-uniform)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.xd + ((tile0 + tt) * CELL_TILE + r) * 4);
    const f32x4 m = *reinterpret_cast<const f32x4*>(a.tmeta + (tile0 + tt) * 4);
    float f[16];
    f[0] = 1.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if constexpr (HOLD_D) dl[tt][c] = v[c];
      else if (h == 0) dsh[wave][tt][r][c] = v[c];
      if (tt == 0) cT[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[c])));  // one cell per wave
      const float dh = (float)(__bf16)v[c];
      const float dm = (float)(__bf16)(v[c] - dh);
      f[1 + 3 * c] = dh;
      f[2 + 3 * c] = dh;
      f[3 + 3 * c] = dm;
    }
    f[10] = v[0] * v[0];
    f[11] = v[1] * v[1];
    f[12] = v[2] * v[2];
    f[13] = v[0] * v[1];
    f[14] = v[0] * v[2];
    f[15] = v[1] * v[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float lo_k = f[j], hi_k = f[8 + j];  // lane half 0: k = j, lane half 1: k = 8 + j
      xb[tt][j] = (__bf16)(h ? hi_k : lo_k);
    }
    live[tt] = ((__builtin_amdgcn_readfirstlane(__float_as_int(m[3])) >> 30) & 1) == 0;
    U[tt] = 0.f;
  }

  float acc[TT][NE];
  double accd[TT][NE];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      acc[tt][e] = 0.f;
      accd[tt][e] = 0.0;
    }

  const int64_t s_begin = (int64_t)seg * a.seg_stages;
  int64_t s_end = s_begin + a.seg_stages;
  if (s_end > a.m_stages) s_end = a.m_stages;

  auto stage_in = [&](int64_t s, int buf) {
    const unsigned char* src = a.img + s * SB;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int piece = (p * WAVES_PER_BLOCK + wave) * 1024;  // wave-uniform LDS offset
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + piece + lane * 16),
          (__attribute__((address_space(3))) void*)(&lds[buf][piece]), 16, 0, 0);
    }
  };
  if (s_begin < s_end) stage_in(s_begin, 0);
  __syncthreads();

  int key_s = -2;  // cell of the source tile U was computed for (no real key is negative... -1 marks pad tiles)
  int in_chunk = 0;
  for (int64_t s = s_begin; s < s_end; ++s) {
    const int buf = (int)((s - s_begin) & 1);
    if (s + 1 < s_end) stage_in(s + 1, buf ^ 1);
    const f32x4* hdr = reinterpret_cast<const f32x4*>(&lds[buf][CELL_HDR_OFF]);
#pragma unroll 1
    for (int qp = 0; qp < CELL_STAGE_TILES / 2; ++qp) {
      {  // W_j b_j of source tiles 2qp (lane half 0) and 2qp+1 (lane half 1), source r on the lane
        const int qh = 2 * qp + h;
        const f32x4 hc = hdr[qh];
        const f32x4 mt = *reinterpret_cast<const f32x4*>(&lds[buf][qh * CELL_TILE_BYTES + CELL_A_BYTES + r * 16]);
        float arg = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) arg = fmaf(mt[c], 2.f * (cT[c] - hc[c]) - mt[c], arg);
        const float w = kexp2(arg * LOG2E);
        wsc[wave][h][0][r] = w * mt[3];
        if constexpr (SIG == SIG_NORM) wsc[wave][h][1][r] = w;
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const int q = 2 * qp + hq;
        const bf16x8 ya = *reinterpret_cast<const bf16x8*>(&lds[buf][q * CELL_TILE_BYTES + r * 32 + 16 * h]);
        const f32x4 cs = hdr[q];
        const int ks = __builtin_amdgcn_readfirstlane(__float_as_int(cs[3]));
        if (ks != key_s) {  // wave-uniform: the source cell changed -> U_i = exp(-|x_i - c_S|^2)
          key_s = ks;
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            f32x4 dv;
            if constexpr (!HOLD_D) dv = *reinterpret_cast<const f32x4*>(&dsh[wave][tt][r][0]);
            float s2 = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              float df;
              if constexpr (HOLD_D) df = dl[tt][c] + (cT[c] - cs[c]);
              else df = dv[c] + (cT[c] - cs[c]);
              s2 = fmaf(df, df, s2);
            }
            U[tt] = kexp2(s2 * -LOG2E);
          }
        }
        // W_j b_j of this source tile in the register layout (registers 4g..4g+3 = source rows
        // 8g+4h .. 8g+4h+3): broadcast reads, shared by the wave's TT target tiles
        const float* wp = &wsc[wave][hq][0][0];
        f32x4 wv[4], wd[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          wv[g] = *reinterpret_cast<const f32x4*>(wp + 8 * g + 4 * h);
          if constexpr (SIG == SIG_NORM) wd[g] = *reinterpret_cast<const f32x4*>(wp + CELL_TILE + 8 * g + 4 * h);
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          if (tt > 0 && !live[tt]) continue;  // wave-uniform
          f32x16 d;
#pragma unroll
          for (int qq = 0; qq < 16; ++qq) d[qq] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, xb[tt], d, 0, 0, 0);
          float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            p0 = fmaf(d[4 * g + 0], wv[g][0], p0);
            p1 = fmaf(d[4 * g + 1], wv[g][1], p1);
            p0 = fmaf(d[4 * g + 2], wv[g][2], p0);
            p1 = fmaf(d[4 * g + 3], wv[g][3], p1);
            if constexpr (SIG == SIG_NORM) {
              q0 = fmaf(d[4 * g + 0], wd[g][0], q0);
              q1 = fmaf(d[4 * g + 1], wd[g][1], q1);
              q0 = fmaf(d[4 * g + 2], wd[g][2], q0);
              q1 = fmaf(d[4 * g + 3], wd[g][3], q1);
            }
          }
          acc[tt][0] = fmaf(U[tt], p0 + p1, acc[tt][0]);
          if constexpr (SIG == SIG_NORM) acc[tt][1] = fmaf(U[tt], q0 + q1, acc[tt][1]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (++in_chunk == a.chunk_stages) {
      in_chunk = 0;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          accd[tt][e] += (double)acc[tt][e];
          acc[tt][e] = 0.f;
        }
    }
    __syncthreads();  // vmcnt(0) + barrier: stage s+1 has landed, stage s is free
  }

#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double v = accd[tt][e] + (double)acc[tt][e];
      v += __shfl_xor(v, 32);
      if (h == 0) a.part[((int64_t)seg * NE + e) * a.n_slots + (tile0 + tt - a.tile_base) * CELL_TILE + r] = v;
    }
}

hipError_t launch_cell_gaussian(int sig, int TT, const CellArgs& args, dim3 grid, hipStream_t stream,
                                const char** kernel_name);

}  // namespace kmvp
