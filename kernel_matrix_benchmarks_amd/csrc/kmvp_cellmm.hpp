// Gaussian pair loop with the kernel value AND the sum over the sources on the matrix cores.
//
// cell_kernel (kmvp_cell.hpp) takes the remainder polynomial of the range-reduced exponential from one
// bf16 MFMA per 32 x 32 pairs and still spends ONE VALU FMA PER PAIR on weighting it with W_j b_j and
// summing over the sources: 22.5 VALU instructions per 32 x 32 tile pair, the VALU saturated, the matrix
// pipe 37 % busy.  Here the weight moves INTO the source operand and the sum INTO the MFMA's accumulator:
//
//     exp(-|x_i - y_j|^2) b_j = U_i(S) * [ W_j(T) b_j * exp(t_ij) ] ,  t_ij = 2 d_i.e_j ,  |t| <= 0.016
//         (x_i = c_T + d_i, y_j = c_S + e_j, D = c_T - c_S, U_i(S) = exp(-|x_i - c_S|^2),
//          W_j(T) = exp(e_j.(2 D - e_j)); see kmvp_cell.hpp)
//     W_j b_j exp(t_ij) = W_j b_j  +  sum_k A_jk(T) B_ki ,   A_jk = psi_k(e_j) * W_j(T) b_j ,  B_ki = phi_k(d_i)
//
// with the fifteen monomial pairs of t + t^2/2 as the K = 16 contraction index of ONE
// v_mfma_f32_32x32x16_f16 per 32 sources x 32 targets, whose fp32 accumulator (row = source, column =
// target) is carried from source tile to source tile of a source cell:  acc += A_tile(T) x B.  The VALU
// touches a pair of tiles only to rebuild A for the wave's target cell -- ~27 instructions per source tile
// and lane, shared by the TT = 8 target tiles of the wave: 3.4 per tile pair instead of 22.5 -- and, once per
// (target tile, source CELL), to fold the accumulator: its 16 rows summed (packed adds, one v_permlane32_swap),
// the change since the last fold times U_i(S) into an fp64 sum.  The constant term
// W_j b_j of the bracket does not depend on the target inside a cell: it is summed per lane in fp32 as the
// rows of A are built and joins the accumulator's row sum at the fold (exact in fp32, where an f16 operand
// would have needed three more slots).  Every pair is still evaluated -- sixteen multiply-adds per pair on
// the matrix pipe -- and nothing is truncated in space.
//
// Precision.  f16 operands (11 significant bits), fp32 products and accumulation.  Slots (k = 8 h + j for
// lane half h, register pair j/2), with f = 2 e, e' = f W b and v = v_h + v_m the two-way f16 split:
//     h = 0:  d_xh e'_xh, d_yh e'_yh, d_xh e'_xm, d_yh e'_ym, d_xm e'_xh, d_ym e'_yh, (d_x^2/2)(f_x^2 Wb), (d_y^2/2)(f_y^2 Wb)
//     h = 1:  d_zh e'_zh, d_xd_z(f_x f_z Wb)_h, d_zh e'_zm, d_xd_z(f_x f_z Wb)_m, d_zm e'_zh, d_yd_z(f_y f_z Wb), (d_z^2/2)(f_z^2 Wb), d_xd_y(f_x f_y Wb)
// Linear terms carry 22 bits: dropped d_m e'_m <= 2^-22 |t| = 4e-9; quadratic terms (<= 1.3e-4) in one f16
// product each: <= 1.3e-7 in the worst corner, ~1e-8 typically; truncation t^3/6 as in cell_kernel (<= 6.8e-7
// corner to corner, ~1e-8 typical).  Operands are scaled by powers of two into the normal range of f16
// (targets x 2^6; the signal by sigma_b = 2^(15 - ceil(log2(0.104 Wmax max|b|))), kmvp_cellmm_pack.hpp),
// undone exactly at the store.
//
// Mapping: a wave owns TT target tiles of ONE cell (cells are padded to multiples of TT tiles); a workgroup
// of 4 waves shares LDS stages of 12 source tiles (32 x (f_x, f_y, f_z, -|e|^2 log2 e), 32 x b, cell
// header) prefetched through registers, double buffered, one barrier per stage.  TT = 8: 128 accumulator
// registers per lane, two waves per SIMD.
//
// Roof.  The kernel is bound by the matrix pipe, and the pipe by POWER: tools/mfma_stream.hip sustains
// 2.48 PFLOP/s (32.4 cycles per MFMA at 2.4 GHz) on zero operands and 1.50-1.61 PFLOP/s (50-53 nominal
// cycles) on random f16 operands -- the chip sits at its power limit and the clock falls to ~2.0 GHz.  With
// the operand build and the per-tile bookkeeping removed from this kernel's loop (timing-only experiment)
// the same launch takes 25.4 ms against 26.0-26.9 ms complete: the VALU work is hidden, what remains is
// the MFMA stream itself (1.30 PFLOP/s, 0.52 of the nominal 2.5, 0.83 of what random data sustains).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_cell.hpp"  // CellGrid, CELL_TILE, CELL_T_MAX, f32x4, f32x16, kexp2, block_to_work

namespace kmvp {

#ifndef CMM_BF16
#define CMM_BF16 0  // 1: bf16 operands (experiment: 7-9 % more MFMA throughput under the power limit, 8 instead of 11 bits)
#endif
#if CMM_BF16
typedef __bf16 cmm_half;
#else
typedef _Float16 cmm_half;
#endif
typedef cmm_half f16x8 __attribute__((ext_vector_type(8)));
typedef cmm_half f16x2 __attribute__((ext_vector_type(2)));

constexpr int CMM_STAGE_TILES = 12;
constexpr int CMM_STAGE_BYTES = 8192;
constexpr int CMM_E_OFF = 0;                                          // [tile][32][f_x, f_y, f_z, g]
constexpr int CMM_B_OFF = CMM_STAGE_TILES * CELL_TILE * 16;           // [tile][32] b sigma_b
constexpr int CMM_HDR_OFF = CMM_B_OFF + CMM_STAGE_TILES * CELL_TILE * 4;  // [tile] c_x, c_y, c_z, key
static_assert(CMM_HDR_OFF + CMM_STAGE_TILES * 16 <= CMM_STAGE_BYTES, "stage image too small");
constexpr float CMM_TARGET_SCALE = 64.f;   // targets' offsets x 2^6: their f16 mid parts stay normal numbers
constexpr int CMM_MAX_WLOG2 = 8;           // W_j <= 2^8 or the path is not taken (kmvp_product.hip)

struct CellmmArgs {
  const float* xd;           // targets [n_slots][4]: d_x, d_y, d_z, 0 (cell-sorted, tiles padded)  -- as cell_kernel
  const float* tmeta;        // target tiles [n_slots / 32][4]: c_x, c_y, c_z, cell key (bit 30: empty tile)
  const unsigned char* img;  // source stages [m_stages][CMM_STAGE_BYTES]
  const float* scale;        // [0] sigma_b, [1] 1 / (sigma_b * 2^6)
  double* part;              // partial sums of THIS launch's tiles [segments][n_slots]
  int64_t n_slots;           // slots (tiles x 32) of this launch
  int64_t tile_base;         // first target tile of this launch (0: the list of whole groups; behind it: the leftover tiles)
  int64_t m_stages;
  int64_t seg_stages;
  int segments;
  int tile_blocks;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

// two floats -> packed f16 pair in ONE instruction (v_cvt_pkrtz_f16_f32, round toward zero: the high parts'
// remainders are formed exactly afterwards; mid parts and quadratic terms lose at most 2^-10 of themselves)
__device__ __forceinline__ f16x2 cellmm_pk(float a, float b) {
#if CMM_BF16
  f16x2 r;
  r[0] = (cmm_half)a;
  r[1] = (cmm_half)b;
  return r;
#else
  return __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
#endif
}

// B operand of one target (column) for lane half h: see the slot table in the header comment
__device__ __forceinline__ f16x8 cellmm_target_operand(const f32x4 d, int h) {
  float s[3], sh[3], sm[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    s[c] = d[c] * CMM_TARGET_SCALE;
    sh[c] = (float)(cmm_half)s[c];
    sm[c] = s[c] - sh[c];
  }
  const float q = CMM_TARGET_SCALE;  // quadratic monomials carry 2^6 as well: A's e' f carries sigma_b only
  float lo[8], hi[8];
  lo[0] = sh[0]; lo[1] = sh[1]; lo[2] = sh[0]; lo[3] = sh[1]; lo[4] = sm[0]; lo[5] = sm[1];
  lo[6] = d[0] * d[0] * (0.5f * q);
  lo[7] = d[1] * d[1] * (0.5f * q);
  hi[0] = sh[2]; hi[1] = d[0] * d[2] * q; hi[2] = sh[2]; hi[3] = d[0] * d[2] * q; hi[4] = sm[2];
  hi[5] = d[1] * d[2] * q;
  hi[6] = d[2] * d[2] * (0.5f * q);
  hi[7] = d[0] * d[1] * q;
  f16x8 out;
#pragma unroll
  for (int j = 0; j < 8; ++j) out[j] = (cmm_half)(h ? hi[j] : lo[j]);
  return out;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Cross-lane sums of the fold without LDS round trips (a ds_bpermute costs ~100 cycles of latency, and every wave
// of the chip reaches a source cell's end at about the same time: nobody is left to hide it).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float cellmm_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
// sum over the 32 lanes of a wave half, as a wave-uniform value (both halves hold the same 32 numbers here)
__device__ __forceinline__ float cellmm_half_sum(float v) {
  v += cellmm_dpp<0x111, 0xf>(v);  // row_shr:1
  v += cellmm_dpp<0x112, 0xf>(v);  // row_shr:2
  v += cellmm_dpp<0x114, 0xf>(v);  // row_shr:4
  v += cellmm_dpp<0x118, 0xf>(v);  // row_shr:8: lane 15 of every row of 16 holds the row's sum
  v += cellmm_dpp<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3: lane 31 holds the sum of lanes 0..31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
}
// v[lane] + v[lane ^ 32] (v_permlane32_swap)
__device__ __forceinline__ float cellmm_cross_half_add(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

template <int TT>
__global__ void __launch_bounds__(BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(TT >= 8 ? 2 : 1)))
cellmm_kernel(const CellmmArgs a) {
  constexpr int SB = CMM_STAGE_BYTES;
  constexpr int PIECES = SB / (16 * BLOCK_THREADS);
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SB];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t tile0 = a.tile_base + ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;

  // TT = 8 keeps the targets' offsets in LDS (needed again only when U is recomputed, once per source cell)
  constexpr bool HOLD_D = TT < 8;
  __shared__ __attribute__((aligned(16))) float dsh[HOLD_D ? 1 : WAVES_PER_BLOCK][HOLD_D ? 1 : TT][CELL_TILE][4];
  float dl[HOLD_D ? TT : 1][3], cT[3], U[TT];
  f16x8 xb[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.xd + ((tile0 + tt) * CELL_TILE + r) * 4);
    const f32x4 m = *reinterpret_cast<const f32x4*>(a.tmeta + (tile0 + tt) * 4);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if constexpr (HOLD_D) dl[tt][c] = v[c];
      else if (h == 0) dsh[wave][tt][r][c] = v[c];
      if (tt == 0) cT[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[c])));  // one cell per wave
    }
    xb[tt] = cellmm_target_operand(v, h);
    U[tt] = 0.f;
  }

  f32x16 acc[TT];
  double outd[TT];
  float vprev[TT];  // row sum of acc[tt] at the last fold
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    outd[tt] = 0.0;
    vprev[tt] = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[tt][k] = 0.f;
  }

  const int64_t s_begin = (int64_t)seg * a.seg_stages;
  int64_t s_end = s_begin + a.seg_stages;
  if (s_end > a.m_stages) s_end = a.m_stages;

  // Stage s + 1 travels through registers: requested when stage s starts, written to LDS when stage s is done.
  // (LDS-DMA would save the registers, but the compiler then waits for the copy -- vmcnt(0) -- in front of the
  // first LDS read after its issue, i.e. at the START of the stage it was supposed to hide behind.)
  f32x4 pre[PIECES];
  auto fetch = [&](int64_t s) {
    const unsigned char* src = a.img + s * SB;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      pre[p] = *reinterpret_cast<const f32x4*>(src + (p * BLOCK_THREADS + (int)threadIdx.x) * 16);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      *reinterpret_cast<f32x4*>(&lds[buf][(p * BLOCK_THREADS + (int)threadIdx.x) * 16]) = pre[p];
  };
  if (s_begin < s_end) {
    fetch(s_begin);
    commit(0);
  }
  __syncthreads();

  int key_s = -2;     // cell of the source tiles the accumulators hold (-2: nothing yet)
  float S0 = 0.f;     // sum of W_j b_j over this lane's source row, tiles of the current cell
  float S0c = 0.f;    // ... and its running compensation
  float D1[3] = {0.f, 0.f, 0.f};  // log2 e (c_T - c_S)

  // Fold of the finished source cell into the fp64 sums.  The accumulators are NOT reset: they run on over the whole
  // segment and a cell's share is the difference of their row sums before and after it (`vprev`) -- resetting them
  // costs 16 v_mov per target tile and fold, a sixth of the fold's instructions; the running sums stay small (the
  // accumulators hold only the t + t^2/2 part, <= 1.6 % of the kernel values) and add ~1e-8 of rounding.
  auto fold = [&]() {
    const float s0 = cellmm_half_sum(S0) * CMM_TARGET_SCALE;  // the accumulators carry the targets' 2^6
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const f32x16 d = acc[tt];
      // the 16 rows with packed adds: 4 + 2 + 1 v_pk_add_f32 and one v_add_f32
      f32x2 p0 = f32x2{d[0], d[1]} + f32x2{d[2], d[3]};
      f32x2 p1 = f32x2{d[4], d[5]} + f32x2{d[6], d[7]};
      f32x2 p2 = f32x2{d[8], d[9]} + f32x2{d[10], d[11]};
      f32x2 p3 = f32x2{d[12], d[13]} + f32x2{d[14], d[15]};
      p0 = p0 + p1;
      p2 = p2 + p3;
      p0 = p0 + p2;
      const float v = cellmm_cross_half_add(p0[0] + p0[1]);
      outd[tt] += (double)(U[tt] * ((v - vprev[tt]) + s0));
      vprev[tt] = v;
    }
    S0 = 0.f;
    S0c = 0.f;
  };
  // a new source cell (wave-uniform): fold the finished one, then D1 and U_i = exp(-|x_i - c_S|^2) for this one
  auto new_cell = [&](const f32x4 cs, int ks) {
    if (key_s >= 0) fold();
    key_s = ks;
#pragma unroll
    for (int c = 0; c < 3; ++c) D1[c] = (cT[c] - cs[c]) * LOG2E;
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
  };

  for (int64_t s = s_begin; s < s_end; ++s) {
    const int buf = (int)((s - s_begin) & 1);
    if (s + 1 < s_end) fetch(s + 1);
    const unsigned char* base = &lds[buf][0];
    // running per-lane read addresses of the two streams (one v_add each per tile)
    const unsigned char* p_ef = base + CMM_E_OFF + r * 16;
    const unsigned char* p_b = base + CMM_B_OFF + r * 4;
    const f32x4* hdr = reinterpret_cast<const f32x4*>(base + CMM_HDR_OFF);
#pragma unroll 1
    for (int q = 0; q < CMM_STAGE_TILES; ++q) {
      const f32x4 cs = hdr[q];
      const int ks = __builtin_amdgcn_readfirstlane(__float_as_int(cs[3]));
      if (ks < 0) break;                    // pad tiles (key -1) only trail the last source tile (wave-uniform)
      if (ks != key_s) new_cell(cs, ks);    // wave-uniform, once per ~30 tiles
      // ---- A = psi_k(e_j) W_j(T) b_j for source row r, k-half h: ~27 VALU instructions
      const f32x4 ef = *reinterpret_cast<const f32x4*>(p_ef);
      const float bq = *reinterpret_cast<const float*>(p_b);
      p_ef += CELL_TILE * 16;
      p_b += CELL_TILE * 4;
      // the lane half's five monomial factors: every entry of A is then ONE product a_k * (W_j b_j)
      const f32x4 a14 = h ? f32x4{ef[2], ef[0] * ef[2], ef[2] * ef[2], ef[0] * ef[1]}
                          : f32x4{ef[0], ef[1], ef[0] * ef[0], ef[1] * ef[1]};
      const float a5 = h ? ef[1] * ef[2] : ef[1];
      const float arg = fmaf(ef[0], D1[0], fmaf(ef[1], D1[1], fmaf(ef[2], D1[2], ef[3])));
      const float wb = kexp2(arg) * bq;  // W_j b_j (x sigma_b); v_exp_f32 runs beside the MFMAs
      {  // compensated (Kahan) sum: this constant term is ~98 % of the kernel values, and a plain fp32 sum over the
         // cell's ~30 tiles was the largest single error of the kernel (max error over 4096 rows 5.7e-7 -> 3.5e-7)
        const float yk = wb - S0c;
        const float tk = S0 + yk;
        S0c = (tk - S0) - yk;
        S0 = tk;
      }
      const float u1 = a14[0] * wb, u2 = a14[1] * wb;   // x, y | z, x z   (two-way split below)
      const f16x2 R0 = cellmm_pk(u1, u2);
      const float c1 = (float)R0[0], c2 = (float)R0[1];
      const f16x2 R1 = cellmm_pk(u1 - c1, u2 - c2);
      const f16x2 R3 = cellmm_pk(a14[2] * wb, a14[3] * wb);  // x x, y y | z z, x y
      const f16x2 R2 = cellmm_pk(c1, a5 * wb);               // (x_h, y_h) | (z_h, y z)
      i32x4 yw;
      yw[0] = __builtin_bit_cast(int, R0);
      yw[1] = __builtin_bit_cast(int, R1);
      yw[2] = __builtin_bit_cast(int, R2);
      yw[3] = __builtin_bit_cast(int, R3);
      const f16x8 ya = __builtin_bit_cast(f16x8, yw);
      // ONE site of accumulating MFMAs and no branch around it (anything else makes the compiler copy
      // accumulator registers by the hundred): the empty tiles that pad a cell to TT tiles run too, their
      // columns are never read back
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#if CMM_BF16
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, xb[tt], acc[tt], 0, 0, 0);
#else
        acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ya, xb[tt], acc[tt], 0, 0, 0);
#endif
      }
    }
    if (s + 1 < s_end) commit(buf ^ 1);  // the other buffer was last read before the previous barrier
    __syncthreads();
  }
  if (key_s >= 0) fold();

  const double inv = (double)a.scale[1];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
    if (h == 0) a.part[(int64_t)seg * a.n_slots + (tile0 + tt - a.tile_base) * CELL_TILE + r] = outd[tt] * inv;
}

// ------------------------------------------------------------------------------------------------------------------------
// The same pair loop on v_mfma_f32_16x16x32_f16 (round 3).  Under the chip's power limit the 16x16x32 shape sustains more
// than 32x32x16 at equal flop: tools/mfma_pingpong.hip, this kernel's shape (an A operand rebuilt on the VALU, 28 VALU
// instructions per trip, random data, two waves per SIMD): 1.50 against 1.32 PFLOP/s; MI355X_MICROARCH 'DVFS give-back' 7.
//
// Algebra as above.  What changes is who holds what: K = 32 is TWO groups of 16 sources x the 16 monomial slots, the output
// tile is 16 row buckets x 16 targets (rows are only buckets: a fold sums them all), so per 32 sources x 32 targets there
// are two MFMAs (target halves) instead of one.
//   lane l:  cc = l & 15,  kg = l >> 4 (k group 8 kg .. 8 kg + 7):  slot half h = kg & 1,  source group G = kg >> 1
//   A operand (sources):  lane (cc, kg) holds the slots (h) of source row 16 G + cc of the tile       -- A[cc][8 kg + j]
//   B operand (targets):  lane (cc, kg) holds the slots (h) of target cc of the half tile             -- B[8 kg + j][cc]
//   accumulator (4 registers):  rows 4 (l >> 4) + reg, column cc; row sum = the 4 registers, then lanes l ^ 16 and l ^ 32
//                               (v_permlane16_swap, v_permlane32_swap)
// A wave owns the same TT tiles of 32 targets = 2 TT half tiles: 4 x 2 TT accumulator registers (64 at TT = 8, where
// cellmm_kernel needs 128) and 4 x 2 TT operand registers (64 against 32).
__device__ __forceinline__ float cellmm16_rows_sum(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);   // + lane ^ 16
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);  // + lane ^ 32
}
// sum over the wave's 32 distinct source rows (held by the lanes of rows 0 and 2 of 16; rows 1 and 3 duplicate them)
__device__ __forceinline__ float cellmm16_sources_sum(float v) {
  v += cellmm_dpp<0x111, 0xf>(v);  // row_shr:1
  v += cellmm_dpp<0x112, 0xf>(v);  // row_shr:2
  v += cellmm_dpp<0x114, 0xf>(v);  // row_shr:4
  v += cellmm_dpp<0x118, 0xf>(v);  // row_shr:8: lane 15 of every row of 16 holds the row's sum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15)) +
         __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
}

template <int TT>
__global__ void __launch_bounds__(BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(TT >= 8 ? 2 : 1)))
cellmm16_kernel(const CellmmArgs a) {
  static_assert(!CMM_BF16, "f16 operands only");
  constexpr int SB = CMM_STAGE_BYTES;
  constexpr int PIECES = SB / (16 * BLOCK_THREADS);
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr int HT = 2 * TT;  // half tiles of 16 targets
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SB];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int cc = lane & 15;
  const int kg = lane >> 4;
  const int h = kg & 1;
  const int rs = 16 * (kg >> 1) + cc;  // this lane's source row in a tile of 32
  const int64_t tile0 = a.tile_base + ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;

  constexpr bool HOLD_D = TT < 8;
  __shared__ __attribute__((aligned(16))) float dsh[HOLD_D ? 1 : WAVES_PER_BLOCK][HOLD_D ? 1 : HT][16][4];
  float dl[HOLD_D ? HT : 1][3], cT[3], U[HT];
  f16x8 xb[HT];
#pragma unroll
  for (int u = 0; u < HT; ++u) {
    const int64_t tile = tile0 + (u >> 1);
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.xd + (tile * CELL_TILE + 16 * (u & 1) + cc) * 4);
    const f32x4 m = *reinterpret_cast<const f32x4*>(a.tmeta + tile * 4);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if constexpr (HOLD_D) dl[u][c] = v[c];
      else if (kg == 0) dsh[wave][u][cc][c] = v[c];
      if (u == 0) cT[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m[c])));  // one cell per wave
    }
    xb[u] = cellmm_target_operand(v, h);
    U[u] = 0.f;
  }

  f32x4 acc[HT];
  double outd[HT];
  float vprev[HT];
#pragma unroll
  for (int u = 0; u < HT; ++u) {
    outd[u] = 0.0;
    vprev[u] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[u][k] = 0.f;
  }

  const int64_t s_begin = (int64_t)seg * a.seg_stages;
  int64_t s_end = s_begin + a.seg_stages;
  if (s_end > a.m_stages) s_end = a.m_stages;

  f32x4 pre[PIECES];
  auto fetch = [&](int64_t s) {
    const unsigned char* src = a.img + s * SB;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      pre[p] = *reinterpret_cast<const f32x4*>(src + (p * BLOCK_THREADS + (int)threadIdx.x) * 16);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p)
      *reinterpret_cast<f32x4*>(&lds[buf][(p * BLOCK_THREADS + (int)threadIdx.x) * 16]) = pre[p];
  };
  if (s_begin < s_end) {
    fetch(s_begin);
    commit(0);
  }
  __syncthreads();

  int key_s = -2;
  float S0 = 0.f, S0c = 0.f;
  float D1[3] = {0.f, 0.f, 0.f};

  auto fold = [&]() {  // as in cellmm_kernel
    const float s0 = cellmm16_sources_sum(S0) * CMM_TARGET_SCALE;
#pragma unroll
    for (int u = 0; u < HT; ++u) {
      const f32x4 d = acc[u];
      const float v = cellmm16_rows_sum((d[0] + d[1]) + (d[2] + d[3]));
      outd[u] += (double)(U[u] * ((v - vprev[u]) + s0));
      vprev[u] = v;
    }
    S0 = 0.f;
    S0c = 0.f;
  };
  auto new_cell = [&](const f32x4 cs, int ks) {
    if (key_s >= 0) fold();
    key_s = ks;
#pragma unroll
    for (int c = 0; c < 3; ++c) D1[c] = (cT[c] - cs[c]) * LOG2E;
#pragma unroll
    for (int u = 0; u < HT; ++u) {
      f32x4 dv;
      if constexpr (!HOLD_D) dv = *reinterpret_cast<const f32x4*>(&dsh[wave][u][cc][0]);
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float df;
        if constexpr (HOLD_D) df = dl[u][c] + (cT[c] - cs[c]);
        else df = dv[c] + (cT[c] - cs[c]);
        s2 = fmaf(df, df, s2);
      }
      U[u] = kexp2(s2 * -LOG2E);
    }
  };

  for (int64_t s = s_begin; s < s_end; ++s) {
    const int buf = (int)((s - s_begin) & 1);
    if (s + 1 < s_end) fetch(s + 1);
    const unsigned char* base = &lds[buf][0];
    const unsigned char* p_ef = base + CMM_E_OFF + rs * 16;
    const unsigned char* p_b = base + CMM_B_OFF + rs * 4;
    const f32x4* hdr = reinterpret_cast<const f32x4*>(base + CMM_HDR_OFF);
#pragma unroll 1
    for (int q = 0; q < CMM_STAGE_TILES; ++q) {
      const f32x4 cs = hdr[q];
      const int ks = __builtin_amdgcn_readfirstlane(__float_as_int(cs[3]));
      if (ks < 0) break;
      if (ks != key_s) new_cell(cs, ks);
      // ---- A = psi_k(e_j) W_j(T) b_j for source row rs, slot half h (as cellmm_kernel)
      const f32x4 ef = *reinterpret_cast<const f32x4*>(p_ef);
      const float bq = *reinterpret_cast<const float*>(p_b);
      p_ef += CELL_TILE * 16;
      p_b += CELL_TILE * 4;
      const f32x4 a14 = h ? f32x4{ef[2], ef[0] * ef[2], ef[2] * ef[2], ef[0] * ef[1]}
                          : f32x4{ef[0], ef[1], ef[0] * ef[0], ef[1] * ef[1]};
      const float a5 = h ? ef[1] * ef[2] : ef[1];
      const float arg = fmaf(ef[0], D1[0], fmaf(ef[1], D1[1], fmaf(ef[2], D1[2], ef[3])));
      const float wb = kexp2(arg) * bq;
      {
        const float yk = wb - S0c;
        const float tk = S0 + yk;
        S0c = (tk - S0) - yk;
        S0 = tk;
      }
      const float u1 = a14[0] * wb, u2 = a14[1] * wb;
      const f16x2 R0 = cellmm_pk(u1, u2);
      const float c1 = (float)R0[0], c2 = (float)R0[1];
      const f16x2 R1 = cellmm_pk(u1 - c1, u2 - c2);
      const f16x2 R3 = cellmm_pk(a14[2] * wb, a14[3] * wb);
      const f16x2 R2 = cellmm_pk(c1, a5 * wb);
      i32x4 yw;
      yw[0] = __builtin_bit_cast(int, R0);
      yw[1] = __builtin_bit_cast(int, R1);
      yw[2] = __builtin_bit_cast(int, R2);
      yw[3] = __builtin_bit_cast(int, R3);
      const f16x8 ya = __builtin_bit_cast(f16x8, yw);
#pragma unroll
      for (int u = 0; u < HT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ya, xb[u], acc[u], 0, 0, 0);
    }
    if (s + 1 < s_end) commit(buf ^ 1);
    __syncthreads();
  }
  if (key_s >= 0) fold();

  const double inv = (double)a.scale[1];
#pragma unroll
  for (int u = 0; u < HT; ++u)
    if (kg == 0)
      a.part[(int64_t)seg * a.n_slots + (tile0 + (u >> 1) - a.tile_base) * CELL_TILE + 16 * (u & 1) + cc] = outd[u] * inv;
}

// shape: 0 = cellmm_kernel (32x32x16), 1 = cellmm16_kernel (16x16x32)
hipError_t launch_cellmm_gaussian(int TT, int shape, const CellmmArgs& args, dim3 grid, hipStream_t stream, const char** kernel_name);

}  // namespace kmvp
