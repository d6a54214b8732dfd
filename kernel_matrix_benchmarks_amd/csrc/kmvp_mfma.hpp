// High-dimensional kernel products on the bf16 matrix cores of gfx950 (CDNA4).
//
//   a_i = sum_j k(x_i, y_j) b_j ,  optionally divided by sum_j k(x_i, y_j)
//
// Arithmetic restated from bruteforce.py:36-49 (the fast squared-distance form
// |x|^2 + |y|^2 - 2 x.y), :18-22 / :8-15 (kernel functions) and :142-145 (numerator
// and denominator in one sweep).  The (N,M) matrix is never formed.
//
// Tile algebra (one wavefront = 32 targets, one source tile = 32 sources):
//
//   1. squared distances straight out of the matrix pipe.  Sources and targets are
//      stored as AUGMENTED bf16 rows of KD = 16*KS entries
//         source j :  [ -2 y_j (D) , |y|^2 hi, mid, lo , 1 , 1 , 1 , 0.. ]
//         target i :  [    x_i (D) ,    1   ,  1 , 1 , |x|^2 hi, mid, lo , 0.. ]
//      so   S[j][i] = <src_j, tgt_i> = |y_j|^2 - 2 x_i.y_j + |x_i|^2   (fp32 accumulate).
//      The norms are computed in fp32 from the bf16-rounded coordinates and split
//      three ways (24 bits), so S is the exact squared distance of the rounded points
//      up to fp32 accumulation error -- no VALU instruction is spent on it.
//      KS MFMAs v_mfma_f32_32x32x16_bf16 with A = source tile (LDS), B = targets (VGPRs).
//   2. S has the TARGET on the lane (column) and the 16 registers run over sources, so
//      the kernel value p = k(max(S,0)) is evaluated elementwise on the VALU, the
//      denominator is a per-lane register sum, and
//   3. P is already the A operand of the second product  O[i][e] += sum_j P[j][i] V[j][e]
//      (accumulator-as-operand: registers 8s..8s+7 converted pairwise to bf16 are the
//      fragment of k-step s, with the k-order permuted: element j of lane half h is
//      tile row 16s + 8(j>>2) + 4h + (j&3)).  The signal tile is stored TRANSPOSED in LDS
//      ([column][source]) so that the matching B fragment is two 8-byte reads.
//      2*NT MFMAs for NT = ceil(E/32) column tiles.
//
// A workgroup is 4 wavefronts = 128 targets sharing one staged source tile; tiles are
// pre-packed by pack_mfma_sources_kernel as ready-to-copy LDS images (padded row strides:
// conflict-free ds_read_b128 / ds_read_b64), copied with coalesced 16-byte loads,
// double buffered, one barrier per tile.  Launch = target tile-blocks x source segments
// as for the low-D kernels.
//
// Roofline: per 32x32 tile and wave, KS + 2 NT MFMAs (32 cycles each on the SIMD's
// matrix pipe) against 16 x (1 max + transcendental(s) + 1/2 cvt + 1 add) VALU slots;
// for exp(-r) (sqrt + exp = 8 quarter-rate slots per pair) the VALU side is the longer
// one -- the transcendental rate, not the MFMA rate, bounds this kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_lowd.hpp"  // K_*, SIG_*, WAVES_PER_BLOCK, BLOCK_THREADS, block_to_work

namespace kmvp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));

constexpr int MFMA_TILE = 32;      // sources per tile = targets per wave
constexpr int MFMA_AUG = 6;        // augmentation columns appended to the D coordinates
constexpr int MFMA_V_STRIDE = 72;  // bytes per transposed-signal row (32 bf16 + 8 pad)

__host__ __device__ constexpr int mfma_ksteps(int D) { return (D + MFMA_AUG + 15) / 16; }
// exp(<x,y>) (K_EXPDOT): no norms; the LAST three columns of the last k-step carry the target's shift and the pad mask
//   source j :  [ y_j (D) , 0.. , 1     , 1     , live ? 0 : -3e38 ]
//   target i :  [ x_i (D) , 0.. , -m_hi , -m_lo , 1                ]        S[j][i] = <x_i, y_j> log2(e) - m_i
constexpr int MFMA_DOT_AUG = 3;
__host__ __device__ constexpr int mfma_ksteps_dot(int D) { return (D + MFMA_DOT_AUG + 15) / 16; }
// the Gaussian with the same shift (K_GAUSSIAN_SHIFTED): the six norm columns as above, the shift in columns 16 KS - 3 / - 2
//   source j :  [ -2 y, |y|^2 (3), 1, 1, 1, 0.. , 1    , 1    , 0 ]
//   target i :  [    x, 1, 1, 1, |x|^2 (3), 0.. , m_hi , m_lo , 0 ]        S[j][i] = |x_i - y_j|^2 + m_i ,  p = 2^-S
__host__ __device__ constexpr int mfma_ksteps_shifted(int D) { return (D + MFMA_AUG + MFMA_DOT_AUG + 15) / 16; }
// kernels that carry the per-target running shift, and the sign of S in log2(p) = SGN * S
template <int KERNEL> __host__ __device__ constexpr bool mfma_online() { return KERNEL == K_EXPDOT || KERNEL == K_GAUSSIAN_SHIFTED; }
template <int KERNEL> __host__ __device__ constexpr int mfma_sgn() { return KERNEL == K_EXPDOT ? 1 : -1; }
__host__ __device__ constexpr int mfma_y_stride(int KS) { return KS * 32 + 16; }  // bytes per source row
// bytes of one tile image, rounded up to a whole number of 16-byte pieces per thread of
// the copying workgroup (no predication in the staging loop)
__host__ __device__ constexpr int mfma_image_bytes(int KS, int NT) {
  return (MFMA_TILE * mfma_y_stride(KS) + NT * 32 * MFMA_V_STRIDE + 4095) / 4096 * 4096;
}

struct MfmaArgs {
  const __bf16* xa;          // augmented targets [n_pad][16*KS]
  const unsigned char* img;  // source tile images [m_tiles][image_bytes]
  float* part;               // partial numerators [segments][n_pad][NT*32]
  float* partd;              // partial denominators [segments][n_pad]
  int64_t n_pad;
  int64_t m_tiles;           // source tiles in this shard
  int64_t seg_tiles;         // tiles per segment
  int segments;
  int tile_blocks;
  int64_t j_offset;
  int64_t m_total;
  float* kexp;               // exp(<x,y>): exponents of the partial sums [segments][n_pad] (sums are at the scale 2^-kexp)
};

template <int KERNEL>
__device__ __forceinline__ float mfma_kval(float s) {
  if constexpr (KERNEL == K_GAUSSIAN || KERNEL == K_GAUSSIAN_SHIFTED) {
    return kexp2(-s);
  } else if constexpr (KERNEL == K_ABSEXP) {
    // |s| instead of max(s, 0): a free source modifier.  s < 0 only by rounding noise of the
    // bf16/fp32 distance (~1e-6), where sqrt(|s|) ~ 1e-3 is as good an answer as 0.
    return kexp2(-__builtin_amdgcn_sqrtf(__builtin_fabsf(s)));
  } else if constexpr (KERNEL == K_EXPDOT) {
    return kexp2(s);  // s = <x,y> log2(e) - m_i straight from the matrix pipe
  } else {
    return __builtin_amdgcn_rsqf(__builtin_fabsf(s));
  }
}

// The same in two stages for the rotated pipeline (mfma_pipe_kernel VAR bit 2): stage 1 maps s to the argument of the last
// transcendental IN PLACE (exp(-r): r = sqrt|s|; the others: s), stage 2 is that transcendental.  A whole scheduling region
// lies between the two, so the second never waits for the first.
template <int KERNEL>
__device__ __forceinline__ float mfma_kval_stage1(float s) {
  if constexpr (KERNEL == K_ABSEXP) return __builtin_amdgcn_sqrtf(__builtin_fabsf(s));
  else return s;
}
template <int KERNEL>
__device__ __forceinline__ float mfma_kval_stage2(float u) {
  if constexpr (KERNEL == K_GAUSSIAN || KERNEL == K_GAUSSIAN_SHIFTED) return kexp2(-u);
  else if constexpr (KERNEL == K_ABSEXP) return kexp2(-u);
  else if constexpr (KERNEL == K_EXPDOT) return kexp2(u);
  else return __builtin_amdgcn_rsqf(__builtin_fabsf(u));
}

// Scheduling pattern of one region: NM times { one MFMA, then its share of NTR transcendentals } (the shares differ by at
// most one; the builtin wants literal constants, hence the recursion)
template <int I, int NM, int NTR>
__device__ __forceinline__ void mfma_sched_pattern() {
  if constexpr (I < NM) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int share = (NTR * (I + 1)) / NM - (NTR * I) / NM;
    if constexpr (share > 0) __builtin_amdgcn_sched_group_barrier(0x400, share, 0);
    mfma_sched_pattern<I + 1, NM, NTR>();
  }
}

// row of the 32x32 accumulator held in register `reg` by lane half `h`
__device__ __forceinline__ constexpr int acc_row(int reg, int h) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * h;
}

// ---- exp(<x,y>): the per-target running shift (softmax attention without a range limit).
// The kernel values of target i are p = 2^(s - m_i), s = <x_i, y_j> log2(e); m_i is an INTEGER that rides in the target's
// MFMA operand (two bf16 columns: a multiple of 128 and a remainder, both exact), so a pair costs no instruction for it.
// bf16 and fp32 share their exponent range, so m_i only has to keep the values inside it: it is set from the row maximum of
// the FIRST source tile a wave sees (p <= 1 there) and touched again only when a tile holds a value whose p would pass 2^60
// -- seen on the tile's s itself (eight v_max3 and a compare per tile; the pipelined kernel takes the verdict on tile t + 1 at
// the end of step t, in the shadow of its last P.V MFMAs, and only tests a scalar at the top of step t + 1), BEFORE any of the
// tile's values exists.  With every p <= 2^60 the sums of up
// to 2^60 sources stay finite.  An event (wave-uniform, rare) moves m_i up by delta_i >= 0: the tile's s is lowered by
// delta, the denominator and the target's output rows are scaled by 2^-delta (exact), the operand columns are patched
// (the next tile's distances are computed after the event: with the new operand).  Values more than 126 binades under a
// row's own maximum flush to zero.  Partial sums leave the kernel with their exponent (MfmaArgs::kexp = -m_i) and are
// merged as (mantissa, exponent) pairs.
constexpr float MFMA_DOT_LIMIT_LOG2 = 60.f;      // a tile with log2(p) beyond this triggers an event
constexpr float MFMA_DOT_MAX_SHIFT = 32000.f;    // |m| < 2^15: m_hi / 128 and m_lo are bf16 integers

// largest log2(p) = SGN * s of a tile's 16 values on this lane
// (v_max3_f32 / v_min3_f32 spelled out: fmaxf() costs a canonicalising v_max_f32 per value on top -- 30 instead of 16
// instructions per step on the unit that bounds these kernels.  A NaN among the values is ignored, as fmaxf would.)
template <int SGN>
__device__ __forceinline__ float mfma_tile_max(const f32x16& s) {
  float t = s[0];
  if constexpr (SGN > 0) {
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(s[1]), "v"(s[2]));
#pragma unroll
    for (int q = 3; q < 15; q += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(s[q]), "v"(s[q + 1]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(s[15]), "v"(s[15]));
    return t;
  } else {
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(s[1]), "v"(s[2]));
#pragma unroll
    for (int q = 3; q < 15; q += 2) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(s[q]), "v"(s[q + 1]));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(s[15]), "v"(s[15]));
    return -t;
  }
}

// one target tile of one wave.  s: the tile's 16 values per lane (target = lane & 31, both lane halves), lowered in place;
// o: the tile's output accumulators (row acc_row(q, h) = target); den: this lane's partial denominator; m: the target's
// shift; xlast: the target's operand of the last k-step; scratch: 32 floats of LDS owned by the wave.
// SGN: log2 of a kernel value is SGN * s (+1: exp(<x,y>), s = logit - m; -1: the shifted Gaussian, s = distance^2 + m).
template <int NT, int SGN>
__device__ __forceinline__ void mfma_dot_event(bool first, f32x16& s, f32x16 (&o)[NT], float& den, float& m, bf16x8& xlast,
                                               float* scratch, int r, int h) {
  float tmax = mfma_tile_max<SGN>(s);
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
  float delta = ceilf(tmax);                       // the tile's largest value becomes <= 1
  if (!first) delta = fmaxf(delta, 0.f);           // afterwards the shift only ever grows
  if (!(delta == delta)) delta = 0.f;              // a NaN row stays NaN
  const float m_new = fminf(fmaxf(m + delta, -MFMA_DOT_MAX_SHIFT), MFMA_DOT_MAX_SHIFT);
  delta = m_new - m;
  m = m_new;
#pragma unroll
  for (int q = 0; q < 16; ++q) s[q] -= (float)SGN * delta;
  const int di = (int)delta;
  den = ldexpf(den, -di);
  // the output rows are targets too, in another layout: the deltas travel through LDS
  if (h == 0) scratch[r] = delta;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const mf32x4 f = *reinterpret_cast<const mf32x4*>(scratch + 8 * g + 4 * h);  // rows 8 g + 4 h + (0..3) = acc_row(4 g + j, h)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[nt][4 * g + j] = ldexpf(o[nt][4 * g + j], -(int)f[j]);
  }
  __builtin_amdgcn_wave_barrier();
  const float m_hi = 128.f * truncf(m_new * (1.f / 128.f));
  if (h == 1) {  // k = 16 KS - 3 and 16 KS - 2: elements 5 and 6 of the upper lane half's fragment; s = ... - SGN m
    xlast[5] = (__bf16)((float)-SGN * m_hi);
    xlast[6] = (__bf16)((float)-SGN * (m_new - m_hi));
  }
}

template <int KERNEL, int KS, int NT, int TW>
__global__ void __launch_bounds__(BLOCK_THREADS) mfma_kernel(const MfmaArgs a) {
  constexpr int WPB = WAVES_PER_BLOCK;
  constexpr int KD = 16 * KS;
  constexpr int YS = mfma_y_stride(KS);
  constexpr int IMG = mfma_image_bytes(KS, NT);
  constexpr int NPIECES = IMG / 1024;               // 1 KiB pieces of an image, dealt to the waves round robin
  constexpr int PIECES = (NPIECES + WPB - 1) / WPB;
  static_assert(IMG % 1024 == 0, "image is a whole number of LDS-DMA pieces");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][IMG];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  // a wave owns TW target tiles of 32; the source fragments it reads from LDS serve all of them
  const int64_t i0 = ((int64_t)tb * WPB + wave) * (MFMA_TILE * TW);

  // B operand of the distance product: this lane's targets, 8 consecutive k per k-step
  bf16x8 xb[TW][KS];
#pragma unroll
  for (int w = 0; w < TW; ++w)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      xb[w][ks] = *reinterpret_cast<const bf16x8*>(a.xa + (i0 + w * MFMA_TILE + r) * KD + ks * 16 + 8 * h);

  // inverse-distance zero column of this lane's targets (local to the shard), and the
  // wave-uniform range of zero columns of the wave's targets (conservative when the mod wraps)
  int64_t jz[TW], jz_lo = 0, jz_hi = -1;
#pragma unroll
  for (int w = 0; w < TW; ++w) jz[w] = -1;
  if constexpr (KERNEL == K_INVDIST) {
#pragma unroll
    for (int w = 0; w < TW; ++w) {
      const int64_t g = (i0 + w * MFMA_TILE + r) % (a.m_total + 1);
      jz[w] = (g < a.m_total) ? g - a.j_offset : (int64_t)-1;
    }
    const int64_t g_lo = i0 % (a.m_total + 1);
    const int64_t g_hi = g_lo + (MFMA_TILE * TW - 1);
    if (g_hi <= a.m_total) {
      jz_lo = g_lo - a.j_offset;
      jz_hi = g_hi - a.j_offset;
    } else {
      jz_lo = INT64_MIN / 2;
      jz_hi = INT64_MAX / 2;
    }
  }

  f32x16 o[TW][NT];
  float den[TW];
  // exp(<x,y>): the targets' shifts, whether a tile has set them yet, and the wave's LDS scratch of an event
  constexpr bool DOT = mfma_online<KERNEL>();
  __shared__ __attribute__((aligned(16))) float dscr[DOT ? WPB : 1][DOT ? MFMA_TILE : 1];
  float msh[TW];
  bool unset[TW];
#pragma unroll
  for (int w = 0; w < TW; ++w) {
    den[w] = 0.f;
    msh[w] = 0.f;
    unset[w] = true;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) o[w][nt][q] = 0.f;
  }

  const int64_t t_begin = (int64_t)seg * a.seg_tiles;
  int64_t t_end = t_begin + a.seg_tiles;
  if (t_end > a.m_tiles) t_end = a.m_tiles;

  // Staging by LDS-DMA (global_load_lds_dwordx4): every wave-instruction copies 1 KiB of
  // the tile image straight into LDS (wave-uniform LDS base + lane*16, per-lane global
  // address), no VGPRs held across the tile.  The image of tile t+1 is requested right
  // after the barrier that ends tile t-1 (its buffer is free then) and is waited for by
  // the vmcnt(0) that __syncthreads() emits at the end of tile t: the global latency is
  // covered by a whole tile of MFMA + VALU work, one barrier per tile.
  auto stage_tile = [&](int64_t t, int buf) {
    const unsigned char* src = a.img + t * IMG;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int piece = (p * WPB + wave) * 1024;  // wave-uniform LDS offset
      if (NPIECES % WPB == 0 || piece < IMG)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + piece + lane * 16),
            (__attribute__((address_space(3))) void*)(&lds[buf][piece]), 16, 0, 0);
    }
  };
  if (t_begin < t_end) stage_tile(t_begin, 0);
  __syncthreads();  // vmcnt(0) + barrier: tile t_begin and the target fragments have landed

  for (int64_t t = t_begin; t < t_end; ++t) {
    const int buf = (int)((t - t_begin) & 1);
    if (t + 1 < t_end) stage_tile(t + 1, buf ^ 1);
    const unsigned char* ly = &lds[buf][0];
    const unsigned char* lv = &lds[buf][MFMA_TILE * YS];

    // source fragments of the tile, read once for all TW target tiles
    bf16x8 ya[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      ya[ks] = *reinterpret_cast<const bf16x8*>(ly + r * YS + (ks * 16 + 8 * h) * 2);
    bf16x8 vb[2][NT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned char* row = lv + (nt * 32 + r) * MFMA_V_STRIDE;
        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(row + (16 * s2 + 4 * h) * 2);
        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(row + (16 * s2 + 8 + 4 * h) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vb[s2][nt][j] = v0[j];
          vb[s2][nt][4 + j] = v1[j];
        }
      }
    const int64_t j0 = t * MFMA_TILE;
    bool check = false;
    if constexpr (KERNEL == K_INVDIST) check = (j0 + MFMA_TILE - 1 >= jz_lo) && (j0 <= jz_hi);

#pragma unroll
    for (int w = 0; w < TW; ++w) {
      // ---- 1. S[j][i]: KS MFMAs, A = source rows (row r of the tile), B = targets
      f32x16 s;
#pragma unroll
      for (int q = 0; q < 16; ++q) s[q] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[ks], xb[w][ks], s, 0, 0, 0);

      // ---- 2. kernel values on the VALU; target on the lane, 16 sources in registers
      if constexpr (DOT) {  // (kmvp_mfma.hpp "the per-target running shift")
        if (unset[w] || __any(mfma_tile_max<mfma_sgn<KERNEL>()>(s) > MFMA_DOT_LIMIT_LOG2)) {  // wave-uniform, rare after the first tile
          mfma_dot_event<NT, mfma_sgn<KERNEL>()>(unset[w], s, o[w], den[w], msh[w], xb[w][KS - 1], &dscr[wave][0], r, h);
          unset[w] = false;
        }
      }
      float p[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float k = mfma_kval<KERNEL>(s[q]);
        if constexpr (KERNEL == K_INVDIST) {
          if (check) k = (j0 + acc_row(q, h) == jz[w]) ? 0.f : k;
        }
        p[q] = k;
        den[w] += k;
      }

      // ---- 3. O[i][e] += sum_j P[j][i] V[j][e]: P registers are the A operand
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pa;
#pragma unroll
        for (int j = 0; j < 8; ++j) pa[j] = (__bf16)p[8 * s2 + j];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          o[w][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb[s2][nt], o[w][nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: numerators [segment][target][column] (128-byte rows per register),
  // denominators: the two lane halves hold the two halves of each target's sources
#pragma unroll
  for (int w = 0; w < TW; ++w) {
    const float dsum = den[w] + __shfl_xor(den[w], 32);
    float* part = a.part + ((int64_t)seg * a.n_pad + i0 + w * MFMA_TILE) * (NT * 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) part[(int64_t)acc_row(q, h) * (NT * 32) + nt * 32 + r] = o[w][nt][q];
    if (h == 0) a.partd[(int64_t)seg * a.n_pad + i0 + w * MFMA_TILE + r] = dsum;
    if constexpr (DOT) {
      if (h == 0) a.kexp[(int64_t)seg * a.n_pad + i0 + w * MFMA_TILE + r] = unset[w] ? INFINITY : -msh[w];
    }
  }
}


// ------------------------------------------------------------------------------------
// Software-pipelined variant (two target tiles per wave, KS <= MFMA_PIPE_MAX_KS,
// NT <= MFMA_PIPE_MAX_NT).  rocprofv3 on mfma_kernel at config 3: VALU active 62 % of the
// cycles, matrix pipe busy 44 %, both at once 12 % -- the two waves of a SIMD run their
// MFMA and VALU phases one after the other.  Here every wave overlaps them by itself: the
// distance MFMAs of source tile t+1 are issued between the transcendentals of tile t (the
// matrix pipe is asynchronous to the wave once an MFMA has issued), and the P.V MFMAs of a
// target tile run under the VALU work of the next one.  Costs: the squared distances of
// the next tile live in registers (+16 TW VGPRs) and LDS holds three tile images
// (t: signal fragments, t+1: source fragments, t+2: in flight).
constexpr int MFMA_PIPE_MAX_KS = 6;
constexpr int MFMA_PIPE_MAX_NT = 2;
// quarter-rate instructions per kernel value: exp2 (Gaussian), sqrt + exp2, rsq
template <int KERNEL>
__host__ __device__ constexpr int MFMA_TRANS_PER_PAIR() { return KERNEL == K_ABSEXP ? 2 : 1; }

// VAR (compile time; the "mfma_variant" option picks the instantiation):
//   bit 0  DEN_MFMA  the denominators sum_j P[j][i] leave the VALU: one more accumulator tile per target tile, fed by
//                    P x (a constant B operand with ones in column 0) -- 2 MFMAs per tile pair instead of 16 v_add_f32
//                    (the VALU, not the matrix pipe, is the busy unit here); the sums are then over the bf16-rounded
//                    kernel values, the very numbers the numerators use
//   bit 2  ROTATE    the loop is rotated by one transcendental stage: the FIRST stage of target tile 0's kernel
//                    values of tile t + 1 (exp(-r): the square roots, in place in the distance registers) runs under the
//                    P.V MFMAs of target tile 1 of tile t, and within a step the two stages of a tile's values always
//                    sit in different scheduling regions (no transcendental waits for the one it depends on):
//                      A: distances of t + 1 (2 KS MFMAs) | stage 2 of target tile 0, stage 1 of target tile 1
//                      B: P.V of target tile 0            | stage 2 of target tile 1
//                      C: P.V of target tile 1            | stage 1 of target tile 0 of tile t + 1
template <int KERNEL, int KS, int NT, int VAR = 0>
__global__ void __launch_bounds__(BLOCK_THREADS) mfma_pipe_kernel(const MfmaArgs a) {
  constexpr bool DEN_MFMA = (VAR & 1) != 0;
  constexpr bool ROTATE = (VAR & 4) != 0;
  constexpr bool DOT = mfma_online<KERNEL>();
  static_assert(!DOT || VAR == 0, "the running shift is built into the plain pipeline only");
  static_assert((VAR & 2) == 0, "bit 1 was the deferred-P.V arm: measured 5 % slower and removed (LAB_NOTES.md)");
  constexpr int TW = 2;
  constexpr int KD = 16 * KS;
  constexpr int YS = mfma_y_stride(KS);
  constexpr int IMG = mfma_image_bytes(KS, NT);
  constexpr int PIECES = IMG / (16 * BLOCK_THREADS);
  __shared__ __attribute__((aligned(16))) unsigned char lds[3][IMG];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t i0 = ((int64_t)tb * WAVES_PER_BLOCK + wave) * (MFMA_TILE * TW);

  bf16x8 xb[TW][KS];
#pragma unroll
  for (int w = 0; w < TW; ++w)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      xb[w][ks] = *reinterpret_cast<const bf16x8*>(a.xa + (i0 + w * MFMA_TILE + r) * KD + ks * 16 + 8 * h);

  int64_t jz[TW], jz_lo = 0, jz_hi = -1;
#pragma unroll
  for (int w = 0; w < TW; ++w) jz[w] = -1;
  if constexpr (KERNEL == K_INVDIST) {
#pragma unroll
    for (int w = 0; w < TW; ++w) {
      const int64_t g = (i0 + w * MFMA_TILE + r) % (a.m_total + 1);
      jz[w] = (g < a.m_total) ? g - a.j_offset : (int64_t)-1;
    }
    const int64_t g_lo = i0 % (a.m_total + 1);
    const int64_t g_hi = g_lo + (MFMA_TILE * TW - 1);
    if (g_hi <= a.m_total) {
      jz_lo = g_lo - a.j_offset;
      jz_hi = g_hi - a.j_offset;
    } else {
      jz_lo = INT64_MIN / 2;
      jz_hi = INT64_MAX / 2;
    }
  }

  f32x16 o[TW][NT];
  f32x16 oden[TW];  // DEN_MFMA: column 0 holds the denominators of target tile w (a tile of its own per target tile: a
                    // shared one would let a NaN row of one tile -- NaN x 0 -- into the other tile's sums)
  float den[TW];
  __shared__ __attribute__((aligned(16))) float dscr[DOT ? WAVES_PER_BLOCK : 1][DOT ? MFMA_TILE : 1];
  float msh[TW];
  bool unset[TW], over[TW];  // over: the NEXT tile holds a value beyond the limit (wave-uniform; the first tile sets the shift anyway)
#pragma unroll
  for (int w = 0; w < TW; ++w) {
    den[w] = 0.f;
    msh[w] = 0.f;
    unset[w] = true;
    over[w] = false;
#pragma unroll
    for (int q = 0; q < 16; ++q) oden[w][q] = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) o[w][nt][q] = 0.f;
  }
  // B operand of the denominator product: B[k][col] = 1 for col == 0, all 16 k of a step
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)(r == 0 ? 1.f : 0.f);

  const int64_t t_begin = (int64_t)seg * a.seg_tiles;
  int64_t t_end = t_begin + a.seg_tiles;
  if (t_end > a.m_tiles) t_end = a.m_tiles;

  auto stage_tile = [&](int64_t t, int buf) {
    const unsigned char* src = a.img + t * IMG;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int piece = (p * WAVES_PER_BLOCK + wave) * 1024;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + piece + lane * 16),
          (__attribute__((address_space(3))) void*)(&lds[buf][piece]), 16, 0, 0);
    }
  };
  // squared distances of one source tile against both target tiles: 2 KS MFMAs
  auto distances = [&](const unsigned char* ly, f32x16 (&s)[TW]) {
    bf16x8 ya[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      ya[ks] = *reinterpret_cast<const bf16x8*>(ly + r * YS + (ks * 16 + 8 * h) * 2);
#pragma unroll
    for (int w = 0; w < TW; ++w)
#pragma unroll
      for (int q = 0; q < 16; ++q) s[w][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int w = 0; w < TW; ++w) s[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[ks], xb[w][ks], s[w], 0, 0, 0);
  };

  if (t_begin >= t_end) {
    // nothing to do for this segment: the partials still have to be written (zeros)
  } else {
    stage_tile(t_begin, 0);
    if (t_begin + 1 < t_end) stage_tile(t_begin + 1, 1);
  }
  __syncthreads();  // vmcnt(0) + barrier: tiles t_begin, t_begin + 1 and the target fragments have landed

  // two register sets for the squared distances, swapped by unrolling the tile loop twice
  // (a copy per tile would cost 16 TW v_mov on the VALU, which is the busier unit)
  f32x16 s_a[TW], s_b[TW];
  if (t_begin < t_end) distances(&lds[0][0], s_a);

  // the P.V (and denominator) MFMAs of target tile w
  auto pv = [&](int w, const bf16x8 (&pa)[2], const bf16x8 (&vb)[2][NT]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) o[w][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s2], vb[s2][nt], o[w][nt], 0, 0, 0);
      if constexpr (DEN_MFMA) oden[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s2], ones, oden[w], 0, 0, 0);
    }
  };
  constexpr int PV_MFMAS = 2 * NT + (DEN_MFMA ? 2 : 0);
  constexpr int TRANS = MFMA_TRANS_PER_PAIR<KERNEL>() * 16;  // quarter-rate instructions per target tile

  int buf = 0;  // image of tile t
  // ROTATE: s_cur[0] arrives with stage 1 applied
  auto step_rot = [&](int64_t t, f32x16 (&s_cur)[TW], f32x16 (&s_next)[TW]) {
    const int buf1 = buf == 2 ? 0 : buf + 1;
    const int buf2 = buf1 == 2 ? 0 : buf1 + 1;
    if (t + 2 < t_end) stage_tile(t + 2, buf2);
    const unsigned char* lv = &lds[buf][MFMA_TILE * YS];
    bf16x8 vb[2][NT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned char* row = lv + (nt * 32 + r) * MFMA_V_STRIDE;
        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(row + (16 * s2 + 4 * h) * 2);
        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(row + (16 * s2 + 8 + 4 * h) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vb[s2][nt][j] = v0[j];
          vb[s2][nt][4 + j] = v1[j];
        }
      }
    const int64_t j0 = t * MFMA_TILE;
    bool check = false;
    if constexpr (KERNEL == K_INVDIST) check = (j0 + MFMA_TILE - 1 >= jz_lo) && (j0 <= jz_hi);
    constexpr int T2 = 16;                                   // stage-2 transcendentals per target tile
    constexpr int T1 = KERNEL == K_ABSEXP ? 16 : 0;          // stage-1 transcendentals per target tile
    auto values = [&](int w, bf16x8 (&pa)[2]) {             // stage 2 + denominators + bf16 fragments
      float p[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float k = mfma_kval_stage2<KERNEL>(s_cur[w][q]);
        if constexpr (KERNEL == K_INVDIST) {
          if (check) k = (j0 + acc_row(q, h) == jz[w]) ? 0.f : k;
        }
        p[q] = k;
        if constexpr (!DEN_MFMA) den[w] += k;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pa[s2][j] = (__bf16)p[8 * s2 + j];
    };
    bf16x8 pa0[2], pa1[2];
    // ---- region A
    distances(&lds[buf1][0], s_next);
    values(0, pa0);
#pragma unroll
    for (int q = 0; q < 16; ++q) s_cur[1][q] = mfma_kval_stage1<KERNEL>(s_cur[1][q]);
    mfma_sched_pattern<0, 2 * KS, T2 + T1>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- region B
    pv(0, pa0, vb);
    values(1, pa1);
    mfma_sched_pattern<0, PV_MFMAS, T2>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- region C
    pv(1, pa1, vb);
    if constexpr (T1 > 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) s_next[0][q] = mfma_kval_stage1<KERNEL>(s_next[0][q]);
      mfma_sched_pattern<0, PV_MFMAS, T1>();
    }
    buf = buf1;
    __syncthreads();
  };
  auto step = [&](int64_t t, f32x16 (&s_cur)[TW], f32x16 (&s_next)[TW]) {
    const int buf1 = buf == 2 ? 0 : buf + 1;   // tile t + 1 (landed: waited for at the end of iteration t - 1)
    const int buf2 = buf1 == 2 ? 0 : buf1 + 1; // tile t + 2 (free: last read in iteration t - 1)
    if (t + 2 < t_end) stage_tile(t + 2, buf2);
    const unsigned char* lv = &lds[buf][MFMA_TILE * YS];

    bf16x8 vb[2][NT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned char* row = lv + (nt * 32 + r) * MFMA_V_STRIDE;
        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(row + (16 * s2 + 4 * h) * 2);
        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(row + (16 * s2 + 8 + 4 * h) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vb[s2][nt][j] = v0[j];
          vb[s2][nt][4 + j] = v1[j];
        }
      }
    const int64_t j0 = t * MFMA_TILE;
    bool check = false;
    if constexpr (KERNEL == K_INVDIST) check = (j0 + MFMA_TILE - 1 >= jz_lo) && (j0 <= jz_hi);

    if constexpr (DOT) {
      // (kmvp_mfma.hpp "the per-target running shift": the verdict on this tile was taken at the end of the previous step,
      // in the shadow of its last P.V MFMAs; the event runs before any of the tile's values exists and before the
      // distances of tile t + 1 are issued -- they see the patched operand)
#pragma unroll
      for (int w = 0; w < TW; ++w) {
        if (unset[w] || over[w]) {
          mfma_dot_event<NT, mfma_sgn<KERNEL>()>(unset[w], s_cur[w], o[w], den[w], msh[w], xb[w][KS - 1], &dscr[wave][0], r, h);
          unset[w] = false;
        }
      }
    }

    // Scheduling regions per source tile, each pairing MFMAs with independent VALU work of the same wave (an MFMA that
    // cannot enter the matrix pipe yet blocks the wave's following instructions, so the MFMAs are spread between the
    // transcendentals, not bunched):
    //   A: distances of tile t + 1 (2 KS MFMAs)      | kernel values of target tile 0
    //   B: P.V of target tile 0 (2 NT MFMAs)         | kernel values of target tile 1
    //   C: P.V of target tile 1 (2 NT MFMAs), barrier
    // (Past the last tile the t + 1 buffer holds stale data; that result is never used.)
    distances(&lds[buf1][0], s_next);

    bf16x8 pa[TW][2];
#pragma unroll
    for (int w = 0; w < TW; ++w) {
      float p[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float k = mfma_kval<KERNEL>(s_cur[w][q]);
        if constexpr (KERNEL == K_INVDIST) {
          if (check) k = (j0 + acc_row(q, h) == jz[w]) ? 0.f : k;
        }
        p[q] = k;
        if constexpr (!DEN_MFMA) den[w] += k;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pa[w][s2][j] = (__bf16)p[8 * s2 + j];
      if (w > 0) pv(w - 1, pa[w - 1], vb);
      // one MFMA, then enough transcendentals to cover its 32 cycles on the matrix pipe
      if (w == 0) {
        constexpr int NM = 2 * KS;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, TRANS / NM > 0 ? TRANS / NM : 1, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < PV_MFMAS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x400, TRANS / PV_MFMAS, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    pv(TW - 1, pa[TW - 1], vb);
    if constexpr (DOT) {  // the next tile's verdict, on the VALU while the matrix pipe works the last P.V MFMAs off
#pragma unroll
      for (int w = 0; w < TW; ++w) over[w] = __any(mfma_tile_max<mfma_sgn<KERNEL>()>(s_next[w]) > MFMA_DOT_LIMIT_LOG2);
    }
    buf = buf1;
    __syncthreads();  // tile t + 2 has landed (vmcnt(0)), nobody reads image t any more
  };
  int64_t t = t_begin;
  if constexpr (ROTATE) {
    if (t_begin < t_end) {
#pragma unroll
      for (int q = 0; q < 16; ++q) s_a[0][q] = mfma_kval_stage1<KERNEL>(s_a[0][q]);
    }
    for (; t + 1 < t_end; t += 2) {
      step_rot(t, s_a, s_b);
      step_rot(t + 1, s_b, s_a);
    }
    if (t < t_end) step_rot(t, s_a, s_b);
  } else {
    for (; t + 1 < t_end; t += 2) {
      step(t, s_a, s_b);
      step(t + 1, s_b, s_a);
    }
    if (t < t_end) step(t, s_a, s_b);
  }

#pragma unroll
  for (int w = 0; w < TW; ++w) {
    float* part = a.part + ((int64_t)seg * a.n_pad + i0 + w * MFMA_TILE) * (NT * 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) part[(int64_t)acc_row(q, h) * (NT * 32) + nt * 32 + r] = o[w][nt][q];
    if constexpr (DEN_MFMA) {
      if (r == 0) {  // column 0 of the tile: rows = targets
#pragma unroll
        for (int q = 0; q < 16; ++q) a.partd[(int64_t)seg * a.n_pad + i0 + w * MFMA_TILE + acc_row(q, h)] = oden[w][q];
      }
    } else {
      const float dsum = den[w] + __shfl_xor(den[w], 32);
      if (h == 0) a.partd[(int64_t)seg * a.n_pad + i0 + w * MFMA_TILE + r] = dsum;
    }
    if constexpr (DOT) {
      if (h == 0) a.kexp[(int64_t)seg * a.n_pad + i0 + w * MFMA_TILE + r] = unset[w] ? INFINITY : -msh[w];
    }
  }
}

}  // namespace kmvp
