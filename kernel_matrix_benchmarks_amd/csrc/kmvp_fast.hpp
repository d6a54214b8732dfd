// Low-dimensional pair loop with the squared distance on the bf16 matrix cores at fp32
// accuracy ("fast_sqdists" of the reference, bruteforce.py:36-49, re-thought for gfx950).
//
//   s_ij = |x_i|^2 + |y_j|^2 - 2 x_i.y_j
//
// is bilinear in augmented coordinates, hence an MFMA.  The f32-input MFMA of gfx950 runs
// at the fp32 VECTOR rate and (measured) does not overlap with VALU work, so it buys
// little.  The bf16 MFMA pipe is a separate unit that runs beside the VALU at 16x the
// rate -- but bf16 has 8 significant bits.  Split-precision restores fp32 accuracy:
// every fp32 coordinate is written EXACTLY as a sum of three bf16 numbers
//     v = v_h + v_m + v_l          (8 + 8 + 8 = 24 significant bits)
// and the product x.y is expanded, keeping the six partial products down to 2^-24:
//     x y ~ x_h y_h + x_m y_h + x_h y_m + x_l y_h + x_m y_m + x_h y_l .
// bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so the tile
// of s that falls out of the matrix pipe has the accuracy of the reference's
// fast_sqdists=True in float32 (error ~ eps32 * (|x'|^2 + |y'|^2); x', y' are centred on
// the cloud's bounding box and carry the kernel's constant).  Rows of K = 6 D + 6:
//     source j : per d [ -2y_h, -2y_h, -2y_m, -2y_h, -2y_m, -2y_l ] , |y'|^2 h,m,l , 1,1,1
//     target i : per d [   x_h,   x_m,   x_h,   x_l,   x_m,   x_h ] ,  1, 1, 1 , |x'|^2 h,m,l
// KS = ceil(K/16) chained v_mfma_f32_32x32x16_bf16 per 32 sources x 32 targets.
//
// What it buys: the difference form costs 11 VALU issue slots per pair (6 for s, 4 for the
// quarter-rate exp2, 1 FMA).  Here the VALU only runs the exp2 and the FMA (5 slots); the
// matrix pipe needs KS * 32 cycles per 1024 pairs, far below the VALU's 160.
//
// Mapping (as kmvp_mfma.hpp): the result tile has the TARGET on the lane (column) and 16
// SOURCES in the lane's registers, so k(s) b_j is one v_exp_f32 + one FMA per register
// with the source's signal read from LDS, summed per lane; the two lane halves hold the
// two halves of the sources and are added by ONE cross-lane shuffle at the end (the
// wave-reduce of the inner sum).  A wave owns TT target tiles (32 TT targets); a
// workgroup of 4 waves shares source stages of ST tiles, pre-packed as LDS images and
// copied by LDS-DMA (global_load_lds), double buffered, one barrier per stage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_mfma.hpp"  // bf16x8, f32x16, acc_row, kexp2 (via kmvp_lowd.hpp)

namespace kmvp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FAST_TILE = 32;   // sources per tile = targets per result column block
// source tiles per LDS stage: four while a stage stays below ~40 KiB (K <= 144), two beyond
__host__ __device__ constexpr int fast_stage_tiles(int KS) { return KS <= 9 ? 4 : 2; }

__host__ __device__ constexpr int fast_ksteps(int D) { return (6 * D + 6 + 15) / 16; }
__host__ __device__ constexpr int fast_row_bytes(int KS) { return KS * 32 + 16; }  // padded source row
__host__ __device__ constexpr int fast_tile_bytes(int KS, int EB) {
  return FAST_TILE * fast_row_bytes(KS) + FAST_TILE * 4 * (EB > 0 ? EB : 0);
}
__host__ __device__ constexpr int fast_stage_bytes(int KS, int EB) {
  return (fast_stage_tiles(KS) * fast_tile_bytes(KS, EB) + 4095) / 4096 * 4096;
}

__host__ __device__ constexpr int fast_target_row(int D) { return (D + 1 + 3) / 4 * 4; }  // floats per target

struct FastArgs {
  const float* xr;           // targets [n_pad][fast_target_row(D)]: centred scaled coordinates, then |x'|^2
  const unsigned char* img;  // source stages [m_stages][stage_bytes]
  double* part;              // partial sums [segments][NE][n_pad]
  int64_t n_pad;
  int64_t m_tiles;           // source tiles of 32
  int64_t m_stages;          // source stages of fast_stage_tiles(KS) tiles
  int64_t seg_stages;        // stages per segment
  int segments;
  int tile_blocks;
  int chunk_stages;          // stages per fp32 chunk
  int64_t j_offset;
  int64_t m_total;
  int same_points;           // targets are the (unsharded) sources: exp(-r) then gives its own pair k = 1 exactly (s = 0 by
                             // construction in the reference's expanded form; here s carries ~1e-7 R^2 of rounding and
                             // sqrt turns that into 3e-4 R)
};

template <int KERNEL>
__device__ __forceinline__ float fast_kval(float s) {
  if constexpr (KERNEL == K_GAUSSIAN) {
    return kexp2(-s);
  } else if constexpr (KERNEL == K_ABSEXP) {
    return kexp2(-__builtin_amdgcn_sqrtf(__builtin_fabsf(s)));  // |s|: free modifier; s<0 only by rounding
  } else {
    return __builtin_amdgcn_rsqf(__builtin_fabsf(s));
  }
}

// bf16 pieces of an fp32 value as fp32 numbers: v == hi + mid + lo exactly
__device__ __forceinline__ void fast_split3f(float v, float& hi, float& mid, float& lo) {
  hi = (float)(__bf16)v;
  const float r1 = v - hi;
  mid = (float)(__bf16)r1;
  lo = (float)(__bf16)(r1 - mid);
}

// element k of a target's augmented row: per d (x_h, x_m, x_h, x_l, x_m, x_h), then 1, 1, 1,
// |x'|^2 h, m, l, then zeros.  hi/mid/lo[D] hold the split of |x'|^2.
template <int D, int K>
__device__ __forceinline__ float fast_target_elem(const float (&hi)[D + 1], const float (&mid)[D + 1],
                                                  const float (&lo)[D + 1]) {
  if constexpr (K < 6 * D) {
    constexpr int d = K / 6, role = K % 6;
    if constexpr (role == 0 || role == 2 || role == 5) return hi[d];
    if constexpr (role == 1 || role == 4) return mid[d];
    return lo[d];
  } else if constexpr (K < 6 * D + 3) {
    return 1.f;
  } else if constexpr (K == 6 * D + 3) {
    return hi[D];
  } else if constexpr (K == 6 * D + 4) {
    return mid[D];
  } else if constexpr (K == 6 * D + 5) {
    return lo[D];
  } else {
    return 0.f;
  }
}

template <int D, int KS, int J>
__device__ __forceinline__ void fast_target_fill(bf16x8& out, int h, const float (&hi)[D + 1],
                                                 const float (&mid)[D + 1], const float (&lo)[D + 1]) {
  if constexpr (J < 8) {
    const float v0 = fast_target_elem<D, KS * 16 + J>(hi, mid, lo);      // lane half 0: k = 16 ks + j
    const float v1 = fast_target_elem<D, KS * 16 + 8 + J>(hi, mid, lo);  // lane half 1: k = 16 ks + 8 + j
    out[J] = (__bf16)(h ? v1 : v0);
    fast_target_fill<D, KS, J + 1>(out, h, hi, mid, lo);
  }
}

template <int D, int KS>
__device__ __forceinline__ void fast_target_operand(bf16x8 (&xb)[fast_ksteps(D)], int h, const float (&hi)[D + 1],
                                                    const float (&mid)[D + 1], const float (&lo)[D + 1]) {
  if constexpr (KS < fast_ksteps(D)) {
    fast_target_fill<D, KS, 0>(xb[KS], h, hi, mid, lo);
    fast_target_operand<D, KS + 1>(xb, h, hi, mid, lo);
  }
}

template <int KERNEL, int D, int SIG, int TT>
__global__ void __launch_bounds__(BLOCK_THREADS) fast_kernel(const FastArgs a) {
  constexpr int KS = fast_ksteps(D);
  constexpr int RD = fast_target_row(D);
  constexpr int EB = (SIG == SIG_DENSITY) ? 0 : 1;
  constexpr int NE = (SIG == SIG_NORM) ? 2 : 1;
  constexpr int RB = fast_row_bytes(KS);
  constexpr int TB = fast_tile_bytes(KS, EB);
  constexpr int SB = fast_stage_bytes(KS, EB);
  constexpr int PIECES = SB / (16 * BLOCK_THREADS);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SB];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t tile0 = ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;  // first target tile of the wave

  // The B operand of this lane's targets is built in registers from D + 1 floats per target
  // (rounded up to whole float4s); a pre-packed operand array would be 16 KS bf16 per target, re-read once per source
  // segment (at the headline shape 3 GB of L2 fills per launch instead of 0.8 GB).
  bf16x8 xb[TT][KS];
  int64_t jz[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const float* row = a.xr + ((tile0 + tt) * FAST_TILE + r) * RD;
    float v[RD];
#pragma unroll
    for (int q = 0; q < RD / 4; ++q) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(row + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * q + j] = w[j];
    }
    float hi[D + 1], mid[D + 1], lo[D + 1];
#pragma unroll
    for (int d = 0; d <= D; ++d) fast_split3f(v[d], hi[d], mid[d], lo[d]);
    fast_target_operand<D, 0>(xb[tt], h, hi, mid, lo);
    if (KERNEL == K_INVDIST || (KERNEL == K_ABSEXP && a.same_points)) {
      const int64_t g = ((tile0 + tt) * FAST_TILE + r) % (a.m_total + 1);  // (same points: the target's own index)
      jz[tt] = (g < a.m_total) ? g - a.j_offset : (int64_t)-1;
    } else {
      jz[tt] = -1;
    }
  }
  int64_t jz_lo = 0, jz_hi = -1;
  if (KERNEL == K_INVDIST || (KERNEL == K_ABSEXP && a.same_points)) {
    const int64_t g_lo = (tile0 * FAST_TILE) % (a.m_total + 1);
    const int64_t g_hi = g_lo + (FAST_TILE * TT - 1);
    if (g_hi <= a.m_total) {
      jz_lo = g_lo - a.j_offset;
      jz_hi = g_hi - a.j_offset;
    } else {
      jz_lo = INT64_MIN / 2;
      jz_hi = INT64_MAX / 2;
    }
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

  int in_chunk = 0;
  for (int64_t s = s_begin; s < s_end; ++s) {
    const int buf = (int)((s - s_begin) & 1);
    if (s + 1 < s_end) stage_in(s + 1, buf ^ 1);
#pragma unroll 1
    for (int q = 0; q < fast_stage_tiles(KS); ++q) {
      const unsigned char* lt = &lds[buf][q * TB];
      const int64_t t = s * fast_stage_tiles(KS) + q;  // source tile index (tiles beyond m_tiles are all-pad)
      // A fragments of the tile (row r, 8 consecutive k per k-step) and the 16 signal values
      // of this lane's rows: registers 4g..4g+3 hold rows 8g+4h .. 8g+4h+3
      bf16x8 ya[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        ya[ks] = *reinterpret_cast<const bf16x8*>(lt + r * RB + (ks * 16 + 8 * h) * 2);
      float bv[16];
      if constexpr (EB > 0) {
        const float* lb = reinterpret_cast<const float*>(lt + FAST_TILE * RB);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(lb + 8 * g + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[4 * g + j] = v[j];
        }
      }
      bool check = false;
      if constexpr (KERNEL == K_INVDIST || KERNEL == K_ABSEXP)
        check = (t * FAST_TILE + FAST_TILE - 1 >= jz_lo) && (t * FAST_TILE <= jz_hi);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        f32x16 d;
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) d[qq] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[ks], xb[tt][ks], d, 0, 0, 0);
        float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;  // two chains each for numerator / denominator
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float k = fast_kval<KERNEL>(d[reg]);
          if constexpr (KERNEL == K_INVDIST) {
            if (check) k = (t * FAST_TILE + acc_row(reg, h) == jz[tt]) ? 0.f : k;
          }
          if constexpr (KERNEL == K_ABSEXP) {  // (same points) the target's own pair: r = 0
            if (check) k = (t * FAST_TILE + acc_row(reg, h) == jz[tt]) ? 1.f : k;
          }
          if constexpr (SIG == SIG_DENSITY) {
            if (reg & 1) p1 += k; else p0 += k;
          } else {
            if (reg & 1) p1 = fmaf(k, bv[reg], p1); else p0 = fmaf(k, bv[reg], p0);
            if constexpr (SIG == SIG_NORM) {
              if (reg & 1) q1 += k; else q0 += k;
            }
          }
        }
        acc[tt][0] += p0 + p1;
        if constexpr (SIG == SIG_NORM) acc[tt][1] += q0 + q1;
      }
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

  // fold the last chunk, add the two lane halves (sources 4h.. of every 8), write
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double v = accd[tt][e] + (double)acc[tt][e];
      v += __shfl_xor(v, 32);
      if (h == 0) a.part[((int64_t)seg * NE + e) * a.n_pad + (tile0 + tt) * FAST_TILE + r] = v;
    }
}

}  // namespace kmvp
