// Low-dimensional pair loop with the squared distance on the bf16 matrix cores, accurate for
// EVERY kernel: the expansion is done around a centre that moves with the source tile.
//
// fast_kernel (kmvp_fast.hpp) expands s = |x|^2 + |y|^2 - 2 x.y around ONE centre for the
// whole cloud: the absolute error of s is eps32 * (|x'|^2 + |y'|^2), fine for the Gaussian
// on a compact cloud, not for 1/sqrt(s) or exp(-sqrt(s)), whose values hinge on the RELATIVE
// accuracy of small s.  Here
//   * the sources are sorted along a Morton curve once per set_points (hipcub radix sort),
//     so that 128 consecutive sources form a spatially compact group with centre c_g;
//   * a group's rows are stored relative to c_g (y' = (y - c_g) * kernel constant, |y'| <= R_g,
//     split three ways into bf16 as in kmvp_fast.hpp); coordinates themselves are never
//     centred or scaled, so clusters far from each other or from the origin lose nothing;
//   * the TARGET operand is rebuilt on the fly for every (target tile, group):
//     x'' = (x - c_g) * kernel constant, split three ways, |x''|^2 split three ways.
// Then |x''|^2 ~ s and the error of s is eps32 * (s + 2 R_g sqrt(s) + 2 R_g^2): RELATIVE
// accuracy ~ eps32 for every pair farther apart than the group radius.  Pairs closer than
// that (s < tau_g = kappa R_g^2; a fraction ~1e-6 of all pairs, plus the diagonal) are
// recomputed exactly in a wave-uniform rare branch from the fp32 coordinates kept in the
// group image -- which is also where the inverse-distance zero rule (bruteforce.py:13-14,
// on ORIGINAL source indices) and an exact s = 0 on the diagonal are applied.
//
// VALU cost per pair: transcendental(s) + 1 FMA + ~0.7 (operand rebuild and reach flag,
// amortised over the 128 sources of a group) ~ 5.7 issue slots against 11-12 for the difference
// form.
//
// Build note: the library is compiled with -fno-slp-vectorize (Makefile).  With LLVM's SLP
// vectoriser packing the two tiles' fp32 chains into v_pk_* pairs, the reach-gated version of
// this kernel returned wrong, run-to-run varying sums for the ODD target tiles of a wave (the
// second vector element) as soon as two tiles were gated; without SLP it is correct, bitwise
// reproducible and faster (tests: test_matrix_core_kernels_reproducible_and_tile_count_independent).  K layout (D <= 4; lane half h owns dimensions h and h + 2):
//   k 0..7   : dim 0 six partial products, |y'|^2_h * 1, |y'|^2_m * 1
//   k 8..15  : dim 1 six partial products, |y'|^2_l * 1, 1 * |x''|^2_h
//   k 16..23 : dim 2 six partial products, 1 * |x''|^2_m, 1 * |x''|^2_l
//   k 24..31 : dim 3 six partial products, 0, 0
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_mfma.hpp"  // bf16x8, f32x16, acc_row, kexp2

namespace kmvp {

typedef float cf32x4 __attribute__((ext_vector_type(4)));

constexpr int CF_GROUP = 128;      // sources per centre group (four MFMA row tiles)
constexpr int CF_STAGE_GROUPS = 1; // groups per LDS stage
constexpr int CF_ROW_BYTES = 80;   // 32 bf16 + 16 bytes pad (conflict-free ds_read_b128)
// group image: [centre c0..c3][tau, pad x3][64 rows x 80 B][64 signal floats][64 x float4 raw
// coordinates][64 x int32 GLOBAL original source index, -1 for pad]
constexpr int CF_HDR = 32;
constexpr int CF_OFF_B = CF_HDR + CF_GROUP * CF_ROW_BYTES;
constexpr int CF_OFF_RAW = CF_OFF_B + CF_GROUP * 4;
constexpr int CF_OFF_IDX = CF_OFF_RAW + CF_GROUP * 16;
constexpr int CF_GROUP_BYTES = CF_OFF_IDX + CF_GROUP * 4;
constexpr int CF_STAGE_BYTES = (CF_STAGE_GROUPS * CF_GROUP_BYTES + 4095) / 4096 * 4096;
constexpr float CF_KAPPA = 0.03f;  // tau_g = kappa * R_g^2: below it a pair is recomputed exactly

struct CfastArgs {
  const float* xraw;         // targets [n_pad][4]: the caller's fp32 coordinates, untouched; unused dims 0
  const unsigned char* img;  // source stages [m_stages][CF_STAGE_BYTES]
  double* part;              // partial sums [segments][NE][n_pad]
  int64_t n_pad;
  int64_t m_stages;
  int64_t seg_stages;
  int segments;
  int tile_blocks;
  int chunk_stages;
  int64_t j_offset;
  int64_t m_total;
  float scale;               // the kernel's constant, applied AFTER a difference is formed
};

template <int KERNEL>
__device__ __forceinline__ float cf_kval(float s) {
  if constexpr (KERNEL == K_GAUSSIAN) {
    return kexp2(-s);
  } else if constexpr (KERNEL == K_ABSEXP) {
    return kexp2(-__builtin_amdgcn_sqrtf(__builtin_fabsf(s)));
  } else {
    return __builtin_amdgcn_rsqf(__builtin_fabsf(s));
  }
}

// v -> (hi, mid, lo) as fp32 values that are exactly representable in bf16, hi + mid + lo == v
__device__ __forceinline__ void cf_split(float v, float& hi, float& mid, float& lo) {
  hi = (float)(__bf16)v;
  const float r1 = v - hi;
  mid = (float)(__bf16)r1;
  lo = (float)(__bf16)(r1 - mid);
}

template <int KERNEL, int SIG, int TT>
__global__ void __launch_bounds__(BLOCK_THREADS) cfast_kernel(const CfastArgs a) {
  constexpr int EB = (SIG == SIG_DENSITY) ? 0 : 1;
  constexpr int NE = (SIG == SIG_NORM) ? 2 : 1;
  constexpr int PIECES = CF_STAGE_BYTES / (16 * BLOCK_THREADS);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][CF_STAGE_BYTES];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t tile0 = ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;

  // the lane's targets (both lane halves hold the same target r of each tile)
  float x[TT][4];
  int jz[TT];  // GLOBAL source index whose pair the target drops (inverse-distance, same points)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int64_t i = (tile0 + tt) * 32 + r;
    // (non-temporal: a workgroup's targets and partial sums are touched once, the 2 MiB source segment its XCD streams is
    // re-read by every workgroup -- they must not push it out of the 4 MiB L2: profiles/r03_c4shard_segments_fetch.txt)
    const cf32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cf32x4*>(a.xraw + i * 4));
#pragma unroll
    for (int d = 0; d < 4; ++d) x[tt][d] = v[d];
    if constexpr (KERNEL == K_INVDIST) {
      const int64_t g = i % (a.m_total + 1);
      jz[tt] = (g < a.m_total) ? (int)g : -2;
    } else {
      jz[tt] = -2;
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
    const unsigned char* src = a.img + s * CF_STAGE_BYTES;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int piece = (p * WAVES_PER_BLOCK + wave) * 1024;
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
    for (int g = 0; g < CF_STAGE_GROUPS; ++g) {
      const unsigned char* lg = &lds[buf][g * CF_GROUP_BYTES];
      const cf32x4 cen = *reinterpret_cast<const cf32x4*>(lg);  // wave-uniform broadcast
      const float tau = *reinterpret_cast<const float*>(lg + 16);
      const float reach2 = *reinterpret_cast<const float*>(lg + 20);  // (R_g + sqrt(tau_g))^2
      // bit tt: some target of tile tt lies within reach of the group
      unsigned near_mask = 0;
      const unsigned char* lrows = lg + CF_HDR;
      const float* lb = reinterpret_cast<const float*>(lg + CF_OFF_B);
      const cf32x4* lraw = reinterpret_cast<const cf32x4*>(lg + CF_OFF_RAW);
      const int* lidx = reinterpret_cast<const int*>(lg + CF_OFF_IDX);

      // ---- target operands relative to the group's centre, one pair of fragments per target
      // tile (this lane half: dims h and h + 2); reused by all row tiles of the group
      bf16x8 xb[TT][2];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        // differences of the caller's coordinates first, the kernel's constant afterwards: a
        // difference of nearby fp32 numbers is (nearly) exact, whatever their distance from the
        // origin; scaling or centring the cloud beforehand would round every coordinate
        const float x0 = (x[tt][0] - cen[0]) * a.scale, x1 = (x[tt][1] - cen[1]) * a.scale;
        const float x2 = (x[tt][2] - cen[2]) * a.scale, x3 = (x[tt][3] - cen[3]) * a.scale;
        const float sq = fmaf(x3, x3, fmaf(x2, x2, fmaf(x1, x1, x0 * x0)));
        const float xa = h ? x1 : x0, xc = h ? x3 : x2;
        float ah, am, al, ch, cm, cl, sh, sm, sl;
        cf_split(xa, ah, am, al);
        cf_split(xc, ch, cm, cl);
        cf_split(sq, sh, sm, sl);
        // ks 0: (a_h, a_m, a_h, a_l, a_m, a_h, e0, e1), e = (1, 1) for h = 0, (1, |x''|^2_h) for h = 1
        // ks 1: (c_h, c_m, c_h, c_l, c_m, c_h, f0, f1), f = (|x''|^2_m, |x''|^2_l) for h = 0, (0, 0) for h = 1
        bf16x8 b0, b1;
        b0[0] = (__bf16)ah; b0[1] = (__bf16)am; b0[2] = (__bf16)ah; b0[3] = (__bf16)al;
        b0[4] = (__bf16)am; b0[5] = (__bf16)ah; b0[6] = (__bf16)1.f; b0[7] = (__bf16)(h ? sh : 1.f);
        b1[0] = (__bf16)ch; b1[1] = (__bf16)cm; b1[2] = (__bf16)ch; b1[3] = (__bf16)cl;
        b1[4] = (__bf16)cm; b1[5] = (__bf16)ch; b1[6] = (__bf16)(h ? 0.f : sm); b1[7] = (__bf16)(h ? 0.f : sl);
        xb[tt][0] = b0;
        xb[tt][1] = b1;
        // (finite and beyond reach, or the tile is searched: NaN / inf from non-finite coordinates
        // must reach the exact branch, where they behave as in the reference)
        near_mask |= (__ballot(!(sq > reach2 && sq <= 3.0e38f)) != 0ull ? 1u : 0u) << tt;
      }

#pragma unroll 1
      for (int rt = 0; rt < CF_GROUP / 32; ++rt) {
        // source operands of this row tile (32 sources), shared by the wave's target tiles
        bf16x8 ya[2];
        float bv[16];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          ya[ks] = *reinterpret_cast<const bf16x8*>(lrows + (rt * 32 + r) * CF_ROW_BYTES + (ks * 16 + 8 * h) * 2);
        if constexpr (EB > 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const cf32x4 v = *reinterpret_cast<const cf32x4*>(lb + rt * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[4 * q + j] = v[j];
          }
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          f32x16 d;
#pragma unroll
          for (int q = 0; q < 16; ++q) d[q] = 0.f;
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[0], xb[tt][0], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[1], xb[tt][1], d, 0, 0, 0);

          // ---- rare branch.  A pair closer than sqrt(tau) needs its target within R_g + sqrt(tau) of
          // the group's centre: the wave-uniform flag of this (target tile, group) skips the search
          // for almost all tiles; where it is set, the minimum over the tile decides (wave-uniform
          // again).  Such pairs get the exact difference form (bruteforce.py:53-54) from the fp32
          // coordinates; for 1/r the pair that carries the target's own index gets s = +inf, i.e.
          // k = 0 (bruteforce.py:13-14).
          const bool gate = (near_mask >> tt) & 1u;
          float dmin = INFINITY;
          if (gate) {
            dmin = fminf(fminf(d[0], d[1]), d[2]);
#pragma unroll
            for (int q = 3; q < 15; q += 2) dmin = fminf(fminf(dmin, d[q]), d[q + 1]);
            dmin = fminf(dmin, d[15]);
          }
          // (negated comparisons: a NaN -- non-finite coordinates poison a whole group's centre --
          // also takes the exact branch, where inf - x gives s = inf, k = 0 as in the reference)
          if (gate && __any(!(dmin > tau))) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              if (!(d[q] > tau)) {
                const int row = rt * 32 + acc_row(q, h);
                const cf32x4 yr = lraw[row];
                const float e0 = (x[tt][0] - yr[0]) * a.scale, e1 = (x[tt][1] - yr[1]) * a.scale;
                const float e2 = (x[tt][2] - yr[2]) * a.scale, e3 = (x[tt][3] - yr[3]) * a.scale;
                float sx = fmaf(e3, e3, fmaf(e2, e2, fmaf(e1, e1, e0 * e0)));
                if constexpr (KERNEL == K_INVDIST) {
                  if (lidx[row] == jz[tt]) sx = INFINITY;
                }
                d[q] = sx;
              }
            }
          }
          float p0 = 0.f, p1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float k = cf_kval<KERNEL>(d[q]);
            if constexpr (SIG == SIG_DENSITY) {
              if (q & 1) p1 += k; else p0 += k;
            } else {
              if (q & 1) p1 = fmaf(k, bv[q], p1); else p0 = fmaf(k, bv[q], p0);
              if constexpr (SIG == SIG_NORM) {
                if (q & 1) q1 += k; else q0 += k;
              }
            }
          }
          acc[tt][0] += p0 + p1;
          if constexpr (SIG == SIG_NORM) acc[tt][1] += q0 + q1;
        }
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
    __syncthreads();
  }

#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double v = accd[tt][e] + (double)acc[tt][e];
      v += __shfl_xor(v, 32);
      if (h == 0) __builtin_nontemporal_store(v, &a.part[((int64_t)seg * NE + e) * a.n_pad + (tile0 + tt) * 32 + r]);
    }
}

}  // namespace kmvp
