// Several signal columns for exp(-r) -- and for the Gaussian on clouds outside fast_kernel's radius rule: cfast_kernel's
// squared distances (kmvp_cfast.hpp: expansion around per-group centres of Morton-sorted sources, exact recomputation of
// the closest pairs, RELATIVE accuracy in s) feeding fastmm_kernel's second product (kmvp_fastmm.hpp: the tile of kernel
// values split into two f16 pieces and multiplied with the pre-packed signal operand by f16 MFMAs, up to 32 columns per
// pass).  float32, D <= 4.
//
// 1/r (KERNEL = K_INVDIST, always with the online shift; needs targets == sources, as cfast_kernel's zero rule does): the
// values are unbounded, so the per-target shift kop = floor(log2 r_min) of "The shift" (kmvp_fastmm.hpp) is what makes the
// f16 pieces possible at all: T = 2^(15 + kop) / r <= 2^15.5.  It costs nothing per pair: s is linear in the TARGET operand
// (every term of |x''|^2 + |y'|^2 - 2 x''.y' carries exactly one target-side factor), the operand is rebuilt per (target
// tile, group) anyway, so it is built times c = 4^-(15 + kop) -- a power of two, exact in the bf16 pieces -- and
// T = rsq(c s) comes out of the one transcendental.  The pair that carries the target's own ORIGINAL index is dropped in
// the exact branch (s = +inf: bruteforce.py:13-14), a coincident other pair gives T = inf -> NaN sums for that target,
// as the reference's inf / nan row.
//
//   per (target tile, group of 128 sources): target operand relative to the group's centre (as cfast_kernel)
//   per row tile of 32 sources:  S = Y~ X~^T (2 bf16 MFMAs)  ->  [near pairs: exact s]  ->  T = 2^15 k(s)
//                                ->  T_h + T_l (f16)  ->  O += B'^T T (f16 MFMAs)
//
// Group image (one per LDS stage): [header 32 B: centre, tau, reach^2][128 rows x 80 B][4 row tiles x signal operands
// (2048 B in MODE 0, 4096 B in MODE 1; layout of pack_fastmm_signal_kernel)][128 x float4 raw coordinates].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_cfast.hpp"
#include "kmvp_fastmm.hpp"

namespace kmvp {

constexpr int CFM_OFF_SIG = CF_HDR + CF_GROUP * CF_ROW_BYTES;
__host__ __device__ constexpr int cfm_off_raw(int MODE) { return CFM_OFF_SIG + (CF_GROUP / 32) * fmm_sig_bytes(MODE); }
__host__ __device__ constexpr int cfm_off_idx(int MODE) { return cfm_off_raw(MODE) + CF_GROUP * 16; }  // int32 GLOBAL original source index, -1: pad
__host__ __device__ constexpr int cfm_group_bytes(int MODE) { return cfm_off_idx(MODE) + CF_GROUP * 4; }
__host__ __device__ constexpr int cfm_stage_bytes(int MODE) { return (cfm_group_bytes(MODE) + 4095) / 4096 * 4096; }

struct CfastmmArgs {
  const float* xraw;         // targets [n_pad][4]: the caller's fp32 coordinates (pack_cfast_targets_kernel)
  const unsigned char* img;  // source stages [m_stages][cfm_stage_bytes(MODE)]
  const double* unscale;     // [32]: 2^-15 / sigma_e per column
  double* part;              // partial sums [segments][NE][n_pad]
  int64_t n_pad;
  int64_t m_stages;
  int64_t seg_stages;
  int segments;
  int tile_blocks;
  int chunk_stages;
  int NE;
  float scale;               // the kernel's constant, applied AFTER a difference is formed
  int64_t m_total;           // 1/r: the zero rule works on global indices (bruteforce.py:13-14)
};

// T = 2^sh k(s), sh = FMM_SHIFT [+ the target's online shift]
template <int KERNEL>
__device__ __forceinline__ float cfm_tval(float s, float sh) {
  if constexpr (KERNEL == K_GAUSSIAN) return kexp2(sh - s);
  else if constexpr (KERNEL == K_ABSEXP) return kexp2(sh - __builtin_amdgcn_sqrtf(__builtin_fabsf(s)));
  else return __builtin_amdgcn_rsqf(__builtin_fabsf(s));  // 1/r: s arrives scaled by 4^-sh (see the header)
}

// ONLINE = 1: per-target running shift kop = floor(smallest exponent seen so far), T = 2^(15 + kop) k -- see "The shift" in
// kmvp_fastmm.hpp.  Here the shift is a per-lane constant of the VALU subtraction that was there anyway; the pair loop
// pays the running minimum of the tile (8 v_min3 + one v_permlane32_swap) and a rare wave-uniform rescale branch.
template <int KERNEL, int MODE, int TT, int ONLINE = 0>
__global__ void __launch_bounds__(BLOCK_THREADS) cfastmm_kernel(const CfastmmArgs a) {
  static_assert(KERNEL != K_INVDIST || ONLINE == 1, "1/r needs the per-target shift");
  constexpr int SB = cfm_stage_bytes(MODE);
  constexpr int PIECES = SB / (16 * BLOCK_THREADS);
  constexpr int NOUT = MODE ? 16 : 8;
  static_assert(SB % (16 * BLOCK_THREADS) == 0, "stage = whole LDS-DMA pieces");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SB];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t tile0 = ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;

  float x[TT][4];
  int jz[TT];      // 1/r: GLOBAL source index whose pair the target drops
  float csc[TT];   // 1/r: c = 4^-(15 + kop), the scale of the target operand (and so of s)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int64_t i = (tile0 + tt) * 32 + r;
    const cf32x4 v = *reinterpret_cast<const cf32x4*>(a.xraw + i * 4);
#pragma unroll
    for (int d = 0; d < 4; ++d) x[tt][d] = v[d];
    const int64_t g = i % (a.m_total + 1);
    jz[tt] = (KERNEL == K_INVDIST && g < a.m_total) ? (int)g : -2;
    csc[tt] = KERNEL == K_INVDIST ? 9.313225746154785e-10f : 1.f;  // 4^-15
  }

  f32x16 acc[TT];
  double accd[TT][NOUT];
  float kop[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    kop[tt] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[tt][q] = 0.f;
#pragma unroll
    for (int q = 0; q < NOUT; ++q) accd[tt][q] = 0.0;
  }
  auto fold_one = [&](int tt) {  // as in fastmm_kernel
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
      if constexpr (MODE == 0) accd[tt][q] += (double)acc[tt][q] + (double)acc[tt][q + 8];
      else accd[tt][q] += (double)acc[tt][q];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[tt][q] = 0.f;
  };
  auto fold = [&]() {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) fold_one(tt);
  };

  const int64_t s_begin = (int64_t)seg * a.seg_stages;
  int64_t s_end = s_begin + a.seg_stages;
  if (s_end > a.m_stages) s_end = a.m_stages;

  auto stage_in = [&](int64_t s, int buf) {
    const unsigned char* src = a.img + s * SB;
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
    const unsigned char* lg = &lds[buf][0];
    const cf32x4 cen = *reinterpret_cast<const cf32x4*>(lg);  // wave-uniform broadcast
    const float tau = *reinterpret_cast<const float*>(lg + 16);
    const float reach2 = *reinterpret_cast<const float*>(lg + 20);  // (R_g + sqrt(tau_g))^2
    unsigned near_mask = 0;  // bit tt: some target of tile tt lies within reach of the group
    const unsigned char* lrows = lg + CF_HDR;
    const cf32x4* lraw = reinterpret_cast<const cf32x4*>(lg + cfm_off_raw(MODE));

    // target operands relative to the group's centre (kmvp_cfast.hpp): this lane half holds dims h and h + 2
    bf16x8 xb[TT][2];
    // (c: 1, or for 1/r the target's power-of-two scale -- every piece is scaled exactly, and so is s)
    auto build_xb = [&](int tt, float c) -> float {
      const float x0 = (x[tt][0] - cen[0]) * a.scale, x1 = (x[tt][1] - cen[1]) * a.scale;
      const float x2 = (x[tt][2] - cen[2]) * a.scale, x3 = (x[tt][3] - cen[3]) * a.scale;
      const float sq = fmaf(x3, x3, fmaf(x2, x2, fmaf(x1, x1, x0 * x0)));
      const float xa = (h ? x1 : x0) * c, xc = (h ? x3 : x2) * c;
      float ah, am, al, ch, cm, cl, sh, sm, sl;
      cf_split(xa, ah, am, al);
      cf_split(xc, ch, cm, cl);
      cf_split(sq * c, sh, sm, sl);
      bf16x8 b0, b1;
      b0[0] = (__bf16)ah; b0[1] = (__bf16)am; b0[2] = (__bf16)ah; b0[3] = (__bf16)al;
      b0[4] = (__bf16)am; b0[5] = (__bf16)ah; b0[6] = (__bf16)c; b0[7] = (__bf16)(h ? sh : c);
      b1[0] = (__bf16)ch; b1[1] = (__bf16)cm; b1[2] = (__bf16)ch; b1[3] = (__bf16)cl;
      b1[4] = (__bf16)cm; b1[5] = (__bf16)ch; b1[6] = (__bf16)(h ? 0.f : sm); b1[7] = (__bf16)(h ? 0.f : sl);
      xb[tt][0] = b0;
      xb[tt][1] = b1;
      return sq;
    };
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float sq = build_xb(tt, csc[tt]);
      near_mask |= (__ballot(!(sq > reach2 && sq <= 3.0e38f)) != 0ull ? 1u : 0u) << tt;
    }
    const int* lidx = reinterpret_cast<const int*>(lg + cfm_off_idx(MODE));

#pragma unroll 1
    for (int rt = 0; rt < CF_GROUP / 32; ++rt) {
      bf16x8 ya[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        ya[ks] = *reinterpret_cast<const bf16x8*>(lrows + (rt * 32 + r) * CF_ROW_BYTES + (ks * 16 + 8 * h) * 2);
      const unsigned char* ls = lg + CFM_OFF_SIG + rt * fmm_sig_bytes(MODE);
      h16x8 ah[2], al[2];
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        ah[g2] = *reinterpret_cast<const h16x8*>(ls + g2 * 1024 + lane * 16);
        if constexpr (MODE == 1) al[g2] = *reinterpret_cast<const h16x8*>(ls + 2048 + g2 * 1024 + lane * 16);
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        f32x16 d;
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0.f;
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[0], xb[tt][0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[1], xb[tt][1], d, 0, 0, 0);

        // rare branch (kmvp_cfast.hpp): pairs closer than sqrt(tau) get the exact difference form
        const bool gate = (near_mask >> tt) & 1u;
        float dmin = INFINITY;
        if (gate || ONLINE) {
          dmin = fminf(fminf(d[0], d[1]), d[2]);
#pragma unroll
          for (int q = 3; q < 15; q += 2) dmin = fminf(fminf(dmin, d[q]), d[q + 1]);
          dmin = fminf(dmin, d[15]);
        }
        const float tau_c = tau * csc[tt];  // (d is c s)
        if (gate && __any(!(dmin > tau_c))) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if (!(d[q] > tau_c)) {
              const int row = rt * 32 + acc_row(q, h);
              const cf32x4 yr = lraw[row];
              const float e0 = (x[tt][0] - yr[0]) * a.scale, e1 = (x[tt][1] - yr[1]) * a.scale;
              const float e2 = (x[tt][2] - yr[2]) * a.scale, e3 = (x[tt][3] - yr[3]) * a.scale;
              float sx = fmaf(e3, e3, fmaf(e2, e2, fmaf(e1, e1, e0 * e0))) * csc[tt];
              if constexpr (KERNEL == K_INVDIST) {
                if (lidx[row] == jz[tt]) sx = INFINITY;  // the target's own index: k = 0 (bruteforce.py:13-14)
              }
              d[q] = sx;
            }
          }
          if constexpr (ONLINE) {  // the minimum the shift follows must not see the dropped pair
            dmin = fminf(fminf(d[0], d[1]), d[2]);
#pragma unroll
            for (int q = 3; q < 15; q += 2) dmin = fminf(fminf(dmin, d[q]), d[q + 1]);
            dmin = fminf(dmin, d[15]);
          }
        }
        if constexpr (ONLINE) {
          // smallest exponent of the tile for this target (both lane halves): s for the Gaussian, r for exp(-r); the
          // exactly recomputed pairs are >= 0 and at most tau away from what dmin saw
          float m = fmm_min_halves(dmin);
          const bool first = s == s_begin && rt == 0;
          if constexpr (KERNEL == K_INVDIST) {
            // m = c s_min; the tile's largest T is rsq(m): 2^15.5 at m = 2^-31.  New shift from the exponent of the
            // UNSCALED s_min = m 4^(15 + kop) = f 2^e, f in [0.5, 1): kop = floor((e - 1) / 2) <= log2 r_min.
            // m == 0 (a coincident pair that is not the target's own): nothing to scale to, T = inf, the row goes NaN.
            const bool need = (first || m < 4.656612873077393e-10f) && m > 0.f && m < 3.0e38f;
            if (__any(need)) {
              fold_one(tt);
              if (need) {
                int e;
                (void)frexpf(m, &e);
                const int kn = ((e - 1) >> 1) + 15 + (int)kop[tt];  // floor((e_unscaled - 1) / 2), e_unscaled = e + 2 (15 + kop)
                const int di = (int)kop[tt] - kn;
                const float f = ldexpf(1.f, 2 * di);            // c_new / c_old = 4^(kop_old - kn)
#pragma unroll
                for (int q = 0; q < 16; ++q) d[q] *= f;
#pragma unroll
                for (int q = 0; q < NOUT; ++q) accd[tt][q] = ldexp(accd[tt][q], -di);
                kop[tt] = (float)kn;
                csc[tt] *= f;
              }
              (void)build_xb(tt, csc[tt]);  // the remaining row tiles of this group
            }
          } else {
            if constexpr (KERNEL == K_ABSEXP) m = __builtin_amdgcn_sqrtf(fmaxf(m, 0.f));
            else m = fmaxf(m, 0.f);
            const bool need = (first || m < kop[tt] - 0.5f) && m < 3.0e38f;  // (hysteresis: T up to 2^15.5 is still an f16 number)
            if (__any(need)) {  // rare
              fold_one(tt);
              if (need) {
                const float kn = floorf(m);
                const int di = (int)(kop[tt] - kn);  // >= 0 except at the first tile (nothing accumulated yet)
#pragma unroll
                for (int q = 0; q < NOUT; ++q) accd[tt][q] = ldexp(accd[tt][q], -di);
                kop[tt] = kn;
              }
            }
          }
        }
        const float sh = (float)FMM_SHIFT + kop[tt];
        h16x8 th[2], tl[2];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const fmm_f32x2 t = {cfm_tval<KERNEL>(d[8 * g2 + 2 * i], sh), cfm_tval<KERNEL>(d[8 * g2 + 2 * i + 1], sh)};
            const h16x2 hh = __builtin_convertvector(t, h16x2);
            const fmm_f32x2 rest = {fmm_minus_lo_half(t[0], hh), fmm_minus_hi_half(t[1], hh)};
            const h16x2 ll = __builtin_convertvector(rest, h16x2);
            th[g2][2 * i] = hh[0];
            th[g2][2 * i + 1] = hh[1];
            tl[g2][2 * i] = ll[0];
            tl[g2][2 * i + 1] = ll[1];
          }
        }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g2], th[g2], acc[tt], 0, 0, 0);
          if constexpr (MODE == 1) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[g2], th[g2], acc[tt], 0, 0, 0);
          acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[g2], tl[g2], acc[tt], 0, 0, 0);
        }
      }
    }
    if (++in_chunk == a.chunk_stages) {
      in_chunk = 0;
      fold();
    }
    __syncthreads();
  }
  fold();

#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
      const int e = 8 * (q >> 2) + 4 * h + (q & 3);
      if (e < a.NE) {
        double v = accd[tt][q] * a.unscale[e];
        if constexpr (ONLINE) v = ldexp(v, -(int)kop[tt]);  // the true scale
        a.part[((int64_t)seg * a.NE + e) * a.n_pad + (tile0 + tt) * 32 + r] = v;
      }
    }
}

}  // namespace kmvp
