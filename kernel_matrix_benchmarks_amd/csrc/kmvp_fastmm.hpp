// Gaussian products with SEVERAL signal columns (low-D attention with E value channels, bruteforce.py:142-153;
// exp(<x,y>) attention through its Gaussian form): both matrix products of "K = exp(-sqdist); a = K @ b" on the
// matrix cores, the kernel matrix never leaving the registers.
//
// fast_kernel (kmvp_fast.hpp) takes s = |x'|^2 + |y'|^2 - 2 x'.y' from split-bf16 MFMAs at fp32 accuracy and then
// spends one VALU FMA per pair AND COLUMN on k(s) b_j: with E columns the difference form (lowd, blocks of four
// columns) and the per-column cell forms all scale with E.  Here the tile of kernel values is handed back to the
// matrix pipe:
//
//   S  (32 sources x 32 targets) = Y~ X~^T          KS bf16 MFMAs   (operands as in kmvp_fast.hpp, plus one column
//                                                                    that subtracts FMM_SHIFT: T = 2^15 exp(-s));
//                                                                    D <= 64, KS = ceil((6 D + 9) / 16) <= 25
//   T  = exp2(-S)                                   16 v_exp_f32 per lane: the one transcendental per PAIR
//                                                   (exp(-r), KERNEL = K_ABSEXP: exp2(15 - sqrt(S)) without the shift
//                                                   column, closest pairs recomputed exactly: see the kernel)
//   T  = T_h + T_l  (two f16: 11 + 11 bits)         v_cvt_pk_f16_f32, v_fma_mix_f32 (T - T_h), v_cvt_pk_f16_f32
//   O (32 columns x 32 targets) += B'^T T           f16 MFMAs, fp32 accumulators: up to 32 signal columns at once
//
// The S tile comes out of the MFMA with the TARGET on the lane and 16 sources in the lane's registers, which IS the
// B-operand layout of the second product (k = source) up to a permutation of k that the pre-packed signal operand
// B' follows -- no cross-lane traffic between the two products.  fp32 accuracy needs the signal split as well,
// b sigma_e = b_h + b_l (sigma_e a power of two per column, |b sigma_e| < 2^14):
//   MODE 0 (<= 16 columns): the A operand holds b_h in rows 0..15 and b_l in rows 16..31 -> two MFMAs per 16 sources
//          (with T_h and T_l), rows e and e + 16 added when the accumulator is folded;
//   MODE 1 (<= 32 columns): A_h, A_l separately -> three MFMAs (A_h T_h, A_l T_h, A_h T_l).
// Every 2 x "chunk" sources (option; default 32 source tiles) the fp32 accumulators are folded into fp64 registers (the
// chain of MFMA accumulations stays short), and the partial sums leave as fp64 like everywhere else.
//
// Per 32 x 32 pairs and lane: 16 v_exp_f32 + 16 v_cvt_pk_f16_f32 + 16 v_fma_mix_f32 against 16 (1 + E) FMAs +
// 16 v_exp_f32 of a VALU sum, and KS + 4 (6) MFMAs of 8 passes beside them (331 issue cycles, measured
// 335).  Error: b is carried to 2^-22; a kernel value T = 2^15 k is carried to 2^-22 RELATIVE while T >= 2^-3 and to
// 2^-25 ABSOLUTE below that (T_l goes subnormal), i.e. to 2^-40 of the LARGEST value the shift was chosen for; s as in
// fast_kernel (eps32 (|x'|^2 + |y'|^2), clouds inside the radius rule) -- measured in tests/test_gpu_parity.py.
//
// The shift (ONLINE).  With the fixed shift 2^15 the largest value is k = 1: right when every target has a source at
// distance ~0 (targets == sources), wrong for a target far from every source -- its whole row sits near or below the
// f16 floor (rows 3.8 / 4.5 away came out 5e-2 off / NaN).  ONLINE = 1 carries a per-TARGET integer shift
// kop, the flash-attention recurrence with the running maximum rounded to an integer power of two:
//     T = 2^(15 + kop) k ,   kop = floor(log2 of 1 / (largest k seen so far for this target))
//  * Gaussian / exp(<x,y>): the shift rides in the MFMA operands -- columns 16 KS - 1 and 16 KS - 2 of the target row
//    hold -k_hi and -k_lo (kop = k_hi + k_lo, k_hi a multiple of 128, |k_lo| < 128: both exact in bf16, |kop| up
//    to 32000, i.e. logits up to 2.2e4), the source rows hold 1 there -- so the pair loop pays only the running
//    minimum of S (8 v_min3 + one v_permlane32_swap per tile);
//  * exp(-r): T = exp2(15 + kop - sqrt(S)), the per-lane constant replaces the literal 15 (free);
//  * when a tile would exceed 2^15 (or at a wave's first tile) the fp32 accumulator is folded into the fp64 sums,
//    these are rescaled by 2^(kop_new - kop_old) <= 1 (exact), the tile's S is shifted and the operand patched:
//    a rare wave-uniform branch.  kop only decreases, so nothing ever overflows; what a smaller, earlier value
//    loses is below 2^-40 of the row's largest term.
// The partial sums leave as fp64 at the true scale (x 2^-kop), or -- exp(<x,y>), FastmmArgs::kexp -- at the scale
// 2^-kop together with kop per (segment, target), for a reduction that never forms exp(max logit)
// (reduce_shifted_kernel in kmvp_product.hip: "no range limit" for row-normalised attention).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_fast.hpp"

namespace kmvp {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float fmm_f32x2 __attribute__((ext_vector_type(2)));

constexpr int FMM_MAX_D = 64;  // K = 6 D + 7 <= 391: 25 k-steps of 16 (the operands of one target tile: 100 VGPRs; the
                               // clouds' centre buffer holds 64 dimensions)
constexpr int FMM_MAX_KS = 25;
constexpr int FMM_MAX_KS_TWO_TILES = 4;  // two target tiles per wave while the operands are small (D <= 9)
// source tiles per LDS stage: four while a stage stays below ~30 KiB, two up to K = 144, one beyond
__host__ __device__ constexpr int fmm_stage_tiles(int KS) { return KS <= 4 ? 4 : (KS <= 9 ? 2 : 1); }
constexpr int FMM_SHIFT = 15;        // T = 2^15 exp(-s) <= 32768 < 65504: small kernel values stay normal f16 numbers
constexpr float FMM_MAX_ONLINE_SHIFT = 32000.f;  // |kop|: what the two bf16 operand columns hold exactly (k_hi = 128 j, |j| <= 255)
constexpr int FMM_MAX_COLS = 32;

// K = 6 D + 7 columns (kmvp_fastmm_pack.hpp) + the two columns of the online shift at 16 KS - 2, 16 KS - 1
__host__ __device__ constexpr int fmm_ksteps(int D) { return (6 * D + 9 + 15) / 16; }
__host__ __device__ constexpr int fmm_row_bytes(int KS) { return KS * 32 + 16; }
__host__ __device__ constexpr int fmm_sig_bytes(int MODE) { return MODE ? 4096 : 2048; }
__host__ __device__ constexpr int fmm_tile_bytes(int KS, int MODE) { return FAST_TILE * fmm_row_bytes(KS) + fmm_sig_bytes(MODE); }
__host__ __device__ constexpr int fmm_stage_bytes(int KS, int MODE) {
  return (fmm_stage_tiles(KS) * fmm_tile_bytes(KS, MODE) + 4095) / 4096 * 4096;
}

struct FastmmArgs {
  const unsigned char* xop;  // target operands [n_pad / 32][KS][64 lanes] x 16 bytes (pack_fastmm_targets_kernel)
  const unsigned char* img;  // source stages [m_stages][fmm_stage_bytes]
  const double* unscale;     // [32]: 2^-15 / sigma_e per column
  double* part;              // partial sums [segments][NE][n_pad]
  int64_t n_pad;
  int64_t m_stages;
  int64_t seg_stages;
  int segments;
  int tile_blocks;
  int chunk_stages;
  int NE;                    // columns written (numerators [+ denominator])
  // exp(-r) only: the caller's coordinates for the exact recomputation of the closest pairs
  const float* xraw;         // targets (N, D)
  const float* yraw;         // sources (M, D)
  int64_t n, m;
  int D;
  float scale;               // the kernel's constant, applied AFTER a difference is formed
  float tau;                 // pairs with s <= tau are recomputed in the difference form
  float* kexp;               // ONLINE only; nullptr: the partial sums are written at the true scale; else they are
                             // written at 2^-kop and kexp[segment][n_pad] receives kop (exp(<x,y>))
};

// t - (float)pair[0] and t - (float)pair[1] in ONE instruction each: v_fma_mix_f32 reads an f16 half of a register as
// an fma operand (the compiler folds fma(h, -1, t) back into v_cvt_f32_f16 + v_sub_f32, a third more VALU work in the
// pair loop)
__device__ __forceinline__ float fmm_minus_lo_half(float t, h16x2 pair) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(t));
  return r;
}
__device__ __forceinline__ float fmm_minus_hi_half(float t, h16x2 pair) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(t));
  return r;
}

// T = 2^sh k(s).  Gaussian: the shift rides in the operands (S = s - sh); exp(-r): S = s, one v_sqrt_f32 more.
template <int KERNEL>
__device__ __forceinline__ float fmm_tval(float S, float sh) {
  if constexpr (KERNEL == K_GAUSSIAN) return kexp2(-S);
  else return kexp2(sh - __builtin_amdgcn_sqrtf(__builtin_fabsf(S)));
}

// min over the two lane halves (lanes l and l ^ 32 hold the two halves of one target's sources): one VALU swap
__device__ __forceinline__ float fmm_min_halves(float m) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fminf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// the difference form of one pair from the caller's coordinates (rare branch of the exp(-r) variant; kept out of line:
// sixteen inlined copies of the loop cost the hot path registers)
__device__ __attribute__((noinline)) float fmm_exact_sqdist(const float* __restrict__ xr, const float* __restrict__ yr,
                                                             int D, float scale) {
  float sx = 0.f;
  for (int c = 0; c < D; ++c) {
    const float e = (xr[c] - yr[c]) * scale;
    sx = fmaf(e, e, sx);
  }
  return sx;
}

template <int KS, int MODE, int TT, int KERNEL = K_GAUSSIAN, int ONLINE = 0>
__global__ void __launch_bounds__(BLOCK_THREADS) fastmm_kernel(const FastmmArgs a) {
  static_assert(KERNEL == K_GAUSSIAN || KERNEL == K_ABSEXP, "bounded kernels only");
  constexpr int RB = fmm_row_bytes(KS);
  constexpr int TB = fmm_tile_bytes(KS, MODE);
  constexpr int SB = fmm_stage_bytes(KS, MODE);
  constexpr int PIECES = SB / (16 * BLOCK_THREADS);
  constexpr int NOUT = MODE ? 16 : 8;  // output rows (columns of the result) per lane
  static_assert(SB % (16 * BLOCK_THREADS) == 0, "stage = whole LDS-DMA pieces");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SB];

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r = lane & 31;
  const int h = lane >> 5;
  const int64_t tile0 = ((int64_t)tb * WAVES_PER_BLOCK + wave) * TT;

  // the B operand of this lane's targets, pre-packed per (target tile, k-step, lane): the kernel depends on the point
  // dimension only through the number of k-steps
  bf16x8 xb[TT][KS];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      xb[tt][ks] = *reinterpret_cast<const bf16x8*>(a.xop + (((tile0 + tt) * KS + ks) * 64 + lane) * 16);

  f32x16 acc[TT];
  double accd[TT][NOUT];
  float kop[TT];  // ONLINE: the target's current shift (integer valued); acc and accd are sums of 2^(15 + kop) k b
  bool kset[TT];  // ... and whether any live source has set it yet
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    kop[tt] = 0.f;
    kset[tt] = false;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[tt][q] = 0.f;
#pragma unroll
    for (int q = 0; q < NOUT; ++q) accd[tt][q] = 0.0;
  }
  // rows of the accumulator registers: register 4g + j holds row 8g + 4h + j.  MODE 0: rows e (g < 2) and e + 16
  // (g + 2) are the two halves of column e; MODE 1: row = column.
  auto fold_one = [&](int tt) {
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
    // The squared distances of tile q + 1 are issued before the kernel values of tile q are worked on: the matrix pipe
    // runs them under the transcendentals and conversions of this very wave (the stage loop is unrolled, so the two
    // register sets swap without copies).
    constexpr int ST = fmm_stage_tiles(KS);
    f32x16 dn[TT];
    auto distances = [&](int q, f32x16 (&d)[TT]) {
      const unsigned char* lt = &lds[buf][q * TB];
      bf16x8 ya[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        ya[ks] = *reinterpret_cast<const bf16x8*>(lt + r * RB + (ks * 16 + 8 * h) * 2);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) d[tt][qq] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          d[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[ks], xb[tt][ks], d[tt], 0, 0, 0);
      }
    };
    distances(0, dn);
#pragma unroll
    for (int q = 0; q < ST; ++q) {
      const unsigned char* lt = &lds[buf][q * TB];
      // signal operands of the two k-steps of 16 sources: [k-step][lane] x 16 bytes (b_h | b_l rows in MODE 0)
      const unsigned char* ls = lt + FAST_TILE * RB;
      h16x8 ah[2], al[2];
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        ah[g2] = *reinterpret_cast<const h16x8*>(ls + g2 * 1024 + lane * 16);
        if constexpr (MODE == 1) al[g2] = *reinterpret_cast<const h16x8*>(ls + 2048 + g2 * 1024 + lane * 16);
      }
      f32x16 dc[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) dc[tt] = dn[tt];
      if (q + 1 < ST) distances(q + 1, dn);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        f32x16 d = dc[tt];
        float dmin = 0.f;
        if constexpr (KERNEL == K_ABSEXP || ONLINE) {
          dmin = fminf(fminf(d[0], d[1]), d[2]);
#pragma unroll
          for (int qq = 3; qq < 15; qq += 2) dmin = fminf(fminf(dmin, d[qq]), d[qq + 1]);
          dmin = fminf(dmin, d[15]);
        }
        if constexpr (KERNEL == K_ABSEXP) {
          // exp(-r) hinges on the RELATIVE accuracy of small s, the expansion around one centre has an absolute error
          // ~1e-7 R^2: pairs with s <= tau = kappa R^4 (the host's bound; in practice coincident and nearly coincident
          // points) are recomputed in the difference form from the caller's coordinates -- a wave-uniform rare branch
          if (__any(!(dmin > a.tau))) {
            const int64_t it = (tile0 + tt) * FAST_TILE + r;
            const int64_t j0 = (s * ST + q) * FAST_TILE;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
              const int64_t j = j0 + acc_row(qq, h);
              if (!(d[qq] > a.tau) && it < a.n && j < a.m) d[qq] = fmm_exact_sqdist(a.xraw + it * a.D, a.yraw + j * a.D, a.D, a.scale);
            }
          }
        }
        if constexpr (ONLINE) {
          // the target's closest source of this tile, over both lane halves.  Gaussian / exp(<x,y>): m = min S, the
          // tile's largest T is 2^-m; exp(-r): m = min r (the exactly recomputed pairs are >= 0 and at most tau away
          // from what dmin saw), the largest T is 2^(15 + kop - m).
          float m = fmm_min_halves(dmin);
          if constexpr (KERNEL == K_ABSEXP) m = __builtin_amdgcn_sqrtf(fmaxf(m, 0.f));
          const bool first = s == s_begin && q == 0;
          // (half a binade of hysteresis: T up to 2^15.5 < 65504 is still an f16 number, and the rounding noise of s at
          // nearly coincident points must not lower the shift of every such target by one)
          const bool need = (first || (KERNEL == K_ABSEXP ? m < kop[tt] - 0.5f : m < -((float)FMM_SHIFT + 0.5f))) && m < 3.0e38f;
          if (__any(need)) {  // rare: see the header
            fold_one(tt);  // (this tile's accumulator only: a target's sums do not depend on what else its wave owns)
            if (need) {
              float kn;
              if constexpr (KERNEL == K_ABSEXP) kn = floorf(m);
              else kn = fminf(fmaxf(kop[tt] + floorf(m + (float)FMM_SHIFT), -FMM_MAX_ONLINE_SHIFT), FMM_MAX_ONLINE_SHIFT);
              const float delta = kop[tt] - kn;  // >= 0 except at the first tile
              if constexpr (KERNEL != K_ABSEXP) {
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) d[qq] += delta;
                if (q + 1 < ST) {  // the next tile's distances were issued with the old operand
#pragma unroll
                  for (int qq = 0; qq < 16; ++qq) dn[tt][qq] += delta;
                }
                // kn = k_hi + k_lo, both exact in bf16 and of kn's sign (no cancellation between the two columns): k_hi a
                // multiple of 128 (8 significant bits up to 32640), |k_lo| < 128
                const float k_hi = 128.f * truncf(kn * 0.0078125f);
                if (h == 1) {
                  xb[tt][KS - 1][7] = (__bf16)(-k_hi);
                  xb[tt][KS - 1][6] = (__bf16)(k_hi - kn);
                }
              }
              const int di = (int)delta;
#pragma unroll
              for (int qq = 0; qq < NOUT; ++qq) accd[tt][qq] = ldexp(accd[tt][qq], -di);
              kop[tt] = kn;
              kset[tt] = true;
            }
          }
        }
        const float sh = (float)FMM_SHIFT + kop[tt];
        h16x8 th[2], tl[2];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const fmm_f32x2 t = {fmm_tval<KERNEL>(d[8 * g2 + 2 * i], sh), fmm_tval<KERNEL>(d[8 * g2 + 2 * i + 1], sh)};
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
    __syncthreads();  // vmcnt(0) + barrier: stage s+1 has landed, stage s is free
  }
  fold();

  // lane (r, h) holds target r of each tile and the columns e = 8g + 4h + j of its registers
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int q = 0; q < NOUT; ++q) {
      const int e = 8 * (q >> 2) + 4 * h + (q & 3);
      if (e < a.NE) {
        double v = accd[tt][q] * a.unscale[e];
        if constexpr (ONLINE) {
          if (!a.kexp) v = ldexp(v, -(int)kop[tt]);  // the true scale
        }
        a.part[((int64_t)seg * a.NE + e) * a.n_pad + (tile0 + tt) * FAST_TILE + r] = v;
      }
    }
  if constexpr (ONLINE) {
    if (a.kexp && h == 0) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)  // +inf: this segment saw no live source for the target
        a.kexp[(int64_t)seg * a.n_pad + (tile0 + tt) * FAST_TILE + r] = kset[tt] ? kop[tt] : INFINITY;
    }
  }
}

}  // namespace kmvp
