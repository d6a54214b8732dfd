// Packing and reduction kernels of the bf16 MFMA path (non-template kernels: included
// by kmvp_api.hip only).  Layouts are documented in kmvp_mfma.hpp.
#pragma once
#include "kmvp_mfma.hpp"

namespace kmvp {

// ---------------------------------------------------------------------------------
// packing (per set_points / set_signal; HBM-bound and small)

__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)v;
  const float r1 = v - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

// targets (N,D) f32 -> augmented bf16 rows [n_pad][KD]; pad targets are the origin (S = |y|^2: finite against live
// sources, +inf against pad sources -- an all-zero row would make that inf x 0 = NaN)
// dot == 1 (exp(<x,y>), kmvp_mfma.hpp MFMA_DOT_AUG): [x, 0.., -m_hi = 0, -m_lo = 0, 1]; pad targets are the origin.
// dot == 2 (the shifted Gaussian): the plain layout, whose zeros at 16 KS - 3 / - 2 are the initial shift.
__global__ void pack_mfma_targets_kernel(const float* __restrict__ x, __bf16* __restrict__ xa,
                                         int64_t n, int64_t n_pad, int D, int KD, float scale, int dot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  __bf16* row = xa + i * KD;
  if (dot == 1) {
    for (int k = 0; k < KD; ++k) row[k] = (__bf16)((k < D && i < n) ? x[i * D + k] * scale : (k == KD - 1 ? 1.f : 0.f));
    return;
  }
  if (i >= n) {
    for (int k = 0; k < KD; ++k) row[k] = (__bf16)((k >= D && k < D + 3) ? 1.f : 0.f);
    return;
  }
  float sq = 0.f;
  for (int d = 0; d < D; ++d) {
    const __bf16 v = (__bf16)(x[i * D + d] * scale);
    row[d] = v;
    sq = fmaf((float)v, (float)v, sq);
  }
  __bf16 hi, mid, lo;
  split3(sq, hi, mid, lo);
  row[D + 0] = (__bf16)1.f;
  row[D + 1] = (__bf16)1.f;
  row[D + 2] = (__bf16)1.f;
  row[D + 3] = hi;
  row[D + 4] = mid;
  row[D + 5] = lo;
  for (int k = D + MFMA_AUG; k < KD; ++k) row[k] = (__bf16)0.f;
}

// sources (M,D) + signal (M,E) f32 -> one LDS image per tile of 32 sources:
//   [32 rows x y_stride bytes: -2y, |y|^2 hi/mid/lo, 1, 1, 1, 0..] [NT*32 rows x 72 bytes: V^T]
// pad sources get |y|^2 = +inf (k = 0 for every kernel) and zero signal.
__global__ void pack_mfma_sources_kernel(const float* __restrict__ y, const float* __restrict__ b,
                                         unsigned char* __restrict__ img, int64_t m, int64_t m_tiles,
                                         int D, int E, int KS, int NT, float scale, int dot) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m_tiles * MFMA_TILE) return;
  const int64_t t = j / MFMA_TILE;
  const int jr = (int)(j % MFMA_TILE);
  const int KD = 16 * KS;
  const int YS = mfma_y_stride(KS);
  unsigned char* base = img + t * (int64_t)mfma_image_bytes(KS, NT);
  __bf16* row = reinterpret_cast<__bf16*>(base + jr * YS);
  const bool live = j < m;
  if (dot == 1) {
    // exp(<x,y>): [y, 0.., 1, 1, mask]; a pad source gets -3e38 through the targets' column of ones: 2^(-3e38 - m) = 0
    // (finite on purpose: -inf would turn a non-finite target coordinate's 0 x inf into NaN for the whole row)
    for (int k = 0; k < KD + 8; ++k) row[k] = (__bf16)((k < D && live) ? y[j * D + k] * scale : 0.f);
    row[KD - 3] = (__bf16)1.f;
    row[KD - 2] = (__bf16)1.f;
    row[KD - 1] = (__bf16)(live ? 0.f : -3.0e38f);
  } else {
    float sq = 0.f;
    for (int d = 0; d < D; ++d) {
      const __bf16 v = live ? (__bf16)(y[j * D + d] * scale) : (__bf16)0.f;
      sq = fmaf((float)v, (float)v, sq);
      row[d] = (__bf16)(-2.f * (float)v);
    }
    __bf16 hi, mid, lo;
    split3(sq, hi, mid, lo);
    row[D + 0] = live ? hi : (__bf16)INFINITY;
    row[D + 1] = live ? mid : (__bf16)0.f;
    row[D + 2] = live ? lo : (__bf16)0.f;
    row[D + 3] = (__bf16)1.f;
    row[D + 4] = (__bf16)1.f;
    row[D + 5] = (__bf16)1.f;
    for (int k = D + MFMA_AUG; k < KD + 8; ++k) row[k] = (__bf16)0.f;  // incl. the 16-byte row pad
    if (dot == 2) {  // the shifted Gaussian: ones against the targets' two shift columns (kmvp_mfma.hpp mfma_ksteps_shifted)
      row[KD - 3] = (__bf16)1.f;
      row[KD - 2] = (__bf16)1.f;
    }
  }
  unsigned char* vt = base + MFMA_TILE * YS;
  for (int e = 0; e < NT * 32; ++e) {
    const float v = (live && b != nullptr && e < E) ? b[j * E + e] : ((live && b == nullptr && e == 0) ? 1.f : 0.f);
    *reinterpret_cast<__bf16*>(vt + e * MFMA_V_STRIDE + jr * 2) = (__bf16)v;
    if (jr < 4) *reinterpret_cast<__bf16*>(vt + e * MFMA_V_STRIDE + 64 + jr * 2) = (__bf16)0.f;  // row pad
  }
}

// [segments][n_pad][NEP] fp32 + [segments][n_pad] -> sums[e][i] fp64 (+ denominator column)
__global__ void mfma_reduce_kernel(const float* __restrict__ part, const float* __restrict__ partd,
                                   double* __restrict__ sums, int64_t n_pad, int NEP, int E,
                                   int segments, int with_den) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cols = E + (with_den ? 1 : 0);
  if (q >= n_pad * cols) return;
  const int64_t i = q / cols;
  const int e = (int)(q % cols);
  double v = 0.0;
  if (e < E) {
    for (int s = 0; s < segments; ++s) v += (double)part[((int64_t)s * n_pad + i) * NEP + e];
  } else {
    for (int s = 0; s < segments; ++s) v += (double)partd[(int64_t)s * n_pad + i];
  }
  sums[(int64_t)e * n_pad + i] = v;
}

// exp(<x,y>): the same with the partial sums of segment s at the scale 2^-kexp[s][i] (kmvp_mfma.hpp): K_i = min_s kexp,
// sums = sum_s part 2^(K_i - kexp) (every factor <= 1), kmin[i] = K_i (+inf: no live source at all)
__global__ void mfma_reduce_shifted_kernel(const float* __restrict__ part, const float* __restrict__ partd,
                                           const float* __restrict__ kexp, double* __restrict__ sums,
                                           double* __restrict__ kmin, int64_t n_pad, int NEP, int E, int segments, int with_den) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int cols = E + (with_den ? 1 : 0);
  if (q >= n_pad * cols) return;
  const int64_t i = q / cols;
  const int e = (int)(q % cols);
  float K = INFINITY;
  for (int s = 0; s < segments; ++s) K = fminf(K, kexp[(int64_t)s * n_pad + i]);
  double v = 0.0;
  for (int s = 0; s < segments; ++s) {
    const float ks = kexp[(int64_t)s * n_pad + i];
    if (!(ks < INFINITY)) continue;
    const double t = e < E ? (double)part[((int64_t)s * n_pad + i) * NEP + e] : (double)partd[(int64_t)s * n_pad + i];
    v += ldexp(t, (int)fmaxf(K - ks, -100000.f));
  }
  sums[(int64_t)e * n_pad + i] = v;
  if (e == 0) kmin[i] = (double)K;
}

}  // namespace kmvp
