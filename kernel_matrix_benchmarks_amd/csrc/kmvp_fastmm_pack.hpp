// Packing kernels of fastmm_kernel (layouts in kmvp_fastmm.hpp; included by kmvp_product.hip only).
#pragma once
#include "kmvp_fast_pack.hpp"
#include "kmvp_fastmm.hpp"

namespace kmvp {

// the part of the stage image the POINTS determine: per tile 32 rows of split-bf16 coordinates as in
// pack_fast_sources_kernel, with a 1 in column 6 D + 6 (against the targets' -FMM_SHIFT) and in columns 16 KS - 2,
// 16 KS - 1 (free by fmm_ksteps; against the two pieces of the targets' online shift); pad sources: |y'|^2 = +inf.
// dot != 0 (k = exp(<x,y>)): the rows hold -2 (y scale) uncentred and a zero norm, so that S = -2 scale <x, y>.
__global__ void pack_fastmm_rows_kernel(const float* __restrict__ y, const float* __restrict__ centre,
                                        unsigned char* __restrict__ img, int64_t m, int64_t m_stages, int D, int KS,
                                        int MODE, float scale, int dot) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m_stages * fmm_stage_tiles(KS) * FAST_TILE) return;
  const int64_t stage = j / (fmm_stage_tiles(KS) * FAST_TILE);
  const int q = (int)((j / FAST_TILE) % fmm_stage_tiles(KS));
  const int r = (int)(j % FAST_TILE);
  const int RB = fmm_row_bytes(KS);
  unsigned char* tile = img + stage * (int64_t)fmm_stage_bytes(KS, MODE) + q * fmm_tile_bytes(KS, MODE);
  __bf16* row = reinterpret_cast<__bf16*>(tile + r * RB);
  const bool live = j < m;
  const __bf16 zero = (__bf16)0.f, one = (__bf16)1.f;
  double sq = 0.0;
  for (int d = 0; d < D; ++d) {
    const float v = live ? (dot ? y[j * D + d] * scale : (y[j * D + d] - centre[d]) * scale) : 0.f;
    if (!dot) sq += (double)v * (double)v;
    __bf16 vh, vm, vl;
    fast_split3(v, vh, vm, vl);
    const __bf16 h2 = (__bf16)(-2.f * (float)vh), m2 = (__bf16)(-2.f * (float)vm), l2 = (__bf16)(-2.f * (float)vl);
    row[6 * d + 0] = h2;
    row[6 * d + 1] = h2;
    row[6 * d + 2] = m2;
    row[6 * d + 3] = h2;
    row[6 * d + 4] = m2;
    row[6 * d + 5] = l2;
  }
  __bf16 sh, sm, sl;
  fast_split3((float)sq, sh, sm, sl);
  row[6 * D + 0] = live ? sh : (__bf16)INFINITY;
  row[6 * D + 1] = live ? sm : zero;
  row[6 * D + 2] = live ? sl : zero;
  row[6 * D + 3] = one;
  row[6 * D + 4] = one;
  row[6 * D + 5] = one;
  row[6 * D + 6] = one;
  for (int k = 6 * D + 7; k < 16 * KS + 8; ++k) row[k] = zero;  // incl. the 16-byte row pad
  row[16 * KS - 1] = one;
  row[16 * KS - 2] = one;
}

// target operands [n_pad / 32][KS][64 lanes] x 16 bytes: lane (r, h) of target tile t holds elements k = 16 ks + 8 h + j
// (j = 0 .. 7) of target 32 t + r's augmented row -- per d (x_h, x_m, x_h, x_l, x_m, x_h) of the centred scaled coordinate
// split three ways into bf16, then 1, 1, 1, |x'|^2 h, m, l (accumulated in double, rounded once), -FMM_SHIFT, zeros.
// One thread per (target, k-step, lane half).
// (shift: FMM_SHIFT for the Gaussian, 0 for exp(-r), whose shift is applied after the square root)
// dot != 0: uncentred, unscaled coordinates and a zero norm (see pack_fastmm_rows_kernel)
__global__ void pack_fastmm_targets_kernel(const float* __restrict__ x, const float* __restrict__ centre,
                                           unsigned char* __restrict__ xop, int64_t n, int64_t n_pad, int D, int KS,
                                           float scale, float shift, int dot) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_pad * KS * 2) return;
  const int64_t i = id / (2 * KS);
  const int ks = (int)((id / 2) % KS), h = (int)(id & 1);
  double sq = 0.0;
  for (int d = 0; d < D && !dot; ++d) {
    const float v = i < n ? (x[i * D + d] - centre[d]) * scale : 0.f;
    sq += (double)v * (double)v;
  }
  __bf16 sh, sm, sl;
  fast_split3((float)sq, sh, sm, sl);
  bf16x8 out;
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * ks + 8 * h + j;
    __bf16 e = (__bf16)0.f;
    if (k < 6 * D) {
      const int d = k / 6, role = k % 6;
      const float v = i < n ? (dot ? x[i * D + d] : (x[i * D + d] - centre[d]) * scale) : 0.f;
      __bf16 vh, vm, vl;
      fast_split3(v, vh, vm, vl);
      e = (role == 0 || role == 2 || role == 5) ? vh : ((role == 1 || role == 4) ? vm : vl);
    } else if (k < 6 * D + 3) {
      e = (__bf16)1.f;
    } else if (k == 6 * D + 3) {
      e = sh;
    } else if (k == 6 * D + 4) {
      e = sm;
    } else if (k == 6 * D + 5) {
      e = sl;
    } else if (k == 6 * D + 6) {
      e = (__bf16)(-shift);
    }
    out[j] = e;
  }
  const int64_t tile = i / 32;
  const int lane = h * 32 + (int)(i % 32);
  *reinterpret_cast<bf16x8*>(xop + ((tile * KS + ks) * 64 + lane) * 16) = out;
}

// one block per column: sigma[e] = 2^(14 - ex) with max |b_e| = f 2^ex, f in [0.5, 1): |b sigma| < 2^14, so that the
// f16 rest of ordinary entries is a normal number; unscale[e] = 2^-FMM_SHIFT / sigma[e], exact.  The denominator column
// of normalised rows (e == E) is all ones: sigma = 1.
// (block of columns col0 .. col0 + nb - 1 of the (m, E) signal; local column e = blockIdx.x)
__global__ void __launch_bounds__(256) fastmm_colscale_kernel(const float* __restrict__ b, int64_t m, int E, int col0,
                                                              int nb, float* __restrict__ sigma,
                                                              double* __restrict__ unscale) {
  __shared__ unsigned wmax[4];
  const int e = blockIdx.x;
  const bool is_signal = e < nb && col0 + e < E;
  unsigned v = 0;
  if (is_signal)
    for (int64_t i = threadIdx.x; i < m; i += blockDim.x) {
      const unsigned u = (unsigned)__float_as_int(b[i * E + col0 + e]) & 0x7fffffffu;
      v = u > v ? u : v;
    }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) v = wmax[w] > v ? wmax[w] : v;
    const float bmax = __int_as_float((int)v);
    int ex = 0;
    if (is_signal && bmax > 0.f && bmax < INFINITY) {
      int eb;
      (void)frexpf(bmax, &eb);
      ex = 14 - eb;
      ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
    }
    sigma[e] = e < nb ? ldexpf(1.f, ex) : 0.f;
    unscale[e] = e < nb ? ldexp(1.0, -FMM_SHIFT - ex) : 0.0;
  }
}

// the signal part of the stage image.  One thread per 16-byte operand piece: (tile, part, k-step g2, lane); lane
// (m = lane & 31, h = lane >> 5) element i holds source row 8 (2 g2 + (i >> 2)) + 4 h + (i & 3) of the tile -- the order
// in which the lane's registers of the S tile hold the sources (acc_row) -- of
//   MODE 0: part 0 only; m < 16: b_h of column m, m >= 16: b_l of column m - 16
//   MODE 1: part 0: b_h of column m, part 1: b_l of column m
// with b sigma_e = b_h + b_l (f16 each), for the block of nb columns that starts at column col0 of the (m, E) signal;
// a column beyond E (the denominator of normalised rows) is all ones; pad sources and unused columns: 0.
__global__ void pack_fastmm_signal_kernel(const float* __restrict__ b, const float* __restrict__ sigma,
                                          unsigned char* __restrict__ img, int64_t m, int64_t m_stages, int E, int col0,
                                          int nb, int KS, int MODE) {
  const int parts = MODE ? 2 : 1;
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_tile = (int64_t)parts * 2 * 64;
  if (id >= m_stages * fmm_stage_tiles(KS) * per_tile) return;
  const int64_t t = id / per_tile;
  const int w = (int)(id % per_tile);
  const int part = w / 128, g2 = (w / 64) & 1, lane = w & 63;
  const int mrow = lane & 31, h = lane >> 5;
  const int col = MODE ? mrow : (mrow & 15);
  const bool want_lo = MODE ? part == 1 : mrow >= 16;
  unsigned char* tile = img + (t / fmm_stage_tiles(KS)) * (int64_t)fmm_stage_bytes(KS, MODE) +
                        (t % fmm_stage_tiles(KS)) * fmm_tile_bytes(KS, MODE);
  h16x8 out;
  for (int i = 0; i < 8; ++i) {
    const int64_t j = t * FAST_TILE + 8 * (2 * g2 + (i >> 2)) + 4 * h + (i & 3);
    float v = 0.f;
    if (j < m && col < nb) v = col0 + col < E ? b[j * E + col0 + col] * sigma[col] : 1.f;
    const _Float16 hi = (_Float16)v;
    out[i] = want_lo ? (_Float16)(v - (float)hi) : hi;
  }
  *reinterpret_cast<h16x8*>(tile + FAST_TILE * fmm_row_bytes(KS) + part * 2048 + g2 * 1024 + lane * 16) = out;
}

}  // namespace kmvp
