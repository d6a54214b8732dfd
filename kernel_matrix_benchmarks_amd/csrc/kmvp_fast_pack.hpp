// Packing kernels of the split-bf16 MFMA low-D path (non-template kernels: included by
// kmvp_api.hip only).  Layouts are documented in kmvp_fast.hpp.
#pragma once
#include "kmvp_fast.hpp"

namespace kmvp {

// Bounding box of both clouds in two launches (once per kmvp_set_points): FAST_BBOX_BLOCKS blocks
// reduce strided subsets per dimension, one block finishes.  aux layout (floats):
//   [0, D) centre = midpoint of the box, [FAST_AUX_RADIUS2] squared half-diagonal,
//   [FAST_AUX_HALF, FAST_AUX_HALF + D) half-widths.
// Subtracting the centre before the |x|^2 + |y|^2 - 2x.y expansion keeps the cancellation error of
// the fast form as small as the data allow; the half-diagonal bounds |x'|^2, |y'|^2 and drives the
// "auto" choice; the Morton keys of kmvp_cfast_pack.hpp are taken relative to the box.
constexpr int FAST_AUX_RADIUS2 = 64;
constexpr int FAST_AUX_HALF = 65;
constexpr int FAST_AUX_FLOATS = 160;
constexpr int FAST_BBOX_BLOCKS = 256;

template <typename real>
__global__ void __launch_bounds__(256) fast_bbox_partial_kernel(const real* __restrict__ y, int64_t m,
                                                               const real* __restrict__ x, int64_t n, int D,
                                                               float* __restrict__ part /* [blocks][D][2] */) {
  __shared__ float lo[256], hi[256];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int d = 0; d < D; ++d) {
    float vmin = INFINITY, vmax = -INFINITY;
    for (int64_t i = first; i < m; i += stride) {
      const float v = (float)y[i * D + d];
      vmin = fminf(vmin, v);
      vmax = fmaxf(vmax, v);
    }
    if (x != nullptr)
      for (int64_t i = first; i < n; i += stride) {
        const float v = (float)x[i * D + d];
        vmin = fminf(vmin, v);
        vmax = fmaxf(vmax, v);
      }
    lo[threadIdx.x] = vmin;
    hi[threadIdx.x] = vmax;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        lo[threadIdx.x] = fminf(lo[threadIdx.x], lo[threadIdx.x + s]);
        hi[threadIdx.x] = fmaxf(hi[threadIdx.x], hi[threadIdx.x + s]);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      part[((int64_t)blockIdx.x * D + d) * 2 + 0] = lo[0];
      part[((int64_t)blockIdx.x * D + d) * 2 + 1] = hi[0];
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(FAST_BBOX_BLOCKS) fast_center_kernel(const float* __restrict__ part, int D,
                                                                      float* __restrict__ centre) {
  __shared__ float lo[FAST_BBOX_BLOCKS], hi[FAST_BBOX_BLOCKS];
  float radius2 = 0.f;
  for (int d = 0; d < D; ++d) {
    lo[threadIdx.x] = part[((int64_t)threadIdx.x * D + d) * 2 + 0];
    hi[threadIdx.x] = part[((int64_t)threadIdx.x * D + d) * 2 + 1];
    __syncthreads();
    for (int s = FAST_BBOX_BLOCKS / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) {
        lo[threadIdx.x] = fminf(lo[threadIdx.x], lo[threadIdx.x + s]);
        hi[threadIdx.x] = fmaxf(hi[threadIdx.x], hi[threadIdx.x + s]);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float c = 0.5f * (lo[0] + hi[0]);
      const bool ok = (c == c && fabsf(c) != INFINITY);
      centre[d] = ok ? c : 0.f;
      const float half = ok ? 0.5f * (hi[0] - lo[0]) : INFINITY;
      radius2 += half * half;
      centre[FAST_AUX_HALF + d] = half;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) centre[FAST_AUX_RADIUS2] = radius2;
}

// exact three-way bf16 split of an fp32 value: v == hi + mid + lo
__device__ __forceinline__ void fast_split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)v;
  const float r1 = v - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

// targets [n_pad][RD]: centred scaled coordinates x'_0 .. x'_{D-1}, then |x'|^2 (accumulated in
// double, rounded once), zeros up to RD = fast_target_row(D); pad targets are all zero.  The
// kernel splits these into the bf16 operand itself.
__global__ void pack_fast_targets_kernel(const float* __restrict__ x, const float* __restrict__ centre,
                                         float* __restrict__ xr, int64_t n, int64_t n_pad, int D, int RD,
                                         float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  float* row = xr + i * RD;
  double sq = 0.0;
  for (int d = 0; d < D; ++d) {
    const float v = i < n ? (x[i * D + d] - centre[d]) * scale : 0.f;
    sq += (double)v * (double)v;
    row[d] = v;
  }
  row[D] = (float)sq;
  for (int d = D + 1; d < RD; ++d) row[d] = 0.f;
}

// source stages: per tile [32 rows x row_bytes] (per d (-2y_h, -2y_h, -2y_m, -2y_h, -2y_m, -2y_l),
// then |y'|^2 h,m,l, 1,1,1) followed by [32] signal floats; pad sources: |y'|^2 = +inf, b = 0.
__global__ void pack_fast_sources_kernel(const float* __restrict__ y, const float* __restrict__ b,
                                         const float* __restrict__ centre, unsigned char* __restrict__ img,
                                         int64_t m, int64_t m_stages, int D, int EB, int KS, float scale) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int ST = fast_stage_tiles(KS);
  if (j >= m_stages * ST * FAST_TILE) return;
  const int64_t stage = j / (ST * FAST_TILE);
  const int q = (int)((j / FAST_TILE) % ST);
  const int r = (int)(j % FAST_TILE);
  const int RB = fast_row_bytes(KS);
  unsigned char* tile = img + stage * (int64_t)fast_stage_bytes(KS, EB) + q * fast_tile_bytes(KS, EB);
  __bf16* row = reinterpret_cast<__bf16*>(tile + r * RB);
  const bool live = j < m;
  const __bf16 zero = (__bf16)0.f, one = (__bf16)1.f;
  double sq = 0.0;
  for (int d = 0; d < D; ++d) {
    const float v = live ? (y[j * D + d] - centre[d]) * scale : 0.f;
    sq += (double)v * (double)v;
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
  for (int k = 6 * D + 6; k < 16 * KS + 8; ++k) row[k] = zero;  // incl. the 16-byte row pad
  if (EB > 0) reinterpret_cast<float*>(tile + FAST_TILE * RB)[r] = live ? b[j] : 0.f;
}

}  // namespace kmvp
