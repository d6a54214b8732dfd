// Morton ordering and packing kernels of the centred split-bf16 path (non-template kernels:
// included by kmvp_product.hip only).  Layouts are documented in kmvp_cfast.hpp.
#pragma once
#include "kmvp_cfast.hpp"
#include "kmvp_fast_pack.hpp"  // aux layout (FAST_AUX_*)

namespace kmvp {

// Morton key of every source from its position inside the clouds' bounding box
// (centre[0..D) and half-widths centre[FAST_AUX_HALF..), written by fast_center_kernel);
// floor(30 / D) bits per dimension.  Pad entries (j >= m) sort last.
__global__ void cfast_morton_kernel(const float* __restrict__ y, const float* __restrict__ centre,
                                    unsigned* __restrict__ keys, int* __restrict__ vals, int64_t m,
                                    int64_t m_alloc, int D) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m_alloc) return;
  vals[j] = (int)j;
  if (j >= m) {
    keys[j] = 0xFFFFFFFFu;
    return;
  }
  const int bits = 30 / D;
  unsigned key = 0;
  unsigned q[4] = {0, 0, 0, 0};
  for (int d = 0; d < D; ++d) {
    const float half = centre[FAST_AUX_HALF + d];
    float t = half > 0.f ? (y[j * D + d] - centre[d]) / (2.f * half) + 0.5f : 0.f;  // [0, 1]
    t = fminf(fmaxf(t, 0.f), 0.999999f);
    q[d] = (unsigned)(t * (float)(1u << bits));
  }
  for (int bit = bits - 1; bit >= 0; --bit)
    for (int d = 0; d < D; ++d) key = (key << 1) | ((q[d] >> bit) & 1u);
  keys[j] = key;
}

// The part of a group image the POINTS determine (shared by cfast_kernel's and cfastmm_kernel's images, which differ in
// where the raw coordinates sit): one workgroup of CF_GROUP threads per group of sorted sources: centre = bounding-box
// midpoint of the group in the caller's coordinates, rows y' = (y - c) * scale, tau = kappa * max |y'|^2, reach^2, the
// caller's fp32 coordinates.  Returns the thread's source index (>= m: pad).
__device__ __forceinline__ int cfast_pack_group_points(const float* __restrict__ y, const int* __restrict__ perm,
                                                       unsigned char* __restrict__ g, int raw_off, int64_t group,
                                                       int64_t m, int D, float scale) {
  const int rr = threadIdx.x;
  const int64_t k = group * CF_GROUP + rr;  // position in the sorted order
  const int src = perm[k];  // pad positions carry indices >= m
  const bool live = src < m;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int d = 0; d < D; ++d) v[d] = live ? y[(int64_t)src * D + d] : 0.f;
  // group centre: midpoint of the live points' bounding box (wave reductions, then across the
  // waves of the workgroup through LDS)
  __shared__ float red_lo[4][CF_GROUP / 64], red_hi[4][CF_GROUP / 64], red_r2[CF_GROUP / 64];
  float c[4];
  for (int d = 0; d < 4; ++d) {
    float lo = live ? v[d] : INFINITY, hi = live ? v[d] : -INFINITY;
    for (int o = 32; o > 0; o >>= 1) {
      lo = fminf(lo, __shfl_xor(lo, o));
      hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((rr & 63) == 0) {
      red_lo[d][rr >> 6] = lo;
      red_hi[d][rr >> 6] = hi;
    }
  }
  __syncthreads();
  for (int d = 0; d < 4; ++d) {
    float lo = red_lo[d][0], hi = red_hi[d][0];
    for (int w = 1; w < CF_GROUP / 64; ++w) {
      lo = fminf(lo, red_lo[d][w]);
      hi = fmaxf(hi, red_hi[d][w]);
    }
    c[d] = (lo <= hi) ? 0.5f * (lo + hi) : 0.f;
  }
  float yr[4];
  double sq = 0.0;
  for (int d = 0; d < 4; ++d) {
    yr[d] = live ? (v[d] - c[d]) * scale : 0.f;
    sq += (double)yr[d] * (double)yr[d];
  }
  float r2 = (float)sq;
  for (int o = 32; o > 0; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
  if ((rr & 63) == 0) red_r2[rr >> 6] = r2;
  __syncthreads();
  for (int w = 0; w < CF_GROUP / 64; ++w) r2 = fmaxf(r2, red_r2[w]);
  if (rr == 0) {
    float* hdr = reinterpret_cast<float*>(g);
    for (int d = 0; d < 4; ++d) hdr[d] = c[d];
    hdr[4] = CF_KAPPA * r2;
    const float reach = sqrtf(r2) + sqrtf(CF_KAPPA * r2);  // no pair below tau for targets farther than this from c
    hdr[5] = reach * reach;
    hdr[6] = hdr[7] = 0.f;
  }
  // row: k 0..7 dim 0 + (|y'|^2_h, |y'|^2_m); 8..15 dim 1 + (|y'|^2_l, 1); 16..23 dim 2 + (1, 1);
  //      24..31 dim 3 + (0, 0); then 8 bf16 of pad
  __bf16* row = reinterpret_cast<__bf16*>(g + CF_HDR + rr * CF_ROW_BYTES);
  const __bf16 zero = (__bf16)0.f, one = (__bf16)1.f;
  __bf16 sh, sm, sl;
  {
    const float sf = (float)sq;
    sh = (__bf16)sf;
    const float r1 = sf - (float)sh;
    sm = (__bf16)r1;
    sl = (__bf16)(r1 - (float)sm);
  }
  for (int d = 0; d < 4; ++d) {
    const __bf16 vh = (__bf16)yr[d];
    const float r1 = yr[d] - (float)vh;
    const __bf16 vm = (__bf16)r1;
    const __bf16 vl = (__bf16)(r1 - (float)vm);
    const __bf16 h2 = (__bf16)(-2.f * (float)vh), m2 = (__bf16)(-2.f * (float)vm), l2 = (__bf16)(-2.f * (float)vl);
    __bf16* blk = row + 8 * d;
    blk[0] = h2; blk[1] = h2; blk[2] = m2; blk[3] = h2; blk[4] = m2; blk[5] = l2;
  }
  row[6] = live ? sh : (__bf16)INFINITY;
  row[7] = live ? sm : zero;
  row[14] = live ? sl : zero;
  row[15] = one;
  row[22] = one;
  row[23] = one;
  row[30] = zero;
  row[31] = zero;
  for (int q = 32; q < 40; ++q) row[q] = zero;
  float* raw = reinterpret_cast<float*>(g + raw_off) + rr * 4;
  for (int d = 0; d < 4; ++d) raw[d] = live ? v[d] : INFINITY;
  return src;
}

// cfast_kernel's group image: the points' part, the signal (one float per source), the original global index
__global__ void __launch_bounds__(CF_GROUP) pack_cfast_sources_kernel(
    const float* __restrict__ y, const float* __restrict__ b, const int* __restrict__ perm,
    unsigned char* __restrict__ img, int64_t m, int D, int EB, float scale, int64_t j_offset) {
  const int64_t group = blockIdx.x;
  const int rr = threadIdx.x;
  const int64_t stage = group / CF_STAGE_GROUPS;
  unsigned char* g = img + stage * (int64_t)CF_STAGE_BYTES + (group % CF_STAGE_GROUPS) * CF_GROUP_BYTES;
  const int src = cfast_pack_group_points(y, perm, g, CF_OFF_RAW, group, m, D, scale);
  const bool live = src < m;
  reinterpret_cast<float*>(g + CF_OFF_B)[rr] = (live && EB > 0) ? b[src] : 0.f;
  reinterpret_cast<int*>(g + CF_OFF_IDX)[rr] = live ? (int)(j_offset + src) : -1;
}

// targets [n_pad][4]: the caller's coordinates (unused dimensions and pad targets 0)
__global__ void pack_cfast_targets_kernel(const float* __restrict__ x, float* __restrict__ xraw, int64_t n,
                                          int64_t n_pad, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  for (int d = 0; d < 4; ++d) xraw[i * 4 + d] = (i < n && d < D) ? x[i * D + d] : 0.f;
}

}  // namespace kmvp
