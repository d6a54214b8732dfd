// Packing kernels of cfastmm_kernel (layout in kmvp_cfastmm.hpp; included by kmvp_product.hip only).  Targets:
// pack_cfast_targets_kernel; Morton order: kmvp_cfast_pack.hpp.
#pragma once
#include "kmvp_cfast_pack.hpp"
#include "kmvp_cfastmm.hpp"
#include "kmvp_fastmm_pack.hpp"

namespace kmvp {

// one workgroup of CF_GROUP threads per group of sorted sources: header, rows, raw coordinates
__global__ void __launch_bounds__(CF_GROUP) pack_cfastmm_points_kernel(const float* __restrict__ y,
                                                                      const int* __restrict__ perm,
                                                                      unsigned char* __restrict__ img, int64_t m, int D,
                                                                      int MODE, float scale, int64_t j_offset) {
  const int64_t group = blockIdx.x;
  unsigned char* g = img + group * (int64_t)cfm_stage_bytes(MODE);
  const int src = cfast_pack_group_points(y, perm, g, cfm_off_raw(MODE), group, m, D, scale);
  reinterpret_cast<int*>(g + cfm_off_idx(MODE))[threadIdx.x] = src < m ? (int)(j_offset + src) : -1;
}

// the signal operands of every row tile, in the sorted order of the sources: pack_fastmm_signal_kernel's layout
// (one thread per 16-byte piece: (group, row tile, part, k-step g2, lane)), block of nb columns from column col0
__global__ void pack_cfastmm_signal_kernel(const float* __restrict__ b, const float* __restrict__ sigma,
                                           const int* __restrict__ perm, unsigned char* __restrict__ img, int64_t m,
                                           int64_t m_groups, int E, int col0, int nb, int MODE) {
  const int parts = MODE ? 2 : 1;
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_tile = (int64_t)parts * 2 * 64;
  if (id >= m_groups * (CF_GROUP / 32) * per_tile) return;
  const int64_t t = id / per_tile;  // row tile in the sorted order
  const int w = (int)(id % per_tile);
  const int part = w / 128, g2 = (w / 64) & 1, lane = w & 63;
  const int mrow = lane & 31, h = lane >> 5;
  const int col = MODE ? mrow : (mrow & 15);
  const bool want_lo = MODE ? part == 1 : mrow >= 16;
  unsigned char* sig = img + (t / (CF_GROUP / 32)) * (int64_t)cfm_stage_bytes(MODE) + CFM_OFF_SIG +
                       (t % (CF_GROUP / 32)) * fmm_sig_bytes(MODE);
  h16x8 out;
  for (int i = 0; i < 8; ++i) {
    const int64_t k = t * 32 + 8 * (2 * g2 + (i >> 2)) + 4 * h + (i & 3);  // position in the sorted order
    const int src = perm[k];
    float v = 0.f;
    if (src < m && col < nb) v = col0 + col < E ? b[(int64_t)src * E + col0 + col] * sigma[col] : 1.f;
    const _Float16 hi = (_Float16)v;
    out[i] = want_lo ? (_Float16)(v - (float)hi) : hi;
  }
  *reinterpret_cast<h16x8*>(sig + part * 2048 + g2 * 1024 + lane * 16) = out;
}

}  // namespace kmvp
