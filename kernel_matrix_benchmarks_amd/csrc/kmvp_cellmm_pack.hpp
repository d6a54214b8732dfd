// Packing kernels of cellmm_kernel (layouts in kmvp_cellmm.hpp; included by kmvp_product.hip only).
// The target side is cell_kernel's (pack_cell_targets_kernel).
#pragma once
#include "kmvp_cellmm.hpp"

namespace kmvp {

// one block of 32 threads per source tile: the part of the stage image the POINTS determine --
// (f, g) = (2 e, -|e|^2 log2 e) per source, the tile's cell centre and key (-1: pad tile)
__global__ void __launch_bounds__(CELL_TILE) pack_cellmm_points_kernel(
    const float* __restrict__ y, const int* __restrict__ perm, const int* __restrict__ gstart,
    const int* __restrict__ gcnt, const unsigned* __restrict__ gkey, int64_t n_groups, int D, CellGrid grid,
    unsigned char* __restrict__ img) {
  const int64_t g = blockIdx.x;
  const int r = threadIdx.x;
  unsigned char* stage = img + (g / CMM_STAGE_TILES) * CMM_STAGE_BYTES;
  const int q = (int)(g % CMM_STAGE_TILES);
  const bool real_tile = g < n_groups;
  const unsigned key = real_tile ? gkey[g] : 0u;
  const bool valid = real_tile && r < gcnt[g];
  const int64_t idx = valid ? perm[gstart[g] + r] : 0;
  float e[3] = {0.f, 0.f, 0.f}, c[3] = {0.f, 0.f, 0.f};
  for (int a = 0; a < D; ++a) {
    c[a] = cell_centre(key, a, grid);
    e[a] = valid ? y[idx * D + a] - c[a] : 0.f;  // the caller's coordinate minus the stored centre: exact to an ulp of e
  }
  const float e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
  *reinterpret_cast<f32x4*>(stage + CMM_E_OFF + (q * CELL_TILE + r) * 16) =
      f32x4{2.f * e[0], 2.f * e[1], 2.f * e[2], e2 * -1.4426950408889634f};
  if (r == 0)
    *reinterpret_cast<f32x4*>(stage + CMM_HDR_OFF + q * 16) =
        f32x4{c[0], c[1], c[2], __int_as_float(real_tile ? (int)key : -1)};
}

// max |b| as the bits of a non-negative float (they order like unsigned integers; a NaN wins)
// (column `col` of the row-major (n, E) signal)
__global__ void cellmm_absmax_kernel(const float* __restrict__ b, int64_t n, int E, int col, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned v = (unsigned)__float_as_int(b[i * E + col]) & 0x7fffffffu;
    m = v > m ? v : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)m, off);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// scale[0] = sigma_b = 2^(15 - ceil(log2(0.104 max|b|)) - wlog2): the largest first-order entry of A,
// f W b sigma_b with |f| <= 0.104 and W <= 2^wlog2, stays below 2^15 (f16 overflows at 65504) and the f16 mid
// parts of ordinary entries stay normal numbers; scale[1] = 1 / (sigma_b 2^6), exact.
__global__ void cellmm_scale_kernel(const unsigned* __restrict__ bmax_bits, int wlog2, int density,
                                    float* __restrict__ scale) {
  float bmax = density ? 1.f : __int_as_float((int)*bmax_bits);
  int ex = 0;
  if (bmax > 0.f && bmax < INFINITY) {
    int eb;
    (void)frexpf(bmax * 0.104f, &eb);  // bmax * 0.104 = m 2^eb, m in [0.5, 1)  ->  ceil(log2) <= eb
    ex = 15 - eb - wlog2;
    ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
  }
  scale[0] = ldexpf(1.f, ex);
  scale[1] = ldexpf(1.f, -ex - 6);
}

// one block of 32 threads per source tile: the signal part of the stage image, b sigma_b (0 for pad sources), for
// column `col` of the row-major (M, E) signal; b == nullptr: b = 1 (density estimation, the denominator of
// normalised rows)
__global__ void __launch_bounds__(CELL_TILE) pack_cellmm_signal_kernel(
    const float* __restrict__ b, int E, int col, const int* __restrict__ perm, const int* __restrict__ gstart,
    const int* __restrict__ gcnt, int64_t n_groups, const float* __restrict__ scale, unsigned char* __restrict__ img) {
  const int64_t g = blockIdx.x;
  const int r = threadIdx.x;
  unsigned char* stage = img + (g / CMM_STAGE_TILES) * CMM_STAGE_BYTES;
  const int q = (int)(g % CMM_STAGE_TILES);
  const bool valid = g < n_groups && r < gcnt[g];
  const float bj = valid ? (b ? b[(int64_t)perm[gstart[g] + r] * E + col] : 1.f) : 0.f;
  *reinterpret_cast<float*>(stage + CMM_B_OFF + (q * CELL_TILE + r) * 4) = bj * scale[0];
}

}  // namespace kmvp
