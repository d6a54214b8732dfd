// Packing kernels of the float64 cell path (layouts in kmvp_cell64.hpp; included by kmvp_product.hip only).
#pragma once
#include "kmvp_cell64.hpp"

namespace kmvp {

// ---- packing (one launch each, HBM-bound) ----------------------------------------------------------------

// cell index of every point (float64 coordinates against the float32 box, clamped to the grid)
__global__ void cell64_keys_kernel(const double* __restrict__ p, int64_t n, int D, CellGrid grid,
                                   unsigned* __restrict__ keys, int* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned key = 0;
  for (int a = 0; a < D; ++a) {
    int c = (int)floor((p[i * D + a] - (double)grid.lo[a]) * (double)grid.inv_h[a]);
    c = c < 0 ? 0 : (c >= grid.g[a] ? grid.g[a] - 1 : c);
    key |= (unsigned)c << (10 * a);
  }
  keys[i] = key;
  vals[i] = (int)i;
}

// one block of 64 threads per target tile; tiles >= n_groups repeat the last tile's cell and are empty
__global__ void __launch_bounds__(CELL64_TILE) pack_cell64_targets_kernel(
    const double* __restrict__ x, const int* __restrict__ perm, const int* __restrict__ gstart,
    const int* __restrict__ gcnt, const unsigned* __restrict__ gkey, int64_t n_groups, int D, CellGrid grid,
    double* __restrict__ xd, double* __restrict__ tmeta, int* __restrict__ slot_of) {
  const int64_t g = blockIdx.x;
  const int r = threadIdx.x;
  const int64_t gg = g < n_groups ? g : n_groups - 1;
  const unsigned key = gkey[gg];
  const bool valid = g < n_groups && r < gcnt[g];
  const int64_t idx = valid ? perm[gstart[g] + r] : 0;
  double d[3] = {0.0, 0.0, 0.0}, c[3] = {0.0, 0.0, 0.0};
  for (int a = 0; a < D; ++a) {
    c[a] = cell64_centre(key, a, grid);
    d[a] = valid ? x[idx * D + a] - c[a] : 0.0;
  }
  *reinterpret_cast<f64x4*>(xd + (g * CELL64_TILE + r) * 4) = f64x4{d[0], d[1], d[2], 0.0};
  if (r == 0) *reinterpret_cast<f64x4*>(tmeta + g * 4) = f64x4{c[0], c[1], c[2], 0.0};
  if (valid) slot_of[idx] = (int)(g * CELL64_TILE + r);
}

// source records in cell order; key_of[p] = cell key of sorted position p; positions >= m: zero tail
__global__ void pack_cell64_sources_kernel(const double* __restrict__ y, const double* __restrict__ b,
                                           const int* __restrict__ perm, const unsigned* __restrict__ key_of,
                                           int64_t m, int64_t m_alloc, int D, CellGrid grid, double* __restrict__ srec) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m_alloc) return;
  f64x4 rec = {0.0, 0.0, 0.0, 0.0};
  if (p < m) {
    const int64_t idx = perm[p];
    const unsigned key = key_of[p];
    for (int a = 0; a < D; ++a) rec[a] = 2.0 * (y[idx * D + a] - cell64_centre(key, a, grid));
    rec[3] = b ? b[idx] : 1.0;
  }
  *reinterpret_cast<f64x4*>(srec + p * 4) = rec;
}

}  // namespace kmvp
