// Packing kernels of the cell-reduced Gaussian path (layouts in kmvp_cell.hpp; included by
// kmvp_product.hip only).
#pragma once
#include "kmvp_cell.hpp"

namespace kmvp {

// cell index of every point: 10 bits per axis, clamped to the grid (the box was measured on these points)
__global__ void cell_keys_kernel(const float* __restrict__ p, int64_t n, int D, CellGrid grid,
                                 unsigned* __restrict__ keys, int* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned key = 0;
  for (int a = 0; a < D; ++a) {
    int c = (int)floorf((p[i * D + a] - grid.lo[a]) * grid.inv_h[a]);
    c = c < 0 ? 0 : (c >= grid.g[a] ? grid.g[a] - 1 : c);
    key |= (unsigned)c << (10 * a);
  }
  keys[i] = key;
  vals[i] = (int)i;
}

// one block of 32 threads per target tile.  tiles >= n_groups are pad tiles (they repeat the
// last tile's cell, so a wave never sees a cell change because of them)
__global__ void __launch_bounds__(CELL_TILE) pack_cell_targets_kernel(
    const float* __restrict__ x, const int* __restrict__ perm, const int* __restrict__ gstart,
    const int* __restrict__ gcnt, const unsigned* __restrict__ gkey, int64_t n_groups, int D, CellGrid grid,
    float* __restrict__ xd, float* __restrict__ tmeta, int* __restrict__ slot_of) {
  const int64_t g = blockIdx.x;
  const int r = threadIdx.x;
  const int64_t gg = g < n_groups ? g : n_groups - 1;
  const unsigned key = gkey[gg];
  const bool valid = g < n_groups && r < gcnt[g];
  const int64_t idx = valid ? perm[gstart[g] + r] : 0;
  float d[4] = {0.f, 0.f, 0.f, 0.f}, c[3] = {0.f, 0.f, 0.f};
  for (int a = 0; a < D; ++a) {
    c[a] = cell_centre(key, a, grid);
    d[a] = valid ? x[idx * D + a] - c[a] : 0.f;
  }
  *reinterpret_cast<f32x4*>(xd + (g * CELL_TILE + r) * 4) = f32x4{d[0], d[1], d[2], 0.f};
  const bool empty = g >= n_groups || gcnt[g] == 0;
  if (r == 0)
    *reinterpret_cast<f32x4*>(tmeta + g * 4) = f32x4{c[0], c[1], c[2], __builtin_bit_cast(float, key | (empty ? 1u << 30 : 0u))};
  if (valid) slot_of[idx] = (int)(g * CELL_TILE + r);
}

// one block of 32 threads per source tile (m_stages * CELL_STAGE_TILES of them); b == nullptr: density
__global__ void __launch_bounds__(CELL_TILE) pack_cell_sources_kernel(
    const float* __restrict__ y, const float* __restrict__ b, const int* __restrict__ perm,
    const int* __restrict__ gstart, const int* __restrict__ gcnt, const unsigned* __restrict__ gkey,
    int64_t n_groups, int D, CellGrid grid, unsigned char* __restrict__ img) {
  const int64_t g = blockIdx.x;
  const int r = threadIdx.x;
  unsigned char* stage = img + (g / CELL_STAGE_TILES) * CELL_STAGE_BYTES;
  const int q = (int)(g % CELL_STAGE_TILES);
  const bool real_tile = g < n_groups;
  const unsigned key = real_tile ? gkey[g] : 0u;
  const bool valid = real_tile && r < gcnt[g];
  const int64_t idx = valid ? perm[gstart[g] + r] : 0;
  float e[3] = {0.f, 0.f, 0.f}, c[3] = {0.f, 0.f, 0.f};
  for (int a = 0; a < D; ++a) {
    c[a] = cell_centre(key, a, grid);
    e[a] = valid ? y[idx * D + a] - c[a] : 0.f;
  }
  const float bj = valid ? (b ? b[idx] : 1.f) : 0.f;
  bf16x8 lo8, hi8;
  {
    float f[16];
    f[0] = valid ? 1.f : 0.f;
    for (int a = 0; a < 3; ++a) {
      const float e2 = 2.f * e[a];
      const float eh = (float)(__bf16)e2;
      const float em = (float)(__bf16)(e2 - eh);
      f[1 + 3 * a] = eh;  // x d_h
      f[2 + 3 * a] = em;  // x d_h
      f[3 + 3 * a] = eh;  // x d_m
    }
    f[10] = 2.f * e[0] * e[0];
    f[11] = 2.f * e[1] * e[1];
    f[12] = 2.f * e[2] * e[2];
    f[13] = 4.f * e[0] * e[1];
    f[14] = 4.f * e[0] * e[2];
    f[15] = 4.f * e[1] * e[2];
    for (int j = 0; j < 8; ++j) {
      lo8[j] = (__bf16)f[j];
      hi8[j] = (__bf16)f[8 + j];
    }
  }
  unsigned char* tile = stage + q * CELL_TILE_BYTES;
  *reinterpret_cast<bf16x8*>(tile + r * 32) = lo8;
  *reinterpret_cast<bf16x8*>(tile + r * 32 + 16) = hi8;
  *reinterpret_cast<f32x4*>(tile + CELL_A_BYTES + r * 16) = f32x4{e[0], e[1], e[2], bj};
  if (r == 0)
    *reinterpret_cast<f32x4*>(stage + CELL_HDR_OFF + q * 16) =
        f32x4{c[0], c[1], c[2], __builtin_bit_cast(float, real_tile ? (int)key : -1)};
}

// sums[e][i] = sorted[e][slot_of[i]]: the way back from cell order to the caller's order (the segments
// were summed in cell order first, coalesced)
__global__ void gather_cells_kernel(const double* __restrict__ sorted, const int* __restrict__ slot_of,
                                    double* __restrict__ sums, int64_t n, int64_t n_slots, int NE) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)NE * n) return;
  const int64_t e = q / n, i = q % n;
  sums[q] = sorted[e * n_slots + slot_of[i]];
}

}  // namespace kmvp
