// float64 Gaussian pair loop with exp() range-reduced by grid cells (the float64 counterpart of
// kmvp_cell.hpp; same algebra, no matrix cores: gfx950's fp64 MFMA runs at the vector rate).
//
//     exp(-|x_i - y_j|^2) = U_i(S) * W_j(T) * exp(t_ij),   t_ij = d_i . (2 e_j),   |t| <= 0.05
//         U_i(S) = exp(-|x_i - c_S|^2)      one software exp per (target, source CELL)
//         W_j(T) = exp(e_j.(2 D - e_j))      one software exp per (source, target CELL)
//         exp(t)  = its degree-7 Taylor polynomial (t^8/8! <= 9.7e-16 at the worst corner pair)
//
// The difference-form kernel spends ~23 fp64 instructions per pair (6 for the squared distance, ~16 for
// the table-and-polynomial exp of kexp_neg_f64, 1 FMA); here a pair costs 3 (t) + 7 (Horner) + 1 (FMA)
// (kmvp_lowd.hpp's count per pair: SQ_INSTS_VALU 23.4 against 13.7 here, padding and per-cell work included).
// Cells of side sqrt(2 * 0.05 / D) (0.18 for D = 3).
//
// Mapping (as lowd_kernel): two targets per lane, a wavefront = one tile of 128 targets of ONE cell (cells
// are padded to whole tiles), sources wave-uniform.  Per source cell the wave computes U, then walks
// the cell's sources 64 at a time: W_j b_j with the source on the lane, then for each source the record
// (2e, b) arrives through wave-uniform loads (scalar cache) and W_j b_j through a broadcast LDS read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_cell.hpp"  // CellGrid
#include "kmvp_lowd.hpp"  // kexp_neg_f64, block_to_work, WAVES_PER_BLOCK

namespace kmvp {

constexpr int CELL64_TPL = 2;                  // targets per lane: halves the scalar-cache and LDS traffic per pair
constexpr int CELL64_TILE = 64 * CELL64_TPL;  // targets per tile = one wavefront
constexpr double CELL64_T_MAX = 0.05;  // bound on |2 d.e|

__host__ __device__ inline double cell64_centre(unsigned key, int a, const CellGrid& grid) {
  return (double)grid.lo[a] + ((double)((key >> (10 * a)) & 1023u) + 0.5) * (double)grid.h[a];
}

struct Cell64Args {
  const double* xd;       // targets [n_slots][4]: d_x, d_y, d_z, 0 (cell order, cells padded to tiles of 64)
  const double* tmeta;    // target tiles [n_slots / 64][4]: c_x, c_y, c_z, 0
  const double* srec;     // sources [m + 64][4] in cell order: 2 e_x, 2 e_y, 2 e_z, b (64 zero records behind the last)
  const int* scell;       // source cells [n_scells][2]: first record, count
  const double* scentre;  // source cells [n_scells][4]: c_x, c_y, c_z, 0
  double* part;           // partial sums [segments][NE][n_slots], NE = 2 for normalised rows
  int64_t n_slots;
  int n_scells;
  int seg_cells;          // source cells per segment
  int segments;
  int tile_blocks;
};

typedef double f64x4 __attribute__((ext_vector_type(4)));

// exp(t) for |t| <= CELL64_T_MAX by Horner (degree 7: t^8/8! <= 9.7e-16 for a corner-to-corner pair of cells,
// below 1e-18 for a typical pair -- under the rounding of the sums themselves)
__device__ __forceinline__ double cell64_exp(double t) {
  double p = fma(t, 1.0 / 5040.0, 1.0 / 720.0);
  p = fma(p, t, 1.0 / 120.0);
  p = fma(p, t, 1.0 / 24.0);
  p = fma(p, t, 1.0 / 6.0);
  p = fma(p, t, 0.5);
  p = fma(p, t, 1.0);
  return fma(p, t, 1.0);
}

// one source against the lane's targets (NE = 2: w[0] = W_j b_j for the numerator, w[1] = W_j for the denominator)
template <int NE>
__device__ __forceinline__ void cell64_pair(const double (&d)[CELL64_TPL][3], const f64x4 rec, const double (&w)[NE],
                                            double (&acc)[CELL64_TPL][NE]) {
#pragma unroll
  for (int k = 0; k < CELL64_TPL; ++k) {
    const double t = fma(d[k][0], rec[0], fma(d[k][1], rec[1], d[k][2] * rec[2]));
    const double p = cell64_exp(t);
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[k][e] = fma(p, w[e], acc[k][e]);
  }
}

// SIG: SIG_PRODUCT (density = product with b = 1, set by the packer) or SIG_NORM (numerator and denominator in one
// sweep: a second sum with W_j alone)
template <int SIG>
__global__ void __launch_bounds__(BLOCK_THREADS) cell64_kernel(const Cell64Args a) {
  __shared__ double exp_tab[64];
  constexpr int NE = SIG == SIG_NORM ? 2 : 1;
  __shared__ double wsh[WAVES_PER_BLOCK][64][NE];  // W_j b_j (and W_j) of the wave's current 64 sources (broadcast reads)
  if (threadIdx.x < 64) exp_tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
  __syncthreads();
  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t tile = (int64_t)tb * WAVES_PER_BLOCK + wave;
  const f64x4 ct = *reinterpret_cast<const f64x4*>(a.tmeta + tile * 4);  // wave-uniform
  double d[CELL64_TPL][3], acc[CELL64_TPL][NE];
#pragma unroll
  for (int k = 0; k < CELL64_TPL; ++k) {
    const f64x4 dv = *reinterpret_cast<const f64x4*>(a.xd + (tile * CELL64_TILE + 64 * k + lane) * 4);
    d[k][0] = dv[0];
    d[k][1] = dv[1];
    d[k][2] = dv[2];
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[k][e] = 0.0;
  }

  const int c_begin = seg * a.seg_cells;
  const int c_end = min(c_begin + a.seg_cells, a.n_scells);
  for (int sc = c_begin; sc < c_end; ++sc) {
    const int first = __builtin_amdgcn_readfirstlane(a.scell[2 * sc]);
    const int count = __builtin_amdgcn_readfirstlane(a.scell[2 * sc + 1]);
    const f64x4 cs = *reinterpret_cast<const f64x4*>(a.scentre + (int64_t)sc * 4);
    const double Dx = ct[0] - cs[0], Dy = ct[1] - cs[1], Dz = ct[2] - cs[2];
    double accc[CELL64_TPL][NE];
#pragma unroll
    for (int k = 0; k < CELL64_TPL; ++k)
#pragma unroll
      for (int e = 0; e < NE; ++e) accc[k][e] = 0.0;
    for (int base = 0; base < count; base += 64) {
      const int nj = min(64, count - base);
      // W_j b_j with the source on the lane (records behind the cell's last one belong to the next
      // cell or are the zero tail: masked by the lane test)
      const f64x4 mine = *reinterpret_cast<const f64x4*>(a.srec + ((int64_t)first + base + lane) * 4);
      // e.(2D - e) with e = rec/2:  rec.(D - rec/4)
      const double arg = fma(mine[0], fma(mine[0], -0.25, Dx), fma(mine[1], fma(mine[1], -0.25, Dy), mine[2] * fma(mine[2], -0.25, Dz)));
      const double wj = lane < nj ? kexp_neg_f64(-arg, exp_tab) : 0.0;
      wsh[wave][lane][0] = wj * mine[3];
      if constexpr (NE == 2) wsh[wave][lane][1] = wj;
      __builtin_amdgcn_wave_barrier();
      const double (*wbp)[NE] = wsh[wave];
      const double* rec0 = a.srec + ((int64_t)first + base) * 4;  // wave-uniform
      // records two at a time through the scalar cache, the next pair requested before this one is used
      int j = 0;
      if (nj >= 2) {
        f64x4 n0 = *reinterpret_cast<const f64x4*>(rec0), n1 = *reinterpret_cast<const f64x4*>(rec0 + 4);
        for (; j + 2 <= nj; j += 2) {
          const f64x4 r0 = n0, r1 = n1;
          if (j + 4 <= nj) {  // wave-uniform
            n0 = *reinterpret_cast<const f64x4*>(rec0 + (j + 2) * 4);
            n1 = *reinterpret_cast<const f64x4*>(rec0 + (j + 3) * 4);
          }
          cell64_pair<NE>(d, r0, wbp[j + 0], accc);
          cell64_pair<NE>(d, r1, wbp[j + 1], accc);
        }
      }
      if (j < nj) {
        const f64x4 r0 = *reinterpret_cast<const f64x4*>(rec0 + j * 4);
        cell64_pair<NE>(d, r0, wbp[j], accc);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < CELL64_TPL; ++k) {
      const double ux = d[k][0] + Dx, uy = d[k][1] + Dy, uz = d[k][2] + Dz;
      const double U = kexp_neg_f64(fma(ux, ux, fma(uy, uy, uz * uz)), exp_tab);
#pragma unroll
      for (int e = 0; e < NE; ++e) acc[k][e] = fma(U, accc[k][e], acc[k][e]);
    }
  }
#pragma unroll
  for (int k = 0; k < CELL64_TPL; ++k)
#pragma unroll
    for (int e = 0; e < NE; ++e)
      a.part[((int64_t)seg * NE + e) * a.n_slots + tile * CELL64_TILE + 64 * k + lane] = acc[k][e];
}

hipError_t launch_cell64_gaussian(int sig, const Cell64Args& args, dim3 grid, hipStream_t stream, const char** kernel_name);

}  // namespace kmvp
