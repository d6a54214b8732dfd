// Low-dimensional pair-loop kernels for gfx950 (CDNA4): a_i = sum_j k(x_i, y_j) b_j.
//
// Arithmetic restated from the reference's dense bruteforce plugin
// (kernel_matrix_benchmarks/algorithms/bruteforce.py): squared distances in the
// difference form of :53-54, kernel functions of :18-22 / :8-15, the four
// query() branches of :130-153.  Nothing here is a translation: the reference
// materialises the (N,M) matrix and calls BLAS, this file never forms it.
//
// Mapping to the hardware
//  * lanes = targets.  One 64-lane wavefront owns one target tile of 64*T points;
//    each lane keeps T targets (coordinates + partial sums) in VGPRs for the
//    whole launch, so a source record is reused 64*T times per wave.
//  * sources are wave-uniform.  They are stored as packed records
//    [y_0..y_{D-1}, b_0..b_{E-1}, pad] (R dwords, written by pack_sources with the
//    kernel's constant folded into the coordinates) and reach the VALU either
//      FEED 0: through the scalar data cache into SGPRs (s_load_dwordxN) -- a
//              source costs no VGPR, no LDS cycle and no vector-memory issue; or
//      FEED 1: through an LDS tile the workgroup fills with coalesced 16-byte
//              global loads, read back as broadcast ds_read_b128.
//  * the launch is a 2-D decomposition: target tile-blocks x source segments.
//    blockIdx is remapped so that all blocks of one source segment share
//    blockIdx % 8, i.e. one XCD and one L2 (placement is a speed matter only).
//  * sums: fp32 inside a chunk of `chunk` sources, folded into an fp64 per-target
//    running sum between chunks; each (segment, target) partial is written once
//    as fp64 and the segments are added in index order by reduce_segments, so
//    results are bitwise reproducible run to run (no atomics).
//  * roofline: the loop is bound by VALU issue, not by HBM: per pair
//    3D-1 full-rate ops for the squared distance, one (gaussian, 1/r) or two
//    (exp(-r)) quarter-rate transcendentals, E (+1 normalised) FMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmvp {

enum : int { K_GAUSSIAN = 0, K_ABSEXP = 1, K_INVDIST = 2, K_EXPDOT = 3, K_GAUSSIAN_SHIFTED = 4 };
// K_EXPDOT: host-side only (run_product) and the bf16 matrix-core kernels; K_GAUSSIAN_SHIFTED: the bf16 matrix-core kernels
// only -- the Gaussian with exp(<x,y>)'s per-target running shift, taken when targets != sources (kmvp_mfma.hpp)

// signal mode of a launch
enum : int {
  SIG_PRODUCT = 0,  // a = K b                 (bruteforce.py:153)
  SIG_NORM = 1,     // numerator K b and denominator K 1 in one sweep (bruteforce.py:142-145)
  SIG_DENSITY = 2   // a = K 1, no signal read (bruteforce.py:150)
};

template <typename real>
struct LowdArgs {
  const real* xs;    // targets, SoA: xs[d * n_pad + i], the caller's coordinates
  const real* rec;   // source records, [m_pad][R]
  double* part;      // partial sums [segments][NE][n_pad]
  int64_t n;         // targets
  int64_t n_pad;     // targets rounded up to the tile-block size
  int64_t m_pad;     // sources rounded up to the batch size (pad records contribute 0)
  int64_t seg_len;   // sources per segment (multiple of the batch size)
  int segments;      // source segments (grid = tile_blocks * segments)
  int tile_blocks;   // target tile-blocks
  int chunk;         // sources per fp32 chunk
  int64_t j_offset;  // global index of source 0 (sharding)
  int64_t m_total;   // global number of sources (inverse-distance zero pattern)
};

__device__ __forceinline__ float kexp2(float v) { return __builtin_amdgcn_exp2f(v); }

// kernel value from the squared distance s of the caller's coordinates (difference form,
// bruteforce.py:53-54).  The constant that turns exp() into the hardware's exp2() multiplies
// the squared distance, not the coordinates: scaling x and y before the subtraction would
// round every coordinate by eps |x| and lose the pairs of clouds far from the origin.
template <int KERNEL>
__device__ __forceinline__ float kval(float s) {
  if constexpr (KERNEL == K_GAUSSIAN) {
    return kexp2(s * -1.4426950408889634f);  // exp(-s) = 2^(-s log2 e)
  } else if constexpr (KERNEL == K_ABSEXP) {
    return kexp2(__builtin_amdgcn_sqrtf(s) * -1.4426950408889634f);
  } else {
    return __builtin_amdgcn_rsqf(s);  // 1/sqrt(0) = inf, as the reference's 1/np.sqrt
  }
}
// exp(-s) in double precision for s >= 0 without the math library's general-purpose exp
// (gfx950 has no fp64 transcendental instruction; ocml's exp costs ~30 fp64 operations):
//   -s = (64 m + j) ln2/64 + r ,  |r| <= ln2/128
//   exp(-s) = 2^m * 2^(j/64) * exp(r)
// with 2^(j/64) from a 64-entry table in LDS (filled once per workgroup) and exp(r) as its
// degree-5 Taylor polynomial (r^6/720 < 4e-17).  Two-constant Cody-Waite reduction keeps r
// exact to ~1e-19 for s < 800; larger s (incl. the +inf of pad records) are clamped there and
// return exactly 0, like exp(): ldexp underflows below 2^-1074.
__device__ __forceinline__ double kexp_neg_f64(double s, const double* __restrict__ tab) {
  s = fmin(s, 800.0);
  const double n = rint(s * (-92.332482616893657));      // -64 / ln 2
  double r = fma(n, -0.010830424696905538, -s);  // ln2/64, high 30 bits: n * hi is exact
  r = fma(n, 6.563929801064195e-13, r);          // ln2/64 - hi = -6.56e-13
  const int ni = (int)n;
  const double t = tab[ni & 63];
  double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(t * p, ni >> 6);
}

template <int KERNEL>
__device__ __forceinline__ double kval(double s, const double* __restrict__ tab) {
  if constexpr (KERNEL == K_GAUSSIAN) {
    return kexp_neg_f64(s, tab);
  } else if constexpr (KERNEL == K_ABSEXP) {
    // sqrt(s) = s / sqrt(s) from the same estimate-and-one-step as 1/sqrt(s) below (the compiler's IEEE sqrt spends about
    // twice the instructions on scaling and a second correction; s is a sum of squares: never negative, denormal only
    // where the result is far below every tolerance); s = 0 and s = inf (pad records) pass through
    const double y0 = __builtin_amdgcn_rsq(s);
    const double e = fma(-s * y0, y0, 1.0);
    const double r = s * fma(y0 * e, fma(e, 0.375, 0.5), y0);
    return kexp_neg_f64((s == 0.0 || s == (double)INFINITY) ? s : r, tab);
  } else {
    // 1/sqrt(s): hardware estimate (v_rsq_f64, ~26 bits) + ONE Newton step in its third-order form
    // y = y0 (1 + e/2 + 3 e^2/8), e = 1 - s y0^2 (|e| <= 2^-25: the neglected term 5 e^3/16 is below 2^-76); the
    // estimate is already exact for s = 0 (inf, as the reference's 1/np.sqrt) and s = inf (0, pad records)
    const double y0 = __builtin_amdgcn_rsq(s);
    const double e = fma(-s * y0, y0, 1.0);
    const double y = fma(y0 * e, fma(e, 0.375, 0.5), y0);
    return (s == 0.0 || s == (double)INFINITY) ? y0 : y;
  }
}
template <int KERNEL>
__device__ __forceinline__ float kval(float s, const double*) {
  return kval<KERNEL>(s);
}

// the constant the matrix-core paths (kmvp_fast.hpp, kmvp_cfast.hpp, kmvp_mfma.hpp) multiply
// CENTRED coordinates with, so that their MFMA tile is the exponent of exp2() directly
template <int KERNEL, typename real>
__host__ __device__ inline real coord_scale() {
  if constexpr (sizeof(real) == 8) return (real)1;
  if constexpr (KERNEL == K_GAUSSIAN) return (real)1.2011224087864498;  // sqrt(log2 e)
  if constexpr (KERNEL == K_ABSEXP) return (real)1.4426950408889634;    // log2 e
  return (real)1;
}

// blockIdx -> (tile-block, segment).  Blocks are dealt round-robin over the 8 XCDs, so
// blockIdx % 8 labels the blocks that share an L2.  With segments % 8 == 0, XCD group g
// only ever touches segments g, g+8, g+16, ... and works through them ONE AT A TIME
// (the segment index varies slowest along the group's block sequence): the segment it
// is streaming (<= ~2 MiB, see choose_segments) stays resident in its 4 MiB L2, so
// HBM sees each source record about once per launch.  Placement only affects speed.
__device__ __forceinline__ void block_to_work(int bid, int segments, int tile_blocks, int& tb,
                                              int& seg) {
  if ((segments & 7) == 0) {
    const int q = bid >> 3;
    seg = (bid & 7) + 8 * (q / tile_blocks);
    tb = q % tile_blocks;
  } else {
    seg = bid % segments;
    tb = bid / segments;
  }
}

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK_THREADS = 64 * WAVES_PER_BLOCK;
constexpr int LDS_TILE = 256;  // source records per LDS tile (FEED 1)

template <int D, int E, int SIG>
struct RecLayout {
  static constexpr int EB = (SIG == SIG_DENSITY) ? 0 : E;  // signal dwords in a record
  static constexpr int R = ((D + EB + 3) / 4) * 4;         // record length in elements
  static constexpr int NE = (SIG == SIG_DENSITY) ? 1 : (SIG == SIG_NORM ? E + 1 : E);
};

// One (target, source) interaction for all T targets of the lane.
template <int KERNEL, int D, int E, int SIG, int T, bool CHECK_DIAG, typename real>
__device__ __forceinline__ void interact(const real (&x)[T][D], real (&acc)[T][RecLayout<D, E, SIG>::NE],
                                         const real* __restrict__ r, const int64_t (&jz)[T],
                                         int64_t j_local, const double* __restrict__ tab) {
  using L = RecLayout<D, E, SIG>;
  real y[D];
#pragma unroll
  for (int d = 0; d < D; ++d) y[d] = r[d];
  real b[L::EB > 0 ? L::EB : 1];
#pragma unroll
  for (int e = 0; e < L::EB; ++e) b[e] = r[D + e];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    real df = x[t][0] - y[0];
    real s = df * df;
#pragma unroll
    for (int d = 1; d < D; ++d) {
      df = x[t][d] - y[d];
      s = fma(df, df, s);
    }
    real k = kval<KERNEL>(s, tab);
    if constexpr (CHECK_DIAG) k = (j_local == jz[t]) ? (real)0 : k;
    if constexpr (SIG == SIG_DENSITY) {
      acc[t][0] += k;
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) acc[t][e] = fma(k, b[e], acc[t][e]);
      if constexpr (SIG == SIG_NORM) acc[t][E] += k;
    }
  }
}

template <int KERNEL, int D, int E, int SIG, int T, int FEED, typename real>
__global__ void __launch_bounds__(BLOCK_THREADS) lowd_kernel(const LowdArgs<real> a) {
  using L = RecLayout<D, E, SIG>;
  constexpr int R = L::R;
  constexpr int NE = L::NE;
  constexpr int U = 4;  // sources per batch; segments start on batch boundaries
  constexpr bool F32 = sizeof(real) == 4;

  // fp64 only: table 2^(j/64) of kexp_neg_f64
  __shared__ double exp_tab_lds[F32 ? 1 : 64];
  const double* exp_tab = exp_tab_lds;
  if constexpr (!F32) {
    if (threadIdx.x < 64) exp_tab_lds[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
  }

  int tb, seg;
  block_to_work((int)blockIdx.x, a.segments, a.tile_blocks, tb, seg);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // first target of this wave's tile
  const int64_t i0 = ((int64_t)tb * WAVES_PER_BLOCK + wave) * (64 * T);

  real x[T][D];
  int64_t jz[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int64_t i = i0 + t * 64 + lane;  // < n_pad by construction; pad targets are 0
#pragma unroll
    for (int d = 0; d < D; ++d) x[t][d] = a.xs[(int64_t)d * a.n_pad + i];
    if constexpr (KERNEL == K_INVDIST) {
      // bruteforce.py:13-14: flat index i*M+j is zeroed when it is a multiple of M+1,
      // i.e. column (i mod (M+1)) if that is < M.  Local column = global - j_offset.
      const int64_t g = i % (a.m_total + 1);
      jz[t] = (g < a.m_total) ? g - a.j_offset : (int64_t)-1;
    } else {
      jz[t] = -1;
    }
  }
  // wave-uniform bounds of the zero columns of this tile (conservative when the
  // mod wraps inside the tile: then every batch is checked)
  int64_t jz_lo = 0, jz_hi = -1;
  if constexpr (KERNEL == K_INVDIST) {
    const int64_t g_lo = i0 % (a.m_total + 1);
    const int64_t g_hi = g_lo + (64 * T - 1);
    if (g_hi <= a.m_total) {
      jz_lo = g_lo - a.j_offset;
      jz_hi = g_hi - a.j_offset;
    } else {
      jz_lo = INT64_MIN / 2;
      jz_hi = INT64_MAX / 2;
    }
  }

  const int64_t seg_begin = (int64_t)seg * a.seg_len;
  int64_t seg_end = seg_begin + a.seg_len;
  if (seg_end > a.m_pad) seg_end = a.m_pad;

  double accd[T][NE];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int e = 0; e < NE; ++e) accd[t][e] = 0.0;

  real acc[T][NE];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[t][e] = 0;

  auto fold = [&]() {
    if constexpr (F32) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          accd[t][e] += (double)acc[t][e];
          acc[t][e] = 0;
        }
    }
  };

  if constexpr (FEED == 0) {
    // ---- scalar-cache stream: the record address depends on blockIdx and the loop
    // counter only, so the loads are s_load_dwordxN into SGPRs.  The next batch is
    // requested before the current one is consumed (the record array carries one
    // spare batch at its end, so the last prefetch stays in bounds).
    const real* __restrict__ segp = a.rec + seg_begin * R;
    const int seg_n = (int)(seg_end - seg_begin);
    const int jz_lo32 = (int)max((int64_t)-1, min(jz_lo - seg_begin, (int64_t)INT32_MAX));
    const int jz_hi32 = (int)max((int64_t)-2, min(jz_hi - seg_begin, (int64_t)INT32_MAX));
    int64_t jzl[T];
#pragma unroll
    for (int t = 0; t < T; ++t) jzl[t] = jz[t] - seg_begin;
    // two SGPR batches in ping-pong: batch B is requested before batch A is consumed
    // and vice versa (segments hold a whole number of 2*U sources).
    auto run_batch = [&](const real (&rb)[U * R], int j) {
      bool check = false;
      if constexpr (KERNEL == K_INVDIST) check = (j + U - 1 >= jz_lo32) && (j <= jz_hi32);
      if (check) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          interact<KERNEL, D, E, SIG, T, true, real>(x, acc, rb + u * R, jzl, (int64_t)(j + u), exp_tab);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
          interact<KERNEL, D, E, SIG, T, false, real>(x, acc, rb + u * R, jzl, (int64_t)(j + u), exp_tab);
      }
    };
    real ra[U * R], rb[U * R];
#pragma unroll
    for (int q = 0; q < U * R; ++q) ra[q] = segp[q];
    for (int c0 = 0; c0 < seg_n; c0 += a.chunk) {
      const int c1 = min(c0 + a.chunk, seg_n);
      for (int j = c0; j < c1; j += 2 * U) {
        const real* __restrict__ pb = segp + (int64_t)(j + U) * R;
#pragma unroll
        for (int q = 0; q < U * R; ++q) rb[q] = pb[q];
        run_batch(ra, j);
        const real* __restrict__ pa = segp + (int64_t)(j + 2 * U) * R;
#pragma unroll
        for (int q = 0; q < U * R; ++q) ra[q] = pa[q];
        run_batch(rb, j + U);
      }
      fold();
    }
  } else {
    // ---- LDS-staged tiles: coalesced 16-byte loads of LDS_TILE records per block,
    // double buffered (one barrier per tile), broadcast reads in the pair loop.
    static_assert((LDS_TILE * R * sizeof(real)) % (16 * BLOCK_THREADS) == 0 ||
                      (LDS_TILE * R * sizeof(real)) < (16 * BLOCK_THREADS),
                  "tile must be a whole number of 16-byte pieces per thread");
    constexpr int TILE_BYTES = LDS_TILE * R * (int)sizeof(real);
    constexpr int PIECES = (TILE_BYTES + 16 * BLOCK_THREADS - 1) / (16 * BLOCK_THREADS);
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2][TILE_BYTES];
    const int64_t n_tiles = (seg_end - seg_begin + LDS_TILE - 1) / LDS_TILE;
    const unsigned char* gbase = reinterpret_cast<const unsigned char*>(a.rec + seg_begin * R);
    const int64_t seg_bytes = (seg_end - seg_begin) * R * (int64_t)sizeof(real);
    uint4 stage[PIECES];
    auto gload = [&](int64_t tile) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int64_t off = tile * TILE_BYTES + ((int64_t)p * BLOCK_THREADS + threadIdx.x) * 16;
        stage[p] = (off < seg_bytes && (p * BLOCK_THREADS + (int)threadIdx.x) * 16 < TILE_BYTES)
                       ? *reinterpret_cast<const uint4*>(gbase + off)
                       : make_uint4(0, 0, 0, 0);
      }
    };
    gload(0);
    int since_fold = 0;
    for (int64_t tile = 0; tile < n_tiles; ++tile) {
      const int buf = (int)(tile & 1);
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int o = (p * BLOCK_THREADS + (int)threadIdx.x) * 16;
        if (o < TILE_BYTES) *reinterpret_cast<uint4*>(&lds_raw[buf][o]) = stage[p];
      }
      __syncthreads();
      if (tile + 1 < n_tiles) gload(tile + 1);
      const int64_t jt = seg_begin + tile * LDS_TILE;
      int cnt = LDS_TILE;
      if (jt + cnt > seg_end) cnt = (int)(seg_end - jt);
      const real* lrec = reinterpret_cast<const real*>(&lds_raw[buf][0]);
      for (int jj = 0; jj < cnt; jj += U) {
        const int64_t j = jt + jj;
        bool check = false;
        if constexpr (KERNEL == K_INVDIST) check = (j + U - 1 >= jz_lo) && (j <= jz_hi);
        if (check) {
#pragma unroll
          for (int u = 0; u < U; ++u)
            interact<KERNEL, D, E, SIG, T, true, real>(x, acc, lrec + (jj + u) * R, jz, j + u, exp_tab);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            interact<KERNEL, D, E, SIG, T, false, real>(x, acc, lrec + (jj + u) * R, jz, j + u, exp_tab);
        }
      }
      since_fold += LDS_TILE;
      if (since_fold >= a.chunk) {
        fold();
        since_fold = 0;
      }
    }
    fold();
  }

  // ---- one fp64 partial per (segment, output column, target); coalesced over targets
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int64_t i = i0 + t * 64 + lane;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double v;
      if constexpr (F32) v = accd[t][e];
      else v = (double)acc[t][e];
      a.part[((int64_t)seg * NE + e) * a.n_pad + i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------
// layout + epilogue kernels (HBM-bound, tiny next to the pair loop)

// Beyond the specialised kernel (D <= 8, compile-time D):
// D <= 128 with BOTH clouds padded with zero coordinates to rows of 8 DCH entries (pad_rows_kernel), so
// that the loop over dimensions has a compile-time length and no guards: the target's coordinates
// live in REGISTERS, a source row arrives
// through wave-uniform loads (scalar cache), fp32 sums are folded into fp64 every MID_CHUNK sources,
// signal columns go in blocks of 8.
constexpr int MID_CHUNK = 64;
// Signal columns per pass over the sources (the squared distances are recomputed per pass): 8, or 32 where the
// registers allow it -- float32 rows of up to 64 coordinates with more than 8 columns (E = 64 at D = 64: two passes
// instead of eight).  The host pads the signal rows to a multiple of it.
__host__ __device__ constexpr int lowd_mid_colblock(int real_bytes, int D, int E) {
  return (real_bytes == 4 && D <= 64 && E > 8) ? 32 : 8;
}

template <int KERNEL, int SIG, typename real, int DCH, int EBW = 8>
__global__ void __launch_bounds__(BLOCK_THREADS) lowd_mid_kernel(
    const real* __restrict__ x /* (N, 8 DCH) */, const real* __restrict__ y /* (M, 8 DCH) */,
    const real* __restrict__ b /* (M, EP), EP = EBW ceil(E / EBW), zero padded; unused for density */,
    double* __restrict__ part, int64_t n, int64_t n_pad, int64_t m, int E, int EP, int NE, int segments,
    int64_t seg_len, int64_t j_offset, int64_t m_total) {
  constexpr int DMAX = 8 * DCH;
  __shared__ double exp_tab_lds[sizeof(real) == 4 ? 1 : 64];
  const double* exp_tab = exp_tab_lds;
  if constexpr (sizeof(real) == 8) {
    if (threadIdx.x < 64) exp_tab_lds[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
  }
  const int seg = (int)(blockIdx.x % segments);
  const int64_t tb = blockIdx.x / segments;
  const int64_t i = tb * BLOCK_THREADS + threadIdx.x;
  const int64_t ic = i < n ? i : n - 1;
  real xr[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) xr[d] = x[ic * DMAX + d];
  int64_t jz = -1;
  if constexpr (KERNEL == K_INVDIST) {
    const int64_t g = ic % (m_total + 1);
    jz = (g < m_total) ? g - j_offset : (int64_t)-1;
  }
  const int64_t j0 = (int64_t)seg * seg_len;
  int64_t j1 = j0 + seg_len;
  if (j1 > m) j1 = m;
  const int blocks = (SIG == SIG_DENSITY) ? 1 : EP / EBW;  // passes over the sources, EBW signal columns each
  for (int blk = 0; blk < blocks; ++blk) {
    double accd[EBW], dend = 0.0;
#pragma unroll
    for (int q = 0; q < EBW; ++q) accd[q] = 0.0;
    for (int64_t jc = j0; jc < j1; jc += MID_CHUNK) {
      const int64_t jend = jc + MID_CHUNK < j1 ? jc + MID_CHUNK : j1;
      real acc[EBW], den = (real)0;
#pragma unroll
      for (int q = 0; q < EBW; ++q) acc[q] = (real)0;
      for (int64_t j = jc; j < jend; ++j) {
        const real* __restrict__ yrow = y + j * DMAX;  // wave-uniform address
        real s = 0;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
          const real df = xr[d] - yrow[d];  // padded dimensions: 0 - 0
          s = fma(df, df, s);
        }
        real k = kval<KERNEL>(s, exp_tab);
        if constexpr (KERNEL == K_INVDIST) k = (j == jz) ? (real)0 : k;
        if constexpr (SIG != SIG_DENSITY) {
          const real* __restrict__ brow = b + j * EP + EBW * blk;
#pragma unroll
          for (int q = 0; q < EBW; ++q) acc[q] = fma(k, brow[q], acc[q]);  // padded columns: k * 0
        }
        if constexpr (SIG != SIG_PRODUCT) den += k;  // denominator / density
      }
#pragma unroll
      for (int q = 0; q < EBW; ++q) accd[q] += (double)acc[q];
      dend += (double)den;
    }
    if (i < n_pad) {
      if constexpr (SIG != SIG_DENSITY) {
#pragma unroll
        for (int q = 0; q < EBW; ++q)
          if (EBW * blk + q < E) part[((int64_t)seg * NE + EBW * blk + q) * n_pad + i] = accd[q];
      }
      if (SIG != SIG_PRODUCT && blk == 0) part[((int64_t)seg * NE + (NE - 1)) * n_pad + i] = dend;
    }
  }
}

// Any D beyond LOWD_MID_MAX_D: the same padded rows (to multiples of BIG_CHUNK coordinates), walked in
// chunks of BIG_CHUNK dimensions for a batch of BIG_BATCH sources at a time -- the target's chunk sits in
// registers (reloaded per batch from its own row, L1/L2 resident), the squared distances of the batch
// accumulate in registers across the chunks, source chunks arrive through wave-uniform loads.
constexpr int BIG_CHUNK = 32;
constexpr int BIG_BATCH = 8;

template <int KERNEL, int SIG, typename real>
__global__ void __launch_bounds__(BLOCK_THREADS) lowd_big_kernel(
    const real* __restrict__ x /* (N, DP) */, const real* __restrict__ y /* (M, DP) */,
    const real* __restrict__ b /* (M, EP), zero padded; unused for density */, double* __restrict__ part, int64_t n,
    int64_t n_pad, int64_t m, int DP, int E, int EP, int NE, int segments, int64_t seg_len, int64_t j_offset,
    int64_t m_total) {
  __shared__ double exp_tab_lds[sizeof(real) == 4 ? 1 : 64];
  const double* exp_tab = exp_tab_lds;
  if constexpr (sizeof(real) == 8) {
    if (threadIdx.x < 64) exp_tab_lds[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 64.0));
    __syncthreads();
  }
  const int seg = (int)(blockIdx.x % segments);
  const int64_t tb = blockIdx.x / segments;
  const int64_t i = tb * BLOCK_THREADS + threadIdx.x;
  const int64_t ic = i < n ? i : n - 1;
  const real* __restrict__ xrow = x + ic * DP;
  int64_t jz = -1;
  if constexpr (KERNEL == K_INVDIST) {
    const int64_t g = ic % (m_total + 1);
    jz = (g < m_total) ? g - j_offset : (int64_t)-1;
  }
  const int64_t j0 = (int64_t)seg * seg_len;
  int64_t j1 = j0 + seg_len;
  if (j1 > m) j1 = m;
  const int blocks = (SIG == SIG_DENSITY) ? 1 : EP / 8;
  for (int blk = 0; blk < blocks; ++blk) {
    double accd[8], dend = 0.0;
    real acc[8], den = (real)0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      accd[q] = 0.0;
      acc[q] = (real)0;
    }
    int in_chunk = 0;
    for (int64_t j = j0; j < j1; j += BIG_BATCH) {
      real sb[BIG_BATCH];
#pragma unroll
      for (int jj = 0; jj < BIG_BATCH; ++jj) sb[jj] = (real)0;
      for (int c = 0; c < DP; c += BIG_CHUNK) {
        real xr[BIG_CHUNK];
#pragma unroll
        for (int d = 0; d < BIG_CHUNK; ++d) xr[d] = xrow[c + d];
#pragma unroll
        for (int jj = 0; jj < BIG_BATCH; ++jj) {
          const int64_t jr = j + jj < j1 ? j + jj : j1 - 1;  // past the end: a valid row, its value is dropped below
          const real* __restrict__ yrow = y + jr * DP + c;   // wave-uniform address
          real s = sb[jj];
#pragma unroll
          for (int d = 0; d < BIG_CHUNK; ++d) {
            const real df = xr[d] - yrow[d];
            s = fma(df, df, s);
          }
          sb[jj] = s;
        }
      }
#pragma unroll
      for (int jj = 0; jj < BIG_BATCH; ++jj) {
        if (j + jj < j1) {  // wave-uniform
          real k = kval<KERNEL>(sb[jj], exp_tab);
          if constexpr (KERNEL == K_INVDIST) k = (j + jj == jz) ? (real)0 : k;
          if constexpr (SIG != SIG_DENSITY) {
            const real* __restrict__ brow = b + (j + jj) * EP + 8 * blk;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fma(k, brow[q], acc[q]);
          }
          if constexpr (SIG != SIG_PRODUCT) den += k;
        }
      }
      if (++in_chunk == MID_CHUNK / BIG_BATCH) {
        in_chunk = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          accd[q] += (double)acc[q];
          acc[q] = (real)0;
        }
        dend += (double)den;
        den = (real)0;
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) accd[q] += (double)acc[q];
    dend += (double)den;
    if (i < n_pad) {
      if constexpr (SIG != SIG_DENSITY) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (8 * blk + q < E) part[((int64_t)seg * NE + 8 * blk + q) * n_pad + i] = accd[q];
      }
      if (SIG != SIG_PRODUCT && blk == 0) part[((int64_t)seg * NE + (NE - 1)) * n_pad + i] = dend;
    }
  }
}

// targets (N,D) row-major -> SoA xs[d*n_pad + i] * scale ; pad targets are 0
template <typename real>
__global__ void pack_targets_kernel(const real* __restrict__ x, real* __restrict__ xs, int64_t n,
                                    int64_t n_pad, int D, real scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;
  for (int d = 0; d < D; ++d) xs[(int64_t)d * n_pad + i] = i < n ? x[i * D + d] * scale : (real)0;
}

// sources (M,D) + signal (M,E) -> records [m_pad][R]; pad records have y = +inf and
// b = 0 so that every kernel evaluates to exactly 0 for them.
template <typename real>
__global__ void pack_sources_kernel(const real* __restrict__ y, const real* __restrict__ b,
                                    real* __restrict__ rec, int64_t m, int64_t m_pad, int D, int EB,
                                    int R, real scale, int ldb = -1, int col0 = 0) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m_pad) return;
  if (ldb < 0) ldb = EB;  // the signal has exactly EB columns; otherwise columns col0 .. col0+EB of ldb
  real* r = rec + j * R;
  for (int d = 0; d < D; ++d) r[d] = j < m ? y[j * D + d] * scale : (real)INFINITY;
  for (int e = 0; e < EB; ++e) r[D + e] = j < m ? b[j * ldb + col0 + e] : (real)0;
  for (int q = D + EB; q < R; ++q) r[q] = (real)0;
}

// out[r][d] = d < D ? in[r][d] : 0 for d < DP (rows padded with zero coordinates)
template <typename real>
__global__ void pad_rows_kernel(const real* __restrict__ in, real* __restrict__ out, int64_t rows, int D, int DP) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= rows * DP) return;
  const int64_t r = q / DP;
  const int d = (int)(q % DP);
  out[q] = d < D ? in[r * D + d] : (real)0;
}

template <typename real>
__global__ void scale_kernel(const real* __restrict__ in, real* __restrict__ out, int64_t count,
                             real scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = in[i] * scale;
}

}  // namespace kmvp
