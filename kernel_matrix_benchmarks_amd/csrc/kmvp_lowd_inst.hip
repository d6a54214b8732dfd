// Instantiations of the low-D pair-loop kernels for ONE kernel function and ONE
// precision.  Compiled six times (see Makefile):
//   -DKMVP_KERNEL={0,1,2}  -DKMVP_REAL={float,double}  -DKMVP_FN=launch_lowd_<k>_<p>
// so that each kernel function is its own set of device kernels (no runtime
// kernel dispatch inside device code) and the six units build in parallel.
#include "kmvp_internal.hpp"

#ifndef KMVP_KERNEL
#error "KMVP_KERNEL, KMVP_REAL and KMVP_FN must be defined"
#endif

namespace kmvp {

using real = KMVP_REAL;
constexpr int KERNEL = KMVP_KERNEL;

#define KMVP_STR2(x) #x
#define KMVP_STR(x) KMVP_STR2(x)

template <int D, int E, int SIG, int T, int FEED>
static hipError_t launch_one(const LowdArgs<real>& args, dim3 grid, hipStream_t stream,
                             const char** kernel_name) {
  hipLaunchKernelGGL((lowd_kernel<KERNEL, D, E, SIG, T, FEED, real>), grid, dim3(BLOCK_THREADS), 0,
                     stream, args);
  if (kernel_name) *kernel_name = "lowd_kernel";
  return hipGetLastError();
}

// T / FEED variants: every shape gets (T=1, LDS feed) -- the measured best and the
// default -- and (T=2, scalar-cache feed); the headline shape (D=3, E=1) additionally
// gets the whole tuning grid so that the choice stays backed by measurements.
template <int D, int E, int SIG>
static hipError_t launch_te(LowdTuning tune, const LowdArgs<real>& args, dim3 grid,
                            hipStream_t stream, const char** kernel_name) {
  const int T = tune.targets_per_lane, F = tune.feed;
  if (F == 1 && T == 1) return launch_one<D, E, SIG, 1, 1>(args, grid, stream, kernel_name);
  if (F == 0 && T == 2) return launch_one<D, E, SIG, 2, 0>(args, grid, stream, kernel_name);
  if constexpr (D == 3 && E == 1) {
    if (F == 0 && T == 1) return launch_one<D, E, SIG, 1, 0>(args, grid, stream, kernel_name);
    if (F == 0 && T == 4) return launch_one<D, E, SIG, 4, 0>(args, grid, stream, kernel_name);
    if (F == 0 && T == 8) return launch_one<D, E, SIG, 8, 0>(args, grid, stream, kernel_name);
    if (F == 1 && T == 2) return launch_one<D, E, SIG, 2, 1>(args, grid, stream, kernel_name);
    if (F == 1 && T == 4) return launch_one<D, E, SIG, 4, 1>(args, grid, stream, kernel_name);
    if (F == 1 && T == 8) return launch_one<D, E, SIG, 8, 1>(args, grid, stream, kernel_name);
  }
  return hipErrorInvalidValue;
}

template <int D, int SIG>
static hipError_t launch_e(int E, LowdTuning tune, const LowdArgs<real>& args, dim3 grid,
                           hipStream_t stream, const char** kernel_name) {
  if constexpr (SIG == SIG_DENSITY) {
    return launch_te<D, 1, SIG>(tune, args, grid, stream, kernel_name);
  } else {
    switch (E) {
      case 1: return launch_te<D, 1, SIG>(tune, args, grid, stream, kernel_name);
      case 2: return launch_te<D, 2, SIG>(tune, args, grid, stream, kernel_name);
      case 3: return launch_te<D, 3, SIG>(tune, args, grid, stream, kernel_name);
      case 4: return launch_te<D, 4, SIG>(tune, args, grid, stream, kernel_name);
      default: return hipErrorInvalidValue;
    }
  }
}

template <int SIG>
static hipError_t launch_d(int D, int E, LowdTuning tune, const LowdArgs<real>& args, dim3 grid,
                           hipStream_t stream, const char** kernel_name) {
  switch (D) {
    case 1: return launch_e<1, SIG>(E, tune, args, grid, stream, kernel_name);
    case 2: return launch_e<2, SIG>(E, tune, args, grid, stream, kernel_name);
    case 3: return launch_e<3, SIG>(E, tune, args, grid, stream, kernel_name);
    case 4: return launch_e<4, SIG>(E, tune, args, grid, stream, kernel_name);
    case 5: return launch_e<5, SIG>(E, tune, args, grid, stream, kernel_name);
    case 6: return launch_e<6, SIG>(E, tune, args, grid, stream, kernel_name);
    case 7: return launch_e<7, SIG>(E, tune, args, grid, stream, kernel_name);
    case 8: return launch_e<8, SIG>(E, tune, args, grid, stream, kernel_name);
    default: return hipErrorInvalidValue;
  }
}

hipError_t KMVP_FN(int D, int E, int sig, LowdTuning tune, const LowdArgs<real>& args, dim3 grid,
                   hipStream_t stream, const char** kernel_name) {
  switch (sig) {
    case SIG_PRODUCT: return launch_d<SIG_PRODUCT>(D, E, tune, args, grid, stream, kernel_name);
    case SIG_NORM: return launch_d<SIG_NORM>(D, E, tune, args, grid, stream, kernel_name);
    case SIG_DENSITY: return launch_d<SIG_DENSITY>(D, E, tune, args, grid, stream, kernel_name);
    default: return hipErrorInvalidValue;
  }
}

#define KMVP_CAT2(a, b) a##b
#define KMVP_CAT(a, b) KMVP_CAT2(a, b)

hipError_t KMVP_CAT(KMVP_FN, _generic)(int sig, const real* x, const real* y, const real* b,
                                       double* part, int64_t n, int64_t n_pad, int64_t m, int D,
                                       int E, int NE, int segments, int64_t seg_len,
                                       int64_t j_offset, int64_t m_total, hipStream_t stream,
                                       const char** kernel_name) {
  const dim3 grid((unsigned)((n_pad / BLOCK_THREADS) * segments));
  if (D <= LOWD_MID_MAX_D) {  // coordinates in registers; x, y and b arrive padded to rows of 8 ceil(. / 8) entries
    if (kernel_name) *kernel_name = "lowd_mid_kernel";
    const int ebw = lowd_mid_colblock((int)sizeof(real), D, E);  // signal columns per pass (the host pads b to it)
    const int EPm = (E + ebw - 1) / ebw * ebw;
#define KMVP_MID(SIGV, DCH)                                                                                           \
  do {                                                                                                                \
    if constexpr (sizeof(real) == 4 && DCH <= 8) {                                                                    \
      if (ebw == 32) {                                                                                                \
        hipLaunchKernelGGL((lowd_mid_kernel<KERNEL, SIGV, real, DCH, 32>), grid, dim3(BLOCK_THREADS), 0, stream, x, y, \
                           b, part, n, n_pad, m, E, EPm, NE, segments, seg_len, j_offset, m_total);                   \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((lowd_mid_kernel<KERNEL, SIGV, real, DCH, 8>), grid, dim3(BLOCK_THREADS), 0, stream, x, y, b,  \
                       part, n, n_pad, m, E, EPm, NE, segments, seg_len, j_offset, m_total);                          \
  } while (0)
#define KMVP_MID_D(SIGV)                \
  switch ((D + 7) / 8) {                \
    case 1: KMVP_MID(SIGV, 1); break;   \
    case 2: KMVP_MID(SIGV, 2); break;   \
    case 3: KMVP_MID(SIGV, 3); break;   \
    case 4: KMVP_MID(SIGV, 4); break;   \
    case 5: KMVP_MID(SIGV, 5); break;   \
    case 6: KMVP_MID(SIGV, 6); break;   \
    case 7: KMVP_MID(SIGV, 7); break;   \
    case 8: KMVP_MID(SIGV, 8); break;   \
    case 9: KMVP_MID(SIGV, 9); break;   \
    case 10: KMVP_MID(SIGV, 10); break; \
    case 11: KMVP_MID(SIGV, 11); break; \
    case 12: KMVP_MID(SIGV, 12); break; \
    case 13: KMVP_MID(SIGV, 13); break; \
    case 14: KMVP_MID(SIGV, 14); break; \
    case 15: KMVP_MID(SIGV, 15); break; \
    default: KMVP_MID(SIGV, 16); break; \
  }
    switch (sig) {
      case SIG_PRODUCT: KMVP_MID_D(SIG_PRODUCT); break;
      case SIG_NORM: KMVP_MID_D(SIG_NORM); break;
      case SIG_DENSITY: KMVP_MID_D(SIG_DENSITY); break;
      default: return hipErrorInvalidValue;
    }
#undef KMVP_MID_D
#undef KMVP_MID
    return hipGetLastError();
  }
  {  // any larger D: chunked walk over padded rows (x, y padded to 32 ceil(D / 32), b to 8 ceil(E / 8))
    if (kernel_name) *kernel_name = "lowd_big_kernel";
    const int DP = (D + BIG_CHUNK - 1) / BIG_CHUNK * BIG_CHUNK, EP = (E + 7) / 8 * 8;
#define KMVP_BIG(SIGV)                                                                                         \
  hipLaunchKernelGGL((lowd_big_kernel<KERNEL, SIGV, real>), grid, dim3(BLOCK_THREADS), 0, stream, x, y, b, part, \
                     n, n_pad, m, DP, E, EP, NE, segments, seg_len, j_offset, m_total)
    switch (sig) {
      case SIG_PRODUCT: KMVP_BIG(SIG_PRODUCT); break;
      case SIG_NORM: KMVP_BIG(SIG_NORM); break;
      case SIG_DENSITY: KMVP_BIG(SIG_DENSITY); break;
      default: return hipErrorInvalidValue;
    }
#undef KMVP_BIG
    return hipGetLastError();
  }
}

}  // namespace kmvp
