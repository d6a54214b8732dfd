// Instantiations of cfastmm_kernel (kmvp_cfastmm.hpp): kernel (Gaussian, exp(-r), 1/r), MODE = 0 (<= 16 columns) / 1 (<= 32),
// TT = target tiles of 32 per wave (1 or 2).
#include "kmvp_internal.hpp"
#include "kmvp_cfastmm.hpp"

namespace kmvp {

template <int KERNEL, int MODE, int ONLINE>
static hipError_t launch_tt(int TT, const CfastmmArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((cfastmm_kernel<KERNEL, MODE, 1, ONLINE>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2: hipLaunchKernelGGL((cfastmm_kernel<KERNEL, MODE, 2, ONLINE>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int KERNEL>
static hipError_t launch_mode(int mode, int TT, int online, const CfastmmArgs& args, dim3 grid, hipStream_t stream) {
  if (online) return mode ? launch_tt<KERNEL, 1, 1>(TT, args, grid, stream) : launch_tt<KERNEL, 0, 1>(TT, args, grid, stream);
  return mode ? launch_tt<KERNEL, 1, 0>(TT, args, grid, stream) : launch_tt<KERNEL, 0, 0>(TT, args, grid, stream);
}

hipError_t launch_cfastmm(int kernel, int mode, int TT, int online, const CfastmmArgs& args, dim3 grid, hipStream_t stream,
                          const char** kernel_name) {
  if (kernel_name) *kernel_name = "cfastmm_kernel";  // (ONLINE = 1 shows in the dispatch note)
  if (kernel == K_GAUSSIAN) return launch_mode<K_GAUSSIAN>(mode, TT, online, args, grid, stream);
  if (kernel == K_ABSEXP) return launch_mode<K_ABSEXP>(mode, TT, online, args, grid, stream);
  if (kernel == K_INVDIST)  // unbounded values: only with the per-target shift
    return mode ? launch_tt<K_INVDIST, 1, 1>(TT, args, grid, stream) : launch_tt<K_INVDIST, 0, 1>(TT, args, grid, stream);
  return hipErrorInvalidValue;
}

}  // namespace kmvp
