// Instantiations of cfastmm_kernel (kmvp_cfastmm.hpp): kernel (Gaussian, exp(-r)), MODE = 0 (<= 16 columns) / 1 (<= 32),
// TT = target tiles of 32 per wave (1 or 2).
#include "kmvp_internal.hpp"
#include "kmvp_cfastmm.hpp"

namespace kmvp {

template <int KERNEL, int MODE>
static hipError_t launch_tt(int TT, const CfastmmArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((cfastmm_kernel<KERNEL, MODE, 1>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2: hipLaunchKernelGGL((cfastmm_kernel<KERNEL, MODE, 2>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_cfastmm(int kernel, int mode, int TT, const CfastmmArgs& args, dim3 grid, hipStream_t stream,
                          const char** kernel_name) {
  if (kernel_name) *kernel_name = "cfastmm_kernel";
  if (kernel == K_GAUSSIAN)
    return mode ? launch_tt<K_GAUSSIAN, 1>(TT, args, grid, stream) : launch_tt<K_GAUSSIAN, 0>(TT, args, grid, stream);
  if (kernel == K_ABSEXP)
    return mode ? launch_tt<K_ABSEXP, 1>(TT, args, grid, stream) : launch_tt<K_ABSEXP, 0>(TT, args, grid, stream);
  return hipErrorInvalidValue;
}

}  // namespace kmvp
