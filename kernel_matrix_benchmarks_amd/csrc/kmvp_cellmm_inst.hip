// Instantiations of cellmm_kernel / cellmm16_kernel (kmvp_cellmm.hpp): TT = target tiles of 32 per wave.
#include "kmvp_internal.hpp"
#include "kmvp_cellmm.hpp"

namespace kmvp {

hipError_t launch_cellmm_gaussian(int TT, int shape, const CellmmArgs& args, dim3 grid, hipStream_t stream, const char** kernel_name) {
  if (kernel_name) *kernel_name = shape == 1 ? "cellmm16_kernel" : "cellmm_kernel";  // the names the profiler shows
  if (shape == 1) {
    switch (TT) {
      case 1: hipLaunchKernelGGL((cellmm16_kernel<1>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
      case 2: hipLaunchKernelGGL((cellmm16_kernel<2>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
      case 4: hipLaunchKernelGGL((cellmm16_kernel<4>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
      case 8: hipLaunchKernelGGL((cellmm16_kernel<8>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (TT) {
    case 1: hipLaunchKernelGGL((cellmm_kernel<1>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2: hipLaunchKernelGGL((cellmm_kernel<2>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 4: hipLaunchKernelGGL((cellmm_kernel<4>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 8: hipLaunchKernelGGL((cellmm_kernel<8>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace kmvp
