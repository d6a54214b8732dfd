// Instantiations of the centred split-bf16 kernel for ONE kernel function (compiled three
// times: -DKMVP_KERNEL={0,1,2} -DKMVP_FN=launch_cfast_<kernel>).  TT = target tiles per wave.
#include "kmvp_internal.hpp"
#include "kmvp_cfast.hpp"

#ifndef KMVP_KERNEL
#error "KMVP_KERNEL and KMVP_FN must be defined"
#endif

namespace kmvp {

template <int SIG>
static hipError_t launch_tt(int TT, const CfastArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((cfast_kernel<KMVP_KERNEL, SIG, 1>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2: hipLaunchKernelGGL((cfast_kernel<KMVP_KERNEL, SIG, 2>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 4: hipLaunchKernelGGL((cfast_kernel<KMVP_KERNEL, SIG, 4>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t KMVP_FN(int sig, int TT, const CfastArgs& args, dim3 grid, hipStream_t stream,
                   const char** kernel_name) {
  if (kernel_name) *kernel_name = "cfast_kernel";
  switch (sig) {
    case SIG_PRODUCT: return launch_tt<SIG_PRODUCT>(TT, args, grid, stream);
    case SIG_NORM: return launch_tt<SIG_NORM>(TT, args, grid, stream);
    case SIG_DENSITY: return launch_tt<SIG_DENSITY>(TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
