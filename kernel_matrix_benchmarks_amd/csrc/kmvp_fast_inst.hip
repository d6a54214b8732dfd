// Instantiations of the split-bf16 MFMA low-D kernel for ONE kernel function (compiled three
// times: -DKMVP_KERNEL={0,1,2} -DKMVP_FN=launch_fast_<kernel>).  KS = k-steps
// (6 D + 6 <= 16 KS), TT = target tiles of 32 per wave.
#include "kmvp_internal.hpp"
#include "kmvp_fast.hpp"

#ifndef KMVP_KERNEL
#error "KMVP_KERNEL and KMVP_FN must be defined"
#endif

namespace kmvp {

template <int KS, int SIG, int TT>
static hipError_t launch_one(const FastArgs& args, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL((fast_kernel<KMVP_KERNEL, KS, SIG, TT>), grid, dim3(BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

template <int KS, int SIG>
static hipError_t launch_tt(int TT, const FastArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: return launch_one<KS, SIG, 1>(args, grid, stream);
    case 2: return launch_one<KS, SIG, 2>(args, grid, stream);
    case 4: return launch_one<KS, SIG, 4>(args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

template <int KS>
static hipError_t launch_sig(int sig, int TT, const FastArgs& args, dim3 grid, hipStream_t stream) {
  switch (sig) {
    case SIG_PRODUCT: return launch_tt<KS, SIG_PRODUCT>(TT, args, grid, stream);
    case SIG_NORM: return launch_tt<KS, SIG_NORM>(TT, args, grid, stream);
    case SIG_DENSITY: return launch_tt<KS, SIG_DENSITY>(TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t KMVP_FN(int KS, int sig, int TT, const FastArgs& args, dim3 grid, hipStream_t stream,
                   const char** kernel_name) {
  if (kernel_name) *kernel_name = "fast_kernel";
  switch (KS) {
    case 1: return launch_sig<1>(sig, TT, args, grid, stream);
    case 2: return launch_sig<2>(sig, TT, args, grid, stream);
    case 3: return launch_sig<3>(sig, TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
