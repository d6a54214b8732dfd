// Instantiations of the split-bf16 MFMA low-D kernel for ONE kernel function (compiled three
// times: -DKMVP_KERNEL={0,1,2} -DKMVP_FN=launch_fast_<kernel>).  D = point dimension
// (K = 6 D + 6 columns, KS = ceil(K / 16) k-steps), TT = target tiles of 32 per wave.
#include "kmvp_internal.hpp"
#include "kmvp_fast.hpp"

#ifndef KMVP_KERNEL
#error "KMVP_KERNEL and KMVP_FN must be defined"
#endif

namespace kmvp {

template <int D, int SIG, int TT>
static hipError_t launch_one(const FastArgs& args, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL((fast_kernel<KMVP_KERNEL, D, SIG, TT>), grid, dim3(BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

template <int D, int SIG>
static hipError_t launch_tt(int TT, const FastArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: return launch_one<D, SIG, 1>(args, grid, stream);
    case 2: return launch_one<D, SIG, 2>(args, grid, stream);
    case 4: return launch_one<D, SIG, 4>(args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

template <int D>
static hipError_t launch_sig(int sig, int TT, const FastArgs& args, dim3 grid, hipStream_t stream) {
  switch (sig) {
    case SIG_PRODUCT: return launch_tt<D, SIG_PRODUCT>(TT, args, grid, stream);
    case SIG_NORM: return launch_tt<D, SIG_NORM>(TT, args, grid, stream);
    case SIG_DENSITY: return launch_tt<D, SIG_DENSITY>(TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t KMVP_FN(int D, int sig, int TT, const FastArgs& args, dim3 grid, hipStream_t stream,
                   const char** kernel_name) {
  if (kernel_name) *kernel_name = "fast_kernel";
  switch (D) {
    case 1: return launch_sig<1>(sig, TT, args, grid, stream);
    case 2: return launch_sig<2>(sig, TT, args, grid, stream);
    case 3: return launch_sig<3>(sig, TT, args, grid, stream);
    case 4: return launch_sig<4>(sig, TT, args, grid, stream);
    case 5: return launch_sig<5>(sig, TT, args, grid, stream);
    case 6: return launch_sig<6>(sig, TT, args, grid, stream);
    case 7: return launch_sig<7>(sig, TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
