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
    case 2:
      if constexpr (D <= FAST_MAX_D_TWO_TILES) return launch_one<D, SIG, 2>(args, grid, stream);
      return hipErrorInvalidValue;
    case 4:
      // four tiles per wave only where the operands fit the register file (K = 6 D + 6 <= 48)
      if constexpr (D <= FAST_MAX_D_FOUR_TILES) return launch_one<D, SIG, 4>(args, grid, stream);
      return hipErrorInvalidValue;
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
    case 8: return launch_sig<8>(sig, TT, args, grid, stream);
    case 9: return launch_sig<9>(sig, TT, args, grid, stream);
    case 10: return launch_sig<10>(sig, TT, args, grid, stream);
    case 11: return launch_sig<11>(sig, TT, args, grid, stream);
    case 12: return launch_sig<12>(sig, TT, args, grid, stream);
    case 13: return launch_sig<13>(sig, TT, args, grid, stream);
    case 14: return launch_sig<14>(sig, TT, args, grid, stream);
    case 15: return launch_sig<15>(sig, TT, args, grid, stream);
    case 16: return launch_sig<16>(sig, TT, args, grid, stream);
    case 17: return launch_sig<17>(sig, TT, args, grid, stream);
    case 18: return launch_sig<18>(sig, TT, args, grid, stream);
    case 19: return launch_sig<19>(sig, TT, args, grid, stream);
    case 20: return launch_sig<20>(sig, TT, args, grid, stream);
    case 21: return launch_sig<21>(sig, TT, args, grid, stream);
    case 22: return launch_sig<22>(sig, TT, args, grid, stream);
    case 23: return launch_sig<23>(sig, TT, args, grid, stream);
    case 24: return launch_sig<24>(sig, TT, args, grid, stream);
    case 25: return launch_sig<25>(sig, TT, args, grid, stream);
    case 26: return launch_sig<26>(sig, TT, args, grid, stream);
    case 27: return launch_sig<27>(sig, TT, args, grid, stream);
    case 28: return launch_sig<28>(sig, TT, args, grid, stream);
    case 29: return launch_sig<29>(sig, TT, args, grid, stream);
    case 30: return launch_sig<30>(sig, TT, args, grid, stream);
    case 31: return launch_sig<31>(sig, TT, args, grid, stream);
    case 32: return launch_sig<32>(sig, TT, args, grid, stream);
    case 33: return launch_sig<33>(sig, TT, args, grid, stream);
    case 34: return launch_sig<34>(sig, TT, args, grid, stream);
    case 35: return launch_sig<35>(sig, TT, args, grid, stream);
    case 36: return launch_sig<36>(sig, TT, args, grid, stream);
    case 37: return launch_sig<37>(sig, TT, args, grid, stream);
    case 38: return launch_sig<38>(sig, TT, args, grid, stream);
    case 39: return launch_sig<39>(sig, TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
