// Instantiations of fastmm_kernel (kmvp_fastmm.hpp): KS = k-steps of the squared-distance product (point dimension D:
// KS = ceil((6 D + 7) / 16), D <= 64), MODE = 0 (<= 16 columns) / 1 (<= 32), TT = target tiles of 32 per wave (two while
// KS <= 4).
// Compiled twice: -DKMVP_FMM_KERNEL=0 -DKMVP_FN=launch_fastmm_gaussian, -DKMVP_FMM_KERNEL=1 -DKMVP_FN=launch_fastmm_absexp.
#include "kmvp_internal.hpp"
#include "kmvp_fastmm.hpp"

#ifndef KMVP_FMM_KERNEL
#error "KMVP_FMM_KERNEL and KMVP_FN must be defined"
#endif

namespace kmvp {

template <int KS, int MODE, int KERNEL, int ONLINE>
static hipError_t launch_tt(int TT, const FastmmArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((fastmm_kernel<KS, MODE, 1, KERNEL, ONLINE>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2:
      if constexpr (KS <= FMM_MAX_KS_TWO_TILES) {
        hipLaunchKernelGGL((fastmm_kernel<KS, MODE, 2, KERNEL, ONLINE>), grid, dim3(BLOCK_THREADS), 0, stream, args);
        break;
      }
      return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int KS>
static hipError_t launch_mode(int mode, int TT, int online, const FastmmArgs& args, dim3 grid, hipStream_t stream) {
  if (online)
    return mode ? launch_tt<KS, 1, KMVP_FMM_KERNEL, 1>(TT, args, grid, stream)
                : launch_tt<KS, 0, KMVP_FMM_KERNEL, 1>(TT, args, grid, stream);
  return mode ? launch_tt<KS, 1, KMVP_FMM_KERNEL, 0>(TT, args, grid, stream)
              : launch_tt<KS, 0, KMVP_FMM_KERNEL, 0>(TT, args, grid, stream);
}

hipError_t KMVP_FN(int KS, int mode, int TT, int online, const FastmmArgs& args, dim3 grid, hipStream_t stream,
                   const char** kernel_name) {
  if (kernel_name) *kernel_name = "fastmm_kernel";  // (ONLINE = 1 shows in the dispatch note and in the profiler's template arguments)
  switch (KS) {
    case 1: return launch_mode<1>(mode, TT, online, args, grid, stream);
    case 2: return launch_mode<2>(mode, TT, online, args, grid, stream);
    case 3: return launch_mode<3>(mode, TT, online, args, grid, stream);
    case 4: return launch_mode<4>(mode, TT, online, args, grid, stream);
    case 5: return launch_mode<5>(mode, TT, online, args, grid, stream);
    case 6: return launch_mode<6>(mode, TT, online, args, grid, stream);
    case 7: return launch_mode<7>(mode, TT, online, args, grid, stream);
    case 8: return launch_mode<8>(mode, TT, online, args, grid, stream);
    case 9: return launch_mode<9>(mode, TT, online, args, grid, stream);
    case 10: return launch_mode<10>(mode, TT, online, args, grid, stream);
    case 11: return launch_mode<11>(mode, TT, online, args, grid, stream);
    case 12: return launch_mode<12>(mode, TT, online, args, grid, stream);
    case 13: return launch_mode<13>(mode, TT, online, args, grid, stream);
    case 14: return launch_mode<14>(mode, TT, online, args, grid, stream);
    case 15: return launch_mode<15>(mode, TT, online, args, grid, stream);
    case 16: return launch_mode<16>(mode, TT, online, args, grid, stream);
    case 17: return launch_mode<17>(mode, TT, online, args, grid, stream);
    case 18: return launch_mode<18>(mode, TT, online, args, grid, stream);
    case 19: return launch_mode<19>(mode, TT, online, args, grid, stream);
    case 20: return launch_mode<20>(mode, TT, online, args, grid, stream);
    case 21: return launch_mode<21>(mode, TT, online, args, grid, stream);
    case 22: return launch_mode<22>(mode, TT, online, args, grid, stream);
    case 23: return launch_mode<23>(mode, TT, online, args, grid, stream);
    case 24: return launch_mode<24>(mode, TT, online, args, grid, stream);
    case 25: return launch_mode<25>(mode, TT, online, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
