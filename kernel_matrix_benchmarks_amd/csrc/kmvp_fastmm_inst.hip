// Instantiations of fastmm_kernel (kmvp_fastmm.hpp): D = point dimension, MODE = 0 (<= 16 columns) / 1 (<= 32),
// TT = target tiles of 32 per wave (1 or 2).
#include "kmvp_internal.hpp"
#include "kmvp_fastmm.hpp"

namespace kmvp {

template <int D, int MODE>
static hipError_t launch_tt(int TT, const FastmmArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((fastmm_kernel<D, MODE, 1>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2: hipLaunchKernelGGL((fastmm_kernel<D, MODE, 2>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int D>
static hipError_t launch_mode(int mode, int TT, const FastmmArgs& args, dim3 grid, hipStream_t stream) {
  return mode ? launch_tt<D, 1>(TT, args, grid, stream) : launch_tt<D, 0>(TT, args, grid, stream);
}

hipError_t launch_fastmm_gaussian(int D, int mode, int TT, const FastmmArgs& args, dim3 grid, hipStream_t stream,
                                  const char** kernel_name) {
  if (kernel_name) *kernel_name = "fastmm_kernel";
  switch (D) {
    case 1: return launch_mode<1>(mode, TT, args, grid, stream);
    case 2: return launch_mode<2>(mode, TT, args, grid, stream);
    case 3: return launch_mode<3>(mode, TT, args, grid, stream);
    case 4: return launch_mode<4>(mode, TT, args, grid, stream);
    case 5: return launch_mode<5>(mode, TT, args, grid, stream);
    case 6: return launch_mode<6>(mode, TT, args, grid, stream);
    case 7: return launch_mode<7>(mode, TT, args, grid, stream);
    case 8: return launch_mode<8>(mode, TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
