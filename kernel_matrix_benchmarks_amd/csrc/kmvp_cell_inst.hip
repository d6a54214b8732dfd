// Instantiations of the cell-reduced Gaussian kernel (kmvp_cell.hpp): TT = target tiles of 32 per wave.
#include "kmvp_internal.hpp"
#include "kmvp_cell.hpp"
#include "kmvp_cell64.hpp"

namespace kmvp {

template <int SIG>
static hipError_t launch_tt(int TT, const CellArgs& args, dim3 grid, hipStream_t stream) {
  switch (TT) {
    case 1: hipLaunchKernelGGL((cell_kernel<SIG, 1>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 2: hipLaunchKernelGGL((cell_kernel<SIG, 2>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 4: hipLaunchKernelGGL((cell_kernel<SIG, 4>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    case 8: hipLaunchKernelGGL((cell_kernel<SIG, 8>), grid, dim3(BLOCK_THREADS), 0, stream, args); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_cell_gaussian(int sig, int TT, const CellArgs& args, dim3 grid, hipStream_t stream,
                                const char** kernel_name) {
  if (kernel_name) *kernel_name = "cell_kernel";
  switch (sig) {
    case SIG_PRODUCT:
    case SIG_DENSITY: return launch_tt<SIG_PRODUCT>(TT, args, grid, stream);  // density: the packer sets b = 1
    case SIG_NORM: return launch_tt<SIG_NORM>(TT, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_cell64_gaussian(int sig, const Cell64Args& args, dim3 grid, hipStream_t stream, const char** kernel_name) {
  if (kernel_name) *kernel_name = "cell64_kernel";
  if (sig == SIG_NORM) hipLaunchKernelGGL((cell64_kernel<SIG_NORM>), grid, dim3(BLOCK_THREADS), 0, stream, args);
  else hipLaunchKernelGGL((cell64_kernel<SIG_PRODUCT>), grid, dim3(BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

}  // namespace kmvp
