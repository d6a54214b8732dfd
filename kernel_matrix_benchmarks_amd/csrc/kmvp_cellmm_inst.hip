// Instantiations of cellmm_kernel / cellmm16_kernel (kmvp_cellmm.hpp): TT = target tiles of 32 per wave.
#include <stdlib.h>
#include "kmvp_internal.hpp"
#include "kmvp_cellmm.hpp"

namespace kmvp {

// diagnostic: KMVP_DBG_LDS=<bytes> of unused dynamic LDS per workgroup lowers the number of resident workgroups per CU
// (tools/cellmm_occupancy.py: what a third wave per SIMD is worth to this loop)
static size_t cmm_dbg_lds() {
  static const size_t v = getenv("KMVP_DBG_LDS") ? (size_t)atol(getenv("KMVP_DBG_LDS")) : 0;
  return v;
}
template <typename K>
static void cmm_launch(K kernel, const CellmmArgs& args, dim3 grid, hipStream_t stream) {
  if (cmm_dbg_lds()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cmm_dbg_lds());
  hipLaunchKernelGGL(kernel, grid, dim3(BLOCK_THREADS), cmm_dbg_lds(), stream, args);
}

hipError_t launch_cellmm_gaussian(int TT, int shape, const CellmmArgs& args, dim3 grid, hipStream_t stream, const char** kernel_name) {
  if (kernel_name) *kernel_name = shape == 1 ? "cellmm16_kernel" : "cellmm_kernel";  // the names the profiler shows
  if (shape == 1) {
    switch (TT) {
      case 1: cmm_launch(cellmm16_kernel<1>, args, grid, stream); break;
      case 2: cmm_launch(cellmm16_kernel<2>, args, grid, stream); break;
      case 4: cmm_launch(cellmm16_kernel<4>, args, grid, stream); break;
      case 8: cmm_launch(cellmm16_kernel<8>, args, grid, stream); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (TT) {
    case 1: cmm_launch(cellmm_kernel<1>, args, grid, stream); break;
    case 2: cmm_launch(cellmm_kernel<2>, args, grid, stream); break;
    case 4: cmm_launch(cellmm_kernel<4>, args, grid, stream); break;
    case 8: cmm_launch(cellmm_kernel<8>, args, grid, stream); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace kmvp
