// Instantiations of the bf16 MFMA kernel for ONE kernel function (compiled five times,
// -DKMVP_KERNEL={0,1,2,3,4} -DKMVP_FN=launch_mfma_<kernel>; 3 = exp(<x,y>): D <= 16*KS - 3; 4 = the shifted Gaussian:
// D <= 16*KS - 9): KS = k-steps of the augmented
// point dimension (D <= 16*KS - 6), NT = 32-column tiles of the signal (E <= 32*NT).
#include <stdlib.h>
#include "kmvp_internal.hpp"
#include "kmvp_mfma.hpp"

#ifndef KMVP_KERNEL
#error "KMVP_KERNEL and KMVP_FN must be defined"
#endif

namespace kmvp {

// diagnostic: KMVP_DBG_LDS=<bytes> of unused dynamic LDS per workgroup lowers the number of resident workgroups per CU
static size_t dbg_lds() {
  static const size_t v = getenv("KMVP_DBG_LDS") ? (size_t)atol(getenv("KMVP_DBG_LDS")) : 0;
  return v;
}
template <typename K>
static void launch_pipe(K kernel, const MfmaArgs& args, dim3 grid, hipStream_t stream) {
  if (dbg_lds()) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dbg_lds());
  hipLaunchKernelGGL(kernel, grid, dim3(BLOCK_THREADS), dbg_lds(), stream, args);
}

template <int KS, int NT>
static hipError_t launch_one(int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream) {
  if constexpr (KS <= MFMA_PIPE_MAX_KS && NT <= MFMA_PIPE_MAX_NT) {
    if (TW >= 3) {  // two target tiles per wave, software-pipelined; TW - 3 = variant (kmvp_mfma.hpp VAR)
      if constexpr (mfma_online<KMVP_KERNEL>()) {  // the running shift lives in the plain pipeline only
        launch_pipe(mfma_pipe_kernel<KMVP_KERNEL, KS, NT, 0>, args, grid, stream);
      } else {
        switch (TW - 3) {
          case 1: launch_pipe(mfma_pipe_kernel<KMVP_KERNEL, KS, NT, 1>, args, grid, stream); break;
          case 4: launch_pipe(mfma_pipe_kernel<KMVP_KERNEL, KS, NT, 4>, args, grid, stream); break;
          case 5: launch_pipe(mfma_pipe_kernel<KMVP_KERNEL, KS, NT, 5>, args, grid, stream); break;
          default: launch_pipe(mfma_pipe_kernel<KMVP_KERNEL, KS, NT, 0>, args, grid, stream); break;
        }
      }
      return hipGetLastError();
    }
  }
  if (TW >= 2)
    hipLaunchKernelGGL((mfma_kernel<KMVP_KERNEL, KS, NT, 2>), grid, dim3(BLOCK_THREADS), 0, stream, args);
  else
    hipLaunchKernelGGL((mfma_kernel<KMVP_KERNEL, KS, NT, 1>), grid, dim3(BLOCK_THREADS), 0, stream, args);
  return hipGetLastError();
}

template <int KS>
static hipError_t launch_nt(int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream) {
  switch (NT) {
    case 1: return launch_one<KS, 1>(TW, args, grid, stream);
    case 2: return launch_one<KS, 2>(TW, args, grid, stream);
    case 3: return launch_one<KS, 3>(TW, args, grid, stream);
    case 4: return launch_one<KS, 4>(TW, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t KMVP_FN(int KS, int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream,
                   const char** kernel_name) {
  if (kernel_name)
    *kernel_name = (TW >= 3 && KS <= MFMA_PIPE_MAX_KS && NT <= MFMA_PIPE_MAX_NT) ? "mfma_pipe_kernel" : "mfma_kernel";
  switch (KS) {
    case 1: return launch_nt<1>(NT, TW, args, grid, stream);
    case 2: return launch_nt<2>(NT, TW, args, grid, stream);
    case 3: return launch_nt<3>(NT, TW, args, grid, stream);
    case 4: return launch_nt<4>(NT, TW, args, grid, stream);
    case 5: return launch_nt<5>(NT, TW, args, grid, stream);
    case 6: return launch_nt<6>(NT, TW, args, grid, stream);
    case 7: return launch_nt<7>(NT, TW, args, grid, stream);
    case 8: return launch_nt<8>(NT, TW, args, grid, stream);
    case 9: return launch_nt<9>(NT, TW, args, grid, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace kmvp
