// Micro-benchmark: 8 f16 MFMAs (32x32x16) per trip plus 32 plain VALU instructions, 2 waves per SIMD:
//   placement   interleaved (4 after every MFMA) or as ONE burst behind the eight MFMAs (cellmm_kernel's shape)
//   dependency  8 independent chains or one dependent chain
//   operation   v_fma_f32 (3 operands), v_mul_f32 (2), v_cvt_pkrtz_f16_f32, v_exp_f32 (transcendental unit)
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize [-mllvm -amdgpu-mfma-vgpr-form=1] -o mfma_valu_burst mfma_valu_burst.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP>
__device__ __forceinline__ float op(float v, float c) {
  if constexpr (OP == 0) return __builtin_fmaf(v, c, 1e-3f);
  else if constexpr (OP == 1) return v * c;
  else if constexpr (OP == 2) { auto p = __builtin_amdgcn_cvt_pkrtz(v, c); return (float)p[0] + 1.0f; }
  else return __builtin_amdgcn_exp2f(v);
}

// PLACE 0: interleaved, 1: burst.  DEP 0: 8 chains, 1: one chain.  R VALU per MFMA.
template <int R, int PLACE, int DEP, int OP, int MFMA_ON>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(1.0f + j); }
  f32x16 acc[8];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 16; ++j) acc[s][j] = 0.f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + threadIdx.x * 1e-6f + j * 1e-3f;
  const float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
    if (PLACE == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (MFMA_ON) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < R; ++e) { const int idx = DEP ? 0 : ((m * R + e) & 7); v[idx] = op<OP>(v[idx], c); }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 8; ++m)
        if (MFMA_ON) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 8 * R; ++e) { const int idx = DEP ? 0 : (e & 7); v[idx] = op<OP>(v[idx], c); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int j = 0; j < 8; ++j) r += v[j];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 16; ++j) r += acc[s][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int R, int PLACE, int DEP, int OP, int MFMA_ON>
int run(const char* name, int waves) {
  float* out;
  const int blocks = 256 * waves, iters = 20000 / waves;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<R, PLACE, DEP, OP, MFMA_ON>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<R, PLACE, DEP, OP, MFMA_ON>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-52s waves/SIMD=%d %8.3f ms  %7.1f nominal cycles per SIMD per trip (%d MFMA + %2d VALU)\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / ((double)waves * iters), 8 * MFMA_ON, 8 * R);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {2}) {
    if (run<4, 0, 0, 0, 0>("fma alone, 8 chains", w)) return 1;
    if (run<4, 0, 1, 0, 0>("fma alone, 1 chain", w)) return 1;
    if (run<4, 0, 0, 1, 0>("mul alone, 8 chains", w)) return 1;
    if (run<4, 0, 0, 2, 0>("cvt_pkrtz+cvt+add alone, 8 chains", w)) return 1;
    if (run<4, 0, 0, 3, 0>("exp alone, 8 chains", w)) return 1;
    if (run<0, 0, 0, 0, 1>("MFMA alone", w)) return 1;
    if (run<4, 0, 0, 0, 1>("MFMA + fma, interleaved, 8 chains", w)) return 1;
    if (run<4, 1, 0, 0, 1>("MFMA + fma, burst, 8 chains", w)) return 1;
    if (run<4, 0, 1, 0, 1>("MFMA + fma, interleaved, 1 chain", w)) return 1;
    if (run<4, 1, 1, 0, 1>("MFMA + fma, burst, 1 chain", w)) return 1;
    if (run<4, 0, 0, 1, 1>("MFMA + mul, interleaved, 8 chains", w)) return 1;
    if (run<4, 1, 0, 1, 1>("MFMA + mul, burst, 8 chains", w)) return 1;
    if (run<4, 0, 0, 2, 1>("MFMA + cvt_pkrtz.., interleaved, 8 chains", w)) return 1;
    if (run<4, 0, 0, 3, 1>("MFMA + exp, interleaved, 8 chains", w)) return 1;
    if (run<4, 1, 0, 3, 1>("MFMA + exp, burst, 8 chains", w)) return 1;
  }
  return 0;
}
