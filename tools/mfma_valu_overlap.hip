// Micro-benchmark: do the bf16 matrix pipe and the VALU (v_exp_f32 / v_fma_f32) of gfx950 run
// concurrently when they come from DIFFERENT waves of one SIMD, and from the SAME wave?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: every wave MFMA only; 1: every wave VALU only; 2: waves 0-3 MFMA, waves 4-7 VALU (2 per SIMD);
// MODE 3: every wave alternates 8 MFMA / 64 exp (same wave, independent data)
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + j); }
  f32x16 acc0 = {0}, acc1 = {0};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0f + threadIdx.x * 1e-6f + j * 1e-3f;
  const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
    }
    if (do_valu) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]);
    }
  }
  float r = 0;
  for (int j = 0; j < 16; ++j) r += v[j] + acc0[j] + acc1[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char* name) {
  float* out;
  const int blocks = 256, iters = 20000;
  CHECK(hipMalloc(&out, sizeof(float) * 512 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-56s %8.3f ms  (%.1f cycles per iteration at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  // per iteration: an MFMA wave issues 8 MFMA (32x32x16 bf16), a VALU wave issues 64 v_exp_f32
  if (run<0>("8 waves/CU x 8 MFMA (2 MFMA waves per SIMD)")) return 1;
  if (run<1>("8 waves/CU x 64 exp (2 VALU waves per SIMD)")) return 1;
  if (run<2>("4 waves MFMA + 4 waves exp (1 + 1 per SIMD)")) return 1;
  if (run<3>("8 waves, each 8 MFMA then 64 exp (2 per SIMD)")) return 1;
  return 0;
}
