// Micro-benchmark: one wave interleaving f16 MFMAs (32x32x16, 8 passes) with PLAIN full-rate VALU work
// (v_fma_f32 / v_cvt_pkrtz), R VALU instructions per MFMA, 1 and 2 waves per SIMD; the VALU stream either
// as 8 independent chains or as ONE dependent chain (the shape of cellmm_kernel's operand build).
// Question: what does an MFMA cost on the VALU issue port, and how much of a dependent VALU chain hides
// in the MFMA's shadow?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_fma_interleave mfma_fma_interleave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int R, int DEP, int MFMA_ON>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(1.0f + j); }
  f32x16 acc[8];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 16; ++j) acc[s][j] = 0.f;
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + threadIdx.x * 1e-6f + j * 1e-3f;
  const float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MFMA_ON) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < R; ++e) {
        const int idx = DEP ? 0 : ((m * R + e) & 7);
        v[idx] = __builtin_fmaf(v[idx], c, 1e-3f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int j = 0; j < 8; ++j) r += v[j];
  for (int s = 0; s < 8; ++s) for (int j = 0; j < 16; ++j) r += acc[s][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int R, int DEP, int MFMA_ON>
int run(const char* name, int waves) {
  float* out;
  const int blocks = 256 * waves, iters = 20000 / waves;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<R, DEP, MFMA_ON>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<R, DEP, MFMA_ON>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-40s waves/SIMD=%d %8.3f ms  %7.1f cycles per SIMD per (%d MFMA + %2d fma) at 2.4 GHz nominal\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / ((double)waves * iters), 8 * MFMA_ON, 8 * R);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 4}) {
    if (run<0, 0, 1>("MFMA only", w)) return 1;
    if (run<4, 0, 0>("fma only R=4, 8 chains", w)) return 1;
    if (run<4, 1, 0>("fma only R=4, 1 chain", w)) return 1;
    if (run<2, 0, 1>("1 MFMA : 2 fma, 8 chains", w)) return 1;
    if (run<4, 0, 1>("1 MFMA : 4 fma, 8 chains", w)) return 1;
    if (run<6, 0, 1>("1 MFMA : 6 fma, 8 chains", w)) return 1;
    if (run<8, 0, 1>("1 MFMA : 8 fma, 8 chains", w)) return 1;
    if (run<8, 0, 0>("fma only R=8, 8 chains", w)) return 1;
    if (run<2, 1, 1>("1 MFMA : 2 fma, 1 chain", w)) return 1;
    if (run<4, 1, 1>("1 MFMA : 4 fma, 1 chain", w)) return 1;
    if (run<8, 1, 1>("1 MFMA : 8 fma, 1 chain", w)) return 1;
  }
  return 0;
}
