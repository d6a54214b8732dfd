// Micro-benchmark: does gfx950's fp64 MFMA (v_mfma_f64_16x16x4_f64) run beside fp64 VALU work (v_fma_f64) of another
// wave of the same SIMD, or do the two share the fp64 datapath?  Also prints the output layout of the MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_overlap mfma_f64_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double f64x4 __attribute__((ext_vector_type(4)));

// MODE 0: every wave MFMA only; 1: every wave fma only; 2: waves 0-3 MFMA, waves 4-7 fma (1 + 1 per SIMD);
// MODE 3: every wave alternates 8 MFMA / 64 fma
template <int MODE>
__global__ void __launch_bounds__(512) k(double* out, int iters) {
  const int wave = threadIdx.x >> 6;
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
  f64x4 acc[8];
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.0;
  double v[16];
  for (int j = 0; j < 16; ++j) v[j] = 1.0 + threadIdx.x * 1e-6 + j * 1e-3;
  const bool do_mfma = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_fma(v[j], 0.999999, 1e-9);
    }
  }
  double r = 0;
  for (int j = 0; j < 16; ++j) r += v[j];
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 4; ++j) r += acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ void layout(double* out) {
  // A[m][k] = 100 m + k, B[k][n] = (k == 0) n + (k == 1) 1000: D[m][n] = 100 m n + 1000 (100 m + 1)
  const int l = threadIdx.x;
  const double a = 100.0 * (l % 16) + (l / 16);
  const double b = (l / 16 == 0) ? (double)(l % 16) : ((l / 16 == 1) ? 1000.0 : 0.0);
  f64x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = c[j];
}

template <int MODE>
int run(const char* name) {
  double* out;
  const int blocks = 256, iters = 20000;
  CHECK(hipMalloc(&out, sizeof(double) * 512 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-60s %8.3f ms  (%.1f cycles per iteration at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  double* out; CHECK(hipMalloc(&out, sizeof(double) * 256));
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
  double h[256]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  // decode: D[m][n] = 100 m n + 1000 (100 m + 1) -> recover (m, n) per (lane, register)
  int ok = 1;
  for (int l = 0; l < 64 && ok; ++l)
    for (int j = 0; j < 4; ++j) {
      const int m = 4 * (l / 16) + j, n = l % 16;  // the guess
      if (h[l * 4 + j] != 100.0 * m * n + 1000.0 * (100.0 * m + 1)) ok = 0;
    }
  printf("output layout: lane l register j holds D[4 (l / 16) + j][l %% 16]: %s\n", ok ? "confirmed" : "NOT confirmed");
  if (!ok) for (int l = 0; l < 64; l += 5) printf("  lane %d: %.0f %.0f %.0f %.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  // per iteration: an MFMA wave issues 8 MFMA (16x16x4 f64), a VALU wave issues 64 v_fma_f64
  if (run<0>("8 waves/CU x 8 f64 MFMA (2 MFMA waves per SIMD)")) return 1;
  if (run<1>("8 waves/CU x 64 v_fma_f64 (2 VALU waves per SIMD)")) return 1;
  if (run<2>("4 waves MFMA + 4 waves fma (1 + 1 per SIMD)")) return 1;
  if (run<3>("8 waves, each 8 MFMA then 64 fma (2 per SIMD)")) return 1;
  return 0;
}
