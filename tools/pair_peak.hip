// Micro-benchmark: the D=3 Gaussian pair loop with NO memory traffic.  Sources are
// wave-uniform values derived from the loop counter with integer SALU ops (so they
// live in SGPRs exactly like the s_load'ed records of lowd_kernel); targets and sums
// are per-lane VGPRs.  Gives the issue ceiling of this instruction stream.
// Build: hipcc --offload-arch=gfx950 -O3 -o pair_peak pair_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T, int KIND>
__global__ void __launch_bounds__(256) pairs(float* out, int iters, int base) {
  float x[T][3], acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    x[t][0] = threadIdx.x * 1e-3f + t;
    x[t][1] = threadIdx.x * 2e-3f + t;
    x[t][2] = threadIdx.x * 3e-3f + t;
    acc[t] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = base + it * 8 + u;  // uniform integer -> SGPR
      const float y0 = __int_as_float(0x3f000000 + q);
      const float y1 = __int_as_float(0x3f100000 + q);
      const float y2 = __int_as_float(0x3f200000 + q);
      const float b = __int_as_float(0x3f300000 + q);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float d = x[t][0] - y0;
        float s = d * d;
        d = x[t][1] - y1;
        s = fmaf(d, d, s);
        d = x[t][2] - y2;
        s = fmaf(d, d, s);
        float k;
        if constexpr (KIND == 0) k = __builtin_amdgcn_exp2f(-s);
        else if constexpr (KIND == 1) k = __builtin_amdgcn_rsqf(s);
        else k = s;  // no transcendental: the full-rate part alone
        acc[t] = fmaf(k, b, acc[t]);
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int t = 0; t < T; ++t) r += acc[t];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int T, int KIND>
int run(const char* name, int blocks, int iters) {
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((pairs<T, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((pairs<T, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double npairs = (double)blocks * 256.0 * iters * 8 * T;
  printf("%-28s T=%d blocks=%6d  %9.3f ms  %.3e pairs/s\n", name, T, blocks, ms, npairs / (ms * 1e-3));
  CHECK(hipFree(out));
  return 0;
}

int main() {
  const int it = 1 << 15;
  for (int wpb : {4, 8}) {  // blocks per CU
    const int blocks = 256 * wpb;
    if (run<1, 0>("gaussian", blocks, it)) return 1;
    if (run<2, 0>("gaussian", blocks, it / 2)) return 1;
    if (run<4, 0>("gaussian", blocks, it / 4)) return 1;
    if (run<8, 0>("gaussian", blocks, it / 8)) return 1;
    if (run<4, 1>("inverse distance", blocks, it / 4)) return 1;
    if (run<4, 2>("no transcendental (7 ops)", blocks, it / 4)) return 1;
  }
  return 0;
}
