// Micro-benchmark: issue ceilings of the fp32 VALU and the transcendental unit on
// gfx950, and of the exact instruction mix of the D=3 Gaussian pair loop
// (3 sub + 1 mul + 2 fma + 1 exp2 + 1 fma per pair).  Prints pairs/s ceilings used
// as the roofline of the low-D kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 65536;

template <int MODE>
__global__ void __launch_bounds__(256) bench(float* out, float seed) {
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE == 0) {         // 8 independent fma chains
        a[i] = fmaf(a[i], b[i], 1.0f);
      } else if constexpr (MODE == 1) {  // 8 independent exp2 chains
        a[i] = __builtin_amdgcn_exp2f(a[i]);
      } else if constexpr (MODE == 2 || MODE == 3) {
        // pair-loop mix, exactly 3 sub, mul, 2 fma, 1 transcendental, fma per body.
        // a[i] = x coordinate(s) of target i (reused for all 3 axes), source = SGPR values
        const float sy0 = __builtin_amdgcn_readfirstlane(__float_as_int(seed + it * 1e-3f)) * 1e-9f;
        float dx = a[i] - sy0, dy = a[i] - (sy0 + 1.0f), dz = a[i] - (sy0 + 2.0f);
        float s = dx * dx; s = fmaf(dy, dy, s); s = fmaf(dz, dz, s);
        float k = MODE == 2 ? __builtin_amdgcn_exp2f(-s) : __builtin_amdgcn_rsqf(s);
        b[i] = fmaf(k, sy0, b[i]);
      } else if constexpr (MODE == 4) {  // 7 fma : 1 exp, all independent
        float k = __builtin_amdgcn_exp2f(a[i]);
        float t = fmaf(a[i], b[i], 1.0f);
        t = fmaf(t, b[i], 1.0f); t = fmaf(t, b[i], 1.0f); t = fmaf(t, b[i], 1.0f);
        t = fmaf(t, b[i], 1.0f); t = fmaf(t, b[i], 1.0f); t = fmaf(t, b[i], k);
        a[i] = t;
      }
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
int run(const char* name, double ops_per_inner, int blocks) {
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double inner = 5.0 * blocks * 256.0 * ITERS * 8;  // lane-level inner bodies
  printf("%-34s blocks=%5d  %8.3f ms  %.3e bodies/s  %.3e lane-ops/s\n", name, blocks, ms / 5,
         inner / (ms * 1e-3), inner * ops_per_inner / (ms * 1e-3));
  CHECK(hipFree(out));
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int blocks : {256 * 4, 256 * 8}) {
    if (run<0>("fma x8 chains", 1, blocks)) return 1;
    if (run<1>("exp2 x8 chains", 1, blocks)) return 1;
    if (run<4>("7 fma + 1 exp2", 8, blocks)) return 1;
    if (run<2>("gaussian pair mix (8 ops/pair)", 1, blocks)) return 1;
    if (run<3>("1/r pair mix (8 ops/pair)", 1, blocks)) return 1;
  }
  return 0;
}
