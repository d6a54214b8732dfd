// Micro-benchmark for cellmm_kernel's shape: every trip a wave BUILDS an f16 A operand on the VALU (NV plain instructions
// + NT transcendentals, random data) and then issues a burst of 8 MFMAs (32x32x16 f16, eight accumulators, eight B
// operands) that read it.  Two waves per SIMD, either free-running (two 256-thread workgroups per CU) or in an explicit
// ping-pong (one 512-thread workgroup per CU: waves 0-3 build while waves 4-7 multiply, s_barrier, swap).
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_pingpong mfma_pingpong.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// the VALU block: NT transcendentals + NV fmas on 8 chains, then 4 packed conversions into the A operand
template <int NV, int NT>
__device__ __forceinline__ void build(float (&v)[8], f16x8& a) {
#pragma unroll
  for (int e = 0; e < NT; ++e) v[e & 7] = __builtin_amdgcn_exp2f(-v[e & 7] * v[e & 7]) + 0.5f;
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e & 7] = __builtin_fmaf(v[e & 7], 0.7310586f, v[(e + 3) & 7] * 0.3f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 p = __builtin_convertvector(f32x2{v[2 * i], v[2 * i + 1]}, f16x2);
    a[2 * i] = p[0];
    a[2 * i + 1] = p[1];
  }
}

template <int NV, int NT, int PINGPONG, int MFMA_ON>
__global__ void __launch_bounds__(PINGPONG ? 512 : 256) k(float* out, int iters, unsigned seed) {
  unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  auto next = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  f16x8 a, b[8];
  float v[8];
  for (int j = 0; j < 8; ++j) { v[j] = 1.0f + next(); a[j] = (_Float16)next(); }
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 8; ++j) b[t][j] = (_Float16)next();
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  const bool second = PINGPONG && (threadIdx.x >> 8);  // waves 4-7: one phase behind
  if (second) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    build<NV, NT>(v, a);
    __builtin_amdgcn_sched_barrier(0);
    if (PINGPONG) __builtin_amdgcn_s_barrier();
    if (MFMA_ON) {
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[m], acc[m], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PINGPONG) __builtin_amdgcn_s_barrier();
  }
  if (PINGPONG && !second) __builtin_amdgcn_s_barrier();
  float r = 0;
  for (int j = 0; j < 8; ++j) r += v[j];
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) r += acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// The same trip on v_mfma_f32_16x16x32_f16 (round 3): 16 MFMAs of 16384 flop read the built operand
// (sixteen accumulators of four registers, sixteen B operands) -- the flop of the eight 32x32x16 above.  A bare stream of
// this shape sustains 1.11x the 32x32x16 stream on random data at two waves per SIMD (profiles/r03_micro_mfma_stream_16x16x32.txt),
// but every MFMA holds the SIMD's vector issue for 8 cycles and there are twice as many of them per flop.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int NV, int NT>
__global__ void __launch_bounds__(256) k16(float* out, int iters, unsigned seed) {
  unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  auto next = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
  f16x8 a, b[16];
  float v[8];
  for (int j = 0; j < 8; ++j) { v[j] = 1.0f + next(); a[j] = (_Float16)next(); }
  for (int t = 0; t < 16; ++t) for (int j = 0; j < 8; ++j) b[t][j] = (_Float16)next();
  f32x4v acc[16];
  for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
    build<NV, NT>(v, a);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[m], acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = 0;
  for (int j = 0; j < 8; ++j) r += v[j];
  for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) r += acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NV, int NT>
int run16(const char* name) {
  float* out;
  const int blocks = 512, iters = 10000;
  CHECK(hipMalloc(&out, sizeof(float) * 512 * 512));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k16<NV, NT>), dim3(blocks), dim3(256), 0, 0, out, iters, 777u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k16<NV, NT>), dim3(blocks), dim3(256), 0, 0, out, iters, 777u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  printf("%-64s %8.3f ms  %7.1f nominal cycles per SIMD per 16 MFMAs (%4.0f TFLOP/s)\n", name, ms, ms * 1e-3 * 2.4e9 / (2.0 * iters),
         2.0 * iters * 16 * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  return 0;
}

template <int NV, int NT, int PINGPONG, int MFMA_ON>
int run(const char* name) {
  float* out;
  const int blocks = PINGPONG ? 256 : 512, iters = 10000;  // two waves per SIMD either way
  CHECK(hipMalloc(&out, sizeof(float) * 512 * 512));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NV, NT, PINGPONG, MFMA_ON>), dim3(blocks), dim3(PINGPONG ? 512 : 256), 0, 0, out, iters, 777u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NV, NT, PINGPONG, MFMA_ON>), dim3(blocks), dim3(PINGPONG ? 512 : 256), 0, 0, out, iters, 777u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  // a SIMD runs 2 waves x iters trips of (8 MFMA + NV + NT + 4 VALU)
  printf("%-64s %8.3f ms  %7.1f nominal cycles per SIMD per 8 MFMAs  (%4.0f TFLOP/s)\n", name, ms, ms * 1e-3 * 2.4e9 / (2.0 * iters),
         MFMA_ON ? 2.0 * iters * 8 * 1024 * 32768.0 / (ms * 1e-3) / 1e12 : 0.0);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  if (run<0, 0, 0, 1>("8 MFMA + 4 cvt, free-running")) return 1;
  if (run<44, 4, 0, 0>("52 VALU (44 fma + 4 exp + 4 cvt) alone")) return 1;
  if (run<44, 4, 0, 1>("8 MFMA + 52 VALU, free-running (2 x 256-thread blocks per CU)")) return 1;
  if (run<44, 4, 1, 1>("8 MFMA + 52 VALU, ping-pong (one 512-thread block per CU)")) return 1;
  if (run<20, 4, 0, 1>("8 MFMA + 28 VALU, free-running")) return 1;
  if (run<20, 4, 1, 1>("8 MFMA + 28 VALU, ping-pong")) return 1;
  if (run16<0, 0>("16 x 16x16x32 MFMA + 4 cvt, free-running")) return 1;
  if (run16<20, 4>("16 x 16x16x32 MFMA + 28 VALU, free-running")) return 1;
  if (run<20, 4, 0, 1>("8 MFMA + 28 VALU, free-running (again)")) return 1;
  if (run16<44, 4>("16 x 16x16x32 MFMA + 52 VALU, free-running")) return 1;
  if (run<76, 4, 0, 1>("8 MFMA + 84 VALU, free-running")) return 1;
  if (run<76, 4, 1, 1>("8 MFMA + 84 VALU, ping-pong")) return 1;
  return 0;
}
