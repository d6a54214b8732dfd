// Micro-benchmark: a wave that interleaves bf16 MFMAs with transcendentals (1 MFMA : R v_exp_f32),
// 1 and 2 waves per SIMD, MFMAs independent or chained through the accumulator -- what overlap
// of the matrix pipe and the VALU can one instruction stream reach on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_interleave mfma_interleave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// per loop trip: 8 MFMAs and 8*R exps.  CHAIN: 0 = 8 independent accumulators (4 used twice),
// 1 = two chains of 4 dependent MFMAs.
template <int R, int CHAIN, int MFMA_ON, int EXP_ON>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(1.0f + j); }
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + threadIdx.x * 1e-6f + j * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MFMA_ON) {
        const int slot = CHAIN ? (m & 1) : (m & 3);
        acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[slot], 0, 0, 0);
      }
      if (EXP_ON) {
#pragma unroll
        for (int e = 0; e < R; ++e) v[(m * R + e) & 7] = __builtin_amdgcn_exp2f(v[(m * R + e) & 7]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int j = 0; j < 8; ++j) r += v[j];
  for (int s = 0; s < 4; ++s) for (int j = 0; j < 16; ++j) r += acc[s][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int R, int CHAIN, int MFMA_ON, int EXP_ON>
int run(const char* name, int waves) {
  float* out;
  const int blocks = 256 * waves, iters = 20000 / waves;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<R, CHAIN, MFMA_ON, EXP_ON>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<R, CHAIN, MFMA_ON, EXP_ON>), dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // cycles per trip of ONE wave slot: a SIMD runs `waves` waves x iters trips
  printf("%-44s waves/SIMD=%d %8.3f ms  %7.1f cycles per SIMD per (8 MFMA + %2d exp) at 2.4 GHz\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / ((double)waves * iters), 8 * R * EXP_ON);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2}) {
    if (run<4, 0, 1, 0>("MFMA only, independent", w)) return 1;
    if (run<4, 1, 1, 0>("MFMA only, 2 chains", w)) return 1;
    if (run<4, 0, 0, 1>("exp only (R=4)", w)) return 1;
    if (run<2, 0, 1, 1>("1 MFMA : 2 exp, independent", w)) return 1;
    if (run<4, 0, 1, 1>("1 MFMA : 4 exp, independent", w)) return 1;
    if (run<4, 1, 1, 1>("1 MFMA : 4 exp, 2 chains", w)) return 1;
    if (run<6, 0, 1, 1>("1 MFMA : 6 exp, independent", w)) return 1;
    if (run<6, 0, 0, 1>("exp only (R=6)", w)) return 1;
  }
  return 0;
}
