// Micro-benchmark: how v_exp_f32 (quarter-rate transcendental) shares the VALU with
// full-rate ops on gfx950, as a function of burst length and waves per SIMD.
// Independent registers only (issue behaviour, not latency).
// Build: hipcc --offload-arch=gfx950 -O3 -o trans_mix trans_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define FMA(r) "v_fma_f32 %" #r ", %" #r ", %" #r ", %" #r "\n\t"
#define EXP(r) "v_exp_f32 %" #r ", %" #r "\n\t"
#define FMA7 FMA(8) FMA(9) FMA(10) FMA(11) FMA(12) FMA(13) FMA(14)
#define OPS  : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
               "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])

// every variant executes 8 exp + 56 fma per loop trip
template <int MODE>
__global__ void __launch_bounds__(256) mix(float* out, int iters) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + threadIdx.x * 1e-6f + i * 1e-3f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {  // burst 1: exp, 7 fma, exp, 7 fma ...
      asm volatile(EXP(0) FMA7 EXP(1) FMA7 EXP(2) FMA7 EXP(3) FMA7 EXP(4) FMA7 EXP(5) FMA7 EXP(6) FMA7 EXP(7) FMA7 OPS);
    } else if constexpr (MODE == 1) {  // burst 2
      asm volatile(EXP(0) EXP(1) FMA7 FMA7 EXP(2) EXP(3) FMA7 FMA7 EXP(4) EXP(5) FMA7 FMA7 EXP(6) EXP(7) FMA7 FMA7 OPS);
    } else if constexpr (MODE == 2) {  // burst 4
      asm volatile(EXP(0) EXP(1) EXP(2) EXP(3) FMA7 FMA7 FMA7 FMA7 EXP(4) EXP(5) EXP(6) EXP(7) FMA7 FMA7 FMA7 FMA7 OPS);
    } else if constexpr (MODE == 3) {  // burst 8
      asm volatile(EXP(0) EXP(1) EXP(2) EXP(3) EXP(4) EXP(5) EXP(6) EXP(7) FMA7 FMA7 FMA7 FMA7 FMA7 FMA7 FMA7 FMA7 OPS);
    } else if constexpr (MODE == 4) {  // fma only (56)
      asm volatile(FMA7 FMA7 FMA7 FMA7 FMA7 FMA7 FMA7 FMA7 OPS);
    } else if constexpr (MODE == 5) {  // exp only (8)
      asm volatile(EXP(0) EXP(1) EXP(2) EXP(3) EXP(4) EXP(5) EXP(6) EXP(7) OPS);
    } else if constexpr (MODE == 6) {  // exp, 3 fma, exp, 4 fma (finer interleave)
      asm volatile(EXP(0) FMA(8) FMA(9) FMA(10) EXP(1) FMA(11) FMA(12) FMA(13) FMA(14) FMA7 EXP(2) FMA(8) FMA(9) FMA(10) EXP(3) FMA(11) FMA(12) FMA(13) FMA(14) FMA7
                   EXP(4) FMA(8) FMA(9) FMA(10) EXP(5) FMA(11) FMA(12) FMA(13) FMA(14) FMA7 EXP(6) FMA(8) FMA(9) FMA(10) EXP(7) FMA(11) FMA(12) FMA(13) FMA(14) FMA7 OPS);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int blocks, int iters) {
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mix<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(mix<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // cycles per loop trip per SIMD at 2.4 GHz, waves per SIMD = blocks/256
  const double trips_per_simd = (double)blocks / 256.0 * iters;  // 4 waves per block, 4 SIMDs per CU
  printf("%-22s waves/SIMD=%d  %8.3f ms  %7.1f cyc/trip/wave-on-SIMD (2.4GHz)\n", name, blocks / 256, ms,
         ms * 1e-3 * 2.4e9 / trips_per_simd);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    const int blocks = 256 * w, iters = 200000 / w;
    if (run<4>("56 fma", blocks, iters)) return 1;
    if (run<5>("8 exp", blocks, iters)) return 1;
    if (run<0>("8 exp + 56 fma burst1", blocks, iters)) return 1;
    if (run<6>("8 exp + 56 fma fine", blocks, iters)) return 1;
    if (run<1>("8 exp + 56 fma burst2", blocks, iters)) return 1;
    if (run<2>("8 exp + 56 fma burst4", blocks, iters)) return 1;
    if (run<3>("8 exp + 56 fma burst8", blocks, iters)) return 1;
  }
  return 0;
}
