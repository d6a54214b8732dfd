// Micro-benchmark: what does a stream of independent 8-pass MFMAs (v_mfma_f32_32x32x16_f16, 8 accumulators,
// 2 waves per SIMD) sustain on gfx950, as a function of
//   B operand: one register quad for all eight MFMAs, or eight different ones (cellmm_kernel: one per target tile)
//   data:      zeros / small integers, or random values (switching power -> clock)
//   accumulators in VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1) or wherever the compiler puts them
// Prints ms and MFMA-pipe cycles per MFMA at the NOMINAL 2.4 GHz (the clock under load is lower).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NB, int RANDOM>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned seed) {
  f16x8 a, b[8];
  unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  auto next = [&]() { s = s * 1664525u + 1013904223u; return RANDOM ? (float)((s >> 8) & 0xffff) / 65536.f - 0.5f : 0.f; };
  // RANDOM 2 / 3: the low 4 / 7 mantissa bits of every f16 operand element cleared; 4: every second k-slot zero
  auto trim = [&](float v, int j) {
    _Float16 hv = (_Float16)v;
    unsigned short bits = __builtin_bit_cast(unsigned short, hv);
    if (RANDOM == 2) bits &= 0xfff0u;
    if (RANDOM == 3) bits &= 0xff80u;
    if (RANDOM == 4 && (j & 1)) bits = 0;
    return __builtin_bit_cast(_Float16, bits);
  };
  for (int j = 0; j < 8; ++j) a[j] = trim(next(), j);
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 8; ++j) b[t][j] = trim(next(), j);
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[NB == 1 ? 0 : m], acc[m], 0, 0, 0);
    if (RANDOM) {  // keep the operand changing from trip to trip, as a rebuilt A does (one VALU instruction per register)
      auto w = __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, a);
      w[it & 3] ^= (RANDOM == 3 ? 0x00800080u : (RANDOM == 2 ? 0x00100010u : 0x00010001u)) << (it & 7);
      if (RANDOM == 4) w[it & 3] &= 0x0000ffffu;
      a = __builtin_bit_cast(f16x8, w);
    }
  }
  float r = 0;
  for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) r += acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// The same stream on v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH 'DVFS give-back' item 7: on random
// data the 16x16x32 shape held a higher clock, ~1.15x the FLOP/s of 32x32x16 at equal cycles per flop): 16 MFMAs of
// 16384 flop per trip = the flop of the eight 32x32x16 above, sixteen accumulators of four registers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int RANDOM>
__global__ void __launch_bounds__(256) k16(float* out, int iters, unsigned seed) {
  f16x8 a, b[16];
  unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  auto next = [&]() { s = s * 1664525u + 1013904223u; return RANDOM ? (float)((s >> 8) & 0xffff) / 65536.f - 0.5f : 0.f; };
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)next();
  for (int t = 0; t < 16; ++t) for (int j = 0; j < 8; ++j) b[t][j] = (_Float16)next();
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[m], acc[m], 0, 0, 0);
    if (RANDOM) {
      auto w = __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, a);
      w[it & 3] ^= 0x00010001u << (it & 7);
      a = __builtin_bit_cast(f16x8, w);
    }
  }
  float r = 0;
  for (int t = 0; t < 16; ++t) for (int j = 0; j < 4; ++j) r += acc[t][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int RANDOM>
int run16(const char* name, int waves) {
  float* out;
  const int blocks = 256 * waves, iters = 40000 / waves;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k16<RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k16<RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double mfmas_per_simd = (double)waves * iters * 16;
  printf("%-44s waves/SIMD=%d %8.3f ms  %6.2f nominal cycles per MFMA  (%.0f TFLOP/s of 2500)\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / mfmas_per_simd, mfmas_per_simd * 1024 * 16384.0 / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  return 0;
}

template <int NB, int RANDOM>
int run(const char* name, int waves) {
  float* out;
  const int blocks = 256 * waves, iters = 40000 / waves;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NB, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NB, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double mfmas_per_simd = (double)waves * iters * 8;
  printf("%-44s waves/SIMD=%d %8.3f ms  %6.2f nominal cycles per MFMA  (%.0f TFLOP/s of 2500)\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / mfmas_per_simd, mfmas_per_simd * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2}) {
    if (run<1, 0>("one B, zero data", w)) return 1;
    if (run<8, 0>("eight B, zero data", w)) return 1;
    if (run<1, 1>("one B, random data", w)) return 1;
    if (run<8, 1>("eight B, random data", w)) return 1;
    if (run<8, 2>("eight B, random data, 7-bit mantissas", w)) return 1;
    if (run<8, 3>("eight B, random data, 4-bit mantissas", w)) return 1;
    if (run<8, 4>("eight B, random data, odd k-slots zero", w)) return 1;
    if (run16<0>("16x16x32: sixteen B, zero data", w)) return 1;
    if (run16<1>("16x16x32: sixteen B, random data", w)) return 1;
    if (run<8, 1>("eight B, random data (again, after 16x16x32)", w)) return 1;
  }
  return 0;
}
