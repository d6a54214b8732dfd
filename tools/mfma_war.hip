// Micro-test: is it safe to overwrite the A (or B) operand registers of a bf16 32x32x16 MFMA
// with a VALU instruction issued N wait states after the MFMA?  Prints, per N, how many of
// the 1024 outputs differ from the undisturbed result.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_war mfma_war.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NOPS, int WHICH>  // WHICH 0: clobber A, 1: clobber B
__global__ void k(float* out) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x * 7 + j * 3) % 13 - 6); b[j] = (__bf16)(float)((threadIdx.x * 5 + j) % 11 - 5); }
  f32x16 d;
  for (int q = 0; q < 16; ++q) d[q] = 0.f;
  // MFMA, NOPS wait states, then VALU writes over the operand registers (fixed v[60:63])
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
  if constexpr (WHICH == 0) {
    asm volatile("v_mov_b32 v60, %1\n\tv_mov_b32 v61, %2\n\tv_mov_b32 v62, %3\n\tv_mov_b32 v63, %4\n\ts_nop 4\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, v[60:63], %5, 0\n\t"
                 ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                 "v_mov_b32 v60, 0x7fc07fc0\n\tv_mov_b32 v61, 0x7fc07fc0\n\tv_mov_b32 v62, 0x7fc07fc0\n\tv_mov_b32 v63, 0x7fc07fc0\n\t"
                 "s_nop 15\n\ts_nop 15"
                 : "=&v"(d) : "v"(ua[0]), "v"(ua[1]), "v"(ua[2]), "v"(ua[3]), "v"(b), "i"(NOPS) : "v60", "v61", "v62", "v63");
  } else {
    asm volatile("v_mov_b32 v60, %1\n\tv_mov_b32 v61, %2\n\tv_mov_b32 v62, %3\n\tv_mov_b32 v63, %4\n\ts_nop 4\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %5, v[60:63], 0\n\t"
                 ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                 "v_mov_b32 v60, 0x7fc07fc0\n\tv_mov_b32 v61, 0x7fc07fc0\n\tv_mov_b32 v62, 0x7fc07fc0\n\tv_mov_b32 v63, 0x7fc07fc0\n\t"
                 "s_nop 15\n\ts_nop 15"
                 : "=&v"(d) : "v"(ub[0]), "v"(ub[1]), "v"(ub[2]), "v"(ub[3]), "v"(a), "i"(NOPS) : "v60", "v61", "v62", "v63");
  }
  for (int q = 0; q < 16; ++q) out[threadIdx.x * 16 + q] = d[q];
}

__global__ void ref(float* out) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)((threadIdx.x * 7 + j * 3) % 13 - 6); b[j] = (__bf16)(float)((threadIdx.x * 5 + j) % 11 - 5); }
  f32x16 d;
  for (int q = 0; q < 16; ++q) d[q] = 0.f;
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
  for (int q = 0; q < 16; ++q) out[threadIdx.x * 16 + q] = d[q];
}

template <int NOPS, int WHICH>
int run(const float* want) {
  float* out; CHECK(hipMalloc(&out, 1024 * 4));
  hipLaunchKernelGGL((k<NOPS, WHICH>), dim3(1), dim3(64), 0, 0, out);
  float h[1024]; CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  int bad = 0, badlo = 0;
  for (int i = 0; i < 1024; ++i) if (!(h[i] == want[i])) { ++bad; }
  printf("clobber %s after %2d wait states: %4d of 1024 outputs differ\n", WHICH ? "B" : "A", NOPS, bad);
  (void)badlo; CHECK(hipFree(out));
  return 0;
}

int main() {
  float* out; CHECK(hipMalloc(&out, 1024 * 4));
  hipLaunchKernelGGL(ref, dim3(1), dim3(64), 0, 0, out);
  float want[1024]; CHECK(hipMemcpy(want, out, sizeof(want), hipMemcpyDeviceToHost));
  if (run<0, 0>(want) || run<1, 0>(want) || run<2, 0>(want) || run<3, 0>(want) || run<4, 0>(want) || run<6, 0>(want) || run<8, 0>(want) || run<12, 0>(want) || run<16, 0>(want)) return 1;
  if (run<0, 1>(want) || run<1, 1>(want) || run<2, 1>(want) || run<3, 1>(want) || run<4, 1>(want) || run<6, 1>(want) || run<8, 1>(want) || run<12, 1>(want) || run<16, 1>(want)) return 1;
  return 0;
}
