// Micro-benchmark: issue cost (cycles per wave64 instruction on one SIMD) of the VALU
// instructions the pair loops are made of, for 1, 2 and 4 resident waves per SIMD:
// v_exp_f32, v_sqrt_f32, v_rsq_f32, v_rcp_f32, v_log_f32, v_fma_f32, v_pk_fma_f32,
// v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_bf16_f32, v_ldexp_f32, v_fract_f32, and the f16 conversions / v_fma_mix_f32 of
// fastmm_kernel.
// 16 independent registers per wave (issue behaviour, not latency).
// Build: hipcc --offload-arch=gfx950 -O3 -o trans_rates trans_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define OPS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
              "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
#define P8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define POPS : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])

#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n\t"
#define I_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n\t"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define I_LOG(i) "v_log_f32 %" #i ", %" #i "\n\t"
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n\t"
#define I_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", 1\n\t"
#define I_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n\t"
#define I_CVT(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %" #i "\n\t"
#define I_CVTH(i) "v_cvt_pk_f16_f32 %" #i ", %" #i ", %" #i "\n\t"
#define I_CVTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %" #i "\n\t"
#define I_MIX(i) "v_fma_mix_f32 %" #i ", %" #i ", -1.0, %" #i " op_sel_hi:[1,0,0]\n\t"
#define I_AND(i) "v_and_b32 %" #i ", 0xffffe000, %" #i "\n\t"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %" #i ", %" #i "\n\t"
#define I_SUB(i) "v_sub_f32 %" #i ", %" #i ", %" #i "\n\t"
#define I_CVTF(i) "v_cvt_f32_f16 %" #i ", %" #i "\n\t"
#define I_EXPH(i) "v_exp_f16 %" #i ", %" #i "\n\t"
#define I_SQRTH(i) "v_sqrt_f16 %" #i ", %" #i "\n\t"
#define I_EXPL(i) "v_exp_legacy_f32 %" #i ", %" #i "\n\t"
#define I_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %" #i "\n\t"
#define I_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %" #i "\n\t"
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %" #i "\n\t"

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float r[16];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = 1.0f + threadIdx.x * 1e-6f + i * 1e-3f;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{1.0f + threadIdx.x * 1e-6f, 1.0f + i * 1e-3f};
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) asm volatile(R16(I_EXP) OPS);
    if constexpr (MODE == 1) asm volatile(R16(I_SQRT) OPS);
    if constexpr (MODE == 2) asm volatile(R16(I_RSQ) OPS);
    if constexpr (MODE == 3) asm volatile(R16(I_RCP) OPS);
    if constexpr (MODE == 4) asm volatile(R16(I_LOG) OPS);
    if constexpr (MODE == 5) asm volatile(R16(I_FMA) OPS);
    if constexpr (MODE == 6) asm volatile(R16(I_LDEXP) OPS);
    if constexpr (MODE == 7) asm volatile(R16(I_FRACT) OPS);
    if constexpr (MODE == 8) asm volatile(R16(I_CVT) OPS);
    if constexpr (MODE == 12) asm volatile(R16(I_CVTH) OPS);
    if constexpr (MODE == 13) asm volatile(R16(I_CVTZ) OPS);
    if constexpr (MODE == 14) asm volatile(R16(I_MIX) OPS);
    if constexpr (MODE == 15) asm volatile(R16(I_AND) OPS);
    if constexpr (MODE == 16) asm volatile(R16(I_PERM) OPS);
    if constexpr (MODE == 17) asm volatile(R16(I_SUB) OPS);
    if constexpr (MODE == 18) asm volatile(R16(I_CVTF) OPS);
    if constexpr (MODE == 19) asm volatile(R16(I_EXPH) OPS);
    if constexpr (MODE == 20) asm volatile(R16(I_SQRTH) OPS);
    if constexpr (MODE == 21) asm volatile(R16(I_EXPL) OPS);
    if constexpr (MODE == 9) asm volatile(P8(I_PKFMA) P8(I_PKFMA) POPS);
    if constexpr (MODE == 10) asm volatile(P8(I_PKADD) P8(I_PKADD) POPS);
    if constexpr (MODE == 11) asm volatile(P8(I_PKMUL) P8(I_PKMUL) POPS);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int waves) {
  float* out;
  const int blocks = 256 * waves, iters = 40000 / waves;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // one SIMD executes waves * iters * 16 instructions
  printf("%-20s waves/SIMD=%d  %8.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz\n", name, waves, ms,
         ms * 1e-3 * 2.4e9 / ((double)waves * iters * 16));
  CHECK(hipFree(out));
  return 0;
}

int main() {
  for (int w : {1, 2, 4}) {
    if (run<0>("v_exp_f32", w)) return 1;
    if (run<1>("v_sqrt_f32", w)) return 1;
    if (run<2>("v_rsq_f32", w)) return 1;
    if (run<3>("v_rcp_f32", w)) return 1;
    if (run<4>("v_log_f32", w)) return 1;
    if (run<5>("v_fma_f32", w)) return 1;
    if (run<6>("v_ldexp_f32", w)) return 1;
    if (run<7>("v_fract_f32", w)) return 1;
    if (run<8>("v_cvt_pk_bf16_f32", w)) return 1;
    if (run<12>("v_cvt_pk_f16_f32", w)) return 1;
    if (run<13>("v_cvt_pkrtz_f16_f32", w)) return 1;
    if (run<14>("v_fma_mix_f32", w)) return 1;
    if (run<15>("v_and_b32", w)) return 1;
    if (run<16>("v_perm_b32", w)) return 1;
    if (run<17>("v_sub_f32", w)) return 1;
    if (run<18>("v_cvt_f32_f16", w)) return 1;
    if (run<19>("v_exp_f16", w)) return 1;
    if (run<20>("v_sqrt_f16", w)) return 1;
    if (run<21>("v_exp_legacy_f32", w)) return 1;
    if (run<9>("v_pk_fma_f32", w)) return 1;
    if (run<10>("v_pk_add_f32", w)) return 1;
    if (run<11>("v_pk_mul_f32", w)) return 1;
  }
  return 0;
}
