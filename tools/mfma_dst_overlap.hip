// Micro-test: v_mfma_f32_32x32x16_bf16 whose destination v[16:31] overlaps its A source v[16:19]
// (the register allocation LLVM produces for "d = mfma(a, b, 0)" when a dies at the MFMA), issued
// (1) into an idle matrix pipe and (2) right behind two other MFMAs, against the same product
// with disjoint registers.  Prints the number of differing results per 16-lane group.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_dst_overlap mfma_dst_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define LOAD_AB \
  "v_mov_b32 v16, %16\n v_mov_b32 v17, %17\n v_mov_b32 v18, %18\n v_mov_b32 v19, %19\n" \
  "v_mov_b32 v48, %20\n v_mov_b32 v49, %21\n v_mov_b32 v50, %22\n v_mov_b32 v51, %23\n" \
  "v_mov_b32 v72, %16\n v_mov_b32 v73, %17\n v_mov_b32 v74, %18\n v_mov_b32 v75, %19\n" \
  "v_mov_b32 v40, %20\n v_mov_b32 v41, %21\n v_mov_b32 v42, %22\n v_mov_b32 v43, %23\n" \
  "s_nop 4\n"
#define QUEUE \
  "v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[40:43], 0\n" \
  "v_mfma_f32_32x32x16_bf16 v[0:15], v[72:75], v[40:43], v[0:15]\n"
#define STORE(base) \
  "s_nop 15\n s_nop 15\n s_nop 15\n" \
  "v_mov_b32 %0, v" #base "\n"
#define OUTS "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]), "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]), \
             "=v"(d[8]), "=v"(d[9]), "=v"(d[10]), "=v"(d[11]), "=v"(d[12]), "=v"(d[13]), "=v"(d[14]), "=v"(d[15])
#define INS "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3])
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
  "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
  "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
  "v48","v49","v50","v51","v72","v73","v74","v75"
#define COPY16(b0,b1,b2,b3,b4,b5,b6,b7,b8,b9,b10,b11,b12,b13,b14,b15) \
  "s_nop 15\n s_nop 15\n s_nop 15\n" \
  "v_mov_b32 %0, v" #b0 "\n v_mov_b32 %1, v" #b1 "\n v_mov_b32 %2, v" #b2 "\n v_mov_b32 %3, v" #b3 "\n" \
  "v_mov_b32 %4, v" #b4 "\n v_mov_b32 %5, v" #b5 "\n v_mov_b32 %6, v" #b6 "\n v_mov_b32 %7, v" #b7 "\n" \
  "v_mov_b32 %8, v" #b8 "\n v_mov_b32 %9, v" #b9 "\n v_mov_b32 %10, v" #b10 "\n v_mov_b32 %11, v" #b11 "\n" \
  "v_mov_b32 %12, v" #b12 "\n v_mov_b32 %13, v" #b13 "\n v_mov_b32 %14, v" #b14 "\n v_mov_b32 %15, v" #b15 "\n"
#define FROM16 COPY16(16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31)
#define FROM32 COPY16(32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47)

#define CHAIN "v_mfma_f32_32x32x16_bf16 v[16:31], v[16:19], v[48:51], 0\n v_mfma_f32_32x32x16_bf16 v[16:31], v[72:75], v[48:51], v[16:31]\n"
#define READ16_NOW \
  "v_mov_b32 %0, v16\n v_mov_b32 %1, v17\n v_mov_b32 %2, v18\n v_mov_b32 %3, v19\n" \
  "v_mov_b32 %4, v20\n v_mov_b32 %5, v21\n v_mov_b32 %6, v22\n v_mov_b32 %7, v23\n" \
  "v_mov_b32 %8, v24\n v_mov_b32 %9, v25\n v_mov_b32 %10, v26\n v_mov_b32 %11, v27\n" \
  "v_mov_b32 %12, v28\n v_mov_b32 %13, v29\n v_mov_b32 %14, v30\n v_mov_b32 %15, v31\n"

// two dependent MFMAs (the second accumulates onto the first), NOPS wait states, then VALU reads:
// how many wait states does the result need, alone and queued behind two other MFMAs?
template <int NOPS, int QUEUED>
__global__ void kl(const unsigned* av, const unsigned* bv, float* out) {
  unsigned a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = av[threadIdx.x * 4 + j]; b[j] = bv[threadIdx.x * 4 + j]; }
  float d[16];
#define NOPCASE(N, STR) \
  if constexpr (NOPS == N && !QUEUED) asm volatile(LOAD_AB CHAIN STR READ16_NOW : OUTS : INS : CLOB); \
  if constexpr (NOPS == N && QUEUED) asm volatile(LOAD_AB QUEUE CHAIN STR READ16_NOW : OUTS : INS : CLOB);
  NOPCASE(0, "")
  NOPCASE(4, "s_nop 3\n")
  NOPCASE(8, "s_nop 7\n")
  NOPCASE(11, "s_nop 10\n")
  NOPCASE(14, "s_nop 13\n")
  NOPCASE(18, "s_nop 15\n s_nop 1\n")
  NOPCASE(24, "s_nop 15\n s_nop 7\n")
  NOPCASE(64, "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n")
  for (int q = 0; q < 16; ++q) out[threadIdx.x * 16 + q] = d[q];
}

template <int MODE>
__global__ void k(const unsigned* av, const unsigned* bv, float* out) {
  unsigned a[4], b[4];
  for (int j = 0; j < 4; ++j) { a[j] = av[threadIdx.x * 4 + j]; b[j] = bv[threadIdx.x * 4 + j]; }
  float d[16];
  if constexpr (MODE == 0)  // disjoint registers, idle pipe
    asm volatile(LOAD_AB "v_mfma_f32_32x32x16_bf16 v[32:47], v[16:19], v[48:51], 0\n" FROM32 : OUTS : INS : CLOB);
  if constexpr (MODE == 1)  // dst overlaps A, idle pipe
    asm volatile(LOAD_AB "v_mfma_f32_32x32x16_bf16 v[16:31], v[16:19], v[48:51], 0\n" FROM16 : OUTS : INS : CLOB);
  if constexpr (MODE == 2)  // disjoint registers, queued behind two MFMAs
    asm volatile(LOAD_AB QUEUE "v_mfma_f32_32x32x16_bf16 v[32:47], v[16:19], v[48:51], 0\n" FROM32 : OUTS : INS : CLOB);
  if constexpr (MODE == 3)  // dst overlaps A, queued behind two MFMAs
    asm volatile(LOAD_AB QUEUE "v_mfma_f32_32x32x16_bf16 v[16:31], v[16:19], v[48:51], 0\n" FROM16 : OUTS : INS : CLOB);
  for (int q = 0; q < 16; ++q) out[threadIdx.x * 16 + q] = d[q];
}

template <int MODE>
int run(const unsigned* a, const unsigned* b, float* out, float* host) {
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, a, b, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(host, out, sizeof(float) * 64 * 16, hipMemcpyDeviceToHost));
  return 0;
}

int main() {
  unsigned ha[256], hb[256];
  unsigned seed = 12345;
  auto rnd_bf16x2 = [&]() {  // two random bf16 in [1, 2) packed
    seed = seed * 1664525u + 1013904223u;
    const unsigned m0 = (seed >> 8) & 0x7f, m1 = (seed >> 20) & 0x7f;
    return (0x3f80u | m0) | ((0x3f80u | m1) << 16);
  };
  for (int i = 0; i < 256; ++i) { ha[i] = rnd_bf16x2(); hb[i] = rnd_bf16x2(); }
  unsigned *a, *b; float* out;
  CHECK(hipMalloc(&a, sizeof(ha))); CHECK(hipMalloc(&b, sizeof(hb))); CHECK(hipMalloc(&out, sizeof(float) * 1024));
  CHECK(hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice));
  static float r[4][1024];
  if (run<0>(a, b, out, r[0]) || run<1>(a, b, out, r[1]) || run<2>(a, b, out, r[2]) || run<3>(a, b, out, r[3])) return 1;
  const char* names[4] = {"disjoint, idle pipe (reference)", "dst overlaps A, idle pipe", "disjoint, queued", "dst overlaps A, queued"};
  for (int m = 1; m < 4; ++m) {
    int bad[4] = {0, 0, 0, 0};
    for (int lane = 0; lane < 64; ++lane)
      for (int q = 0; q < 16; ++q)
        if (r[m][lane * 16 + q] != r[0][lane * 16 + q]) ++bad[lane / 16];
    printf("%-32s differing results in lanes 0-15 / 16-31 / 32-47 / 48-63: %d %d %d %d (of 256 each)\n", names[m], bad[0], bad[1], bad[2], bad[3]);
  }
  // latency probe
  static float ref[1024], got[1024];
  hipLaunchKernelGGL((kl<64, 0>), dim3(1), dim3(64), 0, 0, a, b, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(ref, out, sizeof(ref), hipMemcpyDeviceToHost));
#define PROBE(N, Q) do { \
    hipLaunchKernelGGL((kl<N, Q>), dim3(1), dim3(64), 0, 0, a, b, out); \
    CHECK(hipDeviceSynchronize()); \
    CHECK(hipMemcpy(got, out, sizeof(got), hipMemcpyDeviceToHost)); \
    int bad[4] = {0, 0, 0, 0}; \
    for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 16; ++q) if (got[lane * 16 + q] != ref[lane * 16 + q]) ++bad[lane / 16]; \
    printf("2 chained MFMAs%s, %2d wait states, then VALU reads D: wrong results per 16-lane group %3d %3d %3d %3d\n", Q ? " queued behind 2 MFMAs" : "", N, bad[0], bad[1], bad[2], bad[3]); \
  } while (0)
  PROBE(0, 0); PROBE(4, 0); PROBE(8, 0); PROBE(11, 0); PROBE(14, 0); PROBE(18, 0); PROBE(24, 0);
  PROBE(0, 1); PROBE(4, 1); PROBE(8, 1); PROBE(11, 1); PROBE(14, 1); PROBE(18, 1); PROBE(24, 1); PROBE(64, 1);
  return 0;
}
