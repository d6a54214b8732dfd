// Measurement behind the decision NOT to use symmetric tiles for same_points: when x == y the kernel
// matrix is symmetric, so k(x_i, y_j) could be evaluated once for (i, j) and (j, i).
//
//   full       every wavefront owns 64 targets (one per lane) and sweeps ALL sources (wave-uniform, from LDS):
//              a_i += k_ij b_j in a register.  N^2 kernel values.  (The shape of lowd_kernel.)
//   symmetric  block-triangular: a workgroup owns a tile of 256 targets I and sweeps only the source tiles J >= I.
//              For J > I every kernel value feeds a_i (register, as before) AND a_j += k_ij b_i, which is a sum over
//              the 64 LANES of a wave for every source: a DPP/shuffle butterfly, then per-wave partials in LDS, one
//              atomicAdd per (workgroup, source) into a[j].  N^2 / 2 kernel values.
// Both produce a = K b for the D = 3 Gaussian and are checked against each other.
// Build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o symmetric_probe symmetric_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TILE = 256;  // targets per workgroup = sources per staged tile

__device__ __forceinline__ float kval(float4 x, float4 y) {
  const float dx = x.x - y.x, dy = x.y - y.y, dz = x.z - y.z;
  const float s = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  return __builtin_amdgcn_exp2f(s * -1.4426950408889634f);
}

// rec[j] = (y_x, y_y, y_z, b_j)
__global__ void __launch_bounds__(TILE) full_kernel(const float4* __restrict__ rec, float* __restrict__ a, int n) {
  __shared__ float4 tile[TILE];
  const int i = blockIdx.x * TILE + threadIdx.x;
  const float4 x = rec[i];
  float acc = 0.f;
  for (int j0 = 0; j0 < n; j0 += TILE) {
    __syncthreads();
    tile[threadIdx.x] = rec[j0 + threadIdx.x];
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < TILE; ++j) acc = fmaf(kval(x, tile[j]), tile[j].w, acc);
  }
  a[i] = acc;
}

__global__ void __launch_bounds__(TILE) symmetric_kernel(const float4* __restrict__ rec, float* __restrict__ a, int n) {
  __shared__ float4 tile[TILE];
  __shared__ float colsum[4][TILE];  // per wave: sum over its 64 lanes of k_ij b_i, for every source of the tile
  const int I = blockIdx.x;
  const int i = I * TILE + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 x = rec[i];
  float acc = 0.f;
  for (int J = I; J * TILE < n; ++J) {
    __syncthreads();
    tile[threadIdx.x] = rec[J * TILE + threadIdx.x];
    __syncthreads();
    if (J == I) {  // diagonal tile: plain
#pragma unroll 8
      for (int j = 0; j < TILE; ++j) acc = fmaf(kval(x, tile[j]), tile[j].w, acc);
    } else {
      for (int j = 0; j < TILE; ++j) {
        const float k = kval(x, tile[j]);
        acc = fmaf(k, tile[j].w, acc);
        float c = k * x.w;  // contribution of target i to a_j
        c += __shfl_xor(c, 32);
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 8);
        c += __shfl_xor(c, 4);
        c += __shfl_xor(c, 2);
        c += __shfl_xor(c, 1);
        if (lane == 0) colsum[wave][j] = c;
      }
      __syncthreads();
      const float s = colsum[0][threadIdx.x] + colsum[1][threadIdx.x] + colsum[2][threadIdx.x] + colsum[3][threadIdx.x];
      atomicAdd(&a[J * TILE + threadIdx.x], s);  // one per (workgroup, source)
    }
  }
  atomicAdd(&a[i], acc);
}

int main() {
  for (int n : {65536, 262144}) {
    std::vector<float4> h(n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
    for (auto& r : h) r = make_float4(rnd(), rnd(), rnd(), rnd() - 0.5f);
    float4* rec; float *a1, *a2;
    CHECK(hipMalloc(&rec, sizeof(float4) * n)); CHECK(hipMalloc(&a1, 4 * n)); CHECK(hipMalloc(&a2, 4 * n));
    CHECK(hipMemcpy(rec, h.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms_full = 0, ms_sym = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(full_kernel, dim3(n / TILE), dim3(TILE), 0, 0, rec, a1, n);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_full, e0, e1));
      CHECK(hipMemset(a2, 0, 4 * n));
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(symmetric_kernel, dim3(n / TILE), dim3(TILE), 0, 0, rec, a2, n);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms_sym, e0, e1));
    }
    std::vector<float> r1(n), r2(n);
    CHECK(hipMemcpy(r1.data(), a1, 4 * n, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r2.data(), a2, 4 * n, hipMemcpyDeviceToHost));
    double err = 0, scale = 0;
    for (int i = 0; i < n; ++i) { err = fmax(err, fabs((double)r1[i] - r2[i])); scale = fmax(scale, fabs((double)r1[i])); }
    printf("n = %7d: full %8.3f ms (%.2e pairs/s)   symmetric %8.3f ms (%.2e pairs/s of the FULL matrix)   ratio %.2f   max diff %.1e of %.1e\n",
           n, ms_full, (double)n * n / (ms_full * 1e-3), ms_sym, (double)n * n / (ms_sym * 1e-3), ms_sym / ms_full, err, scale);
    CHECK(hipFree(rec)); CHECK(hipFree(a1)); CHECK(hipFree(a2));
  }
  return 0;
}
