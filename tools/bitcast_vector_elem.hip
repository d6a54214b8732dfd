// Reduced case of the defect noted in DESIGN.md 5.2c: __builtin_bit_cast on an ELEMENT of an ext_vector_type value.
//   hipcc --offload-arch=gfx950 -O3 tools/bitcast_vector_elem.hip -o tools/bitcast_vector_elem && tools/bitcast_vector_elem
// Expected four different integers per row; with clang 19 of ROCm 7.2 the bit_cast row repeats element 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* in, int* out) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(in);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    out[c] = __builtin_bit_cast(int, v[c]);  // reads element 0 four times
    out[4 + c] = __float_as_int(v[c]);       // correct
    const float e = v[c];
    out[8 + c] = __builtin_bit_cast(int, e);  // correct: the operand is a scalar object
  }
}

int main() {
  float h[4] = {1.f, 2.f, 3.f, 4.f}, *d;
  int r[12], *o;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[3] = {"__builtin_bit_cast(int, v[c])", "__float_as_int(v[c])         ", "bit_cast of a scalar copy    "};
  for (int k = 0; k < 3; ++k) printf("%s : %08x %08x %08x %08x\n", names[k], r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
  return 0;
}
