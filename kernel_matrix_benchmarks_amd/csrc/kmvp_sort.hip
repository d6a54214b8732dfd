// Radix sort of (Morton key, source index) pairs for the centred split-bf16 path: a thin
// wrapper around hipcub (kept in its own translation unit; hipcub is a header library).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace kmvp {

// tmp == nullptr: returns the scratch bytes needed in *tmp_bytes
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, const unsigned* keys_in, unsigned* keys_out,
                          const int* vals_in, int* vals_out, int64_t n, hipStream_t stream) {
  return hipcub::DeviceRadixSort::SortPairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0,
                                            32, stream);
}

}  // namespace kmvp
