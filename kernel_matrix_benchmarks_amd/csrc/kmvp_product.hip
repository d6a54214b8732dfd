// The product a = K b [/ K 1]: kernel layouts, launch geometry, the three pair-loop paths
// (lowd_kernel / fast_kernel / mfma_kernel) and their common epilogue
//   segment reduction -> [RCCL all-reduce over the source shards] -> normalise.
// Reference call this serves: BruteForceProductBLAS.query (bruteforce.py:130-153).
#include "kmvp_ctx.hpp"
#include <vector>
#include "kmvp_cell_pack.hpp"
#include "kmvp_cell64_pack.hpp"
#include "kmvp_cellmm_pack.hpp"
#include "kmvp_cfast_pack.hpp"
#include "kmvp_fast_pack.hpp"
#include "kmvp_fastmm_pack.hpp"
#include "kmvp_cfastmm_pack.hpp"
#include "kmvp_mfma_pack.hpp"

namespace kmvp {
namespace {

template <typename real>
hipError_t launch_lowd(int kernel, int D, int E, int sig, LowdTuning tune,
                       const LowdArgs<real>& args, dim3 grid, hipStream_t s, const char** name);
template <>
hipError_t launch_lowd<float>(int kernel, int D, int E, int sig, LowdTuning tune,
                              const LowdArgs<float>& args, dim3 grid, hipStream_t s,
                              const char** name) {
  switch (kernel) {
    case K_GAUSSIAN: return launch_lowd_gaussian_f32(D, E, sig, tune, args, grid, s, name);
    case K_ABSEXP: return launch_lowd_absexp_f32(D, E, sig, tune, args, grid, s, name);
    default: return launch_lowd_invdist_f32(D, E, sig, tune, args, grid, s, name);
  }
}
template <>
hipError_t launch_lowd<double>(int kernel, int D, int E, int sig, LowdTuning tune,
                               const LowdArgs<double>& args, dim3 grid, hipStream_t s,
                               const char** name) {
  switch (kernel) {
    case K_GAUSSIAN: return launch_lowd_gaussian_f64(D, E, sig, tune, args, grid, s, name);
    case K_ABSEXP: return launch_lowd_absexp_f64(D, E, sig, tune, args, grid, s, name);
    default: return launch_lowd_invdist_f64(D, E, sig, tune, args, grid, s, name);
  }
}

template <typename real>
hipError_t launch_generic(int kernel, int sig, const real* x, const real* y, const real* b,
                          double* part, int64_t n, int64_t n_pad, int64_t m, int D, int E, int NE,
                          int segments, int64_t seg_len, int64_t j_offset, int64_t m_total,
                          hipStream_t s, const char** name);
template <>
hipError_t launch_generic<float>(int kernel, int sig, const float* x, const float* y, const float* b,
                                 double* part, int64_t n, int64_t n_pad, int64_t m, int D, int E,
                                 int NE, int segments, int64_t seg_len, int64_t j_offset,
                                 int64_t m_total, hipStream_t s, const char** name) {
  switch (kernel) {
    case K_GAUSSIAN:
      return launch_lowd_gaussian_f32_generic(sig, x, y, b, part, n, n_pad, m, D, E, NE, segments,
                                              seg_len, j_offset, m_total, s, name);
    case K_ABSEXP:
      return launch_lowd_absexp_f32_generic(sig, x, y, b, part, n, n_pad, m, D, E, NE, segments,
                                            seg_len, j_offset, m_total, s, name);
    default:
      return launch_lowd_invdist_f32_generic(sig, x, y, b, part, n, n_pad, m, D, E, NE, segments,
                                             seg_len, j_offset, m_total, s, name);
  }
}
template <>
hipError_t launch_generic<double>(int kernel, int sig, const double* x, const double* y,
                                  const double* b, double* part, int64_t n, int64_t n_pad, int64_t m,
                                  int D, int E, int NE, int segments, int64_t seg_len,
                                  int64_t j_offset, int64_t m_total, hipStream_t s,
                                  const char** name) {
  switch (kernel) {
    case K_GAUSSIAN:
      return launch_lowd_gaussian_f64_generic(sig, x, y, b, part, n, n_pad, m, D, E, NE, segments,
                                              seg_len, j_offset, m_total, s, name);
    case K_ABSEXP:
      return launch_lowd_absexp_f64_generic(sig, x, y, b, part, n, n_pad, m, D, E, NE, segments,
                                            seg_len, j_offset, m_total, s, name);
    default:
      return launch_lowd_invdist_f64_generic(sig, x, y, b, part, n, n_pad, m, D, E, NE, segments,
                                             seg_len, j_offset, m_total, s, name);
  }
}

template <typename real>
real scale_for(int kernel) {
  switch (kernel) {
    case K_GAUSSIAN: return coord_scale<K_GAUSSIAN, real>();
    case K_ABSEXP: return coord_scale<K_ABSEXP, real>();
    default: return coord_scale<K_INVDIST, real>();
  }
}

// Sum over the segments of part[s * stride + q].  Few segments: one thread per sum, index order.  Many (small problems take
// ~100 segments to fill the chip, and one thread's chain of dependent loads then costs more than the pair loop of a 1e4-point
// product): SEG_SPLIT consecutive lanes share a sum -- lane g takes the segments s = g (mod SEG_SPLIT) in index order, a
// butterfly over the group adds the eight partial sums.  A fixed order either way: results stay bitwise reproducible, and
// reduce_segments_kernel (the exchange path) and reduce_finish_kernel (the plain one) add in the SAME order.
constexpr int SEG_SPLIT = 8;
constexpr int SEG_SPLIT_FROM = 16;  // segments from which the split form is taken
__device__ __forceinline__ double seg_sum_one(const double* __restrict__ part, int64_t stride, int64_t q, int segments) {
  double v = 0.0;
  for (int s = 0; s < segments; ++s) v += part[(int64_t)s * stride + q];
  return v;
}
__device__ __forceinline__ double seg_sum_split(const double* __restrict__ part, int64_t stride, int64_t q, int segments, int g) {
  double v = 0.0;
  for (int s = g; s < segments; s += SEG_SPLIT) v += part[(int64_t)s * stride + q];
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

// sums[e][i] = sum over segments of part[s][e][i]  (split != 0: SEG_SPLIT threads per sum, launched accordingly)
__global__ void reduce_segments_kernel(const double* __restrict__ part, double* __restrict__ sums,
                                       int64_t count /* NE*n_pad */, int segments, int split) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (split) {
    const int64_t q = t / SEG_SPLIT;
    if (q >= count) return;  // (whole groups leave together)
    const double v = seg_sum_split(part, count, q, segments, (int)(t % SEG_SPLIT));
    if (t % SEG_SPLIT == 0) sums[q] = v;
    return;
  }
  if (t >= count) return;
  sums[t] = seg_sum_one(part, count, t, segments);
}

// The same for one launch's REGION of partial sums [segment][column][region_slots] (the float32 cell kernels run two
// launches with different numbers of segments): sums[e][slot_base + sl] = sum over segments, sums being [column][n_slots].
__global__ void reduce_region_kernel(const double* __restrict__ part, double* __restrict__ sums, int64_t region_slots,
                                     int NE, int segments, int64_t n_slots, int64_t slot_base) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)NE * region_slots) return;
  const int64_t e = q / region_slots, sl = q % region_slots;
  double v = 0.0;
  for (int s = 0; s < segments; ++s) v += part[((int64_t)s * NE + e) * region_slots + sl];
  sums[e * n_slots + slot_base + sl] = v;
}

// One launch for the single-GPU epilogue: out[i*E + e] = sum over segments (index order) of
// part[s][e][i], divided by the same sum of column E when normalised.  Same additions in the
// same order as reduce_segments_kernel + finish_kernel.
__global__ void reduce_finish_kernel(const double* __restrict__ part, double* __restrict__ out, int64_t n,
                                     int64_t n_pad, int E, int NE, int segments, int normalise, int split) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = split ? t / SEG_SPLIT : t;
  const int g = split ? (int)(t % SEG_SPLIT) : 0;
  if (i >= n) return;  // (split: whole groups leave together)
  const int64_t count = (int64_t)NE * n_pad;
  double den = 1.0;
  if (normalise)
    den = split ? seg_sum_split(part, count, (int64_t)E * n_pad + i, segments, g) : seg_sum_one(part, count, (int64_t)E * n_pad + i, segments);
  for (int e = 0; e < E; ++e) {
    const double v = split ? seg_sum_split(part, count, (int64_t)e * n_pad + i, segments, g)
                           : seg_sum_one(part, count, (int64_t)e * n_pad + i, segments);
    if (g == 0) out[i * E + e] = normalise ? v / den : v;
  }
}

// Column-blocked products: block partials part[s][e][i] (e < NEk) summed over the segments into the
// full sums array: column e < Ek goes to col0 + e, the extra column (denominator) to den_col.
__global__ void reduce_block_kernel(const double* __restrict__ part, double* __restrict__ sums, int64_t n_pad,
                                    int NEk, int Ek, int segments, int col0, int den_col) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t count = (int64_t)NEk * n_pad;
  if (q >= count) return;
  const int e = (int)(q / n_pad);
  const int64_t i = q % n_pad;
  double v = 0.0;
  for (int s = 0; s < segments; ++s) v += part[(int64_t)s * count + q];
  sums[(int64_t)(e < Ek ? col0 + e : den_col) * n_pad + i] = v;
}

// out[i*E + e] = sums[e][i]  (/ sums[E][i] when normalised).  kshift (exp(<x,y>), or nullptr): the sums of target i are
// at the scale 2^-kshift[i] -- it cancels in normalised rows and is applied to plain ones here (+inf: no source at all).
__global__ void finish_kernel(const double* __restrict__ sums, double* __restrict__ out, int64_t n,
                              int64_t n_pad, int E, int normalise, const double* __restrict__ kshift) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double den = normalise ? sums[(int64_t)E * n_pad + i] : 1.0;
  int back = 0;
  bool none = false;
  if (kshift && !normalise) {
    const double k = kshift[i];
    none = !(k < 1.0e300);
    back = none ? 0 : (int)fmin(fmax(-k, -100000.0), 100000.0);
  }
  for (int e = 0; e < E; ++e) {
    const double v = sums[(int64_t)e * n_pad + i];
    out[i * E + e] = normalise ? v / den : (none ? 0.0 : (back ? ldexp(v, back) : v));
  }
}

// exp(<x,y>) (fastmm_kernel with FastmmArgs::kexp): partial sums part[s][e][i] at the scale 2^-kexp[s][i].
//   K_i = min_s kexp[s][i]  (the exponent of the row's largest term so far; +inf: segment s had no live source)
//   sums[col][i] = sum_s part[s][e][i] 2^(K_i - kexp[s][i])        (every factor <= 1: nothing overflows)
// Columns as in reduce_block_kernel (e < Ek -> col0 + e, the extra column -> den_col); kmin[i] = K_i.
__global__ void reduce_shifted_kernel(const double* __restrict__ part, const float* __restrict__ kexp,
                                      double* __restrict__ sums, double* __restrict__ kmin, int64_t n_pad, int NEk, int Ek,
                                      int segments, int col0, int den_col) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)NEk * n_pad) return;
  const int e = (int)(q / n_pad);
  const int64_t i = q % n_pad;
  float K = INFINITY;
  for (int s = 0; s < segments; ++s) K = fminf(K, kexp[(int64_t)s * n_pad + i]);
  double v = 0.0;
  for (int s = 0; s < segments; ++s) {
    const float ks = kexp[(int64_t)s * n_pad + i];
    if (ks < INFINITY) v += ldexp(part[((int64_t)s * NEk + e) * n_pad + i], (int)fmaxf(K - ks, -100000.f));
  }
  sums[(int64_t)(e < Ek ? col0 + e : den_col) * n_pad + i] = v;
  if (e == 0) kmin[i] = (double)K;
}

// after the all-reduce(min) of the exponents over the ranks: this rank's sums move to the common scale 2^-kglobal
__global__ void rescale_shifted_kernel(double* __restrict__ sums, const double* __restrict__ klocal,
                                       const double* __restrict__ kglobal, int64_t n, int64_t n_pad, int NE) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)NE * n) return;
  const int64_t e = q / n, i = q % n;
  const double kl = klocal[i], kg = kglobal[i];
  double& v = sums[e * n_pad + i];
  v = (kl < 1.0e300) ? ldexp(v, (int)fmax(kg - kl, -100000.0)) : 0.0;
}

// canon[e][i] = sums[e][i] for i < n: drops the path-specific padding before the all-reduce
__global__ void unpad_kernel(const double* __restrict__ sums, double* __restrict__ canon, int64_t n,
                             int64_t n_pad, int64_t NE) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= NE * n) return;
  canon[q] = sums[(q / n) * n_pad + q % n];
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}


// Timing marks of a synchronous product; an asynchronous one (solver iteration, possibly inside a
// stream capture) records nothing.
inline hipError_t mark(kmvp_ctx* c, int i) { return c->async_product ? hipSuccess : hipEventRecord(c->ev[i], c->stream); }

// In-place all-reduce of `count` doubles on the device over the ranks of the attached exchange: ncclAllReduce on the
// context's stream, or -- rehearsal transport (kmvp_comm_init_host) -- device -> host, the caller's callback (e.g. gloo),
// host -> device: same buffer, same place in the stream, only the wire differs.
int exchange_f64(kmvp_ctx* c, double* buf, int64_t count, int op) {
  if (c->comm) {
    ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat64, op == KMVP_OP_MIN ? ncclMin : ncclSum, c->comm,
                                      c->stream);
    if (r != ncclSuccess) return fail(c, KMVP_E_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    return KMVP_OK;
  }
  const size_t cnt = (size_t)count;
  c->host_buf.resize(std::max<size_t>(cnt, 1));
  HIP_TRY(c, hipMemcpyAsync(c->host_buf.data(), buf, cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int hr = c->host_xchg(c->host_user, c->host_buf.data(), (int64_t)cnt, op);
  if (hr != 0) return fail(c, KMVP_E_COMM, "host all-reduce callback failed with status " + std::to_string(hr));
  HIP_TRY(c, hipMemcpyAsync(buf, c->host_buf.data(), cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
  return KMVP_OK;
}

// Common tail of every path, after the path's own reduction has left sums[column][n_pad]
// (fp64) in c->sums: one RCCL all-reduce over the source shards when a communicator is
// attached, normalisation / transposition into (N,E), event bookkeeping, and the stream
// synchronisation that makes the entry point synchronous (runner.py:138-140 times it).
int finish_product(kmvp_ctx* c, int64_t count, int64_t N, int64_t n_pad, int E, int sig, const double* kshift = nullptr) {
  int rc;
  const double* sums = (const double*)c->sums.p;
  if (c->exchanges()) {  // also with world == 1: the one-GPU tests then run the real exchange path
    // The padded length n_pad belongs to the kernel a rank happened to choose (tile sizes differ
    // between the paths, and the auto policy looks at the rank's own clouds), so the exchange uses
    // the canonical unpadded layout [column][N]: every rank contributes exactly NE * N doubles.
    const int64_t NE = count / n_pad;
    if ((rc = ensure(c, c->xchg, (size_t)NE * std::max<int64_t>(N, 1) * sizeof(double)))) return rc;
    hipLaunchKernelGGL(unpad_kernel, dim3(blocks_for(NE * N)), dim3(256), 0, c->stream, sums, (double*)c->xchg.p,
                       N, n_pad, NE);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, mark(c, 3));
    if ((rc = exchange_f64(c, (double*)c->xchg.p, NE * N, KMVP_OP_SUM))) return rc;
    HIP_TRY(c, mark(c, 4));
    sums = (const double*)c->xchg.p;
    n_pad = N;
  }
  if ((rc = ensure(c, c->out, (size_t)std::max<int64_t>(N, 1) * E * sizeof(double)))) return rc;
  hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(std::max<int64_t>(N, 1))), dim3(256), 0, c->stream,
                     sums, (double*)c->out.p, N, n_pad, E, sig == SIG_NORM ? 1 : 0, kshift);
  HIP_TRY(c, hipGetLastError());
  c->out_n = N;
  c->out_e = E;
  if (c->async_product) return KMVP_OK;  // the caller keeps working on the stream
  HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipEventElapsedTime(&c->last_kernel_ms, c->ev[0], c->ev[1]));
  HIP_TRY(c, hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]));
  c->last_allreduce_ms = 0.f;
  if (c->exchanges()) HIP_TRY(c, hipEventElapsedTime(&c->last_allreduce_ms, c->ev[3], c->ev[4]));
  return KMVP_OK;
}


// Tail of the exp(<x,y>) path: c->sums [NE][n_pad] at the per-target scale 2^-K_i with K in c->kshift [n_pad].  Sharded:
// the ranks first agree on K (all-reduce MIN: the flash-attention (m, l, o) merge with m an integer exponent), move their
// sums to it, and then take the common tail (all-reduce SUM, normalise).
int finish_product_shifted(kmvp_ctx* c, int64_t N, int64_t n_pad, int E, int sig) {
  const int NE = sig == SIG_NORM ? E + 1 : E;
  int rc;
  const double* kshift = (const double*)c->kshift.p;
  if (c->exchanges()) {
    if ((rc = ensure(c, c->xchgk, (size_t)std::max<int64_t>(N, 1) * sizeof(double)))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->xchgk.p, c->kshift.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if ((rc = exchange_f64(c, (double*)c->xchgk.p, N, KMVP_OP_MIN))) return rc;
    hipLaunchKernelGGL(rescale_shifted_kernel, dim3(blocks_for((int64_t)NE * N)), dim3(256), 0, c->stream, (double*)c->sums.p,
                       (const double*)c->kshift.p, (const double*)c->xchgk.p, N, n_pad, NE);
    HIP_TRY(c, hipGetLastError());
    kshift = (const double*)c->xchgk.p;
  }
  return finish_product(c, (int64_t)NE * n_pad, N, n_pad, E, sig, kshift);
}

// Epilogue of the paths with fp64 partials [segment][column][n_pad] in c->part.  Without a
// communicator one fused launch does it; with one, the segments are summed into c->sums first and
// finish_product() exchanges them.
int reduce_and_finish(kmvp_ctx* c, int segments, int NE, int64_t N, int64_t n_pad, int E, int sig) {
  int rc;
  const int64_t count = (int64_t)NE * n_pad;
  const int split = segments >= SEG_SPLIT_FROM ? 1 : 0;
  if (c->exchanges()) {
    if ((rc = ensure(c, c->sums, (size_t)count * sizeof(double)))) return rc;
    hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks_for(count * (split ? SEG_SPLIT : 1))), dim3(256), 0, c->stream,
                       (const double*)c->part.p, (double*)c->sums.p, count, segments, split);
    HIP_TRY(c, hipGetLastError());
    return finish_product(c, count, N, n_pad, E, sig);
  }
  if ((rc = ensure(c, c->out, (size_t)std::max<int64_t>(N, 1) * E * sizeof(double)))) return rc;
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(blocks_for(std::max<int64_t>(N, 1) * (split ? SEG_SPLIT : 1))), dim3(256), 0, c->stream,
                     (const double*)c->part.p, (double*)c->out.p, N, n_pad, E, NE, segments, sig == SIG_NORM ? 1 : 0, split);
  HIP_TRY(c, hipGetLastError());
  c->out_n = N;
  c->out_e = E;
  if (c->async_product) return KMVP_OK;  // the caller keeps working on the stream
  HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipEventElapsedTime(&c->last_kernel_ms, c->ev[0], c->ev[1]));
  HIP_TRY(c, hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]));
  c->last_allreduce_ms = 0.f;
  return KMVP_OK;
}

// Number of source segments of a launch (specialised kernels).  Three pulls:
//  * L2 residency: with segments % 8 == 0 each XCD streams one segment at a time
//    (block_to_work), so a segment of <= 2 MiB of records stays in its 4 MiB L2;
//  * parallelism: tile_blocks * segments should be many rounds of the 2048 resident
//    blocks (256 CUs x 8), which only matters when there are few target tiles;
//  * the fp64 partial buffer segments * NE * n_pad * 8 bytes stays bounded.
int choose_segments(const kmvp_ctx* c, int64_t tile_blocks, int64_t m_pad, int NE, int64_t n_pad,
                    int64_t rec_bytes, int64_t min_seg, bool small = false, int64_t l2_seg_bytes = 2 << 20,
                    int64_t big_target_blocks = 16384) {
  int64_t seg;
  if (c->opt_segments > 0) {
    seg = c->opt_segments;
  } else {
    seg = 8 * std::max<int64_t>(1, (m_pad * rec_bytes + 8 * l2_seg_bytes - 1) / (8 * l2_seg_bytes));
    const int64_t target_blocks = small ? 4096 : big_target_blocks;  // small problems: about two rounds of resident blocks
    const int64_t for_parallelism = (target_blocks + tile_blocks - 1) / tile_blocks;
    if (for_parallelism > seg) seg = (for_parallelism + 7) / 8 * 8;
    const int64_t cap_len = std::max<int64_t>(1, m_pad / min_seg);              // segment >= min_seg sources
    const int64_t cap_mem = std::max<int64_t>(1, (int64_t)(4e9 / ((double)NE * n_pad * 8)));
    seg = std::min(seg, std::min(cap_len, cap_mem));
    if (seg >= 8) seg = seg / 8 * 8;
  }
  seg = std::max<int64_t>(1, std::min<int64_t>(seg, 65535));
  return (int)seg;
}

// Segments are whole numbers of `units` (stages, tiles): `seg` requested segments become ceil(units / ceil(units / seg)),
// which may be one or two fewer -- and no longer a multiple of 8.  block_to_work() streams one segment per XCD at a time only
// when the count IS a multiple of 8 (measured, cfast_kernel at 2e5 points: 23 segments 5.65 ms, 16: 5.05, 32: 4.94), so the
// nearest multiple of 8 that survives the rounding is taken (the request itself below 8).
int settle_segments(int64_t units, int seg) {
  auto settled = [&](int64_t cand) { return (units + (units + cand - 1) / cand - 1) / ((units + cand - 1) / cand); };
  if (seg >= 8 && units >= 8)
    for (int step = 0; step <= 64; step += 8)
      for (int64_t cand : {(int64_t)seg + step, (int64_t)seg - step})
        if (cand >= 8 && cand <= units && cand % 8 == 0 && settled(cand) == cand) return (int)cand;
  seg = (int)std::max<int64_t>(1, std::min<int64_t>(seg, std::max<int64_t>(units, 1)));
  return (int)settled(seg);
}

// The whole product: everything query() times.  `sig` as in kmvp_lowd.hpp.
template <typename real>
int run_product_t(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D;
  const int E = sig == SIG_DENSITY ? 1 : c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  const bool specialised = D <= LOWD_MAX_D && E <= LOWD_MAX_E;
  const real scale = (real)1;  // difference form: the caller's coordinates, untouched (kval scales s)
  const real* x_raw = (const real*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  int rc;

  int64_t n_pad;
  int segments;
  int64_t seg_len;
  HIP_TRY(c, mark(c, 0));
  if (specialised) {
    LowdTuning tune;
    tune.feed = c->opt_feed >= 0 ? c->opt_feed : DEFAULT_FEED;
    tune.targets_per_lane = c->opt_T > 0 ? c->opt_T : (tune.feed == 1 ? DEFAULT_TARGETS_PER_LANE : 2);
    if (!(D == 3 && E == 1)) {
      // only the headline shape carries the whole (T, feed) grid; elsewhere a requested T keeps its
      // instantiated feed (T = 1: LDS tiles, T = 2: scalar-cache stream), anything else is the default
      if (tune.targets_per_lane == 2 && c->opt_T > 0) tune.feed = 0;
      else if (tune.feed == 0 && c->opt_T <= 0) tune.targets_per_lane = 2;
      else {
        tune.targets_per_lane = DEFAULT_TARGETS_PER_LANE;
        tune.feed = DEFAULT_FEED;
      }
    }
    const int T = tune.targets_per_lane;
    const int EB = sig == SIG_DENSITY ? 0 : E;
    const int R = (D + EB + 3) / 4 * 4;
    const int64_t tile = 64 * (int64_t)T * WAVES_PER_BLOCK;
    n_pad = round_up(std::max<int64_t>(N, 1), tile);
    const int64_t tile_blocks = n_pad / tile;
    const int64_t batch = 8;  // two ping-pong batches of 4 records
    const int64_t m_pad = round_up(std::max<int64_t>(M, 1), batch);
    // few targets: short segments, so that the launch still covers the chip (n = 2000, fp64: 121 -> 9 us)
    const bool small = N < SMALL_PROBLEM_TARGETS;
    segments = choose_segments(c, tile_blocks, m_pad, NE, n_pad, (int64_t)R * sizeof(real), small ? 32 : 1024, small);
    seg_len = round_up((m_pad + segments - 1) / segments, batch);
    segments = (int)((m_pad + seg_len - 1) / seg_len);

    // (re)pack the kernel layouts when the points, the signal, the kernel or T changed
    const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                           c->packed_layout != LAYOUT_LOWD || c->packed_T != T;
    const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
    if (pts_stale) {
      if ((rc = ensure(c, c->xs, (size_t)D * n_pad * sizeof(real)))) return rc;
      hipLaunchKernelGGL((pack_targets_kernel<real>), dim3(blocks_for(n_pad)), dim3(256), 0,
                         c->stream, x_raw, (real*)c->xs.p, N, n_pad, D, scale);
    }
    if (sig_stale) {
      // one spare batch behind the last record keeps the prefetch in bounds
      if ((rc = ensure(c, c->rec, (size_t)(m_pad + batch) * R * sizeof(real)))) return rc;
      hipLaunchKernelGGL((pack_sources_kernel<real>), dim3(blocks_for(m_pad + batch)), dim3(256), 0,
                         c->stream, (const real*)c->y_raw.p, (const real*)c->b_raw.p,
                         (real*)c->rec.p, M, m_pad + batch, D, EB, R, scale);
    }
    HIP_TRY(c, hipGetLastError());
    c->packed_points_ver = c->points_ver;
    c->packed_signal_ver = c->signal_ver;
    c->packed_kernel = kernel;
    c->packed_sig = sig;
    c->packed_layout = LAYOUT_LOWD;
    c->packed_T = T;

    if ((rc = ensure(c, c->part, (size_t)segments * NE * n_pad * sizeof(double)))) return rc;
    LowdArgs<real> a;
    a.xs = (const real*)c->xs.p;
    a.rec = (const real*)c->rec.p;
    a.part = (double*)c->part.p;
    a.n = N;
    a.n_pad = n_pad;
    a.m_pad = m_pad;
    a.seg_len = seg_len;
    a.segments = segments;
    a.tile_blocks = (int)tile_blocks;
    a.chunk = (int)round_up(std::max(c->opt_chunk, 8), batch);
    a.j_offset = c->j_offset;
    a.m_total = c->m_total;
    const int64_t nblocks = tile_blocks * segments;
    if (nblocks > 0x7fffffff) return fail(c, KMVP_E_UNSUPPORTED, "launch grid too large");
    HIP_TRY(c, mark(c, 0));
    hipError_t le = launch_lowd<real>(kernel, D, E, sig, tune, a, dim3((unsigned)nblocks), c->stream,
                                      &c->last_kernel_name);
    if (le == hipErrorInvalidValue)
      return fail(c, KMVP_E_UNSUPPORTED, "no kernel instantiated for this (D, E, targets_per_lane, feed)");
    HIP_TRY(c, le);
  } else {
    // generic fallback: scaled copies of the points, one target per lane
    n_pad = round_up(std::max<int64_t>(N, 1), BLOCK_THREADS);
    // both clouds as rows of DP entries, zero padded: 8 ceil(D / 8) for lowd_mid_kernel, 32 ceil(D / 32) for
    // lowd_big_kernel
    const int DP = D <= LOWD_MID_MAX_D ? (D + 7) / 8 * 8 : (D + 31) / 32 * 32;  // lowd_big_kernel: chunks of 32
    if (c->gen_points_ver != c->points_ver || c->gen_kernel != kernel) {
      if ((rc = ensure(c, c->y_scaled, (size_t)M * DP * sizeof(real)))) return rc;
      hipLaunchKernelGGL((pad_rows_kernel<real>), dim3(blocks_for(M * DP)), dim3(256), 0, c->stream,
                         (const real*)c->y_raw.p, (real*)c->y_scaled.p, M, D, DP);
      if (!c->same_points) {
        if ((rc = ensure(c, c->x_scaled, (size_t)N * DP * sizeof(real)))) return rc;
        hipLaunchKernelGGL((pad_rows_kernel<real>), dim3(blocks_for(N * DP)), dim3(256), 0, c->stream, x_raw,
                           (real*)c->x_scaled.p, N, D, DP);
      }
      HIP_TRY(c, hipGetLastError());
      c->gen_points_ver = c->points_ver;
      c->gen_kernel = kernel;
    }
    const int64_t tile_blocks = n_pad / BLOCK_THREADS;
    segments = choose_segments(c, tile_blocks, M, NE, n_pad, (int64_t)D * sizeof(real), 256);
    seg_len = (M + segments - 1) / segments;
    segments = (int)((M + seg_len - 1) / seg_len);
    if ((rc = ensure(c, c->part, (size_t)segments * NE * n_pad * sizeof(double)))) return rc;
    const real* xg = (const real*)(c->same_points ? c->y_scaled.p : c->x_scaled.p);
    const real* bg = sig == SIG_DENSITY ? nullptr : (const real*)c->b_raw.p;
    if (sig != SIG_DENSITY) {  // signal rows padded to whole column blocks of the kernel (8, or 32: lowd_mid_colblock)
      const int cb = D <= LOWD_MID_MAX_D ? lowd_mid_colblock((int)sizeof(real), D, c->E) : 8;
      const int EP = (c->E + cb - 1) / cb * cb;
      if ((rc = ensure(c, c->rec, (size_t)M * EP * sizeof(real)))) return rc;
      hipLaunchKernelGGL((pad_rows_kernel<real>), dim3(blocks_for(M * EP)), dim3(256), 0, c->stream,
                         (const real*)c->b_raw.p, (real*)c->rec.p, M, c->E, EP);
      HIP_TRY(c, hipGetLastError());
      c->packed_layout = -1;  // rec no longer holds a specialised layout
      bg = (const real*)c->rec.p;
    }
    HIP_TRY(c, mark(c, 0));
    HIP_TRY(c, launch_generic<real>(kernel, sig, xg, (const real*)c->y_scaled.p, bg,
                                    (double*)c->part.p, N, n_pad, M, D, c->E, NE, segments, seg_len,
                                    c->j_offset, c->m_total, c->stream, &c->last_kernel_name));
  }
  HIP_TRY(c, mark(c, 1));

  // ---- epilogue: segments -> sums, [all-reduce over the source shards], normalise
  return reduce_and_finish(c, segments, NE, N, n_pad, E, sig);
}

// Low D, many signal columns (D <= LOWD_MAX_D, E > LOWD_MAX_E): the specialised pair loop run
// once per block of LOWD_MAX_E columns (the kernel values are recomputed per block: E = 16 costs
// four passes of 14 issue slots per pair, against a generic kernel that is 4-5x slower).  The
// denominator of normalised rows comes from the first block.
template <typename real>
int run_product_blocked(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D, E = c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  const real* x_raw = (const real*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  int rc;
  LowdTuning tune;
  tune.feed = DEFAULT_FEED;
  tune.targets_per_lane = DEFAULT_TARGETS_PER_LANE;
  const int T = tune.targets_per_lane;
  const int R = (D + LOWD_MAX_E + 3) / 4 * 4;
  const int64_t tile = 64 * (int64_t)T * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(std::max<int64_t>(N, 1), tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t batch = 8;
  const int64_t m_pad = round_up(std::max<int64_t>(M, 1), batch);
  const bool small = N < SMALL_PROBLEM_TARGETS;
  int segments = choose_segments(c, tile_blocks, m_pad, LOWD_MAX_E + 1, n_pad, (int64_t)R * sizeof(real),
                                 small ? 32 : 1024, small);
  const int64_t seg_len = round_up((m_pad + segments - 1) / segments, batch);
  segments = (int)((m_pad + seg_len - 1) / seg_len);

  HIP_TRY(c, mark(c, 0));
  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                         c->packed_layout != LAYOUT_LOWD || c->packed_T != T;
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)D * n_pad * sizeof(real)))) return rc;
    hipLaunchKernelGGL((pack_targets_kernel<real>), dim3(blocks_for(n_pad)), dim3(256), 0, c->stream, x_raw,
                       (real*)c->xs.p, N, n_pad, D, (real)1);
  }
  c->packed_points_ver = c->points_ver;
  c->packed_kernel = kernel;
  c->packed_layout = LAYOUT_LOWD;
  c->packed_T = T;
  c->packed_sig = -1;  // the source records below hold one column block: never reusable as they are
  if ((rc = ensure(c, c->rec, (size_t)(m_pad + batch) * R * sizeof(real)))) return rc;
  if ((rc = ensure(c, c->part, (size_t)segments * (LOWD_MAX_E + 1) * n_pad * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->sums, (size_t)NE * n_pad * sizeof(double)))) return rc;

  const int64_t nblocks = tile_blocks * segments;
  if (nblocks > 0x7fffffff) return fail(c, KMVP_E_UNSUPPORTED, "launch grid too large");
  HIP_TRY(c, mark(c, 0));
  for (int e0 = 0; e0 < E; e0 += LOWD_MAX_E) {
    const int Ek = std::min(LOWD_MAX_E, E - e0);
    const int sigk = (sig == SIG_NORM && e0 == 0) ? SIG_NORM : SIG_PRODUCT;
    const int NEk = sigk == SIG_NORM ? Ek + 1 : Ek;
    const int Rk = (D + Ek + 3) / 4 * 4;
    hipLaunchKernelGGL((pack_sources_kernel<real>), dim3(blocks_for(m_pad + batch)), dim3(256), 0, c->stream,
                       (const real*)c->y_raw.p, (const real*)c->b_raw.p, (real*)c->rec.p, M, m_pad + batch, D, Ek,
                       Rk, (real)1, E, e0);
    LowdArgs<real> a;
    a.xs = (const real*)c->xs.p;
    a.rec = (const real*)c->rec.p;
    a.part = (double*)c->part.p;
    a.n = N;
    a.n_pad = n_pad;
    a.m_pad = m_pad;
    a.seg_len = seg_len;
    a.segments = segments;
    a.tile_blocks = (int)tile_blocks;
    a.chunk = (int)round_up(std::max(c->opt_chunk, 8), batch);
    a.j_offset = c->j_offset;
    a.m_total = c->m_total;
    hipError_t le = launch_lowd<real>(kernel, D, Ek, sigk, tune, a, dim3((unsigned)nblocks), c->stream,
                                      &c->last_kernel_name);
    if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "no kernel instantiated for this column block");
    HIP_TRY(c, le);
    hipLaunchKernelGGL(reduce_block_kernel, dim3(blocks_for((int64_t)NEk * n_pad)), dim3(256), 0, c->stream,
                       (const double*)c->part.p, (double*)c->sums.p, n_pad, NEk, Ek, segments, e0, E);
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, mark(c, 1));
  return finish_product(c, (int64_t)NE * n_pad, N, n_pad, E, sig);
}

// split-bf16 MFMA low-D path (kmvp_fast.hpp): float32, D <= 7, E == 1, selected by the
// "fast_sqdists" option (the reference's constructor flag of the same name).
int run_product_fast(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D;
  const int E = 1;
  const int NE = sig == SIG_NORM ? 2 : 1;
  const int EB = sig == SIG_DENSITY ? 0 : 1;
  const int64_t N = c->N, M = c->M;
  const int KS = fast_ksteps(D);
  // few targets (the reference's own datasets have n <= 1e4): one tile per wave and segments of a
  // single stage spread the launch over more CUs; from ~3e4 targets on the big tiles win
  const bool small = N < SMALL_PROBLEM_TARGETS;
  const int tt_max = D > FAST_MAX_D_TWO_TILES ? 1 : (D > FAST_MAX_D_FOUR_TILES ? 2 : 4);  // what is instantiated
  const int TT = c->opt_fast_tiles > 0 ? std::min(c->opt_fast_tiles, tt_max) : (small ? 1 : std::min(FAST_DEFAULT_TT, tt_max));
  const int64_t SB = fast_stage_bytes(KS, EB);
  const float scale = scale_for<float>(kernel);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int64_t tile = (int64_t)FAST_TILE * TT * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(N, tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t m_tiles = (M + FAST_TILE - 1) / FAST_TILE;
  const int ST = fast_stage_tiles(KS);
  const int64_t m_stages = (m_tiles + ST - 1) / ST;
  int rc;

  int segments = choose_segments(c, tile_blocks, m_stages, NE, n_pad, SB, small ? 1 : 4, small);
  segments = settle_segments(m_stages, segments);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;

  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                         c->packed_layout != LAYOUT_FAST || c->packed_T != TT;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  float* centre = (float*)c->aux.p;  // written by kmvp_set_points
  if (pts_stale) {
    const int RD = fast_target_row(D);
    if ((rc = ensure(c, c->xs, (size_t)n_pad * RD * sizeof(float)))) return rc;
    hipLaunchKernelGGL(pack_fast_targets_kernel, dim3(blocks_for(n_pad)), dim3(256), 0, c->stream, x_raw,
                       centre, (float*)c->xs.p, N, n_pad, D, RD, scale);
  }
  if (sig_stale) {
    if ((rc = ensure(c, c->rec, (size_t)m_stages * SB))) return rc;
    hipLaunchKernelGGL(pack_fast_sources_kernel, dim3(blocks_for(m_stages * ST * FAST_TILE)),
                       dim3(256), 0, c->stream, (const float*)c->y_raw.p, (const float*)c->b_raw.p, centre,
                       (unsigned char*)c->rec.p, M, m_stages, D, EB, KS, scale);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = kernel;
  c->packed_sig = sig;
  c->packed_layout = LAYOUT_FAST;
  c->packed_T = TT;

  if ((rc = ensure(c, c->part, (size_t)segments * NE * n_pad * sizeof(double)))) return rc;
  FastArgs a;
  a.xr = (const float*)c->xs.p;
  a.img = (const unsigned char*)c->rec.p;
  a.part = (double*)c->part.p;
  a.n_pad = n_pad;
  a.m_tiles = m_tiles;
  a.m_stages = m_stages;
  a.seg_stages = seg_stages;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  a.chunk_stages = std::max(1, c->opt_chunk / (FAST_TILE * ST));
  a.j_offset = c->j_offset;
  a.m_total = c->m_total;
  a.same_points = (c->same_points || c->opt_same_global) ? 1 : 0;
  const dim3 grid((unsigned)(tile_blocks * segments));
  HIP_TRY(c, mark(c, 0));
  hipError_t le;
  switch (kernel) {
    case K_GAUSSIAN: le = launch_fast_gaussian(D, sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
    case K_ABSEXP: le = launch_fast_absexp(D, sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
    default: le = launch_fast_invdist(D, sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
  }
  if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "fast_tiles must be 1, 2 or 4");
  HIP_TRY(c, le);
  HIP_TRY(c, mark(c, 1));

  // ---- epilogue: segments -> sums, [all-reduce over the source shards], normalise
  return reduce_and_finish(c, segments, NE, N, n_pad, E, sig);
}

// Gaussian products with several signal columns, both matrix products on the matrix cores (kmvp_fastmm.hpp): float32,
// D <= 64, any E (blocks of up to 32 columns, the denominator of normalised rows being one more column).
// dot: k = exp(<x,y>) -- the Gaussian instantiation on operands without norms (S = -log2(e) <x,y>), always with the
// online shift, partial sums as (mantissa, exponent) pairs (FastmmArgs::kexp) and the shifted tail.
int run_product_fastmm(kmvp_ctx* c, int kernel, int sig, bool dot = false) {
  const int D = c->D, E = c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  const int KS = fmm_ksteps(D);
  const int MODE = NE > 16 ? 1 : 0;
  const bool small = N < SMALL_PROBLEM_TARGETS;
  const int tt_max = KS <= FMM_MAX_KS_TWO_TILES ? 2 : 1;  // (four tiles: no faster, 256 VGPRs)
  const int TT = c->opt_fast_tiles > 0 ? std::min(c->opt_fast_tiles, tt_max) : (small ? 1 : tt_max);
  const int64_t SB = fmm_stage_bytes(KS, MODE);
  // dot: the source rows hold -2 (y scale), so scale = log2(e) / 2 gives S = -log2(e) <x, y>
  const float scale = dot ? 0.72134752044448170f : scale_for<float>(kernel);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int64_t tile = (int64_t)FAST_TILE * TT * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(N, tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t m_tiles = (M + FAST_TILE - 1) / FAST_TILE;
  const int64_t m_stages = (m_tiles + fmm_stage_tiles(KS) - 1) / fmm_stage_tiles(KS);
  const int nb_max = std::min(NE, FMM_MAX_COLS);
  // the fixed shift 2^15 assumes every target has a source at distance ~0, i.e. targets == sources (also when the
  // sources are sharded: the all-reduced row then contains the self term); otherwise the per-target running shift
  const int online = (dot || !(c->same_points || c->opt_same_global)) ? 1 : 0;
  const int layout_kernel = dot ? K_EXPDOT : kernel;
  int rc;

  // (the kernel's time does not depend on the segment count between 8 and 48 at 1e5 points, the fp64 partial sums --
  // segments x columns x N x 8 bytes -- and their reduction do: about 4096 workgroups instead of 16384)
  int segments = choose_segments(c, tile_blocks, m_stages, nb_max, n_pad, SB, small ? 1 : 4, small, 2 << 20, 4096);
  segments = settle_segments(m_stages, segments);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;
  if (tile_blocks * segments > 0x7fffffff) return fail(c, KMVP_E_UNSUPPORTED, "launch grid too large");

  const int layout_T = TT + 16 * MODE;
  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != layout_kernel ||
                         c->packed_layout != LAYOUT_FASTMM || c->packed_T != layout_T;
  const bool one_block = NE <= FMM_MAX_COLS;
  const bool sig_stale = pts_stale || !one_block || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  float* centre = (float*)c->aux.p;  // written by kmvp_set_points
  if ((rc = ensure(c, c->cell_scale, 1024))) return rc;
  float* sigma = (float*)((char*)c->cell_scale.p + 256);
  double* unscale = (double*)((char*)c->cell_scale.p + 512);
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_pad * KS * 32))) return rc;
    if ((rc = ensure(c, c->rec, (size_t)m_stages * SB))) return rc;
    hipLaunchKernelGGL(pack_fastmm_targets_kernel, dim3(blocks_for(n_pad * KS * 2)), dim3(256), 0, c->stream, x_raw,
                       centre, (unsigned char*)c->xs.p, N, n_pad, D, KS, scale,
                       kernel == K_GAUSSIAN ? (float)FMM_SHIFT : 0.f, dot ? 1 : 0);
    hipLaunchKernelGGL(pack_fastmm_rows_kernel, dim3(blocks_for(m_stages * fmm_stage_tiles(KS) * FAST_TILE)), dim3(256), 0,
                       c->stream, (const float*)c->y_raw.p, centre, (unsigned char*)c->rec.p, M, m_stages, D, KS, MODE,
                       scale, dot ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
  }
  c->packed_points_ver = c->points_ver;
  c->packed_kernel = layout_kernel;
  c->packed_layout = LAYOUT_FASTMM;
  c->packed_T = layout_T;
  c->packed_signal_ver = c->signal_ver;
  c->packed_sig = one_block ? sig : -1;

  if ((rc = ensure(c, c->part, (size_t)segments * nb_max * n_pad * sizeof(double)))) return rc;
  if ((!one_block || dot) && (rc = ensure(c, c->sums, (size_t)NE * n_pad * sizeof(double)))) return rc;
  if (dot) {
    if ((rc = ensure(c, c->kexp, (size_t)segments * n_pad * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->kshift, (size_t)n_pad * sizeof(double)))) return rc;
  }
  FastmmArgs a;
  a.kexp = dot ? (float*)c->kexp.p : nullptr;
  a.xop = (const unsigned char*)c->xs.p;
  a.img = (const unsigned char*)c->rec.p;
  a.unscale = unscale;
  a.part = (double*)c->part.p;
  a.n_pad = n_pad;
  a.m_stages = m_stages;
  a.seg_stages = seg_stages;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  a.chunk_stages = std::max(1, 2 * c->opt_chunk / (FAST_TILE * fmm_stage_tiles(KS)));
  a.xraw = x_raw;
  a.yraw = (const float*)c->y_raw.p;
  a.n = N;
  a.m = M;
  a.D = D;
  a.scale = scale;
  {
    const float r2 = c->cloud_radius2 * scale * scale;  // scaled squared radius of the clouds
    a.tau = kernel == K_ABSEXP ? FMM_ABSEXP_KAPPA * r2 * r2 : 0.f;
  }
  const dim3 grid((unsigned)(tile_blocks * segments));
  const int64_t pieces = m_stages * fmm_stage_tiles(KS) * (MODE ? 2 : 1) * 2 * 64;
  HIP_TRY(c, mark(c, 0));
  for (int col0 = 0; col0 < NE; col0 += FMM_MAX_COLS) {
    const int nb = std::min(FMM_MAX_COLS, NE - col0);
    if (sig_stale) {
      hipLaunchKernelGGL(fastmm_colscale_kernel, dim3(FMM_MAX_COLS), dim3(256), 0, c->stream, (const float*)c->b_raw.p,
                         M, E, col0, nb, sigma, unscale);
      hipLaunchKernelGGL(pack_fastmm_signal_kernel, dim3(blocks_for(pieces)), dim3(256), 0, c->stream,
                         (const float*)c->b_raw.p, (const float*)sigma, (unsigned char*)c->rec.p, M, m_stages, E, col0,
                         nb, KS, MODE);
      HIP_TRY(c, hipGetLastError());
    }
    a.NE = nb;
    hipError_t le = kernel == K_ABSEXP ? launch_fastmm_absexp(KS, MODE, TT, online, a, grid, c->stream, &c->last_kernel_name)
                                       : launch_fastmm_gaussian(KS, MODE, TT, online, a, grid, c->stream, &c->last_kernel_name);
    if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "no fastmm_kernel for this dimension / tile count");
    HIP_TRY(c, le);
    if (dot) {
      // (every pass runs the same distances in the same order: its exponents are the previous pass's, bit for bit)
      const int Ek = std::max(0, std::min(nb, E - col0));
      hipLaunchKernelGGL(reduce_shifted_kernel, dim3(blocks_for((int64_t)nb * n_pad)), dim3(256), 0, c->stream,
                         (const double*)c->part.p, (const float*)c->kexp.p, (double*)c->sums.p, (double*)c->kshift.p, n_pad,
                         nb, Ek, segments, col0, E);
      HIP_TRY(c, hipGetLastError());
    } else if (!one_block) {
      const int Ek = std::min(nb, E - col0);  // signal columns of the block; the one beyond them is the denominator
      hipLaunchKernelGGL(reduce_block_kernel, dim3(blocks_for((int64_t)nb * n_pad)), dim3(256), 0, c->stream,
                         (const double*)c->part.p, (double*)c->sums.p, n_pad, nb, Ek, segments, col0, E);
      HIP_TRY(c, hipGetLastError());
    }
  }
  HIP_TRY(c, mark(c, 1));
  if (dot) c->note = "exp(<x,y>): fastmm_kernel with the per-target online shift, (mantissa, exponent) partial sums";
  else if (online) c->note = "fastmm_kernel with the per-target online shift (targets != sources)";
  if (dot) return finish_product_shifted(c, N, n_pad, E, sig);
  if (one_block) return reduce_and_finish(c, segments, NE, N, n_pad, E, sig);
  return finish_product(c, (int64_t)NE * n_pad, N, n_pad, E, sig);
}

// Morton order of the sources (independent of the kernel; cfast_kernel and cfastmm_kernel): keys -> radix sort -> c->perm
int morton_order(kmvp_ctx* c, int64_t m_alloc) {
  int rc;
  const float* centre = (const float*)c->aux.p;  // written by measure_clouds()
  if (c->perm_ver == c->points_ver && c->perm.cap >= (size_t)m_alloc * sizeof(int)) return KMVP_OK;
  if ((rc = ensure(c, c->perm, (size_t)m_alloc * sizeof(int)))) return rc;
  size_t tmp_bytes = 0;
  HIP_TRY(c, sort_pairs_u32(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, m_alloc, c->stream));
  const size_t keys_bytes = (size_t)m_alloc * sizeof(unsigned);
  if ((rc = ensure(c, c->sortbuf, 3 * keys_bytes + tmp_bytes + 256))) return rc;
  unsigned* keys_in = (unsigned*)c->sortbuf.p;
  unsigned* keys_out = keys_in + m_alloc;
  int* vals_in = (int*)(keys_out + m_alloc);
  void* tmp = (void*)((((uintptr_t)(vals_in + m_alloc)) + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(cfast_morton_kernel, dim3(blocks_for(m_alloc)), dim3(256), 0, c->stream,
                     (const float*)c->y_raw.p, centre, keys_in, vals_in, c->M, m_alloc, c->D);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sort_pairs_u32(tmp, &tmp_bytes, keys_in, keys_out, vals_in, (int*)c->perm.p, m_alloc, c->stream));
  c->perm_ver = c->points_ver;
  c->packed_layout = -1;  // force a re-pack
  return KMVP_OK;
}

// centred split-bf16 MFMA path (kmvp_cfast.hpp): float32, D <= 4, E == 1, every kernel.
int run_product_cfast(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D;
  const int E = 1;
  const int NE = sig == SIG_NORM ? 2 : 1;
  const int EB = sig == SIG_DENSITY ? 0 : 1;
  const int64_t N = c->N, M = c->M;
  const bool small = N < SMALL_PROBLEM_TARGETS;  // see run_product_fast
  const int TT = c->opt_fast_tiles > 0 ? std::min(c->opt_fast_tiles, 4) : (small ? 1 : CFAST_DEFAULT_TT);
  const float scale = scale_for<float>(kernel);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int64_t tile = (int64_t)32 * TT * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(N, tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t per_stage = (int64_t)CF_GROUP * CF_STAGE_GROUPS;
  const int64_t m_stages = (M + per_stage - 1) / per_stage;
  const int64_t m_alloc = m_stages * per_stage;
  if (c->m_total > 0x7fffffff) return fail(c, KMVP_E_UNSUPPORTED, "more than 2^31 sources");
  int rc;

  // No L2-residency pull here (l2_seg_bytes = infinity): cfast_kernel runs as fast on 8 segments as on 40 - 48 at 1e6 and at
  // 1e7 x 1.25e6 points (tools/c4_segments.py: 117.2 / 116.5 ms, 1442 / 1447 ms), and every segment costs N x 16 bytes of
  // partial sums written and read back -- config 4's shard: 48 -> 8 segments, 7.7 -> 1.3 GB.  Few targets still get more
  // segments for parallelism (2e5 points: 16).
  int segments = choose_segments(c, tile_blocks, m_stages, NE, n_pad, CF_STAGE_BYTES, small ? 1 : 4, small, 2 << 20, 24576);
  segments = settle_segments(m_stages, segments);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;

  if ((rc = morton_order(c, m_alloc))) return rc;
  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                         c->packed_layout != LAYOUT_CFAST || c->packed_T != TT;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_pad * 4 * sizeof(float)))) return rc;
    hipLaunchKernelGGL(pack_cfast_targets_kernel, dim3(blocks_for(n_pad)), dim3(256), 0, c->stream, x_raw,
                       (float*)c->xs.p, N, n_pad, D);
  }
  if (sig_stale) {
    if ((rc = ensure(c, c->rec, (size_t)m_stages * CF_STAGE_BYTES))) return rc;
    hipLaunchKernelGGL(pack_cfast_sources_kernel, dim3((unsigned)(m_alloc / CF_GROUP)), dim3(CF_GROUP), 0,
                       c->stream, (const float*)c->y_raw.p, (const float*)c->b_raw.p, (const int*)c->perm.p,
                       (unsigned char*)c->rec.p, M, D, EB, scale, c->j_offset);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = kernel;
  c->packed_sig = sig;
  c->packed_layout = LAYOUT_CFAST;
  c->packed_T = TT;

  if ((rc = ensure(c, c->part, (size_t)segments * NE * n_pad * sizeof(double)))) return rc;
  CfastArgs a;
  a.xraw = (const float*)c->xs.p;
  a.scale = scale;
  a.img = (const unsigned char*)c->rec.p;
  a.part = (double*)c->part.p;
  a.n_pad = n_pad;
  a.m_stages = m_stages;
  a.seg_stages = seg_stages;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  a.chunk_stages = std::max<int>(1, c->opt_chunk / (int)per_stage);
  a.j_offset = c->j_offset;
  a.m_total = c->m_total;
  const dim3 grid((unsigned)(tile_blocks * segments));
  HIP_TRY(c, mark(c, 0));
  hipError_t le;
  switch (kernel) {
    case K_GAUSSIAN: le = launch_cfast_gaussian(sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
    case K_ABSEXP: le = launch_cfast_absexp(sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
    default: le = launch_cfast_invdist(sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
  }
  if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "fast_tiles must be 1, 2 or 4");
  HIP_TRY(c, le);
  HIP_TRY(c, mark(c, 1));

  // ---- epilogue: segments -> sums, [all-reduce over the source shards], normalise
  return reduce_and_finish(c, segments, NE, N, n_pad, E, sig);
}

// ---- cell-reduced Gaussian path (kmvp_cell.hpp): float32, D <= 3, E == 1 -----------------------------

// Several signal columns on cfast_kernel's distances (kmvp_cfastmm.hpp): exp(-r), and the Gaussian outside the radius
// rule; float32, D <= 4, any E (blocks of up to 32 columns, the denominator of normalised rows being one more column).
int run_product_cfastmm(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D, E = c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  const int MODE = NE > 16 ? 1 : 0;
  const bool small = N < SMALL_PROBLEM_TARGETS;
  const int TT = c->opt_fast_tiles > 0 ? std::min(c->opt_fast_tiles, 2) : (small ? 1 : 2);
  const int64_t SB = cfm_stage_bytes(MODE);
  const float scale = scale_for<float>(kernel);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int64_t tile = (int64_t)32 * TT * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(N, tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t m_stages = (M + CF_GROUP - 1) / CF_GROUP;  // one group per stage
  const int64_t m_alloc = m_stages * CF_GROUP;
  const int nb_max = std::min(NE, FMM_MAX_COLS);
  if (c->m_total > 0x7fffffff) return fail(c, KMVP_E_UNSUPPORTED, "more than 2^31 sources");
  const int online = (kernel == K_INVDIST || !(c->same_points || c->opt_same_global)) ? 1 : 0;  // see run_product_fastmm; 1/r: always
  int rc;

  int segments = choose_segments(c, tile_blocks, m_stages, nb_max, n_pad, SB, small ? 1 : 4, small, 2 << 20, 4096);
  segments = settle_segments(m_stages, segments);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;
  if (tile_blocks * segments > 0x7fffffff) return fail(c, KMVP_E_UNSUPPORTED, "launch grid too large");

  if ((rc = morton_order(c, m_alloc))) return rc;
  const int layout_T = TT + 16 * MODE;
  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                         c->packed_layout != LAYOUT_CFASTMM || c->packed_T != layout_T;
  const bool one_block = NE <= FMM_MAX_COLS;
  const bool sig_stale = pts_stale || !one_block || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  if ((rc = ensure(c, c->cell_scale, 1024))) return rc;
  float* sigma = (float*)((char*)c->cell_scale.p + 256);
  double* unscale = (double*)((char*)c->cell_scale.p + 512);
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_pad * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->rec, (size_t)m_stages * SB))) return rc;
    hipLaunchKernelGGL(pack_cfast_targets_kernel, dim3(blocks_for(n_pad)), dim3(256), 0, c->stream, x_raw,
                       (float*)c->xs.p, N, n_pad, D);
    hipLaunchKernelGGL(pack_cfastmm_points_kernel, dim3((unsigned)m_stages), dim3(CF_GROUP), 0, c->stream,
                       (const float*)c->y_raw.p, (const int*)c->perm.p, (unsigned char*)c->rec.p, M, D, MODE, scale, c->j_offset);
    HIP_TRY(c, hipGetLastError());
  }
  c->packed_points_ver = c->points_ver;
  c->packed_kernel = kernel;
  c->packed_layout = LAYOUT_CFASTMM;
  c->packed_T = layout_T;
  c->packed_signal_ver = c->signal_ver;
  c->packed_sig = one_block ? sig : -1;

  if ((rc = ensure(c, c->part, (size_t)segments * nb_max * n_pad * sizeof(double)))) return rc;
  if (!one_block && (rc = ensure(c, c->sums, (size_t)NE * n_pad * sizeof(double)))) return rc;
  CfastmmArgs a;
  a.xraw = (const float*)c->xs.p;
  a.img = (const unsigned char*)c->rec.p;
  a.unscale = unscale;
  a.part = (double*)c->part.p;
  a.n_pad = n_pad;
  a.m_stages = m_stages;
  a.seg_stages = seg_stages;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  a.chunk_stages = std::max(1, 2 * c->opt_chunk / CF_GROUP);
  a.scale = scale;
  a.m_total = c->m_total;
  const dim3 grid((unsigned)(tile_blocks * segments));
  const int64_t pieces = m_stages * (CF_GROUP / 32) * (MODE ? 2 : 1) * 2 * 64;
  HIP_TRY(c, mark(c, 0));
  for (int col0 = 0; col0 < NE; col0 += FMM_MAX_COLS) {
    const int nb = std::min(FMM_MAX_COLS, NE - col0);
    if (sig_stale) {
      hipLaunchKernelGGL(fastmm_colscale_kernel, dim3(FMM_MAX_COLS), dim3(256), 0, c->stream, (const float*)c->b_raw.p,
                         M, E, col0, nb, sigma, unscale);
      hipLaunchKernelGGL(pack_cfastmm_signal_kernel, dim3(blocks_for(pieces)), dim3(256), 0, c->stream,
                         (const float*)c->b_raw.p, (const float*)sigma, (const int*)c->perm.p, (unsigned char*)c->rec.p,
                         M, m_stages, E, col0, nb, MODE);
      HIP_TRY(c, hipGetLastError());
    }
    a.NE = nb;
    hipError_t le = launch_cfastmm(kernel, MODE, TT, online, a, grid, c->stream, &c->last_kernel_name);
    if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "no cfastmm_kernel for this kernel / tile count");
    HIP_TRY(c, le);
    if (!one_block) {
      const int Ek = std::max(0, std::min(nb, E - col0));
      hipLaunchKernelGGL(reduce_block_kernel, dim3(blocks_for((int64_t)nb * n_pad)), dim3(256), 0, c->stream,
                         (const double*)c->part.p, (double*)c->sums.p, n_pad, nb, Ek, segments, col0, E);
      HIP_TRY(c, hipGetLastError());
    }
  }
  HIP_TRY(c, mark(c, 1));
  if (online) c->note = "cfastmm_kernel with the per-target online shift (targets != sources)";
  if (one_block) return reduce_and_finish(c, segments, NE, N, n_pad, E, sig);
  return finish_product(c, (int64_t)NE * n_pad, N, n_pad, E, sig);
}

// Cell order of one cloud: keys -> radix sort; the sorted keys come back to the host, where the
// tile lists are built (one pass over n keys, once per kmvp_set_points).
int cell_sort(kmvp_ctx* c, const void* pts, int64_t n, const CellGrid& grid, kmvp_ctx::DevBuf& perm,
              std::vector<unsigned>& keys, kmvp_ctx::DevBuf* keys_dev = nullptr) {
  int rc;
  const int D = c->D;
  size_t tmp_bytes = 0;
  HIP_TRY(c, sort_pairs_u32(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, n, c->stream));
  const size_t keys_bytes = (size_t)n * sizeof(unsigned);
  if ((rc = ensure(c, c->sortbuf, 3 * keys_bytes + tmp_bytes + 256))) return rc;
  if ((rc = ensure(c, perm, (size_t)n * sizeof(int)))) return rc;
  unsigned* keys_in = (unsigned*)c->sortbuf.p;
  unsigned* keys_out = keys_in + n;
  int* vals_in = (int*)(keys_out + n);
  void* tmp = (void*)((((uintptr_t)(vals_in + n)) + 255) & ~(uintptr_t)255);
  if (c->dtype == KMVP_F64)
    hipLaunchKernelGGL(cell64_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, (const double*)pts, n, D, grid,
                       keys_in, vals_in);
  else
    hipLaunchKernelGGL(cell_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, (const float*)pts, n, D, grid,
                       keys_in, vals_in);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, sort_pairs_u32(tmp, &tmp_bytes, keys_in, keys_out, vals_in, (int*)perm.p, n, c->stream));
  keys.resize((size_t)n);
  HIP_TRY(c, hipMemcpyAsync(keys.data(), keys_out, keys_bytes, hipMemcpyDeviceToHost, c->stream));
  if (keys_dev) {  // the float64 path keeps the sorted keys on the device too (source records are packed from them)
    if ((rc = ensure(c, *keys_dev, keys_bytes))) return rc;
    HIP_TRY(c, hipMemcpyAsync(keys_dev->p, keys_out, keys_bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return KMVP_OK;
}

// Tiles of <= 32 points that never straddle a cell; every cell gets a multiple of `mult` tiles (the
// extra ones are empty), so that the `mult` target tiles of a wavefront always share their cell.
// Device layout of the list: [start][count][key], n_tiles entries each.
int cell_tiles(kmvp_ctx* c, const std::vector<unsigned>& keys, int mult, kmvp_ctx::DevBuf& grp, int64_t* n_tiles,
               int tile = CELL_TILE) {
  const int64_t n = (int64_t)keys.size();
  std::vector<int> start, count;
  std::vector<unsigned> gkey;
  start.reserve((size_t)n / 24 + 16);
  count.reserve((size_t)n / 24 + 16);
  gkey.reserve((size_t)n / 24 + 16);
  for (int64_t p = 0; p < n;) {
    int64_t e = p + 1;
    while (e < n && keys[(size_t)e] == keys[(size_t)p]) ++e;
    int tiles = 0;
    for (int64_t t = p; t < e; t += tile, ++tiles) {
      start.push_back((int)t);
      count.push_back((int)std::min<int64_t>(tile, e - t));
      gkey.push_back(keys[(size_t)p]);
    }
    for (; tiles % mult; ++tiles) {
      start.push_back((int)p);
      count.push_back(0);
      gkey.push_back(keys[(size_t)p]);
    }
    p = e;
  }
  const size_t G = start.size();
  int rc;
  if ((rc = ensure(c, grp, 3 * G * sizeof(int)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(grp.p, start.data(), G * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync((int*)grp.p + G, count.data(), G * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync((int*)grp.p + 2 * G, gkey.data(), G * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  *n_tiles = (int64_t)G;
  return KMVP_OK;
}

// Target tiles of the float32 cell kernels, in TWO lists that share one array.  A wavefront owns TT tiles of ONE cell,
// so a cell of `tiles` tiles used to be padded to a multiple of TT with empty tiles -- at the headline shape (cells of
// 1000 +- 32 points: 32 tiles, or 33-34 for a fifth of them) 5.6 % of all tile pairs were such padding, and the kernel
// is bound by the matrix pipe.  Now a cell's tiles are split: whole groups of TT go to the MAIN list; a remainder of at
// most TT/2 tiles goes to the REST list in groups of two (a larger remainder is still padded to a whole group: the
// second launch with two tiles per wavefront is ~1.6x less efficient per tile).  Both lists are padded to whole
// workgroups (4 wavefronts) with empty tiles that repeat the preceding key.  Layout as cell_tiles().
int cell_tiles_split(kmvp_ctx* c, const std::vector<unsigned>& keys, int TT, kmvp_ctx::DevBuf& grp, int64_t* n_main,
                     int64_t* n_rest) {
  const int64_t n = (int64_t)keys.size();
  const int RT = 2;  // tiles per wavefront of the second launch
  std::vector<int> start, count, rstart, rcount;
  std::vector<unsigned> gkey, rkey;
  start.reserve((size_t)n / 24 + 64);
  count.reserve((size_t)n / 24 + 64);
  gkey.reserve((size_t)n / 24 + 64);
  for (int64_t p = 0; p < n;) {
    int64_t e = p + 1;
    while (e < n && keys[(size_t)e] == keys[(size_t)p]) ++e;
    const int64_t tiles = (e - p + CELL_TILE - 1) / CELL_TILE;
    const int64_t rem = tiles % TT;
    const bool split = TT > RT && rem > 0 && rem <= TT / 2;
    const int64_t main_tiles = split ? tiles - rem : tiles;
    int64_t t = p, k = 0;
    for (; k < main_tiles; ++k, t += CELL_TILE) {
      start.push_back((int)t);
      count.push_back((int)std::min<int64_t>(CELL_TILE, e - t));
      gkey.push_back(keys[(size_t)p]);
    }
    for (; !split && k % TT; ++k) {  // an unsplit remainder: empty tiles up to a whole group
      start.push_back((int)p);
      count.push_back(0);
      gkey.push_back(keys[(size_t)p]);
    }
    if (split) {
      int64_t r = 0;
      for (; r < rem; ++r, t += CELL_TILE) {
        rstart.push_back((int)t);
        rcount.push_back((int)std::min<int64_t>(CELL_TILE, e - t));
        rkey.push_back(keys[(size_t)p]);
      }
      for (; r % RT; ++r) {
        rstart.push_back((int)p);
        rcount.push_back(0);
        rkey.push_back(keys[(size_t)p]);
      }
    }
    p = e;
  }
  auto pad_to = [](std::vector<int>& st, std::vector<int>& ct, std::vector<unsigned>& ky, size_t mult) {
    while (!st.empty() && st.size() % mult) {
      st.push_back(st.back());
      ct.push_back(0);
      ky.push_back(ky.back());
    }
  };
  pad_to(start, count, gkey, (size_t)TT * WAVES_PER_BLOCK);
  pad_to(rstart, rcount, rkey, (size_t)RT * WAVES_PER_BLOCK);
  *n_main = (int64_t)start.size();
  *n_rest = (int64_t)rstart.size();
  start.insert(start.end(), rstart.begin(), rstart.end());
  count.insert(count.end(), rcount.begin(), rcount.end());
  gkey.insert(gkey.end(), rkey.begin(), rkey.end());
  const size_t G = start.size();
  int rc;
  if ((rc = ensure(c, grp, 3 * std::max<size_t>(G, 1) * sizeof(int)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(grp.p, start.data(), G * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync((int*)grp.p + G, count.data(), G * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync((int*)grp.p + 2 * G, gkey.data(), G * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // the host vectors go out of scope
  return KMVP_OK;
}

// The grid of the cell paths: the bounding box (aux) divided, per axis, into the smallest number of equal cells
// whose side stays within h_max (so boundary cells are as full as the others).  false: more than 1024 cells on an axis.
bool cell_make_grid(const float* aux, int D, float h_max, CellGrid& grid) {
  for (int a = 0; a < 3; ++a) {
    grid.lo[a] = 0.f;
    grid.g[a] = 1;
    grid.h[a] = h_max;
    grid.inv_h[a] = 1.f / h_max;
  }
  for (int a = 0; a < D; ++a) {
    const float half = aux[FAST_AUX_HALF + a];
    grid.lo[a] = aux[a] - half;
    const double cells = std::max(1.0, std::ceil(2.0 * half / h_max));
    if (!(cells <= CELL_MAX_GRID)) return false;
    grid.g[a] = (int)cells;
    if (half > 0.f) {
      grid.h[a] = (float)(2.0 * half / cells);
      grid.inv_h[a] = 1.f / grid.h[a];
    }
  }
  return true;
}
void cell_store_grid(kmvp_ctx* c, const CellGrid& grid) {
  for (int a = 0; a < 3; ++a) {
    c->cell_lo[a] = grid.lo[a];
    c->cell_g[a] = grid.g[a];
    c->cell_hh[a] = grid.h[a];
  }
}
void cell_load_grid(const kmvp_ctx* c, CellGrid& grid) {
  for (int a = 0; a < 3; ++a) {
    grid.lo[a] = c->cell_lo[a];
    grid.g[a] = c->cell_g[a];
    grid.h[a] = c->cell_hh[a];
    grid.inv_h[a] = 1.f / c->cell_hh[a];
  }
}

// Tiles of the two target lists cell_tiles_split() builds for groups of TT (before the padding to whole workgroups)
void cell_split_count(const std::vector<unsigned>& keys, int TT, int64_t* n_main, int64_t* n_rest) {
  const int64_t n = (int64_t)keys.size();
  const int RT = 2;
  *n_main = *n_rest = 0;
  for (int64_t p = 0; p < n;) {
    int64_t e = p + 1;
    while (e < n && keys[(size_t)e] == keys[(size_t)p]) ++e;
    const int64_t tiles = (e - p + CELL_TILE - 1) / CELL_TILE;
    const int64_t rem = tiles % TT;
    if (TT > RT && rem > 0 && rem <= TT / 2) {
      *n_main += tiles - rem;
      *n_rest += round_up(rem, RT);
    } else {
      *n_main += round_up(tiles, TT);
    }
    p = e;
  }
}

// picoseconds per 32 x 32 tile of pairs of the float32 cell kernels, whole chip, by the tiles a wavefront owns
// (cellmm_kernel at the headline shape: TT = 8 / 4 / 2: 26.9 / 30.7 / 44.2 ms for 1.03e9 tiles)
inline double cell_ps_per_tile(int TT) { return TT >= 8 ? CMM_PS_PER_TILE_MAIN : (TT >= 4 ? CMM_PS_PER_TILE_TT4 : CMM_PS_PER_TILE_REST); }

// Grid and cell order of both clouds for the current points (cached per points version).
// TT > 0: target tiles per wavefront as requested; TT == 0: the largest of 8, 4, 2 whose padding of the
// target cells stays within CELL_AUTO_MAX_PAD (2 if none does).
// Leaves c->cell_state = 1 when the path can run, -1 when it cannot (D > 3, non-finite box, more
// than 1024 cells along an axis).
int cell_prepare(kmvp_ctx* c, int TT) {
  if (c->cell_ver == c->points_ver && c->cell_state != 0 && (c->cell_state < 0 || c->cell_tt_req == TT)) return KMVP_OK;
  c->cell_tt_req = TT;
  c->cell_ver = c->points_ver;
  c->cell_state = -1;
  const int D = c->D;
  if (D > CELL_MAX_D || !(c->cloud_radius2 < INFINITY) || c->N > 0x3fffffff || c->M > 0x3fffffff) return KMVP_OK;
  float aux[FAST_AUX_FLOATS];
  HIP_TRY(c, hipMemcpyAsync(aux, c->aux.p, sizeof(aux), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  CellGrid grid;
  if (!cell_make_grid(aux, D, (float)(std::sqrt(2.f * CELL_T_MAX / (float)D)), grid)) return KMVP_OK;  // |2 d.e| <= D h^2 / 2 <= the bound
  int rc;
  std::vector<unsigned> keys;
  if ((rc = cell_sort(c, c->y_raw.p, c->M, grid, c->cell_sperm, keys))) return rc;
  if ((rc = cell_tiles(c, keys, 1, c->cell_sgrp, &c->cell_m_tiles))) return rc;
  if (!c->same_points && (rc = cell_sort(c, c->x_raw.p, c->N, grid, c->cell_tperm, keys))) return rc;
  if (TT == 0) {
    // auto: the group size whose two lists cost least (sparse cells of 3 - 5 tiles: groups of four and a leftover list
    // beat groups of two; dense cells: groups of eight)
    double best = INFINITY;
    for (int t : {8, 4, 2}) {
      int64_t nm, nr;
      cell_split_count(keys, t, &nm, &nr);
      const double cost = cell_ps_per_tile(t) * (double)nm + cell_ps_per_tile(2) * (double)nr;
      if (cost < best) {
        best = cost;
        TT = t;
      }
    }
  }
  if ((rc = cell_tiles_split(c, keys, TT, c->cell_tgrp, &c->cell_n_main, &c->cell_n_rest))) return rc;
  c->cell_n_tiles = c->cell_n_main + c->cell_n_rest;  // (both lists already padded to whole workgroups)
  c->cell_tt = TT;
  cell_store_grid(c, grid);
  c->cell_state = 1;
  c->packed_layout = -1;  // xs / rec are re-packed by whoever runs next
  return KMVP_OK;
}

// share of slots over points after padding the last tile of every cell (both clouds)
double cell_padding(const kmvp_ctx* c) {  // (empty tiles included)
  const double t = (double)c->cell_n_tiles * CELL_TILE / (double)std::max<int64_t>(c->N, 1);
  const double s = (double)c->cell_m_tiles * CELL_TILE / (double)std::max<int64_t>(c->M, 1);
  return std::max(t, s);
}

int run_product_cell(kmvp_ctx* c, int sig) {
  const int D = c->D;
  const int E = 1;
  const int NE = sig == SIG_NORM ? 2 : 1;
  const int64_t N = c->N;
  const bool small = N < SMALL_PROBLEM_TARGETS;
  const int TT = c->cell_tt;  // the target tile lists were built for it (cell_prepare)
  const int64_t n_tiles = c->cell_n_tiles;  // whole groups of TT, then leftover tiles in groups of 2 (cell_tiles_split)
  const int64_t n_slots = n_tiles * CELL_TILE;
  const int64_t tile_blocks = c->cell_n_main / (TT * WAVES_PER_BLOCK);
  const int64_t rest_blocks = c->cell_n_rest / (2 * WAVES_PER_BLOCK);
  const int64_t m_stages = (c->cell_m_tiles + CELL_STAGE_TILES - 1) / CELL_STAGE_TILES;
  int rc;
  CellGrid grid;
  cell_load_grid(c, grid);

  int segments = choose_segments(c, std::max<int64_t>(1, tile_blocks), m_stages, NE, n_slots, CELL_STAGE_BYTES,
                                 small ? 1 : 4, small);
  segments = settle_segments(m_stages, segments);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;
  // the launch over the leftover tiles: its own, finer split of the sources and its own region of partial sums
  int rest_segments = rest_blocks > 0 ? choose_segments(c, rest_blocks, m_stages, NE, n_slots, CELL_STAGE_BYTES, 1, small, 2 << 20, 1536) : 0;
  const int64_t rest_seg_stages = rest_blocks > 0 ? (m_stages + rest_segments - 1) / rest_segments : 1;
  if (rest_blocks > 0) rest_segments = (int)((m_stages + rest_seg_stages - 1) / rest_seg_stages);
  const int64_t main_slots = c->cell_n_main * CELL_TILE, rest_slots = c->cell_n_rest * CELL_TILE;

  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != K_GAUSSIAN ||
                         c->packed_layout != LAYOUT_CELL || c->packed_T != TT;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int* tperm = (const int*)(c->same_points ? c->cell_sperm.p : c->cell_tperm.p);
  const int* tgrp = (const int*)c->cell_tgrp.p;
  const int* sgrp = (const int*)c->cell_sgrp.p;
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_slots * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->cell_tmeta, (size_t)n_tiles * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->cell_slot, (size_t)N * sizeof(int)))) return rc;
    hipLaunchKernelGGL(pack_cell_targets_kernel, dim3((unsigned)n_tiles), dim3(CELL_TILE), 0, c->stream, x_raw, tperm,
                       tgrp, tgrp + c->cell_n_tiles, (const unsigned*)(tgrp + 2 * c->cell_n_tiles), c->cell_n_tiles, D,
                       grid, (float*)c->xs.p, (float*)c->cell_tmeta.p, (int*)c->cell_slot.p);
  }
  if (sig_stale) {
    if ((rc = ensure(c, c->rec, (size_t)m_stages * CELL_STAGE_BYTES))) return rc;
    hipLaunchKernelGGL(pack_cell_sources_kernel, dim3((unsigned)(m_stages * CELL_STAGE_TILES)), dim3(CELL_TILE), 0,
                       c->stream, (const float*)c->y_raw.p,
                       sig == SIG_DENSITY ? (const float*)nullptr : (const float*)c->b_raw.p,
                       (const int*)c->cell_sperm.p, sgrp, sgrp + c->cell_m_tiles,
                       (const unsigned*)(sgrp + 2 * c->cell_m_tiles), c->cell_m_tiles, D, grid,
                       (unsigned char*)c->rec.p);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = K_GAUSSIAN;
  c->packed_sig = sig;
  c->packed_layout = LAYOUT_CELL;
  c->packed_T = TT;

  if ((rc = ensure(c, c->part, ((size_t)segments * main_slots + (size_t)rest_segments * rest_slots) * NE * sizeof(double)))) return rc;
  double* part_main = (double*)c->part.p;
  double* part_rest = part_main + (size_t)segments * NE * main_slots;
  CellArgs a;
  a.xd = (const float*)c->xs.p;
  a.tmeta = (const float*)c->cell_tmeta.p;
  a.img = (const unsigned char*)c->rec.p;
  a.m_stages = m_stages;
  a.chunk_stages = std::max(1, c->opt_chunk / (CELL_TILE * CELL_STAGE_TILES));
  HIP_TRY(c, mark(c, 0));
  hipError_t le = hipSuccess;
  if (tile_blocks > 0) {
    a.part = part_main;
    a.n_slots = main_slots;
    a.tile_base = 0;
    a.seg_stages = seg_stages;
    a.segments = segments;
    a.tile_blocks = (int)tile_blocks;
    le = launch_cell_gaussian(sig, TT, a, dim3((unsigned)(tile_blocks * segments)), c->stream, &c->last_kernel_name);
  }
  if (le == hipSuccess && rest_blocks > 0) {  // the cells' leftover tiles, two per wavefront
    a.part = part_rest;
    a.n_slots = rest_slots;
    a.tile_base = c->cell_n_main;
    a.seg_stages = rest_seg_stages;
    a.segments = rest_segments;
    a.tile_blocks = (int)rest_blocks;
    le = launch_cell_gaussian(sig, 2, a, dim3((unsigned)(rest_blocks * rest_segments)), c->stream, &c->last_kernel_name);
  }
  if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "fast_tiles must be 1, 2, 4 or 8");
  HIP_TRY(c, le);
  HIP_TRY(c, mark(c, 1));

  // ---- epilogue: segments -> sums in the caller's order, [all-reduce over the source shards], normalise
  if ((rc = ensure(c, c->sums, (size_t)NE * N * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->cell_sums, (size_t)NE * n_slots * sizeof(double)))) return rc;
  if (tile_blocks > 0)
    hipLaunchKernelGGL(reduce_region_kernel, dim3(blocks_for((int64_t)NE * main_slots)), dim3(256), 0, c->stream,
                       (const double*)part_main, (double*)c->cell_sums.p, main_slots, NE, segments, n_slots, (int64_t)0);
  if (rest_blocks > 0)
    hipLaunchKernelGGL(reduce_region_kernel, dim3(blocks_for((int64_t)NE * rest_slots)), dim3(256), 0, c->stream,
                       (const double*)part_rest, (double*)c->cell_sums.p, rest_slots, NE, rest_segments, n_slots, main_slots);
  hipLaunchKernelGGL(gather_cells_kernel, dim3(blocks_for((int64_t)NE * N)), dim3(256), 0, c->stream,
                     (const double*)c->cell_sums.p, (const int*)c->cell_slot.p, (double*)c->sums.p, N, n_slots, NE);
  HIP_TRY(c, hipGetLastError());
  return finish_product(c, (int64_t)NE * N, N, N, E, sig);
}

// ---- cell form with the sum over the sources in the MFMA accumulator (kmvp_cellmm.hpp): float32, D <= 3,
// E == 1, plain product / density.  Shares grid, cell order, tile lists and the target layout with cell_kernel.

// log2 of the largest W_j(T) = exp(e_j.(2 D - e_j)) on this grid: |e_a| <= h_a / 2, |D_a| <= g_a h_a
int cellmm_wlog2(const kmvp_ctx* c) {
  double arg = 0.0;
  for (int a = 0; a < c->D; ++a) arg += (double)c->cell_hh[a] * ((double)c->cell_g[a] * c->cell_hh[a]);
  return (int)std::ceil(arg * 1.4426950408889634);
}

// One launch per signal column (and one with b = 1 for the denominator of normalised rows): the weights W_j b_j
// live in the MFMA's source operand, so a column is an operand, not an extra fma.  E columns cost E launches of
// ~27 us per 1e9 pairs each -- against 3.5 E VALU slots per pair for the column-blocked difference form.
int run_product_cellmm(kmvp_ctx* c, int sig) {
  const int D = c->D;
  const int E = sig == SIG_DENSITY ? 1 : c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N;
  const bool small = N < SMALL_PROBLEM_TARGETS;
  const int TT = c->cell_tt;  // the target tile lists were built for it (cell_prepare)
  const int64_t n_tiles = c->cell_n_tiles;  // whole groups of TT, then leftover tiles in groups of 2 (cell_tiles_split)
  const int64_t n_slots = n_tiles * CELL_TILE;
  const int64_t tile_blocks = c->cell_n_main / (TT * WAVES_PER_BLOCK);
  const int64_t rest_blocks = c->cell_n_rest / (2 * WAVES_PER_BLOCK);
  const int64_t m_stages = (c->cell_m_tiles + CMM_STAGE_TILES - 1) / CMM_STAGE_TILES;
  int rc;
  CellGrid grid;
  cell_load_grid(c, grid);

  // segments of <= 3 MiB of source image (one per XCD at a time in its 4 MiB L2) and ~14 rounds of resident
  // workgroups: 8 at the headline shape.  tools/cellmm_segments.py: 8 ... 32 segments run alike (28.3-28.5 ms),
  // 4 is slower (29.0); every segment costs n_slots x 8 bytes of partial sums written and read back and one more
  // pass over the targets, so the fewest that keep the chip full are taken (HBM-side traffic 0.44 -> 0.25 GB).
  int segments = choose_segments(c, std::max<int64_t>(1, tile_blocks), m_stages, 1, n_slots, CMM_STAGE_BYTES,
                                 small ? 1 : 2, small, 3 << 20, 7168);
  segments = settle_segments(m_stages, segments);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;
  // the launch over the leftover tiles (two per wavefront, few workgroups) gets its own, finer split of the sources --
  // it is latency-bound per workgroup, so it needs many of them -- and its own region of partial sums
  int rest_segments = rest_blocks > 0 ? choose_segments(c, rest_blocks, m_stages, 1, n_slots, CMM_STAGE_BYTES, 1, small, 3 << 20, 1536) : 0;
  const int64_t rest_seg_stages = rest_blocks > 0 ? (m_stages + rest_segments - 1) / rest_segments : 1;
  if (rest_blocks > 0) rest_segments = (int)((m_stages + rest_seg_stages - 1) / rest_seg_stages);
  const int64_t main_slots = c->cell_n_main * CELL_TILE, rest_slots = c->cell_n_rest * CELL_TILE;

  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != K_GAUSSIAN ||
                         c->packed_layout != LAYOUT_CELLMM || c->packed_T != TT;
  // the image holds ONE column's signal: it can be reused only by single-column products
  const bool sig_stale = pts_stale || NE > 1 || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int* tperm = (const int*)(c->same_points ? c->cell_sperm.p : c->cell_tperm.p);
  const int* tgrp = (const int*)c->cell_tgrp.p;
  const int* sgrp = (const int*)c->cell_sgrp.p;
  if ((rc = ensure(c, c->cell_scale, 64))) return rc;
  float* scale = (float*)c->cell_scale.p;
  unsigned* bmax = (unsigned*)c->cell_scale.p + 4;
  HIP_TRY(c, mark(c, 0));
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_slots * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->cell_tmeta, (size_t)n_tiles * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->cell_slot, (size_t)N * sizeof(int)))) return rc;
    if ((rc = ensure(c, c->rec, (size_t)m_stages * CMM_STAGE_BYTES))) return rc;
    hipLaunchKernelGGL(pack_cell_targets_kernel, dim3((unsigned)n_tiles), dim3(CELL_TILE), 0, c->stream, x_raw, tperm,
                       tgrp, tgrp + c->cell_n_tiles, (const unsigned*)(tgrp + 2 * c->cell_n_tiles), c->cell_n_tiles, D,
                       grid, (float*)c->xs.p, (float*)c->cell_tmeta.p, (int*)c->cell_slot.p);
    hipLaunchKernelGGL(pack_cellmm_points_kernel, dim3((unsigned)(m_stages * CMM_STAGE_TILES)), dim3(CELL_TILE), 0,
                       c->stream, (const float*)c->y_raw.p, (const int*)c->cell_sperm.p, sgrp, sgrp + c->cell_m_tiles,
                       (const unsigned*)(sgrp + 2 * c->cell_m_tiles), c->cell_m_tiles, D, grid, (unsigned char*)c->rec.p);
    HIP_TRY(c, hipGetLastError());
  }
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = K_GAUSSIAN;
  c->packed_sig = NE > 1 ? -1 : sig;
  c->packed_layout = LAYOUT_CELLMM;
  c->packed_T = TT;

  if ((rc = ensure(c, c->part, ((size_t)segments * main_slots + (size_t)rest_segments * rest_slots) * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->sums, (size_t)NE * N * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->cell_sums, (size_t)NE * n_slots * sizeof(double)))) return rc;
  double* part_main = (double*)c->part.p;
  double* part_rest = part_main + (size_t)segments * main_slots;
  // MFMA shape: 16x16x32 (cellmm16_kernel) sustains more under the power limit, but issues twice the MFMAs per flop:
  // 1e6 points 27.7 -> 26.6 ms, 5e5 equal, 2e5 (smaller cells' tile groups) 1.73 -> 1.96 ms (tools/cellmm_shapes.py).  What
  // counts is how full the TARGET cells are, not the pair count: one of eight ranks' share of config 2 (1e6 targets x 125 000
  // sources) gains 3 % as well (3.72 -> 3.60 ms, tools/c2_shard.py --shape)
  const int shape = c->opt_cellmm_shape >= 0 ? c->opt_cellmm_shape : ((TT >= 8 && c->N >= 500000 && c->M >= 100000) ? 1 : 0);
  CellmmArgs a;
  a.xd = (const float*)c->xs.p;
  a.tmeta = (const float*)c->cell_tmeta.p;
  a.img = (const unsigned char*)c->rec.p;
  a.scale = scale;
  a.m_stages = m_stages;
  if (NE == 1 && !sig_stale) HIP_TRY(c, mark(c, 0));  // resident signal: the timed region is the pair loop alone
  for (int col = 0; col < NE; ++col) {
    if (sig_stale) {  // the signal part of the image: max |b| -> sigma_b -> b sigma_b in tile order
      const float* b = (sig == SIG_DENSITY || col == E) ? (const float*)nullptr : (const float*)c->b_raw.p;
      HIP_TRY(c, hipMemsetAsync(bmax, 0, sizeof(unsigned), c->stream));
      if (b)
        hipLaunchKernelGGL(cellmm_absmax_kernel, dim3((unsigned)std::min<int64_t>(1024, (c->M + 255) / 256)), dim3(256), 0,
                           c->stream, b, c->M, E, col, bmax);
      hipLaunchKernelGGL(cellmm_scale_kernel, dim3(1), dim3(1), 0, c->stream, (const unsigned*)bmax, cellmm_wlog2(c),
                         b ? 0 : 1, scale);
      hipLaunchKernelGGL(pack_cellmm_signal_kernel, dim3((unsigned)(m_stages * CMM_STAGE_TILES)), dim3(CELL_TILE), 0,
                         c->stream, b, E, col, (const int*)c->cell_sperm.p, sgrp, sgrp + c->cell_m_tiles, c->cell_m_tiles,
                         (const float*)scale, (unsigned char*)c->rec.p);
      HIP_TRY(c, hipGetLastError());
      if (NE == 1) HIP_TRY(c, mark(c, 0));
    }
    hipError_t le = hipSuccess;
    if (tile_blocks > 0) {
      a.tile_base = 0;
      a.tile_blocks = (int)tile_blocks;
      a.part = part_main;
      a.n_slots = main_slots;
      a.seg_stages = seg_stages;
      a.segments = segments;
      le = launch_cellmm_gaussian(TT, shape, a, dim3((unsigned)(tile_blocks * segments)), c->stream, &c->last_kernel_name);
    }
    if (le == hipSuccess && rest_blocks > 0) {  // the cells' leftover tiles, two per wavefront
      a.tile_base = c->cell_n_main;
      a.tile_blocks = (int)rest_blocks;
      a.part = part_rest;
      a.n_slots = rest_slots;
      a.seg_stages = rest_seg_stages;
      a.segments = rest_segments;
      le = launch_cellmm_gaussian(2, shape, a, dim3((unsigned)(rest_blocks * rest_segments)), c->stream, &c->last_kernel_name);
    }
    if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "fast_tiles must be 1, 2, 4 or 8");
    HIP_TRY(c, le);
    if (NE == 1) HIP_TRY(c, mark(c, 1));
    // segments -> this column's sums in cell order (each launch's region with its own number of segments)
    double* col_sums = (double*)c->cell_sums.p + (size_t)col * n_slots;
    if (tile_blocks > 0)
      hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks_for(main_slots)), dim3(256), 0, c->stream,
                         (const double*)part_main, col_sums, main_slots, segments, 0);
    if (rest_blocks > 0)
      hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks_for(rest_slots)), dim3(256), 0, c->stream,
                         (const double*)part_rest, col_sums + main_slots, rest_slots, rest_segments, 0);
    HIP_TRY(c, hipGetLastError());
  }
  if (NE > 1) HIP_TRY(c, mark(c, 1));  // several columns: the "kernel" time covers every column's pack + pair loop

  // ---- epilogue: back to the caller's order, [all-reduce over the source shards], normalise
  hipLaunchKernelGGL(gather_cells_kernel, dim3(blocks_for((int64_t)NE * N)), dim3(256), 0, c->stream,
                     (const double*)c->cell_sums.p, (const int*)c->cell_slot.p, (double*)c->sums.p, N, n_slots, NE);
  HIP_TRY(c, hipGetLastError());
  return finish_product(c, (int64_t)NE * N, N, N, E, sig);
}

// ---- float64 cell path (kmvp_cell64.hpp): Gaussian, D <= 3, E == 1, plain product ----------------------

// Grid, cell order, target tiles of 64 and the list of source cells for the current points.
int cell64_prepare(kmvp_ctx* c) {
  if (c->cell_ver == c->points_ver && c->cell_state != 0) return KMVP_OK;
  c->cell_ver = c->points_ver;
  c->cell_state = -1;
  const int D = c->D;
  if (D > CELL_MAX_D || !(c->cloud_radius2 < INFINITY) || c->N > 0x3fffffff || c->M > 0x3fffffff) return KMVP_OK;
  float aux[FAST_AUX_FLOATS];
  HIP_TRY(c, hipMemcpyAsync(aux, c->aux.p, sizeof(aux), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  CellGrid grid;
  if (!cell_make_grid(aux, D, (float)((float)std::sqrt(2.0 * CELL64_T_MAX / (double)D)), grid)) return KMVP_OK;  // |2 d.e| <= D h^2 / 2 <= the bound
  int rc;
  std::vector<unsigned> keys;
  if ((rc = cell_sort(c, c->y_raw.p, c->M, grid, c->cell_sperm, keys, &c->cell_skey))) return rc;
  // source cells: [first record, count] and centres
  std::vector<int> runs;
  std::vector<double> centres;
  for (int64_t p = 0; p < c->M;) {
    int64_t e = p + 1;
    while (e < c->M && keys[(size_t)e] == keys[(size_t)p]) ++e;
    runs.push_back((int)p);
    runs.push_back((int)(e - p));
    for (int a = 0; a < 3; ++a) centres.push_back(a < D ? cell64_centre(keys[(size_t)p], a, grid) : 0.0);
    centres.push_back(0.0);
    p = e;
  }
  c->cell_m_tiles = (int64_t)runs.size() / 2;  // number of source cells
  if ((rc = ensure(c, c->cell_sgrp, runs.size() * sizeof(int)))) return rc;
  if ((rc = ensure(c, c->cell_scentre, centres.size() * sizeof(double)))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->cell_sgrp.p, runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->cell_scentre.p, centres.data(), centres.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!c->same_points && (rc = cell_sort(c, c->x_raw.p, c->N, grid, c->cell_tperm, keys))) return rc;
  if ((rc = cell_tiles(c, keys, 1, c->cell_tgrp, &c->cell_n_tiles, CELL64_TILE))) return rc;
  c->cell_tt = 1;
  cell_store_grid(c, grid);
  c->cell_state = 1;
  c->packed_layout = -1;
  return KMVP_OK;
}

int run_product_cell64(kmvp_ctx* c, int sig) {
  const int D = c->D;
  const int NE = sig == SIG_NORM ? 2 : 1;
  const int64_t N = c->N, M = c->M;
  const int64_t n_tiles = round_up(c->cell_n_tiles, WAVES_PER_BLOCK);
  const int64_t n_slots = n_tiles * CELL64_TILE;
  const int64_t tile_blocks = n_tiles / WAVES_PER_BLOCK;
  const int n_scells = (int)c->cell_m_tiles;
  int rc;
  CellGrid grid;
  cell_load_grid(c, grid);
  // segments of whole source cells: enough workgroups for a few rounds of the chip
  int64_t seg = c->opt_segments > 0 ? c->opt_segments : (8192 + tile_blocks - 1) / tile_blocks;
  seg = std::max<int64_t>(1, std::min<int64_t>(seg, n_scells));
  const int seg_cells = (int)((n_scells + seg - 1) / seg);
  const int segments = (n_scells + seg_cells - 1) / seg_cells;

  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != K_GAUSSIAN ||
                         c->packed_layout != LAYOUT_CELL64;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  const double* x_raw = (const double*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int* tperm = (const int*)(c->same_points ? c->cell_sperm.p : c->cell_tperm.p);
  const int* tgrp = (const int*)c->cell_tgrp.p;
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_slots * 4 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->cell_tmeta, (size_t)n_tiles * 4 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->cell_slot, (size_t)N * sizeof(int)))) return rc;
    hipLaunchKernelGGL(pack_cell64_targets_kernel, dim3((unsigned)n_tiles), dim3(CELL64_TILE), 0, c->stream, x_raw, tperm,
                       tgrp, tgrp + c->cell_n_tiles, (const unsigned*)(tgrp + 2 * c->cell_n_tiles), c->cell_n_tiles, D,
                       grid, (double*)c->xs.p, (double*)c->cell_tmeta.p, (int*)c->cell_slot.p);
  }
  if (sig_stale) {
    const int64_t m_alloc = M + 64;  // zero records behind the last cell: the W pass reads whole groups of 64
    if ((rc = ensure(c, c->rec, (size_t)m_alloc * 4 * sizeof(double)))) return rc;
    hipLaunchKernelGGL(pack_cell64_sources_kernel, dim3(blocks_for(m_alloc)), dim3(256), 0, c->stream,
                       (const double*)c->y_raw.p, sig == SIG_DENSITY ? (const double*)nullptr : (const double*)c->b_raw.p,
                       (const int*)c->cell_sperm.p, (const unsigned*)c->cell_skey.p, M, m_alloc, D, grid, (double*)c->rec.p);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = K_GAUSSIAN;
  c->packed_sig = sig;
  c->packed_layout = LAYOUT_CELL64;
  c->packed_T = 1;

  if ((rc = ensure(c, c->part, (size_t)segments * NE * n_slots * sizeof(double)))) return rc;
  Cell64Args a;
  a.xd = (const double*)c->xs.p;
  a.tmeta = (const double*)c->cell_tmeta.p;
  a.srec = (const double*)c->rec.p;
  a.scell = (const int*)c->cell_sgrp.p;
  a.scentre = (const double*)c->cell_scentre.p;
  a.part = (double*)c->part.p;
  a.n_slots = n_slots;
  a.n_scells = n_scells;
  a.seg_cells = seg_cells;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  const dim3 grid_dim((unsigned)(tile_blocks * segments));
  HIP_TRY(c, mark(c, 0));
  HIP_TRY(c, launch_cell64_gaussian(sig, a, grid_dim, c->stream, &c->last_kernel_name));
  HIP_TRY(c, mark(c, 1));

  if ((rc = ensure(c, c->sums, (size_t)NE * N * sizeof(double)))) return rc;
  if ((rc = ensure(c, c->cell_sums, (size_t)NE * n_slots * sizeof(double)))) return rc;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks_for((int64_t)NE * n_slots)), dim3(256), 0, c->stream,
                     (const double*)c->part.p, (double*)c->cell_sums.p, (int64_t)NE * n_slots, segments, 0);
  hipLaunchKernelGGL(gather_cells_kernel, dim3(blocks_for((int64_t)NE * N)), dim3(256), 0, c->stream,
                     (const double*)c->cell_sums.p, (const int*)c->cell_slot.p, (double*)c->sums.p, N, n_slots, NE);
  HIP_TRY(c, hipGetLastError());
  return finish_product(c, (int64_t)NE * N, N, N, 1, sig);
}

// bf16 MFMA path (kmvp_mfma.hpp): host arrays are float32, points and signal are packed
// to augmented bf16 rows / LDS tile images, sums come back as fp32 partials.
int run_product_mfma(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D;
  const int E = sig == SIG_DENSITY ? 1 : c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  // exp(<x,y>) (include/kmvp.h kmvp_expdot, bfloat16): S = X Y^T log2(e) straight from the matrix pipe, the per-target
  // running shift of kmvp_mfma.hpp, partial sums as (mantissa, exponent) pairs and the shifted tail
  // The Gaussian with targets != sources takes the same running shift (K_GAUSSIAN_SHIFTED): a target farther than ~9 kernel
  // lengths from every source would otherwise have its whole row under the float32 range (0, or 0/0 when normalised) -- the
  // bf16 counterpart of what fastmm_kernel / cfastmm_kernel do for float32.  Targets == sources keep the plain kernel (every
  // row contains k = 1); exp(-r) would need r > 87 and stays plain.
  const bool shifted = kernel == K_GAUSSIAN && !(c->same_points || c->opt_same_global) && c->opt_mfma_variant < 0 &&
                       mfma_ksteps_shifted(D) <= MFMA_MAX_KS;  // (an explicit "mfma_variant" asks for the plain kernel)
  if (shifted) kernel = K_GAUSSIAN_SHIFTED;
  const bool dot = kernel == K_EXPDOT || shifted;  // (mantissa, exponent) partial sums and the shifted tail
  const int KS = kernel == K_EXPDOT ? mfma_ksteps_dot(D) : (shifted ? mfma_ksteps_shifted(D) : mfma_ksteps(D));
  const int NT = (E + 31) / 32;
  if (KS > MFMA_MAX_KS || NT > MFMA_MAX_NT)
    return fail(c, KMVP_E_UNSUPPORTED, "bf16 MFMA path is instantiated for D <= 138 (exp(<x,y>): 141) and E <= 128");
  const int KD = 16 * KS;
  const int NEP = NT * 32;
  const int64_t IMG = mfma_image_bytes(KS, NT);
  const float scale = kernel == K_EXPDOT ? 1.2011224087864498f /* sqrt(log2 e) on both sides */
                                         : scale_for<float>(shifted ? (int)K_GAUSSIAN : kernel);
  const int pack_mode = kernel == K_EXPDOT ? 1 : (shifted ? 2 : 0);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  // target tiles of 32 per wave ("targets_per_lane" option): 1, 2, or (default, where instantiated)
  // 2 with the software-pipelined kernel
  const int TW = c->opt_T == 1 ? 1 : 2;
  const bool pipelined = TW == 2 && c->opt_T != 2 && KS <= MFMA_PIPE_MAX_KS && NT <= MFMA_PIPE_MAX_NT;
  // variant of the pipelined kernel (kmvp_mfma.hpp VAR): exp(-r) (two transcendentals per pair) by default with the loop
  // rotated by one transcendental stage; the denominators stay on the VALU -- on the matrix pipe they save 2 % of the
  // cycles and nothing of the time under the power limit (profiles/r03_c3_variants.txt); the single-transcendental kernels
  // gain nothing from either
  const int variant = dot ? 0 : (c->opt_mfma_variant >= 0 ? c->opt_mfma_variant : (kernel == K_ABSEXP ? 4 : 0));
  const int64_t tile = (int64_t)MFMA_TILE * TW * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(N, tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t m_tiles = (M + MFMA_TILE - 1) / MFMA_TILE;
  int rc;

  // segments: enough workgroups for >= 4 per CU, at least 8 source tiles each
  int64_t seg = c->opt_segments > 0 ? c->opt_segments : (1024 + tile_blocks - 1) / tile_blocks;
  seg = std::max<int64_t>(1, std::min<int64_t>(seg, std::max<int64_t>(1, m_tiles / 8)));
  if (seg >= 8) seg = seg / 8 * 8;
  const int64_t seg_tiles = (m_tiles + seg - 1) / seg;
  const int segments = (int)((m_tiles + seg_tiles - 1) / seg_tiles);

  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                         c->packed_layout != LAYOUT_MFMA || c->packed_T != TW;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  HIP_TRY(c, mark(c, 0));
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_pad * KD * 2))) return rc;
    hipLaunchKernelGGL(pack_mfma_targets_kernel, dim3(blocks_for(n_pad)), dim3(256), 0, c->stream,
                       x_raw, (__bf16*)c->xs.p, N, n_pad, D, KD, scale, pack_mode);
  }
  if (sig_stale) {
    if ((rc = ensure(c, c->rec, (size_t)m_tiles * IMG))) return rc;
    hipLaunchKernelGGL(pack_mfma_sources_kernel, dim3(blocks_for(m_tiles * MFMA_TILE)), dim3(256), 0,
                       c->stream, (const float*)c->y_raw.p,
                       sig == SIG_DENSITY ? (const float*)nullptr : (const float*)c->b_raw.p,
                       (unsigned char*)c->rec.p, M, m_tiles, D, E, KS, NT, scale, pack_mode);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = kernel;
  c->packed_sig = sig;
  c->packed_layout = LAYOUT_MFMA;
  c->packed_T = TW;

  if ((rc = ensure(c, c->part, (size_t)segments * n_pad * NEP * sizeof(float)))) return rc;
  if ((rc = ensure(c, c->partd, (size_t)segments * n_pad * sizeof(float)))) return rc;
  MfmaArgs a;
  a.xa = (const __bf16*)c->xs.p;
  a.img = (const unsigned char*)c->rec.p;
  a.part = (float*)c->part.p;
  a.partd = (float*)c->partd.p;
  a.n_pad = n_pad;
  a.m_tiles = m_tiles;
  a.seg_tiles = seg_tiles;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  a.j_offset = c->j_offset;
  a.m_total = c->m_total;
  a.kexp = nullptr;
  if (dot) {
    if ((rc = ensure(c, c->kexp, (size_t)segments * n_pad * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->kshift, (size_t)n_pad * sizeof(double)))) return rc;
    a.kexp = (float*)c->kexp.p;
  }
  const dim3 grid((unsigned)(tile_blocks * segments));
  HIP_TRY(c, mark(c, 0));
  hipError_t le;
  switch (kernel) {
    case K_EXPDOT: le = launch_mfma_expdot(KS, NT, pipelined ? 3 : TW, a, grid, c->stream, &c->last_kernel_name); break;
    case K_GAUSSIAN_SHIFTED: le = launch_mfma_gaussian_shifted(KS, NT, pipelined ? 3 : TW, a, grid, c->stream, &c->last_kernel_name); break;
    case K_GAUSSIAN: le = launch_mfma_gaussian(KS, NT, pipelined ? 3 + variant : TW, a, grid, c->stream, &c->last_kernel_name); break;
    case K_ABSEXP: le = launch_mfma_absexp(KS, NT, pipelined ? 3 + variant : TW, a, grid, c->stream, &c->last_kernel_name); break;
    default: le = launch_mfma_invdist(KS, NT, pipelined ? 3 + variant : TW, a, grid, c->stream, &c->last_kernel_name); break;
  }
  HIP_TRY(c, le);
  HIP_TRY(c, mark(c, 1));

  const int64_t count = (int64_t)NE * n_pad;
  if ((rc = ensure(c, c->sums, (size_t)count * sizeof(double)))) return rc;
  if (dot) {
    hipLaunchKernelGGL(mfma_reduce_shifted_kernel, dim3(blocks_for(count)), dim3(256), 0, c->stream,
                       (const float*)c->part.p, (const float*)c->partd.p, (const float*)c->kexp.p, (double*)c->sums.p,
                       (double*)c->kshift.p, n_pad, NEP, E, segments, sig == SIG_NORM ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    c->note = shifted ? "bf16 Gaussian with the per-target online shift (targets != sources)"
                      : "exp(<x,y>) on the bf16 matrix cores with the per-target online shift, (mantissa, exponent) partial sums";
    return finish_product_shifted(c, N, n_pad, E, sig);
  }
  hipLaunchKernelGGL(mfma_reduce_kernel, dim3(blocks_for(count)), dim3(256), 0, c->stream,
                     (const float*)c->part.p, (const float*)c->partd.p, (double*)c->sums.p, n_pad, NEP,
                     E, segments, sig == SIG_NORM ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  return finish_product(c, count, N, n_pad, E, sig);
}

}  // namespace

// Centre and squared half-diagonal of the bounding box of the clouds just uploaded by
// kmvp_set_points (float32, D <= 7 only: the inputs of the split-bf16 path and of its "auto"
// rule).  The launch and the 4-byte read-back are queued on the context's stream; the caller
// synchronises.
int measure_clouds(kmvp_ctx* c, int dtype, int64_t M, int64_t N, int D) {
  c->cloud_radius2 = INFINITY;
  // float32: inputs of the split-bf16 paths and of their "auto" rule; float64: the grid of the cell path (D <= 3)
  const bool f64_cells = dtype == KMVP_F64 && D <= CELL_MAX_D;
  if ((dtype != KMVP_F32 && !f64_cells) || D > FMM_MAX_D || M <= 0 || N <= 0) return KMVP_OK;  // (aux holds D <= 64 centres)
  int rc;
  if ((rc = ensure(c, c->aux, (FAST_AUX_FLOATS + 2 * (size_t)FAST_BBOX_BLOCKS * D) * sizeof(float)))) return rc;
  float* part = (float*)c->aux.p + FAST_AUX_FLOATS;
  if (f64_cells)  // box in float32 (the cells only need it to a grid spacing), outward rounding is not needed: keys are clamped
    hipLaunchKernelGGL((fast_bbox_partial_kernel<double>), dim3(FAST_BBOX_BLOCKS), dim3(256), 0, c->stream,
                       (const double*)c->y_raw.p, M, c->same_points ? (const double*)nullptr : (const double*)c->x_raw.p, N,
                       D, part);
  else
    hipLaunchKernelGGL((fast_bbox_partial_kernel<float>), dim3(FAST_BBOX_BLOCKS), dim3(256), 0, c->stream,
                       (const float*)c->y_raw.p, M, c->same_points ? (const float*)nullptr : (const float*)c->x_raw.p, N,
                       D, part);
  hipLaunchKernelGGL(fast_center_kernel, dim3(1), dim3(FAST_BBOX_BLOCKS), 0, c->stream, (const float*)part, D,
                     (float*)c->aux.p);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(&c->cloud_radius2, (float*)c->aux.p + FAST_AUX_RADIUS2, sizeof(float),
                            hipMemcpyDeviceToHost, c->stream));
  c->centre_ver = c->points_ver + 1;  // kmvp_set_points bumps points_ver once the upload is complete
  return KMVP_OK;
}

// kmvp_fit: the cell structures of the Gaussian paths depend on the points only, so they can be built
// before any signal exists (the conditions are those of run_product() with E = 1 assumed).
int prepare_points(kmvp_ctx* c, int kernel) {
  if (kernel != K_GAUSSIAN || c->D > CELL_MAX_D || c->N == 0 || c->M == 0 || c->centre_ver != c->points_ver ||
      c->opt_fast == 0 || c->opt_fast == 1 || c->opt_fast == 2)
    return KMVP_OK;
  const bool big = c->N >= SMALL_PROBLEM_TARGETS && c->M >= SMALL_PROBLEM_TARGETS;
  if (c->dtype == KMVP_F32) {
    const float sc = scale_for<float>(kernel);
    const bool global_ok = c->cloud_radius2 * sc * sc <= FAST_AUTO_RADIUS2;
    if (c->opt_fast >= 3 || (global_ok && big))
      return cell_prepare(c, c->opt_fast_tiles > 0 ? c->opt_fast_tiles : (c->N < SMALL_PROBLEM_TARGETS ? 1 : 0));
  } else if (c->dtype == KMVP_F64) {
    if (c->opt_fast == 3 || big) return cell64_prepare(c);
  }
  return KMVP_OK;
}

// Why a float32 Gaussian product at D <= 3 did not take the cell form (kmvp_last_dispatch_note): the fastest paths
// are narrow, and a caller who gets 5e12 instead of 3.7e13 pairs/s should be told which condition failed.
static void note_no_cells(kmvp_ctx* c) {
  char buf[256];
  const float sc = scale_for<float>(K_GAUSSIAN);
  if (c->opt_fast >= 0 && c->opt_fast < 3) {
    snprintf(buf, sizeof buf, "cell form not considered: fast_sqdists = %d was requested", c->opt_fast);
  } else if (!(c->cloud_radius2 * sc * sc <= FAST_AUTO_RADIUS2)) {
    snprintf(buf, sizeof buf, "cell form not taken: squared half-diagonal of the clouds' bounding box %.3g > %.3g (radius rule)",
             (double)(c->cloud_radius2 * sc * sc), (double)FAST_AUTO_RADIUS2);
  } else if (c->N < SMALL_PROBLEM_TARGETS || c->M < SMALL_PROBLEM_TARGETS) {
    snprintf(buf, sizeof buf, "cell form not taken: fewer than %lld points in a cloud (N = %lld, M = %lld)",
             (long long)SMALL_PROBLEM_TARGETS, (long long)c->N, (long long)c->M);
  } else if (c->cell_state != 1) {
    snprintf(buf, sizeof buf, "cell form not taken: the grid does not apply (non-finite box or more than %d cells along an axis)",
             CELL_MAX_GRID);
  } else {
    snprintf(buf, sizeof buf, "cell form not taken: padding the cells' tiles would use %.0f %% of the slots (limit %.0f %%): too few "
             "points per grid cell", 100.0 * cell_padding(c), 100.0 * CELL_AUTO_MAX_PAD);
  }
  c->note = buf;
}

int run_product(kmvp_ctx* c, int kernel, bool normalise) {
  if (!c) return KMVP_E_INVALID;
  c->note.clear();
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points has not been called");
  if (!c->have_signal) return fail(c, KMVP_E_INVALID, "kmvp_set_signal has not been called");
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->M < c->m_total && !(c->exchanges() && c->world > 1) && !c->opt_partial)
    // a slice of the sources and nobody to sum the shards with: the result would be this rank's partial
    // sums passed off as the product
    return fail(c, KMVP_E_INVALID,
                "the sources are a shard (M < M_total) but no multi-rank communicator is attached: call kmvp_comm_init, "
                "or set option partial_shard = 1 to get this shard's partial sums on purpose");
  if (c->M == 0 && c->N > 0 && c->exchanges() && c->world > 1) {
    // a rank whose source slice is empty still owes the other ranks its (zero) share of the sums
    const int sig0 = c->density ? SIG_DENSITY : (normalise ? SIG_NORM : SIG_PRODUCT);
    const int E = c->density ? 1 : c->E;
    const int64_t NE = sig0 == SIG_NORM ? E + 1 : E;
    int rc = ensure(c, c->sums, (size_t)NE * c->N * sizeof(double));
    if (rc) return rc;
    HIP_TRY(c, mark(c, 0));
    HIP_TRY(c, hipMemsetAsync(c->sums.p, 0, (size_t)NE * c->N * sizeof(double), c->stream));
    HIP_TRY(c, mark(c, 1));
    c->last_kernel_name = "none";
    if (c->density && normalise) {  // every rank takes the all-ones shortcut below: no exchange
    } else if (kernel == K_EXPDOT) {  // ... and its (absent) exponents: +inf loses every all-reduce(min)
      if ((rc = ensure(c, c->kshift, (size_t)c->N * sizeof(double)))) return rc;
      hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(c->N)), dim3(256), 0, c->stream, (double*)c->kshift.p, c->N,
                         (double)INFINITY);
      HIP_TRY(c, hipGetLastError());
      return finish_product_shifted(c, c->N, c->N, E, sig0);
    } else {
      return finish_product(c, NE * c->N, c->N, c->N, E, sig0);
    }
  }
  if (c->N == 0 || c->M == 0) {
    // empty clouds: a = 0 (N,E); nothing to launch
    const int E = c->density ? 1 : c->E;
    int rc = ensure(c, c->out, (size_t)std::max<int64_t>(c->N, 1) * E * sizeof(double));
    if (rc) return rc;
    if (c->N > 0) {
      hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(c->N * E)), dim3(256), 0, c->stream,
                         (double*)c->out.p, c->N * E,
                         normalise ? std::nan("") : 0.0);  // 0/0 in the reference
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->out_n = c->N;
    c->out_e = E;
    c->last_kernel_ms = c->last_total_ms = 0.f;
    return KMVP_OK;
  }
  if (c->density && normalise) {
    // bruteforce.py:134-138: the rows of a normalised matrix sum to one
    int rc = ensure(c, c->out, (size_t)c->N * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(c->N)), dim3(256), 0, c->stream,
                       (double*)c->out.p, c->N, 1.0);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->out_n = c->N;
    c->out_e = 1;
    c->last_kernel_ms = c->last_total_ms = 0.f;
    c->last_kernel_name = "fill_kernel";
    return KMVP_OK;
  }
  const int sig = c->density ? SIG_DENSITY : (normalise ? SIG_NORM : SIG_PRODUCT);
  if (kernel == K_EXPDOT) {
    // k = exp(<x,y>) (include/kmvp.h kmvp_expdot): float32 on fastmm_kernel's matrix-core path only
    if (c->density) return fail(c, KMVP_E_UNSUPPORTED, "exp(<x,y>): pass a signal of ones for density estimation");
    if (c->dtype == KMVP_BF16) return run_product_mfma(c, K_EXPDOT, sig);  // D <= 141, E <= 128
    if (c->dtype != KMVP_F32 || c->D > FMM_MAX_D)
      return fail(c, KMVP_E_UNSUPPORTED,
                  "exp(<x,y>) is built for float32 at D <= 64 and for bfloat16 (other cases: the plugin's Gaussian identity)");
    return run_product_fastmm(c, K_GAUSSIAN, sig, true);
  }
  if (c->dtype == KMVP_BF16) return run_product_mfma(c, kernel, sig);
  if (c->dtype == KMVP_F32 && kernel == K_GAUSSIAN && !c->density && (c->E > 1 || c->D > FAST_MAX_D) &&
      c->centre_ver == c->points_ver && (c->opt_fast < 0 || c->opt_fast == 1 || c->opt_fast == 3)) {
    // Several signal columns (low-D attention with E value channels) -- and single columns beyond fast_kernel's D = 39,
    // which fastmm_kernel serves up to D = 64.  Two forms on the matrix cores:
    //   fastmm_kernel  the tile of kernel values goes back to the matrix pipe for the product with the signal: up to 32
    //                  columns per pass at a cost per pair that does not depend on the column count (D <= 64);
    //   cellmm_kernel  one launch per column where the cell form applies (same rule as for E = 1 below; D <= 3) --
    //                  a seventh of fastmm's cost per pair and column where the cells are well filled.
    // auto takes the cheaper one by the tile counts (picoseconds per 32 x 32 tile on the whole chip, measured at
    // 1e5 .. 1e6 points: tools/fmm_probe.py); fast_sqdists 1 / 3 force one of them.
    const float sc = scale_for<float>(kernel);
    const bool global_ok = c->cloud_radius2 * sc * sc <= FAST_AUTO_RADIUS2;
    const int cols = c->E + (normalise ? 1 : 0);
    const bool fmm_ok = c->D <= FMM_MAX_D && (c->opt_fast == 1 || (c->opt_fast < 0 && global_ok));
    double t_cell = INFINITY;
    if (c->D <= CELL_MAX_D && global_ok && c->opt_fast != 1 &&
        (c->opt_fast == 3 || (c->N >= SMALL_PROBLEM_TARGETS && c->M >= SMALL_PROBLEM_TARGETS))) {
      const int TT = c->opt_fast_tiles > 0 ? c->opt_fast_tiles : (c->N < SMALL_PROBLEM_TARGETS ? 1 : 0);
      int rc = cell_prepare(c, TT);
      if (rc) return rc;
      if (c->cell_state == 1 && (c->opt_fast == 3 || cell_padding(c) <= CELL_AUTO_MAX_PAD) &&
          cellmm_wlog2(c) <= CMM_MAX_WLOG2)
        t_cell = (double)cols * (double)c->cell_m_tiles *
                 (cell_ps_per_tile(c->cell_tt) * (double)c->cell_n_main + cell_ps_per_tile(2) * (double)c->cell_n_rest);
    }
    // (the inputs of the choice go into the dispatch note: the constants were measured on one power-limited box)
    char cost[224];
    double t_fmm = INFINITY;
    if (fmm_ok) {
      const double tiles = std::ceil((double)c->N / FAST_TILE) * std::ceil((double)c->M / FAST_TILE);
      t_fmm = tiles * ((cols + FMM_MAX_COLS - 1) / FMM_MAX_COLS) * (cols > 16 ? FMM_PS_PER_TILE_32 : FMM_PS_PER_TILE_16);
    }
    snprintf(cost, sizeof cost, " [cost model, %d columns: fastmm_kernel %.3g ms, cellmm_kernel %.3g ms (%lld + %lld target x %lld source tiles)]",
             cols, t_fmm * 1e-9, t_cell * 1e-9, (long long)c->cell_n_main, (long long)c->cell_n_rest, (long long)c->cell_m_tiles);
    if (fmm_ok && (c->opt_fast == 1 || t_fmm <= t_cell)) {
      c->note.clear();
      const int rc = run_product_fastmm(c, K_GAUSSIAN, sig);
      if (c->opt_fast < 0) c->note += cost;
      return rc;
    }
    if (t_cell < INFINITY) {
      c->note.clear();
      const int rc = run_product_cellmm(c, sig);
      if (c->opt_fast < 0) c->note += cost;
      return rc;
    }
  }
  if (c->dtype == KMVP_F32 && kernel == K_ABSEXP && !c->density && c->D > CFAST_MAX_D && c->D <= FMM_MAX_D &&
      c->centre_ver == c->points_ver && (c->opt_fast < 0 || c->opt_fast == 1)) {
    // exp(-r) beyond the centred forms' D = 4: fastmm_kernel's expansion around one centre with the closest pairs
    // recomputed exactly (any number of signal columns, one included), on clouds inside the radius rule
    // (auto: beyond the specialised difference form's D = 8, or where that form would go column by column -- at
    // D = 5 with up to four columns the two are equally fast)
    const float sc = scale_for<float>(kernel);
    const bool pays = c->D > LOWD_MAX_D || c->E + (normalise ? 1 : 0) > LOWD_MAX_E;
    if (c->cloud_radius2 * sc * sc <= FAST_AUTO_RADIUS2 && (c->opt_fast == 1 || pays)) {
      c->note.clear();
      return run_product_fastmm(c, K_ABSEXP, sig);
    }
  }
  // (1/r: cfast_kernel's zero rule drops the pair with the target's own index among the COINCIDENT pairs, so the targets
  // must be the sources -- all of them, also when the sources are sharded)
  const bool invdist_square = kernel == K_INVDIST && (c->same_points || c->opt_same_global) && c->N == c->m_total;
  if (c->dtype == KMVP_F32 && (kernel == K_GAUSSIAN || kernel == K_ABSEXP || invdist_square) && !c->density && c->E > 1 &&
      c->D <= CFAST_MAX_D && c->centre_ver == c->points_ver &&
      (c->opt_fast == 2 || (c->opt_fast < 0 && c->E + (normalise ? 1 : 0) >= CFMM_AUTO_MIN_COLS))) {
    // several signal columns where fastmm_kernel does not apply: exp(-r) and 1/r, which need relative accuracy in s, and
    // the Gaussian on clouds outside the radius rule -- cfast_kernel's distances, the same second product
    c->note.clear();
    return run_product_cfastmm(c, kernel, sig);
  }
  if (c->dtype == KMVP_F32 && kernel == K_GAUSSIAN && c->D <= CELL_MAX_D && c->centre_ver == c->points_ver) note_no_cells(c);
  if (c->dtype == KMVP_F32 && c->D <= FAST_MAX_D && (c->density || c->E == 1) && c->centre_ver == c->points_ver) {
    // "fast_sqdists": squared distances in the expanded form on the matrix cores.
    //   fast_kernel  one centre for the whole cloud: absolute error eps32 (|x'|^2 + |y'|^2) in s
    //                -> as accurate as the difference form only for the Gaussian (smooth in s)
    //                on clouds of small scaled radius;
    //   cfast_kernel centre per source group + exact recomputation of the closest pairs:
    //                relative accuracy in s -> every kernel, any radius (D <= 4).  The
    //                inverse-distance zero rule is applied there to coincident pairs only, so
    //                it needs targets == sources.
    // auto (-1) picks the cheapest form that is as accurate as the difference form.
    const float sc = scale_for<float>(kernel);
    const bool global_ok = kernel == K_GAUSSIAN && c->cloud_radius2 * sc * sc <= FAST_AUTO_RADIUS2;
    const bool same = c->same_points || c->opt_same_global;
    const bool centred_ok = c->D <= CFAST_MAX_D && (kernel != K_INVDIST || (same && c->N == c->m_total));
    // cell_kernel: exp() range-reduced by grid cells, the polynomial remainder on the matrix cores
    // (Gaussian, D <= 3).  auto: when the clouds fill the cells well enough that padding stays small.
    // (also inside a solver iteration: grid, cell order and tile lists belong to the points and were built by
    // kmvp_fit or by the first product -- long before a burst of iterations is captured into a hipGraph -- and the
    // launches of the cell paths themselves are asynchronous)
    if (kernel == K_GAUSSIAN && c->D <= CELL_MAX_D &&
        (c->opt_fast >= 3 || (c->opt_fast < 0 && global_ok && c->N >= SMALL_PROBLEM_TARGETS && c->M >= SMALL_PROBLEM_TARGETS))) {
      const int TT = c->opt_fast_tiles > 0 ? c->opt_fast_tiles : (c->N < SMALL_PROBLEM_TARGETS ? 1 : 0);  // 0: by the padding
      int rc = cell_prepare(c, TT);
      if (rc) return rc;
      if (c->cell_state == 1 && (c->opt_fast >= 3 || cell_padding(c) <= CELL_AUTO_MAX_PAD)) {
        // cellmm_kernel (weights in the operand, sum in the accumulator) where its f16 operands have the range:
        // clouds inside the radius rule; cell_kernel otherwise (wide clouds) or on request (fast_sqdists = 4)
        c->note.clear();
        if (c->opt_fast != 4 && global_ok && cellmm_wlog2(c) <= CMM_MAX_WLOG2)
          return run_product_cellmm(c, sig);  // normalised rows: a second launch with b = 1 for the denominator
        if (c->opt_fast != 4) c->note = "cell_kernel instead of cellmm_kernel: the clouds are outside the radius rule (f16 operand range)";
        return run_product_cell(c, sig);
      }
    }
    if (c->opt_fast == 1 || (c->opt_fast < 0 && global_ok)) return run_product_fast(c, kernel, sig);
    if (centred_ok && (c->opt_fast == 2 || c->opt_fast < 0)) return run_product_cfast(c, kernel, sig);
  }
  if (c->dtype == KMVP_F64 && kernel == K_GAUSSIAN && c->D <= CELL_MAX_D && (c->density || c->E == 1) &&
      c->centre_ver == c->points_ver &&
      (c->opt_fast == 3 || (c->opt_fast < 0 && c->N >= SMALL_PROBLEM_TARGETS && c->M >= SMALL_PROBLEM_TARGETS))) {
    // float64 cell form (kmvp_cell64.hpp): 12 fp64 instructions per pair instead of ~23
    int rc = cell64_prepare(c);
    if (rc) return rc;
    if (c->cell_state == 1 &&
        (c->opt_fast == 3 || (double)c->cell_n_tiles * CELL64_TILE <= CELL_AUTO_MAX_PAD * (double)c->N))
      return run_product_cell64(c, sig);
  }
  if (c->D <= LOWD_MAX_D && !c->density && c->E > LOWD_MAX_E && c->note.empty())
    c->note = "E > 4 signal columns: the difference form runs once per block of four columns";
  if (c->D <= LOWD_MAX_D && !c->density && c->E > LOWD_MAX_E)  // low D, many signal columns
    return c->dtype == KMVP_F64 ? run_product_blocked<double>(c, kernel, sig) : run_product_blocked<float>(c, kernel, sig);
  if (c->dtype == KMVP_F64) return run_product_t<double>(c, kernel, sig);
  return run_product_t<float>(c, kernel, sig);
}


}  // namespace kmvp
