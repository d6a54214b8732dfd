// libkmvp.so -- host side of the C ABI declared in include/kmvp.h.
//
// Owns the per-GPU context (device buffers, stream, events, RCCL communicator),
// re-lays-out the caller's arrays for the kernels, picks the launch geometry and
// runs:  pair-loop kernel -> segment reduction -> [RCCL all-reduce] -> finish.
// Reference call order this serves: runner.py:70-148 (prepare_data, fit,
// prepare_query, query, get_result) through the plugin in
// kernel_matrix_benchmarks_amd/algorithms/mi355x.py.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is dlopen'ed on first use
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/kmvp.h"
#include "kmvp_internal.hpp"
#include "kmvp_mfma_pack.hpp"
#include "kmvp_fast_pack.hpp"

using namespace kmvp;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// RCCL entry points, resolved lazily so that single-GPU use never loads the library
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error, path;
  bool load() {
    if (handle) return true;
    // RCCL must drive the SAME HIP runtime as this library: a process may hold two ROCm
    // stacks (e.g. the system one and the copy bundled with PyTorch), and a bare
    // dlopen("librccl.so.1") returns whichever copy happens to be loaded already.  So look
    // next to the libamdhip64 this library is bound to first, by absolute path.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      char resolved[4096];
      std::string hip_path = realpath(info.dli_fname, resolved) ? resolved : info.dli_fname;
      const size_t slash = hip_path.rfind('/');
      if (slash != std::string::npos) {
        const std::string dir = hip_path.substr(0, slash + 1);
        names.push_back(dir + "librccl.so.1");
        names.push_back(dir + "librccl.so");
      }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& n : names) {
      handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (handle) {
        path = n;
        break;
      }
    }
    if (!handle) {
      error = std::string("cannot load librccl: ") + dlerror();
      return false;
    }
    GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
    AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy || !GetErrorString) {
      error = "librccl lacks an expected symbol";
      return false;
    }
    return true;
  }
};
Rccl g_rccl;

}  // namespace

struct kmvp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  std::string err;

  // problem
  int dtype = -1;
  int D = 0, E = 0;
  int64_t M = 0, N = 0, j_offset = 0, m_total = 0;
  bool same_points = false;
  bool have_points = false, have_signal = false, density = false;

  DevBuf y_raw, x_raw, b_raw;   // caller's arrays in the working precision
  DevBuf xs, rec;               // kernel layouts (specialised path; bf16 path: augmented targets, tile images)
  DevBuf partd;                 // bf16 path: partial denominators
  DevBuf aux;                   // fast path: |x'|^2 per target + cloud centre
  DevBuf x_scaled, y_scaled;    // scaled copies (generic path)
  DevBuf part, sums, out;       // fp64 partials, reduced sums, final (N,E)
  DevBuf scratch;               // CG vectors / dot products
  uint64_t points_ver = 0, signal_ver = 0;
  // what xs / rec / scaled copies currently hold
  int packed_kernel = -1, packed_sig = -1, packed_T = -1;
  uint64_t packed_points_ver = 0, packed_signal_ver = 0;
  int gen_kernel = -1;
  uint64_t gen_points_ver = 0;
  int64_t out_n = 0;
  int out_e = 0;

  // tuning (kmvp_set_option)
  int opt_feed = -1, opt_T = 0, opt_segments = 0, opt_chunk = 512;
  int opt_fast = -1, opt_fast_tiles = 0;  // fast_sqdists: -1 auto, 0 never, 1 always
  float cloud_radius2 = INFINITY;          // squared half-diagonal of the clouds' bounding box
  uint64_t centre_ver = 0;

  // sharding
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;

  float last_kernel_ms = 0.f, last_total_ms = 0.f;
  const char* last_kernel_name = "";
};

namespace {

int fail(kmvp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIP_TRY(c, expr)                                                                     \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail((c), e_ == hipErrorOutOfMemory ? KMVP_E_NOMEM : KMVP_E_DEVICE,             \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                        \
  } while (0)

int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }
size_t elem_size(int dtype) { return dtype == KMVP_F64 ? 8 : 4; }

int ensure(kmvp_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return KMVP_OK;
  if (b.p) {
    HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  if (bytes == 0) bytes = 16;
  HIP_TRY(c, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return KMVP_OK;
}

void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

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

// sums[e][i] = sum over segments (index order) of part[s][e][i]
__global__ void reduce_segments_kernel(const double* __restrict__ part, double* __restrict__ sums,
                                       int64_t count /* NE*n_pad */, int segments) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= count) return;
  double v = 0.0;
  for (int s = 0; s < segments; ++s) v += part[(int64_t)s * count + q];
  sums[q] = v;
}

// out[i*E + e] = sums[e][i]  (/ sums[E][i] when normalised)
__global__ void finish_kernel(const double* __restrict__ sums, double* __restrict__ out, int64_t n,
                              int64_t n_pad, int E, int normalise) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double den = normalise ? sums[(int64_t)E * n_pad + i] : 1.0;
  for (int e = 0; e < E; ++e) {
    const double v = sums[(int64_t)e * n_pad + i];
    out[i * E + e] = normalise ? v / den : v;
  }
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

unsigned blocks_for(int64_t n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }

// Number of source segments of a launch (specialised kernels).  Three pulls:
//  * L2 residency: with segments % 8 == 0 each XCD streams one segment at a time
//    (block_to_work), so a segment of <= 2 MiB of records stays in its 4 MiB L2;
//  * parallelism: tile_blocks * segments should be many rounds of the 2048 resident
//    blocks (256 CUs x 8), which only matters when there are few target tiles;
//  * the fp64 partial buffer segments * NE * n_pad * 8 bytes stays bounded.
int choose_segments(const kmvp_ctx* c, int64_t tile_blocks, int64_t m_pad, int NE, int64_t n_pad,
                    int64_t rec_bytes, int64_t min_seg) {
  int64_t seg;
  if (c->opt_segments > 0) {
    seg = c->opt_segments;
  } else {
    const int64_t l2_seg_bytes = 2 << 20;
    seg = 8 * std::max<int64_t>(1, (m_pad * rec_bytes + 8 * l2_seg_bytes - 1) / (8 * l2_seg_bytes));
    const int64_t target_blocks = 16384;
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

// The whole product: everything query() times.  `sig` as in kmvp_lowd.hpp.
template <typename real>
int run_product_t(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D;
  const int E = sig == SIG_DENSITY ? 1 : c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  const bool specialised = D <= LOWD_MAX_D && E <= LOWD_MAX_E;
  const real scale = scale_for<real>(kernel);
  const real* x_raw = (const real*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  int rc;

  int64_t n_pad;
  int segments;
  int64_t seg_len;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (specialised) {
    LowdTuning tune;
    tune.feed = c->opt_feed >= 0 ? c->opt_feed : DEFAULT_FEED;
    tune.targets_per_lane = c->opt_T > 0 ? c->opt_T : (tune.feed == 1 ? DEFAULT_TARGETS_PER_LANE : 2);
    const int T = tune.targets_per_lane;
    const int EB = sig == SIG_DENSITY ? 0 : E;
    const int R = (D + EB + 3) / 4 * 4;
    const int64_t tile = 64 * (int64_t)T * WAVES_PER_BLOCK;
    n_pad = round_up(std::max<int64_t>(N, 1), tile);
    const int64_t tile_blocks = n_pad / tile;
    const int64_t batch = 8;  // two ping-pong batches of 4 records
    const int64_t m_pad = round_up(std::max<int64_t>(M, 1), batch);
    segments = choose_segments(c, tile_blocks, m_pad, NE, n_pad, (int64_t)R * sizeof(real), 1024);
    seg_len = round_up((m_pad + segments - 1) / segments, batch);
    segments = (int)((m_pad + seg_len - 1) / seg_len);

    // (re)pack the kernel layouts when the points, the signal, the kernel or T changed
    const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                           c->packed_T != T;
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
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    hipError_t le = launch_lowd<real>(kernel, D, E, sig, tune, a, dim3((unsigned)nblocks), c->stream,
                                      &c->last_kernel_name);
    if (le == hipErrorInvalidValue)
      return fail(c, KMVP_E_UNSUPPORTED, "no kernel instantiated for this (D, E, targets_per_lane, feed)");
    HIP_TRY(c, le);
  } else {
    // generic fallback: scaled copies of the points, one target per lane
    n_pad = round_up(std::max<int64_t>(N, 1), BLOCK_THREADS);
    if (c->gen_points_ver != c->points_ver || c->gen_kernel != kernel) {
      if ((rc = ensure(c, c->y_scaled, (size_t)M * D * sizeof(real)))) return rc;
      hipLaunchKernelGGL((scale_kernel<real>), dim3(blocks_for(M * D)), dim3(256), 0, c->stream,
                         (const real*)c->y_raw.p, (real*)c->y_scaled.p, M * D, scale);
      if (!c->same_points) {
        if ((rc = ensure(c, c->x_scaled, (size_t)N * D * sizeof(real)))) return rc;
        hipLaunchKernelGGL((scale_kernel<real>), dim3(blocks_for(N * D)), dim3(256), 0, c->stream,
                           x_raw, (real*)c->x_scaled.p, N * D, scale);
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
    HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    HIP_TRY(c, launch_generic<real>(kernel, sig, xg, (const real*)c->y_scaled.p,
                                    sig == SIG_DENSITY ? nullptr : (const real*)c->b_raw.p,
                                    (double*)c->part.p, N, n_pad, M, D, c->E, NE, segments, seg_len,
                                    c->j_offset, c->m_total, c->stream, &c->last_kernel_name));
  }
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));

  // ---- epilogue: segments -> sums, [all-reduce over the source shards], normalise
  const int64_t count = (int64_t)NE * n_pad;
  if ((rc = ensure(c, c->sums, (size_t)count * sizeof(double)))) return rc;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks_for(count)), dim3(256), 0, c->stream,
                     (const double*)c->part.p, (double*)c->sums.p, count, segments);
  HIP_TRY(c, hipGetLastError());
  if (c->comm && c->world > 1) {
    ncclResult_t r = g_rccl.AllReduce(c->sums.p, c->sums.p, (size_t)count, ncclFloat64, ncclSum,
                                      c->comm, c->stream);
    if (r != ncclSuccess)
      return fail(c, KMVP_E_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  }
  if ((rc = ensure(c, c->out, (size_t)std::max<int64_t>(N, 1) * E * sizeof(double)))) return rc;
  hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(std::max<int64_t>(N, 1))), dim3(256), 0,
                     c->stream, (const double*)c->sums.p, (double*)c->out.p, N, n_pad, E,
                     sig == SIG_NORM ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipEventElapsedTime(&c->last_kernel_ms, c->ev[0], c->ev[1]));
  HIP_TRY(c, hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]));
  c->out_n = N;
  c->out_e = E;
  return KMVP_OK;
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
  const int TT = c->opt_fast_tiles > 0 ? c->opt_fast_tiles : FAST_DEFAULT_TT;
  const int64_t SB = fast_stage_bytes(KS, EB);
  const float scale = scale_for<float>(kernel);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int64_t tile = (int64_t)FAST_TILE * TT * WAVES_PER_BLOCK;
  const int64_t n_pad = round_up(N, tile);
  const int64_t tile_blocks = n_pad / tile;
  const int64_t m_tiles = (M + FAST_TILE - 1) / FAST_TILE;
  const int64_t m_stages = (m_tiles + FAST_STAGE - 1) / FAST_STAGE;
  int rc;

  int segments = choose_segments(c, tile_blocks, m_stages, NE, n_pad, SB, 4);
  const int64_t seg_stages = (m_stages + segments - 1) / segments;
  segments = (int)((m_stages + seg_stages - 1) / seg_stages);

  const bool pts_stale = c->packed_points_ver != c->points_ver || c->packed_kernel != kernel ||
                         c->packed_T != -3 - TT;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  float* centre = (float*)c->aux.p;  // written by kmvp_set_points
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_pad * KS * 16 * 2))) return rc;
    hipLaunchKernelGGL(pack_fast_targets_kernel, dim3(blocks_for(n_pad)), dim3(256), 0, c->stream, x_raw,
                       centre, (__bf16*)c->xs.p, N, n_pad, D, KS, scale);
  }
  if (sig_stale) {
    if ((rc = ensure(c, c->rec, (size_t)m_stages * SB))) return rc;
    hipLaunchKernelGGL(pack_fast_sources_kernel, dim3(blocks_for(m_stages * FAST_STAGE * FAST_TILE)),
                       dim3(256), 0, c->stream, (const float*)c->y_raw.p, (const float*)c->b_raw.p, centre,
                       (unsigned char*)c->rec.p, M, m_stages, D, EB, KS, scale);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = kernel;
  c->packed_sig = sig;
  c->packed_T = -3 - TT;  // marks the fast-path layouts

  if ((rc = ensure(c, c->part, (size_t)segments * NE * n_pad * sizeof(double)))) return rc;
  FastArgs a;
  a.xa = (const __bf16*)c->xs.p;
  a.img = (const unsigned char*)c->rec.p;
  a.part = (double*)c->part.p;
  a.n_pad = n_pad;
  a.m_tiles = m_tiles;
  a.m_stages = m_stages;
  a.seg_stages = seg_stages;
  a.segments = segments;
  a.tile_blocks = (int)tile_blocks;
  a.chunk_stages = std::max(1, c->opt_chunk / (FAST_TILE * FAST_STAGE));
  a.j_offset = c->j_offset;
  a.m_total = c->m_total;
  const dim3 grid((unsigned)(tile_blocks * segments));
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  hipError_t le;
  switch (kernel) {
    case K_GAUSSIAN: le = launch_fast_gaussian(KS, sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
    case K_ABSEXP: le = launch_fast_absexp(KS, sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
    default: le = launch_fast_invdist(KS, sig, TT, a, grid, c->stream, &c->last_kernel_name); break;
  }
  if (le == hipErrorInvalidValue) return fail(c, KMVP_E_UNSUPPORTED, "fast_tiles must be 1, 2 or 4");
  HIP_TRY(c, le);
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));

  const int64_t count = (int64_t)NE * n_pad;
  if ((rc = ensure(c, c->sums, (size_t)count * sizeof(double)))) return rc;
  hipLaunchKernelGGL(reduce_segments_kernel, dim3(blocks_for(count)), dim3(256), 0, c->stream,
                     (const double*)c->part.p, (double*)c->sums.p, count, segments);
  HIP_TRY(c, hipGetLastError());
  if (c->comm && c->world > 1) {
    ncclResult_t r = g_rccl.AllReduce(c->sums.p, c->sums.p, (size_t)count, ncclFloat64, ncclSum,
                                      c->comm, c->stream);
    if (r != ncclSuccess)
      return fail(c, KMVP_E_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  }
  if ((rc = ensure(c, c->out, (size_t)N * E * sizeof(double)))) return rc;
  hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(N)), dim3(256), 0, c->stream,
                     (const double*)c->sums.p, (double*)c->out.p, N, n_pad, E, sig == SIG_NORM ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipEventElapsedTime(&c->last_kernel_ms, c->ev[0], c->ev[1]));
  HIP_TRY(c, hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]));
  c->out_n = N;
  c->out_e = E;
  return KMVP_OK;
}

// bf16 MFMA path (kmvp_mfma.hpp): host arrays are float32, points and signal are packed
// to augmented bf16 rows / LDS tile images, sums come back as fp32 partials.
int run_product_mfma(kmvp_ctx* c, int kernel, int sig) {
  const int D = c->D;
  const int E = sig == SIG_DENSITY ? 1 : c->E;
  const int NE = sig == SIG_NORM ? E + 1 : E;
  const int64_t N = c->N, M = c->M;
  const int KS = mfma_ksteps(D);
  const int NT = (E + 31) / 32;
  if (KS > MFMA_MAX_KS || NT > MFMA_MAX_NT)
    return fail(c, KMVP_E_UNSUPPORTED, "bf16 MFMA path is instantiated for D <= 138 and E <= 128");
  const int KD = 16 * KS;
  const int NEP = NT * 32;
  const int64_t IMG = mfma_image_bytes(KS, NT);
  const float scale = scale_for<float>(kernel);
  const float* x_raw = (const float*)(c->same_points ? c->y_raw.p : c->x_raw.p);
  const int TW = c->opt_T == 1 ? 1 : 2;  // target tiles of 32 per wave ("targets_per_lane" option: 1 or 2)
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
                         c->packed_T != -2 - 100 * TW;
  const bool sig_stale = pts_stale || c->packed_signal_ver != c->signal_ver || c->packed_sig != sig;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  if (pts_stale) {
    if ((rc = ensure(c, c->xs, (size_t)n_pad * KD * 2))) return rc;
    hipLaunchKernelGGL(pack_mfma_targets_kernel, dim3(blocks_for(n_pad)), dim3(256), 0, c->stream,
                       x_raw, (__bf16*)c->xs.p, N, n_pad, D, KD, scale);
  }
  if (sig_stale) {
    if ((rc = ensure(c, c->rec, (size_t)m_tiles * IMG))) return rc;
    hipLaunchKernelGGL(pack_mfma_sources_kernel, dim3(blocks_for(m_tiles * MFMA_TILE)), dim3(256), 0,
                       c->stream, (const float*)c->y_raw.p,
                       sig == SIG_DENSITY ? (const float*)nullptr : (const float*)c->b_raw.p,
                       (unsigned char*)c->rec.p, M, m_tiles, D, E, KS, NT, scale);
  }
  HIP_TRY(c, hipGetLastError());
  c->packed_points_ver = c->points_ver;
  c->packed_signal_ver = c->signal_ver;
  c->packed_kernel = kernel;
  c->packed_sig = sig;
  c->packed_T = -2 - 100 * TW;  // marks the bf16 layouts

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
  const dim3 grid((unsigned)(tile_blocks * segments));
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  hipError_t le;
  switch (kernel) {
    case K_GAUSSIAN: le = launch_mfma_gaussian(KS, NT, TW, a, grid, c->stream, &c->last_kernel_name); break;
    case K_ABSEXP: le = launch_mfma_absexp(KS, NT, TW, a, grid, c->stream, &c->last_kernel_name); break;
    default: le = launch_mfma_invdist(KS, NT, TW, a, grid, c->stream, &c->last_kernel_name); break;
  }
  HIP_TRY(c, le);
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));

  const int64_t count = (int64_t)NE * n_pad;
  if ((rc = ensure(c, c->sums, (size_t)count * sizeof(double)))) return rc;
  hipLaunchKernelGGL(mfma_reduce_kernel, dim3(blocks_for(count)), dim3(256), 0, c->stream,
                     (const float*)c->part.p, (const float*)c->partd.p, (double*)c->sums.p, n_pad, NEP,
                     E, segments, sig == SIG_NORM ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  if (c->comm && c->world > 1) {
    ncclResult_t r = g_rccl.AllReduce(c->sums.p, c->sums.p, (size_t)count, ncclFloat64, ncclSum,
                                      c->comm, c->stream);
    if (r != ncclSuccess)
      return fail(c, KMVP_E_COMM, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  }
  if ((rc = ensure(c, c->out, (size_t)N * E * sizeof(double)))) return rc;
  hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(N)), dim3(256), 0, c->stream,
                     (const double*)c->sums.p, (double*)c->out.p, N, n_pad, E, sig == SIG_NORM ? 1 : 0);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipEventElapsedTime(&c->last_kernel_ms, c->ev[0], c->ev[1]));
  HIP_TRY(c, hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]));
  c->out_n = N;
  c->out_e = E;
  return KMVP_OK;
}

int run_product(kmvp_ctx* c, int kernel, bool normalise) {
  if (!c) return KMVP_E_INVALID;
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points has not been called");
  if (!c->have_signal) return fail(c, KMVP_E_INVALID, "kmvp_set_signal has not been called");
  HIP_TRY(c, hipSetDevice(c->device));
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
  if (c->dtype == KMVP_BF16) return run_product_mfma(c, kernel, sig);
  if (c->dtype == KMVP_F32 && c->D <= FAST_MAX_D && (c->density || c->E == 1) && c->centre_ver == c->points_ver) {
    // "fast_sqdists": expanded squared distances on the matrix cores.  auto = only where the
    // expansion is as accurate as the difference form to working precision: the Gaussian
    // (smooth in s; exp(-sqrt(s)) and 1/sqrt(s) amplify the absolute error of s near
    // coincident points, as they do in the reference's own fast form) on clouds whose scaled
    // radius keeps eps32 * (|x'|^2 + |y'|^2) ~ 1e-6.
    const float sc = scale_for<float>(kernel);
    const bool accurate = kernel == K_GAUSSIAN && c->cloud_radius2 * sc * sc <= FAST_AUTO_RADIUS2;
    if (c->opt_fast == 1 || (c->opt_fast < 0 && accurate)) return run_product_fast(c, kernel, sig);
  }
  if (c->dtype == KMVP_F64) return run_product_t<double>(c, kernel, sig);
  return run_product_t<float>(c, kernel, sig);
}


// ------------------------------------------------------------------------------------
// conjugate gradients on K b = a with the on-the-fly product as the operator

constexpr int CG_BLOCKS = 256;

// partial[block][e] = sum over the block's rows of u[i][e] * v[i][e]
__global__ void cg_dot_kernel(const double* __restrict__ u, const double* __restrict__ v, int64_t m,
                              int E, double* __restrict__ partial) {
  __shared__ double red[256];
  for (int e = 0; e < E; ++e) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * blockDim.x)
      acc += u[i * E + e] * v[i * E + e];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * E + e] = red[0];
    __syncthreads();
  }
}

// out[i][e] = u[i][e] + coef[e] * v[i][e]
__global__ void cg_axpy_kernel(double* __restrict__ out, const double* __restrict__ u,
                               const double* __restrict__ v, const double* __restrict__ coef,
                               int64_t m, int E) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  out[q] = u[q] + coef[q % E] * v[q];
}

template <typename real>
__global__ void cg_cast_kernel(const double* __restrict__ in, real* __restrict__ out, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = (real)in[q];
}
template <typename real>
__global__ void cg_widen_kernel(const real* __restrict__ in, double* __restrict__ out, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) out[q] = (double)in[q];
}

struct CgWork {
  double *x, *r, *p, *partial, *coef;
};

int cg_dots(kmvp_ctx* c, const double* u, const double* v, int64_t m, int E, const CgWork& w,
            std::vector<double>& host_partial, std::vector<double>& out) {
  hipLaunchKernelGGL(cg_dot_kernel, dim3(CG_BLOCKS), dim3(256), 0, c->stream, u, v, m, E, w.partial);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(host_partial.data(), w.partial, sizeof(double) * CG_BLOCKS * E,
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out.assign(E, 0.0);
  for (int b = 0; b < CG_BLOCKS; ++b)
    for (int e = 0; e < E; ++e) out[e] += host_partial[(size_t)b * E + e];
  return KMVP_OK;
}

int cg_axpy(kmvp_ctx* c, double* out, const double* u, const double* v,
            const std::vector<double>& coef, int64_t m, int E, const CgWork& w) {
  HIP_TRY(c, hipMemcpyAsync(w.coef, coef.data(), sizeof(double) * E, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // coef is a host temporary
  hipLaunchKernelGGL(cg_axpy_kernel, dim3(blocks_for(m * E)), dim3(256), 0, c->stream, out, u, v,
                     w.coef, m, E);
  HIP_TRY(c, hipGetLastError());
  return KMVP_OK;
}

// K applied to the device vector v (M,E) double; the result lands in c->out (M,E) double.
int cg_apply(kmvp_ctx* c, int kernel, const double* v, int64_t m, int E) {
  int rc = ensure(c, c->b_raw, (size_t)m * E * elem_size(c->dtype));
  if (rc) return rc;
  if (c->dtype == KMVP_F64)
    hipLaunchKernelGGL((cg_cast_kernel<double>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream, v,
                       (double*)c->b_raw.p, m * E);
  else
    hipLaunchKernelGGL((cg_cast_kernel<float>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream, v,
                       (float*)c->b_raw.p, m * E);
  HIP_TRY(c, hipGetLastError());
  c->density = false;
  c->E = E;
  c->have_signal = true;
  ++c->signal_ver;
  return run_product(c, kernel, false);
}

int cg_solve(kmvp_ctx* c, int kernel, const void* a_host, int E, double rtol, int maxit,
             double* out_b, int* iters, double* resid) {
  if (!c) return KMVP_E_INVALID;
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points has not been called");
  if (!c->same_points) return fail(c, KMVP_E_INVALID, "the solver needs x == y (pass x_or_null = NULL)");
  if (c->world > 1) return fail(c, KMVP_E_UNSUPPORTED, "the solver is single-GPU in this build");
  if (!a_host || !out_b || E < 1 || maxit < 0 || !(rtol > 0)) return fail(c, KMVP_E_INVALID, "bad solver arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t m = c->M;
  const size_t vec = (size_t)m * E * sizeof(double);
  int rc = ensure(c, c->scratch, 3 * vec + sizeof(double) * (CG_BLOCKS + 1) * E);
  if (rc) return rc;
  CgWork w;
  w.x = (double*)c->scratch.p;
  w.r = w.x + (size_t)m * E;
  w.p = w.r + (size_t)m * E;
  w.partial = w.p + (size_t)m * E;
  w.coef = w.partial + (size_t)CG_BLOCKS * E;
  std::vector<double> hp((size_t)CG_BLOCKS * E), rs, rs_new, pap, anorm2, coef(E);

  // r = p = a (widened to double), x = 0
  rc = ensure(c, c->b_raw, (size_t)m * E * elem_size(c->dtype));
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->b_raw.p, a_host, (size_t)m * E * elem_size(c->dtype), hipMemcpyHostToDevice, c->stream));
  if (c->dtype == KMVP_F64)
    hipLaunchKernelGGL((cg_widen_kernel<double>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream,
                       (const double*)c->b_raw.p, w.r, m * E);
  else
    hipLaunchKernelGGL((cg_widen_kernel<float>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream,
                       (const float*)c->b_raw.p, w.r, m * E);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(w.p, w.r, vec, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(w.x, 0, vec, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if ((rc = cg_dots(c, w.r, w.r, m, E, w, hp, rs))) return rc;
  anorm2 = rs;

  auto worst = [&](const std::vector<double>& r2) {
    double wv = 0.0;
    for (int e = 0; e < E; ++e)
      wv = std::max(wv, anorm2[e] > 0 ? std::sqrt(r2[e] / anorm2[e]) : 0.0);
    return wv;
  };

  int it = 0;
  double rel = worst(rs);
  while (it < maxit && rel > rtol) {
    if ((rc = cg_apply(c, kernel, w.p, m, E))) return rc;
    const double* Ap = (const double*)c->out.p;
    if ((rc = cg_dots(c, w.p, Ap, m, E, w, hp, pap))) return rc;
    for (int e = 0; e < E; ++e) coef[e] = (pap[e] != 0.0 && rs[e] > 0.0) ? rs[e] / pap[e] : 0.0;
    if ((rc = cg_axpy(c, w.x, w.x, w.p, coef, m, E, w))) return rc;
    for (int e = 0; e < E; ++e) coef[e] = -coef[e];
    if ((rc = cg_axpy(c, w.r, w.r, Ap, coef, m, E, w))) return rc;
    if ((rc = cg_dots(c, w.r, w.r, m, E, w, hp, rs_new))) return rc;
    for (int e = 0; e < E; ++e) coef[e] = rs[e] > 0.0 ? rs_new[e] / rs[e] : 0.0;
    if ((rc = cg_axpy(c, w.p, w.r, w.p, coef, m, E, w))) return rc;
    rs = rs_new;
    rel = worst(rs);
    ++it;
  }

  // true residual ||a - K x|| / ||a|| with one more product
  if ((rc = cg_apply(c, kernel, w.x, m, E))) return rc;
  // w.p = a (widened again) - K x
  HIP_TRY(c, hipMemcpyAsync(c->b_raw.p, a_host, (size_t)m * E * elem_size(c->dtype), hipMemcpyHostToDevice, c->stream));
  if (c->dtype == KMVP_F64)
    hipLaunchKernelGGL((cg_widen_kernel<double>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream,
                       (const double*)c->b_raw.p, w.p, m * E);
  else
    hipLaunchKernelGGL((cg_widen_kernel<float>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream,
                       (const float*)c->b_raw.p, w.p, m * E);
  HIP_TRY(c, hipGetLastError());
  for (int e = 0; e < E; ++e) coef[e] = -1.0;
  if ((rc = cg_axpy(c, w.p, w.p, (const double*)c->out.p, coef, m, E, w))) return rc;
  if ((rc = cg_dots(c, w.p, w.p, m, E, w, hp, rs_new))) return rc;
  const double true_rel = worst(rs_new);

  HIP_TRY(c, hipMemcpyAsync(out_b, w.x, vec, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_signal = false;  // b_raw was used as scratch
  if (iters) *iters = it;
  if (resid) *resid = true_rel;
  if (true_rel > rtol * 1.5 && rel > rtol) {
    c->err = "conjugate gradients reached maxit before the requested residual";
    return KMVP_E_NOT_CONVERGED;
  }
  return KMVP_OK;
}


// ------------------------------------------------------------------------------------
// MINRES (Paige & Saunders) for the symmetric INDEFINITE inverse-distance systems (zero
// diagonal), where conjugate gradients does not apply.  One product per iteration.

// out[i][e] = ca[e] * a[i][e] + cb[e] * b[i][e] + cc[e] * c[i][e]   (coefficients: [3][E])
__global__ void vec_lin3_kernel(double* __restrict__ out, const double* __restrict__ a,
                                const double* __restrict__ b, const double* __restrict__ c,
                                const double* __restrict__ coef, int64_t m, int E) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  const int e = (int)(q % E);
  double v = coef[e] * a[q];
  if (b) v += coef[E + e] * b[q];
  if (c) v += coef[2 * E + e] * c[q];
  out[q] = v;
}

int minres_solve(kmvp_ctx* c, int kernel, const void* a_host, int E, double rtol, int maxit,
                 double* out_b, int* iters, double* resid) {
  if (!c) return KMVP_E_INVALID;
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points has not been called");
  if (!c->same_points) return fail(c, KMVP_E_INVALID, "the solver needs x == y (pass x_or_null = NULL)");
  if (c->world > 1) return fail(c, KMVP_E_UNSUPPORTED, "the solver is single-GPU in this build");
  if (!a_host || !out_b || E < 1 || maxit < 0 || !(rtol > 0)) return fail(c, KMVP_E_INVALID, "bad solver arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t m = c->M;
  const size_t n = (size_t)m * E;
  const size_t vec = n * sizeof(double);
  int rc = ensure(c, c->scratch, 8 * vec + sizeof(double) * ((CG_BLOCKS + 3) * (size_t)E));
  if (rc) return rc;
  double* base = (double*)c->scratch.p;
  double *x = base, *r1 = base + n, *r2 = base + 2 * n, *y = base + 3 * n, *v = base + 4 * n;
  double *w = base + 5 * n, *w1 = base + 6 * n, *w2 = base + 7 * n;
  CgWork wk;
  wk.x = wk.r = wk.p = nullptr;
  wk.partial = base + 8 * n;
  wk.coef = wk.partial + (size_t)CG_BLOCKS * E;  // 3*E coefficients
  std::vector<double> hp((size_t)CG_BLOCKS * E), dots, coef(3 * (size_t)E);
  auto lin3 = [&](double* out, const double* pa, const double* pb, const double* pc) -> int {
    HIP_TRY(c, hipMemcpyAsync(wk.coef, coef.data(), sizeof(double) * 3 * E, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL(vec_lin3_kernel, dim3(blocks_for((int64_t)n)), dim3(256), 0, c->stream, out, pa, pb, pc,
                       wk.coef, m, E);
    HIP_TRY(c, hipGetLastError());
    return KMVP_OK;
  };

  // r1 = r2 = y = a (widened), x = w = w2 = 0
  rc = ensure(c, c->b_raw, n * elem_size(c->dtype));
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->b_raw.p, a_host, n * elem_size(c->dtype), hipMemcpyHostToDevice, c->stream));
  if (c->dtype == KMVP_F64)
    hipLaunchKernelGGL((cg_widen_kernel<double>), dim3(blocks_for((int64_t)n)), dim3(256), 0, c->stream,
                       (const double*)c->b_raw.p, y, (int64_t)n);
  else
    hipLaunchKernelGGL((cg_widen_kernel<float>), dim3(blocks_for((int64_t)n)), dim3(256), 0, c->stream,
                       (const float*)c->b_raw.p, y, (int64_t)n);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(r1, y, vec, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(r2, y, vec, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(x, 0, vec, c->stream));
  HIP_TRY(c, hipMemsetAsync(w, 0, vec, c->stream));
  HIP_TRY(c, hipMemsetAsync(w2, 0, vec, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if ((rc = cg_dots(c, y, y, m, E, wk, hp, dots))) return rc;

  std::vector<double> beta1(E), beta(E), oldb(E, 0.0), dbar(E, 0.0), epsln(E, 0.0), phibar(E), cs(E, -1.0),
      sn(E, 0.0), alfa(E), oldeps(E), delta(E), gbar(E), gamma(E), phi(E);
  std::vector<char> done(E, 0);
  for (int e = 0; e < E; ++e) {
    beta1[e] = std::sqrt(dots[e]);
    beta[e] = beta1[e];
    phibar[e] = beta1[e];
    if (!(beta1[e] > 0.0)) done[e] = 1;  // zero right-hand side: x = 0
  }
  auto worst = [&]() {
    double wv = 0.0;
    for (int e = 0; e < E; ++e)
      if (beta1[e] > 0.0) wv = std::max(wv, phibar[e] / beta1[e]);
    return wv;
  };

  int it = 0;
  double rel = worst();
  while (it < maxit && rel > rtol) {
    ++it;
    // v = y / beta
    for (int e = 0; e < E; ++e) coef[e] = (!done[e] && beta[e] > 0.0) ? 1.0 / beta[e] : 0.0;
    if ((rc = lin3(v, y, nullptr, nullptr))) return rc;
    // y = K v - (beta / oldb) r1
    if ((rc = cg_apply(c, kernel, v, m, E))) return rc;
    for (int e = 0; e < E; ++e) {
      coef[e] = 1.0;
      coef[E + e] = (it >= 2 && oldb[e] > 0.0) ? -beta[e] / oldb[e] : 0.0;
    }
    if ((rc = lin3(y, (const double*)c->out.p, r1, nullptr))) return rc;
    if ((rc = cg_dots(c, v, y, m, E, wk, hp, dots))) return rc;
    for (int e = 0; e < E; ++e) alfa[e] = dots[e];
    // y = y - (alfa / beta) r2 ; then r1 <- r2, r2 <- y (buffer rotation)
    for (int e = 0; e < E; ++e) {
      coef[e] = 1.0;
      coef[E + e] = beta[e] > 0.0 ? -alfa[e] / beta[e] : 0.0;
    }
    if ((rc = lin3(r1, y, r2, nullptr))) return rc;  // written into the old r1 buffer
    {
      double* newy = r1;
      r1 = r2;
      r2 = newy;
      // y must alias r2's content for the next iteration's "v = y / beta": keep y as its own buffer
      HIP_TRY(c, hipMemcpyAsync(y, r2, vec, hipMemcpyDeviceToDevice, c->stream));
    }
    if ((rc = cg_dots(c, r2, r2, m, E, wk, hp, dots))) return rc;
    for (int e = 0; e < E; ++e) {
      oldb[e] = beta[e];
      beta[e] = std::sqrt(std::max(dots[e], 0.0));
      oldeps[e] = epsln[e];
      delta[e] = cs[e] * dbar[e] + sn[e] * alfa[e];
      gbar[e] = sn[e] * dbar[e] - cs[e] * alfa[e];
      epsln[e] = sn[e] * beta[e];
      dbar[e] = -cs[e] * beta[e];
      gamma[e] = std::max(std::sqrt(gbar[e] * gbar[e] + beta[e] * beta[e]), 1e-300);
      cs[e] = gbar[e] / gamma[e];
      sn[e] = beta[e] / gamma[e];
      phi[e] = cs[e] * phibar[e];
      phibar[e] = sn[e] * phibar[e];
    }
    // w_new = (v - oldeps w1 - delta w2) / gamma with w1 <- w2, w2 <- w
    {
      double* t = w1;
      w1 = w2;
      w2 = w;
      w = t;
    }
    for (int e = 0; e < E; ++e) {
      const double dn = done[e] ? 0.0 : 1.0 / gamma[e];
      coef[e] = dn;
      coef[E + e] = -oldeps[e] * dn;
      coef[2 * E + e] = -delta[e] * dn;
    }
    if ((rc = lin3(w, v, w1, w2))) return rc;
    // x = x + phi w
    for (int e = 0; e < E; ++e) {
      coef[e] = 1.0;
      coef[E + e] = done[e] ? 0.0 : phi[e];
    }
    if ((rc = lin3(x, x, w, nullptr))) return rc;
    for (int e = 0; e < E; ++e)
      if (!done[e] && (phibar[e] <= rtol * beta1[e] || beta[e] == 0.0)) done[e] = 1;
    rel = worst();
  }

  // true residual ||a - K x|| / ||a||
  if ((rc = cg_apply(c, kernel, x, m, E))) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->b_raw.p, a_host, n * elem_size(c->dtype), hipMemcpyHostToDevice, c->stream));
  if (c->dtype == KMVP_F64)
    hipLaunchKernelGGL((cg_widen_kernel<double>), dim3(blocks_for((int64_t)n)), dim3(256), 0, c->stream,
                       (const double*)c->b_raw.p, v, (int64_t)n);
  else
    hipLaunchKernelGGL((cg_widen_kernel<float>), dim3(blocks_for((int64_t)n)), dim3(256), 0, c->stream,
                       (const float*)c->b_raw.p, v, (int64_t)n);
  HIP_TRY(c, hipGetLastError());
  for (int e = 0; e < E; ++e) {
    coef[e] = 1.0;
    coef[E + e] = -1.0;
  }
  if ((rc = lin3(v, v, (const double*)c->out.p, nullptr))) return rc;
  if ((rc = cg_dots(c, v, v, m, E, wk, hp, dots))) return rc;
  double true_rel = 0.0;
  for (int e = 0; e < E; ++e)
    if (beta1[e] > 0.0) true_rel = std::max(true_rel, std::sqrt(dots[e]) / beta1[e]);

  HIP_TRY(c, hipMemcpyAsync(out_b, x, vec, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_signal = false;
  if (iters) *iters = it;
  if (resid) *resid = true_rel;
  if (true_rel > rtol * 1.5 && rel > rtol) {
    c->err = "MINRES reached maxit before the requested residual";
    return KMVP_E_NOT_CONVERGED;
  }
  return KMVP_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

int kmvp_abi_version(void) { return KMVP_ABI_VERSION; }

int kmvp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -KMVP_E_DEVICE;
  return n;
}

kmvp_ctx* kmvp_create(int device, int* status) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fail(nullptr, KMVP_E_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (status) *status = KMVP_E_DEVICE;
    return nullptr;
  }
  if (device < 0 || device >= n) {
    fail(nullptr, KMVP_E_INVALID, "device index out of range");
    if (status) *status = KMVP_E_INVALID;
    return nullptr;
  }
  kmvp_ctx* c = new kmvp_ctx();
  c->device = device;
  bool ok = hipSetDevice(device) == hipSuccess &&
            hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok && i < 3; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
  if (!ok) {
    fail(nullptr, KMVP_E_DEVICE, "cannot create stream/events on the device");
    if (status) *status = KMVP_E_DEVICE;
    kmvp_destroy(c);
    return nullptr;
  }
  if (status) *status = KMVP_OK;
  return c;
}

void kmvp_destroy(kmvp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  for (DevBuf* b : {&c->y_raw, &c->x_raw, &c->b_raw, &c->xs, &c->rec, &c->x_scaled, &c->y_scaled,
                    &c->part, &c->partd, &c->aux, &c->sums, &c->out, &c->scratch})
    release(*b);
  for (int i = 0; i < 3; ++i)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* kmvp_last_error(const kmvp_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int kmvp_set_points(kmvp_ctx* c, const void* y, int64_t M, const void* x_or_null, int64_t N, int D,
                    int dtype, int64_t j_offset, int64_t M_total) {
  if (!c) return KMVP_E_INVALID;
  if (dtype != KMVP_F32 && dtype != KMVP_F64 && dtype != KMVP_BF16)
    return fail(c, KMVP_E_INVALID, "unknown dtype");
  if (M < 0 || N < 0 || D < 1) return fail(c, KMVP_E_INVALID, "bad shape");
  if ((M > 0 && !y) || (N > 0 && !x_or_null && N != M))
    return fail(c, KMVP_E_INVALID, "same_points (x == NULL) needs N == M");
  if (M_total < M || j_offset < 0 || j_offset + M > M_total)
    return fail(c, KMVP_E_INVALID, "shard [j_offset, j_offset+M) is outside [0, M_total)");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t es = elem_size(dtype);
  int rc;
  if ((rc = ensure(c, c->y_raw, (size_t)M * D * es))) return rc;
  if (M > 0) HIP_TRY(c, hipMemcpyAsync(c->y_raw.p, y, (size_t)M * D * es, hipMemcpyHostToDevice, c->stream));
  c->same_points = (x_or_null == nullptr);
  if (!c->same_points) {
    if ((rc = ensure(c, c->x_raw, (size_t)N * D * es))) return rc;
    if (N > 0) HIP_TRY(c, hipMemcpyAsync(c->x_raw.p, x_or_null, (size_t)N * D * es, hipMemcpyHostToDevice, c->stream));
  }
  c->cloud_radius2 = INFINITY;
  if (dtype == KMVP_F32 && D <= FAST_MAX_D && M > 0 && N > 0) {
    // bounding box of the clouds for the split-bf16 path (centre + squared half-diagonal)
    if ((rc = ensure(c, c->aux, 16 * sizeof(float)))) return rc;
    hipLaunchKernelGGL(fast_center_kernel, dim3(1), dim3(1024), 0, c->stream, (const float*)c->y_raw.p, M,
                       c->same_points ? (const float*)nullptr : (const float*)c->x_raw.p, N, D,
                       (float*)c->aux.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&c->cloud_radius2, (float*)c->aux.p + 8, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    c->centre_ver = c->points_ver + 1;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // host buffers are only read during the call
  c->dtype = dtype;
  c->D = D;
  c->M = M;
  c->N = N;
  c->j_offset = j_offset;
  c->m_total = M_total;
  c->have_points = true;
  c->have_signal = false;
  ++c->points_ver;
  return KMVP_OK;
}

int kmvp_set_signal(kmvp_ctx* c, const void* b_or_null, int E) {
  if (!c) return KMVP_E_INVALID;
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points must come first");
  if (E < 1) return fail(c, KMVP_E_INVALID, "E must be >= 1");
  if (!b_or_null && E != 1) return fail(c, KMVP_E_INVALID, "density estimation (b == NULL) needs E == 1");
  HIP_TRY(c, hipSetDevice(c->device));
  c->density = (b_or_null == nullptr);
  c->E = E;
  if (!c->density) {
    const size_t bytes = (size_t)c->M * E * elem_size(c->dtype);
    int rc = ensure(c, c->b_raw, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(c, hipMemcpyAsync(c->b_raw.p, b_or_null, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->have_signal = true;
  ++c->signal_ver;
  return KMVP_OK;
}

int kmvp_gaussian(kmvp_ctx* c) { return run_product(c, K_GAUSSIAN, false); }
int kmvp_gaussian_norm(kmvp_ctx* c) { return run_product(c, K_GAUSSIAN, true); }
int kmvp_absexp(kmvp_ctx* c) { return run_product(c, K_ABSEXP, false); }
int kmvp_absexp_norm(kmvp_ctx* c) { return run_product(c, K_ABSEXP, true); }
int kmvp_invdist(kmvp_ctx* c) { return run_product(c, K_INVDIST, false); }
int kmvp_invdist_norm(kmvp_ctx* c) { return run_product(c, K_INVDIST, true); }

int kmvp_get_result(kmvp_ctx* c, double* out, int64_t out_len) {
  if (!c) return KMVP_E_INVALID;
  const int64_t n = c->out_n * c->out_e;
  if (!c->out.p && n > 0) return fail(c, KMVP_E_INVALID, "no result: run a product first");
  if (out_len < n || (n > 0 && !out)) return fail(c, KMVP_E_INVALID, "output buffer too small");
  HIP_TRY(c, hipSetDevice(c->device));
  if (n > 0) {
    HIP_TRY(c, hipMemcpyAsync(out, c->out.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return KMVP_OK;
}

int kmvp_gaussian_cg_solve(kmvp_ctx* c, const void* a, int E, double rtol, int maxit, double* out_b,
                           int* iters, double* resid) {
  return cg_solve(c, K_GAUSSIAN, a, E, rtol, maxit, out_b, iters, resid);
}
int kmvp_absexp_cg_solve(kmvp_ctx* c, const void* a, int E, double rtol, int maxit, double* out_b,
                         int* iters, double* resid) {
  return cg_solve(c, K_ABSEXP, a, E, rtol, maxit, out_b, iters, resid);
}

int kmvp_invdist_minres_solve(kmvp_ctx* c, const void* a, int E, double rtol, int maxit, double* out_b,
                              int* iters, double* resid) {
  return minres_solve(c, K_INVDIST, a, E, rtol, maxit, out_b, iters, resid);
}

int kmvp_comm_get_unique_id(void* id128) {
  if (!id128) return KMVP_E_INVALID;
  if (!g_rccl.load()) return fail(nullptr, KMVP_E_COMM, g_rccl.error);
  static_assert(sizeof(ncclUniqueId) == KMVP_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(nullptr, KMVP_E_COMM, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r));
  memcpy(id128, &id, sizeof(id));
  return KMVP_OK;
}

int kmvp_comm_init(kmvp_ctx* c, const void* id128, int rank, int world) {
  if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, KMVP_E_INVALID, "bad communicator arguments");
  if (!g_rccl.load()) return fail(c, KMVP_E_COMM, g_rccl.error);
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->comm) {
    g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return fail(c, KMVP_E_COMM, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  c->rank = rank;
  c->world = world;
  return KMVP_OK;
}

int kmvp_set_option(kmvp_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return KMVP_E_INVALID;
  const std::string k(key);
  if (k == "feed") {
    if (value < -1 || value > 1) return fail(c, KMVP_E_INVALID, "feed must be -1 (auto), 0 or 1");
    c->opt_feed = (int)value;
  } else if (k == "targets_per_lane") {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
      return fail(c, KMVP_E_INVALID, "targets_per_lane must be 0 (auto), 1, 2, 4 or 8");
    c->opt_T = (int)value;
  } else if (k == "segments") {
    if (value < 0 || value > 65535) return fail(c, KMVP_E_INVALID, "segments out of range");
    c->opt_segments = (int)value;
  } else if (k == "fast_sqdists") {
    if (value < -1 || value > 1) return fail(c, KMVP_E_INVALID, "fast_sqdists must be -1 (auto), 0 or 1");
    c->opt_fast = (int)value;
  } else if (k == "fast_tiles") {
    if (value != 0 && value != 1 && value != 2 && value != 4)
      return fail(c, KMVP_E_INVALID, "fast_tiles must be 0 (auto), 1, 2 or 4");
    c->opt_fast_tiles = (int)value;
  } else if (k == "chunk") {
    if (value < 8 || value > (1 << 24)) return fail(c, KMVP_E_INVALID, "chunk out of range");
    c->opt_chunk = (int)value;
  } else {
    return fail(c, KMVP_E_INVALID, "unknown option " + k);
  }
  return KMVP_OK;
}

int64_t kmvp_device_bytes(const kmvp_ctx* c) {
  if (!c) return 0;
  size_t t = 0;
  for (const DevBuf* b : {&c->y_raw, &c->x_raw, &c->b_raw, &c->xs, &c->rec, &c->x_scaled,
                          &c->y_scaled, &c->part, &c->partd, &c->aux, &c->sums, &c->out, &c->scratch})
    t += b->cap;
  return (int64_t)t;
}
double kmvp_last_kernel_ms(const kmvp_ctx* c) { return c ? c->last_kernel_ms : 0.0; }
double kmvp_last_total_ms(const kmvp_ctx* c) { return c ? c->last_total_ms : 0.0; }
const char* kmvp_last_kernel_name(const kmvp_ctx* c) { return c ? c->last_kernel_name : ""; }

}  // extern "C"
