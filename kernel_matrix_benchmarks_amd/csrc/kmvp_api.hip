// libkmvp.so -- the extern "C" surface declared in include/kmvp.h.
//
// Owns the per-GPU context (device buffers, stream, events, RCCL communicator); the
// compute paths live in kmvp_product.hip and kmvp_solvers.hip.  Reference call order this
// serves: runner.py:70-148 (prepare_data, fit, prepare_query, query, get_result) through the
// plugin in kernel_matrix_benchmarks_amd/algorithms/mi355x.py.
#include "kmvp_ctx.hpp"

namespace kmvp {

Rccl g_rccl;

namespace {
thread_local std::string g_create_error;
}

int fail(kmvp_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  else g_create_error = msg;
  return code;
}
const char* create_error() { return g_create_error.c_str(); }

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

}  // namespace kmvp

using namespace kmvp;

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
  for (int i = 0; ok && i < 5; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
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
                    &c->part, &c->partd, &c->aux, &c->sortbuf, &c->perm, &c->sums, &c->out, &c->scratch, &c->xchg, &c->kexp, &c->kshift, &c->xchgk,
                    &c->cell_tperm, &c->cell_sperm, &c->cell_tgrp, &c->cell_sgrp, &c->cell_slot, &c->cell_tmeta, &c->cell_sums, &c->cell_skey, &c->cell_scentre, &c->cell_scale})
    release(*b);
  for (int i = 0; i < 5; ++i)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* kmvp_last_error(const kmvp_ctx* c) { return c ? c->err.c_str() : create_error(); }

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
  if ((rc = measure_clouds(c, dtype, M, N, D))) return rc;  // bounding box for the split-bf16 path
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

int kmvp_fit(kmvp_ctx* c, int kernel) {
  if (!c) return KMVP_E_INVALID;
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points has not been called");
  if (kernel < 0 || kernel > 2) return fail(c, KMVP_E_INVALID, "kernel must be 0, 1 or 2");
  HIP_TRY(c, hipSetDevice(c->device));
  return prepare_points(c, kernel);
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
int kmvp_expdot(kmvp_ctx* c) { return run_product(c, K_EXPDOT, false); }
int kmvp_expdot_norm(kmvp_ctx* c) { return run_product(c, K_EXPDOT, true); }

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
  c->host_xchg = nullptr;
  c->host_user = nullptr;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    return fail(c, KMVP_E_COMM, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  }
  // what RCCL itself believes: a communicator whose size or rank differs from what the caller asked
  // for would hand back partial sums as results (or hang in the first all-reduce)
  int count = -1, user_rank = -1, dev = -1;
  ncclResult_t r1 = g_rccl.CommCount(c->comm, &count);
  ncclResult_t r2 = g_rccl.CommUserRank(c->comm, &user_rank);
  (void)g_rccl.CommCuDevice(c->comm, &dev);  // reported in the message only
  if (r1 != ncclSuccess || r2 != ncclSuccess || count != world || user_rank != rank) {
    g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    c->comm_count = 1;
    c->rank = 0;
    c->world = 1;
    return fail(c, KMVP_E_COMM, "RCCL communicator mismatch: asked for rank " + std::to_string(rank) + " of " +
                                    std::to_string(world) + " on device " + std::to_string(c->device) + ", RCCL reports rank " +
                                    std::to_string(user_rank) + " of " + std::to_string(count) + " on device " +
                                    std::to_string(dev));
  }
  c->comm_count = count;
  c->rank = rank;
  c->world = world;
  return KMVP_OK;
}

int kmvp_comm_init_host(kmvp_ctx* c, kmvp_host_allreduce_fn fn, void* user, int rank, int world) {
  if (!c || !fn || world < 1 || rank < 0 || rank >= world) return fail(c, KMVP_E_INVALID, "bad communicator arguments");
  if (c->comm) {
    g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
  }
  c->host_xchg = fn;
  c->host_user = user;
  c->comm_count = world;
  c->rank = rank;
  c->world = world;
  return KMVP_OK;
}

int kmvp_comm_world(const kmvp_ctx* c) { return (c && c->exchanges()) ? c->comm_count : 1; }
int kmvp_comm_rank(const kmvp_ctx* c) { return (c && c->exchanges()) ? c->rank : 0; }

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
    if (value < -1 || value > 4) return fail(c, KMVP_E_INVALID, "fast_sqdists must be -1 (auto), 0, 1, 2, 3 or 4");
    c->opt_fast = (int)value;
  } else if (k == "cellmm_shape") {
    if (value < -1 || value > 1) return fail(c, KMVP_E_INVALID, "cellmm_shape must be -1 (by size), 0 (32x32x16) or 1 (16x16x32)");
    c->opt_cellmm_shape = (int)value;
  } else if (k == "mfma_variant") {
    if (value != -1 && value != 0 && value != 1 && value != 4 && value != 5)
      return fail(c, KMVP_E_INVALID, "mfma_variant must be -1 (by kernel), 0, 1, 4 or 5");
    c->opt_mfma_variant = (int)value;
  } else if (k == "same_points_global") {
    c->opt_same_global = value != 0;
  } else if (k == "partial_shard") {
    c->opt_partial = value != 0;
  } else if (k == "fast_tiles") {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
      return fail(c, KMVP_E_INVALID, "fast_tiles must be 0 (auto), 1, 2, 4 or 8");
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
                          &c->y_scaled, &c->part, &c->partd, &c->aux, &c->sortbuf, &c->perm, &c->sums, &c->out, &c->scratch, &c->xchg, &c->kexp, &c->kshift, &c->xchgk,
                          &c->cell_tperm, &c->cell_sperm, &c->cell_tgrp, &c->cell_sgrp, &c->cell_slot, &c->cell_tmeta, &c->cell_sums, &c->cell_skey, &c->cell_scentre, &c->cell_scale})
    t += b->cap;
  return (int64_t)t;
}
double kmvp_last_kernel_ms(const kmvp_ctx* c) { return c ? c->last_kernel_ms : 0.0; }
double kmvp_last_total_ms(const kmvp_ctx* c) { return c ? c->last_total_ms : 0.0; }
double kmvp_last_allreduce_ms(const kmvp_ctx* c) { return c ? c->last_allreduce_ms : 0.0; }
const char* kmvp_last_kernel_name(const kmvp_ctx* c) { return c ? c->last_kernel_name : ""; }
const char* kmvp_last_dispatch_note(const kmvp_ctx* c) { return c ? c->note.c_str() : ""; }

}  // extern "C"
