// Shared host-side state of libkmvp.so: the per-GPU context, small helpers and the
// functions the translation units call across each other.
//   kmvp_api.hip      the extern "C" surface of include/kmvp.h
//   kmvp_product.hip  layouts, launch geometry and the three pair-loop paths
//   kmvp_solvers.hip  conjugate gradients and MINRES on the product
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is dlopen'ed on first use
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/kmvp.h"
#include "kmvp_internal.hpp"

namespace kmvp {

// which path's layouts the shared xs / rec buffers hold
enum : int { LAYOUT_LOWD = 0, LAYOUT_FAST = 1, LAYOUT_MFMA = 2, LAYOUT_CFAST = 3, LAYOUT_CELL = 4, LAYOUT_CELL64 = 5, LAYOUT_CELLMM = 6, LAYOUT_FASTMM = 7, LAYOUT_CFASTMM = 8 };

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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
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
    CommCount = (decltype(CommCount))dlsym(handle, "ncclCommCount");
    CommUserRank = (decltype(CommUserRank))dlsym(handle, "ncclCommUserRank");
    CommCuDevice = (decltype(CommCuDevice))dlsym(handle, "ncclCommCuDevice");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy || !GetErrorString || !CommCount || !CommUserRank ||
        !CommCuDevice) {
      error = "librccl lacks an expected symbol";
      return false;
    }
    return true;
  }
};extern Rccl g_rccl;

}  // namespace kmvp

struct kmvp_ctx {
  using DevBuf = kmvp::DevBuf;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // product start, pair loop end, end; all-reduce start, end
  std::string err;

  // problem
  int dtype = -1;
  int D = 0, E = 0;
  int64_t M = 0, N = 0, j_offset = 0, m_total = 0;
  bool same_points = false;
  bool have_points = false, have_signal = false, density = false;
  bool async_product = false;   // solvers: run_product() leaves the result on the stream, no host synchronisation

  DevBuf y_raw, x_raw, b_raw;   // caller's arrays in the working precision
  DevBuf xs, rec;               // kernel layouts (specialised path; bf16 path: augmented targets, tile images)
  DevBuf partd;                 // bf16 path: partial denominators
  DevBuf aux;                   // cloud centre, squared radius, half-widths (fast_center_kernel)
  DevBuf sortbuf, perm;         // centred path: radix-sort scratch, Morton order of the sources
  DevBuf x_scaled, y_scaled;    // scaled copies (generic path)
  // cell-reduced Gaussian path (kmvp_cell.hpp): cell order of targets / sources, their tiles
  // ([start][count][key] per tile), slot of every target in cell order, target tile centres,
  // segment-reduced sums in cell order
  DevBuf cell_tperm, cell_sperm, cell_tgrp, cell_sgrp, cell_slot, cell_tmeta, cell_sums, cell_skey, cell_scentre, cell_scale;
  DevBuf part, sums, out;       // fp64 partials, reduced sums, final (N,E)
  DevBuf xchg;                  // sharded runs: sums in the canonical unpadded layout [column][N] for the all-reduce
  DevBuf kexp, kshift, xchgk;   // exp(<x,y>): exponents per (segment, target) / per target / per target after the all-reduce(min)
  DevBuf scratch;               // CG vectors / dot products
  uint64_t points_ver = 0, signal_ver = 0;
  // what xs / rec / scaled copies currently hold
  int packed_layout = -1;  // LAYOUT_* of the path that owns xs / rec right now
  int packed_kernel = -1, packed_sig = -1, packed_T = -1;
  uint64_t packed_points_ver = 0, packed_signal_ver = 0;
  int gen_kernel = -1;
  uint64_t gen_points_ver = 0;
  int64_t out_n = 0;
  int out_e = 0;

  // tuning (kmvp_set_option)
  int opt_feed = -1, opt_T = 0, opt_segments = 0, opt_chunk = 512;
  int opt_fast = -1, opt_fast_tiles = 0;  // fast_sqdists: -1 auto, 0 never, 1 always, 2 always the centred form, 3 always the cell form
  int opt_cellmm_shape = -1;              // cellmm_kernel's MFMA shape: 0 = 32x32x16, 1 = 16x16x32 (cellmm16_kernel), -1 = by size
  int opt_mfma_variant = -1;              // bf16 path: VAR of mfma_pipe_kernel (kmvp_mfma.hpp); -1 = by kernel
  int opt_same_global = 0;                // the targets ARE the (unsharded) sources although x was passed explicitly
  int opt_partial = 0;                    // a source slice (M < M_total) may run WITHOUT a communicator: the caller sums the shards
  uint64_t perm_ver = 0;                  // points version the Morton order belongs to
  uint64_t cell_ver = 0;                  // points version the cell structures belong to
  int cell_state = 0;                     // for cell_ver: 0 not examined, 1 built, -1 the path does not apply
  int cell_tt = 0, cell_tt_req = 0;       // target tiles per wavefront the target tile list was padded for; as requested (0 = auto)
  float cell_lo[3] = {0.f, 0.f, 0.f}, cell_hh[3] = {1.f, 1.f, 1.f};  // grid origin and cell sides per axis
  int cell_g[3] = {1, 1, 1};
  int64_t cell_n_tiles = 0, cell_m_tiles = 0;  // tiles of 32 (targets: both lists, padded to workgroups / sources)
  int64_t cell_n_main = 0, cell_n_rest = 0;    // float32 cell kernels: target tiles in groups of cell_tt / leftover tiles in groups of 2
  float cloud_radius2 = INFINITY;          // squared half-diagonal of the clouds' bounding box
  uint64_t centre_ver = 0;

  // sharding
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  // rehearsal transport (kmvp_comm_init_host): the same exchange staged through host memory and summed by a callback
  kmvp_host_allreduce_fn host_xchg = nullptr;
  void* host_user = nullptr;
  std::vector<double> host_buf;
  bool exchanges() const { return comm != nullptr || host_xchg != nullptr; }

  int comm_count = 1;                      // ranks the RCCL communicator itself reports (ncclCommCount)
  float last_kernel_ms = 0.f, last_total_ms = 0.f, last_allreduce_ms = 0.f;
  const char* last_kernel_name = "";
  std::string note;  // why the last product did not take a faster form ("" when it did): kmvp_last_dispatch_note
};

namespace kmvp {

int fail(kmvp_ctx* c, int code, const std::string& msg);
const char* create_error();

#define HIP_TRY(c, expr)                                                                     \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return kmvp::fail((c), e_ == hipErrorOutOfMemory ? KMVP_E_NOMEM : KMVP_E_DEVICE,       \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }
inline size_t elem_size(int dtype) { return dtype == KMVP_F64 ? 8 : 4; }
inline unsigned blocks_for(int64_t n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }

int ensure(kmvp_ctx* c, DevBuf& b, size_t bytes);  // grow-only device buffer
void release(DevBuf& b);

// kmvp_product.hip: the product a = K b [/ K 1] with everything query() times, and the
// bounding box of the freshly uploaded clouds (asynchronous on the context's stream)
int run_product(kmvp_ctx* c, int kernel, bool normalise);
int measure_clouds(kmvp_ctx* c, int dtype, int64_t M, int64_t N, int D);
int prepare_points(kmvp_ctx* c, int kernel);  // kmvp_fit: whatever can be built from the points alone

// kmvp_solvers.hip
int cg_solve(kmvp_ctx* c, int kernel, const void* a_host, int E, double rtol, int maxit, double* out_b,
             int* iters, double* resid);
int minres_solve(kmvp_ctx* c, int kernel, const void* a_host, int E, double rtol, int maxit,
                 double* out_b, int* iters, double* resid);

}  // namespace kmvp
