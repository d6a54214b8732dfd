/* kmvp.h -- C ABI of libkmvp.so, the MI355X (gfx950) kernel matrix-vector product
 * backend:   a_i = sum_j k(x_i, y_j) b_j   computed on the fly (no N x M matrix).
 *
 * The reference (kernel-matrix-benchmarks) has NO native interface: its plugin
 * boundary is the Python class API of
 *     kernel_matrix_benchmarks/algorithms/base.py:51-167   (BaseProduct / BaseSolver)
 * whose only computing implementation is
 *     kernel_matrix_benchmarks/algorithms/bruteforce.py:61-207.
 * This header is therefore what a ctypes binding of that class API binds; each
 * entry point cites the reference method it stands behind.  The Python plugin
 * (kernel_matrix_benchmarks_amd/algorithms/mi355x.py) is the reference-side
 * binding; INTEGRATION.md shows the stub a maintainer adds to the reference tree.
 *
 * Conventions
 *  - extern "C", plain C types only; no C++ exception crosses the boundary.
 *  - every int-returning function returns KMVP_OK (0) or a KMVP_E_* code; the
 *    message is available from kmvp_last_error(ctx) (ctx may be NULL for errors
 *    of kmvp_create).
 *  - host buffers are caller-owned, read during the call only and never
 *    modified (the runner reuses its numpy arrays across instances,
 *    runner.py:31-34,70-93); device buffers are ctx-owned.
 *  - a ctx is bound to ONE GPU and is not thread-safe (the reference's caller
 *    is single-threaded, main.py:303-308).  Nothing touches the GPU before
 *    kmvp_create (HIP contexts do not survive the runner's fork).
 *  - all compute entry points are synchronous: the result is complete in
 *    device memory when they return (runner.py:138-140 times query() with a
 *    wall clock).
 */
#ifndef KMVP_H
#define KMVP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMVP_ABI_VERSION 1

enum kmvp_status {
  KMVP_OK = 0,
  KMVP_E_INVALID = 1,     /* bad argument or call order                     */
  KMVP_E_UNSUPPORTED = 2, /* shape / dtype this build has no kernel for     */
  KMVP_E_DEVICE = 3,      /* HIP runtime error                              */
  KMVP_E_COMM = 4,        /* RCCL error                                     */
  KMVP_E_NOMEM = 5,
  KMVP_E_NOT_CONVERGED = 6
};

/* working precision of the arithmetic; base.py:9 `precision`, algos.yaml:157-162 */
enum kmvp_dtype {
  KMVP_F32 = 0, /* host arrays float32, fp32 VALU kernels (fp64 cross-chunk sums)   */
  KMVP_F64 = 1, /* host arrays float64, fp64 VALU kernels                           */
  KMVP_BF16 = 2 /* host arrays float32, bf16 MFMA tiles with fp32 accumulation
                   (high-D path, D >= 16 only)                                      */
};

typedef struct kmvp_ctx kmvp_ctx;

int kmvp_abi_version(void);
/* number of visible GPUs, or a negative kmvp_status; does not create a context */
int kmvp_device_count(void);

/* BaseAlgorithm.__init__ (base.py:8-29) -- binds the ctx to GPU `device`. */
kmvp_ctx* kmvp_create(int device, int* status);
/* BaseAlgorithm.done (base.py:31-33) -- frees every device buffer and the communicator. */
void kmvp_destroy(kmvp_ctx* ctx);
const char* kmvp_last_error(const kmvp_ctx* ctx);

/* BaseProduct.prepare_data (base.py:56-80, bruteforce.py:89-111) and
 * BaseSolver.prepare_data (base.py:124-133).
 *   y: (M,D) row-major source points of THIS shard, x: (N,D) target points or
 *   NULL for same_points (then N must equal M_total and, when sharded, x is the
 *   FULL cloud passed explicitly -- see below).
 *   dtype: element type of the host arrays / working precision (kmvp_dtype).
 *   j_offset, M_total: global index of y[0] and global number of sources; pass
 *   0 and M when the sources are not sharded.  They only matter for the
 *   inverse-distance kernel, whose zero pattern is defined on the GLOBAL flat
 *   index (bruteforce.py:13-14).
 * Uploads and re-lays-out the points (untimed in the harness). */
int kmvp_set_points(kmvp_ctx* ctx, const void* y, int64_t M, const void* x_or_null, int64_t N,
                    int D, int dtype, int64_t j_offset, int64_t M_total);

/* BaseAlgorithm.fit (base.py:84, bruteforce.py:113-120: the reference builds its kernel matrix
 * there, timed as build_time).  Nothing of the matrix is ever built here; what CAN be built from the
 * points alone is: for the Gaussian on clouds that qualify for the cell form (fast_sqdists 3 / auto),
 * the grid, the cell order (radix sort) and the tile lists.  kernel: 0 gaussian, 1 absexp, 2 invdist.
 * Optional: the first query does the same work when fit was not called. */
int kmvp_fit(kmvp_ctx* ctx, int kernel);

/* BaseProduct.prepare_query (base.py:86-98, bruteforce.py:122-128).
 *   b: (M,E) row-major signal of this shard in the dtype given to
 *   kmvp_set_points, or NULL for density estimation (b == 1, E must be 1). */
int kmvp_set_signal(kmvp_ctx* ctx, const void* b_or_null, int E);

/* BaseProduct.query (base.py:100-106, bruteforce.py:130-153): one entry point per
 * kernel x normalisation -- there is no kernel-selection branch in device code.
 *   kmvp_<kernel>       a = K b            (bruteforce.py:150,153)
 *   kmvp_<kernel>_norm  a = (K b) / (K 1)  (bruteforce.py:134-145)
 * kernels (bruteforce.py:18-22): gaussian exp(-s); absexp exp(-sqrt(s));
 * invdist 1/sqrt(s) with the flat-index diagonal zeroed (bruteforce.py:8-15).
 * With a communicator attached (kmvp_comm_init) the (N,E[+1]) partial sums of
 * all ranks are all-reduced over RCCL before the normalisation. */
int kmvp_gaussian(kmvp_ctx* ctx);
int kmvp_gaussian_norm(kmvp_ctx* ctx);
int kmvp_absexp(kmvp_ctx* ctx);
int kmvp_absexp_norm(kmvp_ctx* ctx);
int kmvp_invdist(kmvp_ctx* ctx);
int kmvp_invdist_norm(kmvp_ctx* ctx);
/* k(x, y) = exp(<x, y>): the attention kernel the reference's README defines (README.md:51-59; no reference plugin
 * computes it: parity unpinned).  float32 at D <= 64 or bfloat16 at D <= 141, E <= 128; a signal must be set (density
 * estimation: pass ones).  S = X Y^T comes straight from the matrix cores (float32: 3-way split bf16 operands, fp32
 * accuracy, kmvp_fastmm.hpp; bfloat16: plain bf16 operands and a bf16 second product, kmvp_mfma.hpp), the kernel values are
 * taken relative to a per-target running exponent (flash-attention recurrence with integer exponents) and leave the pair
 * loop as (mantissa sums, exponent) pairs, so that
 *   kmvp_expdot_norm  softmax attention  a_i = sum_j e^<x_i,y_j> b_j / sum_j e^<x_i,y_j>   has NO range limit on <x, y>;
 *   kmvp_expdot       plain product: overflows to inf exactly where exp(<x, y>) leaves the float64 range.
 * Sharded: all-reduce(min) of the exponents, then the usual all-reduce(sum).  Other dtypes / D: KMVP_E_UNSUPPORTED
 * (the plugin then uses the Gaussian identity, with its range check). */
int kmvp_expdot(kmvp_ctx* ctx);
int kmvp_expdot_norm(kmvp_ctx* ctx);

/* BaseProduct.get_result (base.py:107-116): (N,E) float64 row-major. */
int kmvp_get_result(kmvp_ctx* ctx, double* out, int64_t out_len);

/* BaseSolver.prepare_query + query (base.py:140-156, bruteforce.py:201-207): solves
 * K b = a on the point cloud given to kmvp_set_points (x_or_null == NULL) by
 * conjugate gradients with the on-the-fly product as operator.  The reference
 * uses a dense lstsq; parity is judged on the residual.
 *   a: (M,E) in the ctx dtype.  rtol: target ||K b - a|| / ||a|| (per column).
 *   out_b (M,E) float64 receives the iterate; *iters / *resid the iteration
 *   count and the worst final TRUE relative residual (from one more product).  Returns KMVP_OK
 *   only if that residual is finite and <= 1.5 rtol (the iteration stops on the recurrence residual;
 *   where the two drift apart it is restarted from the true one); otherwise KMVP_E_NOT_CONVERGED with
 *   out_b still written -- also for a non-finite operator or right-hand side. */
int kmvp_gaussian_cg_solve(kmvp_ctx* ctx, const void* a, int E, double rtol, int maxit,
                           double* out_b, int* iters, double* resid);
int kmvp_absexp_cg_solve(kmvp_ctx* ctx, const void* a, int E, double rtol, int maxit,
                         double* out_b, int* iters, double* resid);
/* Same contract for the inverse-distance kernel, whose matrix (zero diagonal,
 * bruteforce.py:13-14) is symmetric but INDEFINITE: MINRES instead of CG. */
int kmvp_invdist_minres_solve(kmvp_ctx* ctx, const void* a, int E, double rtol, int maxit,
                              double* out_b, int* iters, double* resid);

/* Source sharding over the GPUs of one node, one process per GPU:
 * rank 0 calls kmvp_comm_get_unique_id and hands the 128 bytes to every rank
 * out of band; every rank then calls kmvp_comm_init.  world == 1 is allowed. */
#define KMVP_UNIQUE_ID_BYTES 128
int kmvp_comm_get_unique_id(void* id128);
int kmvp_comm_init(kmvp_ctx* ctx, const void* id128, int rank, int world);
/* What the attached RCCL communicator ITSELF reports (ncclCommCount / ncclCommUserRank; kmvp_comm_init
 * fails with KMVP_E_COMM when they differ from what it was asked for).  1 / 0 without a communicator.
 * bench.py prints the count as "rccl_ranks" so that a multi-GPU line shows the collective really spanned
 * N ranks. */
int kmvp_comm_world(const kmvp_ctx* ctx);
int kmvp_comm_rank(const kmvp_ctx* ctx);
/* REHEARSAL transport for the same exchange (not the product: RCCL is).  RCCL refuses two ranks on one device,
 * so the multi-rank path of this library -- shard-local kernels with j_offset / M_total, the canonical unpadded
 * [column][N] layout, exchange, normalisation, the sharded solvers -- cannot be exercised with world > 1 on a
 * one-GPU box through kmvp_comm_init.  With this entry the all-reduce is staged through host memory instead:
 * `fn(user, buf, count, op)` must reduce `count` doubles in place over all ranks -- op = KMVP_OP_SUM, or KMVP_OP_MIN
 * (the exponents of the exp(<x,y>) path) -- e.g. by torch.distributed.all_reduce over gloo, and return 0.  Everything else -- what is exchanged, where in the stream, what happens before and
 * after -- is the code path kmvp_comm_init uses.  Never selected implicitly. */
enum kmvp_reduce_op { KMVP_OP_SUM = 0, KMVP_OP_MIN = 1 };
typedef int (*kmvp_host_allreduce_fn)(void* user, double* buf, int64_t count, int op);
int kmvp_comm_init_host(kmvp_ctx* ctx, kmvp_host_allreduce_fn fn, void* user, int rank, int world);

/* BaseAlgorithm.set_query_arguments (base.py:40-42): tuning knobs, all optional.
 *   "feed"             -1 = auto (default), 0 = scalar-cache source stream, 1 = LDS-staged tiles
 *   "targets_per_lane" 0 = auto (default), 1, 2, 4 or 8 (difference form); bf16 path: 1 or 2 target
 *                      tiles per wavefront without software pipelining, 0 = pipelined where instantiated
 *   "segments"         number of source segments a launch is split into (0 = auto)
 *   "chunk"            sources summed in fp32 before folding into the fp64 sum
 *   "fast_sqdists"     squared distances in the expanded form |x|^2+|y|^2-2x.y on the matrix
 *                      cores (bruteforce.py:36-49 `fast_sqdists`; float32):
 *                      1 = always, around one centre for the whole cloud: fast_kernel (E == 1, D <= 39); with several
 *                          signal columns, or beyond D = 39 (Gaussian, D <= 64; exp(-r) at 5 <= D <= 64 inside the
 *                          radius rule, closest pairs recomputed exactly): fastmm_kernel, where the tile of kernel values goes
 *                          back to the matrix cores for the product with the signal, up to 32 columns per pass
 *                          (the denominator of normalised rows is one more column);
 *                      2 = always, around per-group centres of Morton-sorted sources with exact
 *                          recomputation of the closest pairs (cfast_kernel, D <= 4; with several signal
 *                          columns, Gaussian and exp(-r): cfastmm_kernel, the second product of fastmm_kernel
 *                          on these distances);
 *                      3 = always the cell form (Gaussian, D <= 3): exp() is range-reduced by the cells
 *                          of a regular grid, exp(-|x-y|^2) = U_i(S) W_j(T) exp(2 d.e), and the remainder
 *                          polynomial 1 + t + t^2/2 of t = 2 d.e (|t| <= 0.016) comes out of one MFMA per
 *                          32 x 32 pairs -- cellmm_kernel: f16 MFMA that also carries the weights W_j b_j
 *                          and the sum over the sources (any E: one launch per signal column; normalised
 *                          rows: one more with b = 1), on clouds inside the radius rule; cell_kernel
 *                          (bf16 MFMA for the polynomial, one VALU fma per pair; E == 1) otherwise;
 *                          float64 (cell64_kernel, E == 1): the degree-7 polynomial on the VALU;
 *                      4 = as 3 but always cell_kernel (float32);
 *                      0 = never (difference form, bruteforce.py:53-54);
 *                      -1 = auto (default): the cheapest form that is as accurate as the difference form for the
 *                          shape at hand -- which one that is: docs/DISPATCH.md (generated by asking the library),
 *                          the rules themselves: csrc/kmvp_product.hip run_product()
 *   "same_points_global" 1 when the targets passed to kmvp_set_points are the unsharded
 *                      sources (sharded same_points): enables form 2 for inverse-distance
 *   "partial_shard"    1: a source slice (M < M_total) may be run WITHOUT a multi-rank communicator and
 *                      returns that shard's partial sums (the caller adds the shards up); default 0: such
 *                      a call fails with KMVP_E_INVALID instead of passing partial sums off as the product
 *   "fast_tiles"       target tiles of 32 per wavefront in that kernel: 0 = auto, 1, 2, 4, 8
 *                      (clamped to what is instantiated: fast_kernel 4 up to D = 7, 2 up to D = 23,
 *                      1 beyond; cfast_kernel 4; cell_kernel and cellmm_kernel 8) */
/*   "cellmm_shape"     MFMA shape of the float32 cell form: -1 = by size (default: 16x16x32 from 5e5 targets and 1e5 sources
 *                      on with eight target tiles per wave, else 32x32x16), 0 = 32x32x16 (cellmm_kernel), 1 = 16x16x32
 *                      (cellmm16_kernel)
 *   "mfma_variant"     bf16 path, software-pipelined kernel: -1 = by kernel (default: exp(-r) 4, others 0), 0 = plain,
 *                      1 = denominators on the matrix pipe (one more accumulator tile per target tile), 4 = loop rotated by one
 *                      transcendental stage, 5 = both (csrc/kmvp_mfma.hpp, mfma_pipe_kernel VAR) */
int kmvp_set_option(kmvp_ctx* ctx, const char* key, int64_t value);

/* BaseAlgorithm.get_memory_usage / get_additional (base.py:35-46): bytes of device
 * memory the ctx holds, and HIP-event timings of the last compute call:
 *   kmvp_last_kernel_ms  the dominant (pair-loop) kernel alone
 *   kmvp_last_total_ms   every launch of the call incl. reduction + all-reduce */
int64_t kmvp_device_bytes(const kmvp_ctx* ctx);
double kmvp_last_kernel_ms(const kmvp_ctx* ctx);
double kmvp_last_total_ms(const kmvp_ctx* ctx);
/*   kmvp_last_allreduce_ms  the RCCL all-reduce of the (N,E[+1]) sums alone (0 without a communicator) */
double kmvp_last_allreduce_ms(const kmvp_ctx* ctx);
/* name of the pair-loop kernel the last compute call launched (for rocprof matching) */
const char* kmvp_last_kernel_name(const kmvp_ctx* ctx);
/* The fastest forms are narrow (kmvp_set_option "fast_sqdists").  When the last product did NOT take the cell form
 * although kernel, precision and dimension would allow it, this says which condition failed (too few points per
 * cloud, too few points per grid cell, the radius rule, ...); "" when nothing faster applied.  get_additional()
 * stores it as "dispatch_note". */
const char* kmvp_last_dispatch_note(const kmvp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* KMVP_H */
