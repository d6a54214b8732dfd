// Krylov solvers for K b = a with the on-the-fly product as the operator: conjugate
// gradients (SPD Gaussian / exp(-r) matrices) and MINRES (symmetric indefinite
// inverse-distance matrix).  The reference solves densely with lstsq (bruteforce.py:205-207);
// parity is judged on the residual.
#include <cstdlib>

#include "kmvp_ctx.hpp"

namespace kmvp {

// ------------------------------------------------------------------------------------
// conjugate gradients on K b = a with the on-the-fly product as the operator

constexpr int CG_BLOCKS = 256;

// The system matrix is K(y, y) on ALL points.  Single GPU: x == y (x_or_null = NULL).  Sharded:
// every rank holds all points as targets and its slice of them as sources
// (same_points_global, N == M_total), and a communicator is attached.
static int solver_shape(kmvp_ctx* c) {
  if (c->same_points && c->world == 1) return KMVP_OK;
  if (c->world > 1 && !c->exchanges()) return fail(c, KMVP_E_INVALID, "sharded solver without kmvp_comm_init");
  const bool sharded_square = c->opt_same_global && c->N == c->m_total && c->j_offset + c->M <= c->m_total &&
                             (c->world > 1 || c->M == c->m_total);  // one rank must hold every source
  if (c->same_points && c->world > 1)
    return fail(c, KMVP_E_INVALID, "sharded solver: pass all points as targets and this rank's slice as sources");
  if (!sharded_square) return fail(c, KMVP_E_INVALID, "the solver needs x == y (pass x_or_null = NULL)");
  return KMVP_OK;
}

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

// Coefficients of a vector update passed BY VALUE in the kernel argument block (E <= 8): no
// host-to-device copy and no stream synchronisation per update (a small solve is otherwise
// dominated by them: seven synchronisations per CG iteration instead of three).
constexpr int COEF_INLINE_E = 8;
struct Coef8 { double v[COEF_INLINE_E]; };
struct Coef24 { double v[3 * COEF_INLINE_E]; };

__global__ void cg_axpy_inline_kernel(double* __restrict__ out, const double* __restrict__ u,
                                      const double* __restrict__ v, const Coef8 coef, int64_t m, int E) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  out[q] = u[q] + coef.v[q % E] * v[q];
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

// Sum of the CG_BLOCKS partials of column e by the whole block (fixed tree order: deterministic).
// A single thread walking the 256 partials took ~9 us per call; this takes ~2.
__device__ __forceinline__ double block_sum_partials(const double* __restrict__ partial, int E, int e, double* red) {
  red[threadIdx.x] = threadIdx.x < CG_BLOCKS ? partial[(int64_t)threadIdx.x * E + e] : 0.0;
  __syncthreads();
  for (int s2 = CG_BLOCKS / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  const double v = red[0];
  __syncthreads();
  return v;
}

constexpr int CG_CHECK = 8;  // iterations between two looks at the residual on the host
constexpr int CG_GRAPH_AFTER = 64;  // bursts (512 iterations) before the burst is captured into a hipGraph

// One block.  mode 1: pAp[e] = sum of the partials -> alpha[e] = rs_old[e] / pAp[e];
// mode 2: rs_new[e] = sum -> beta[e] = rs_new[e] / rs_old[e], rs_old[e] = rs_new[e], iteration count + 1,
// and the stopping test max_e sqrt(rs_new / |a|^2) <= rtol: once it holds, `stop` is set and every
// later kernel of the burst returns at once, so the iterate the host reads is the one of the FIRST
// iteration that met the tolerance (the residual of an ill-conditioned system does not fall
// monotonically: at config 5 it meets 1e-6 at iteration 132 and not again before 240).
// scal = [rs_old | alpha | beta | |a|^2] x E, then stop, iterations.  Partials are added in block
// order, as the host would.
__global__ void __launch_bounds__(CG_BLOCKS) cg_scalars_kernel(const double* __restrict__ partial,
                                                              double* __restrict__ scal, int E, int mode,
                                                              double rtol) {
  __shared__ double red[CG_BLOCKS];
  double* stop = scal + 4 * E;
  if (*stop != 0.0) return;  // uniform: every thread reads the same word
  bool not_met = false;
  for (int e = 0; e < E; ++e) {
    const double v = block_sum_partials(partial, E, e, red);
    if (threadIdx.x != 0) continue;
    const double rs_old = scal[e];
    if (mode == 1) {
      scal[E + e] = (v != 0.0 && rs_old > 0.0) ? rs_old / v : 0.0;
    } else {
      scal[2 * E + e] = rs_old > 0.0 ? v / rs_old : 0.0;
      scal[e] = v;
      const double a2 = scal[3 * E + e];
      if (a2 > 0.0 && !(sqrt(v / a2) <= rtol)) not_met = true;  // NaN / inf: not met
    }
  }
  if (mode == 2 && threadIdx.x == 0) {
    scal[4 * E + 1] += 1.0;
    if (!not_met) *stop = 1.0;
  }
}

// x += alpha p ; r -= alpha Ap
__global__ void cg_update_xr_kernel(double* __restrict__ x, double* __restrict__ r, const double* __restrict__ p,
                                    const double* __restrict__ Ap, const double* __restrict__ scal, int64_t m,
                                    int E) {
  if (scal[4 * E] != 0.0) return;  // stopped
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  const double a = scal[E + q % E];
  x[q] = x[q] + a * p[q];
  r[q] = r[q] + (-a) * Ap[q];
}

// p = r + beta p
__global__ void cg_update_p_kernel(double* __restrict__ p, const double* __restrict__ r,
                                   const double* __restrict__ scal, int64_t m, int E) {
  if (scal[4 * E] != 0.0) return;  // stopped
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  p[q] = r[q] + scal[2 * E + q % E] * p[q];
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
  if (E <= COEF_INLINE_E) {
    Coef8 k;
    for (int e = 0; e < COEF_INLINE_E; ++e) k.v[e] = e < E ? coef[e] : 0.0;
    hipLaunchKernelGGL(cg_axpy_inline_kernel, dim3(blocks_for(m * E)), dim3(256), 0, c->stream, out, u, v, k, m, E);
    HIP_TRY(c, hipGetLastError());
    return KMVP_OK;
  }
  HIP_TRY(c, hipMemcpyAsync(w.coef, coef.data(), sizeof(double) * E, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // coef is a host temporary
  hipLaunchKernelGGL(cg_axpy_kernel, dim3(blocks_for(m * E)), dim3(256), 0, c->stream, out, u, v,
                     w.coef, m, E);
  HIP_TRY(c, hipGetLastError());
  return KMVP_OK;
}

// K applied to the device vector v (n,E) double, n = all points; the result lands in c->out
// (n,E) double.  With source sharding the Krylov vectors are replicated on every
// rank, the operator is sharded: this rank's signal is its own slice v[j_offset .. j_offset+M)
// and run_product() ends with the all-reduce of the (n,E) sums, so every rank continues with
// bitwise the same vectors.
int cg_apply(kmvp_ctx* c, int kernel, const double* v, int64_t n, int E) {
  const int64_t m = c->M;  // sources of this rank
  (void)n;
  v += (size_t)c->j_offset * E;
  int rc = ensure(c, c->b_raw, (size_t)n * E * elem_size(c->dtype));  // also holds the right-hand side
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
  if (int rc0 = solver_shape(c)) return rc0;
  if (!a_host || !out_b || E < 1 || maxit < 0 || !(rtol > 0)) return fail(c, KMVP_E_INVALID, "bad solver arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t m = c->N;  // length of the Krylov vectors: all points
  const size_t vec = (size_t)m * E * sizeof(double);
  int rc = ensure(c, c->scratch, 3 * vec + sizeof(double) * ((CG_BLOCKS + 1 + 4) * (size_t)E + 2));
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
    for (int e = 0; e < E; ++e) {
      const double v = anorm2[e] > 0 ? std::sqrt(r2[e] / anorm2[e]) : (anorm2[e] == 0 ? 0.0 : NAN);
      if (!(v <= wv)) wv = v;  // a NaN column makes the whole verdict NaN (std::max would drop it)
    }
    return wv;
  };

  // ---- the iteration lives on the device: the step lengths are computed by one-block kernels from
  // the dot products' partial sums (same additions in the same order as the host would do) and the
  // vector updates read them from device memory, so an iteration is a sequence of launches with no
  // host synchronisation; the host looks at the residual every CG_CHECK iterations only.
  double* scal = w.partial + (size_t)CG_BLOCKS * E + E;  // [rs_old | alpha | beta | |a|^2] x E, stop, iterations
  {
    std::vector<double> init((size_t)4 * E + 2, 0.0);
    for (int e = 0; e < E; ++e) {
      init[e] = rs[e];
      init[3 * E + e] = anorm2[e];
    }
    HIP_TRY(c, hipMemcpyAsync(scal, init.data(), sizeof(double) * init.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  const unsigned vblocks = blocks_for(m * E);
  std::vector<double> state((size_t)4 * E + 2);
  int it = 0;
  double rel = worst(rs);
  c->async_product = true;
  // one burst = CG_CHECK iterations of ~10 launches each.  A solve that is still running after
  // CG_GRAPH_AFTER bursts replays the burst as a hipGraph from then on (instantiating the 80-node
  // graph costs ~70 ms on this stack, so short solves never pay for it; single GPU only: with a
  // communicator the all-reduce stays out of graphs).  KMVP_NO_GRAPH=1 disables it.
  auto run_burst = [&](int burst) -> int {
    for (int k = 0; k < burst; ++k) {
      int rcb = cg_apply(c, kernel, w.p, m, E);
      if (rcb) return rcb;
      const double* Ap = (const double*)c->out.p;
      hipLaunchKernelGGL(cg_dot_kernel, dim3(CG_BLOCKS), dim3(256), 0, c->stream, w.p, Ap, m, E, w.partial);
      hipLaunchKernelGGL(cg_scalars_kernel, dim3(1), dim3(CG_BLOCKS), 0, c->stream, w.partial, scal, E, 1, rtol);
      hipLaunchKernelGGL(cg_update_xr_kernel, dim3(vblocks), dim3(256), 0, c->stream, w.x, w.r, w.p, Ap, scal, m, E);
      hipLaunchKernelGGL(cg_dot_kernel, dim3(CG_BLOCKS), dim3(256), 0, c->stream, w.r, w.r, m, E, w.partial);
      hipLaunchKernelGGL(cg_scalars_kernel, dim3(1), dim3(CG_BLOCKS), 0, c->stream, w.partial, scal, E, 2, rtol);
      hipLaunchKernelGGL(cg_update_p_kernel, dim3(vblocks), dim3(256), 0, c->stream, w.p, w.r, scal, m, E);
    }
    return KMVP_OK;
  };
  hipGraphExec_t gexec = nullptr;
  bool try_graph = !c->exchanges() && getenv("KMVP_NO_GRAPH") == nullptr;
  int full_bursts = 0;
  double true_rel = NAN, prev_true = INFINITY;
  // The iteration stops on the RECURRENCE residual; the verdict is on the TRUE one, a - K x, from one more
  // product.  Where the two have drifted apart (float32 operator, ill-conditioned Gaussian matrices) the
  // recurrence is restarted from the true residual (r = p = a - K x: "residual replacement"), at most
  // CG_MAX_RESTARTS times and only while that still halves the true residual.
  constexpr int CG_MAX_RESTARTS = 3;
  for (int pass = 0;; ++pass) {
    while (it < maxit && rel > rtol) {
      const int burst = std::min(CG_CHECK, maxit - it);
      if (gexec && burst == CG_CHECK) {
        if (hipGraphLaunch(gexec, c->stream) != hipSuccess) rc = fail(c, KMVP_E_DEVICE, "hipGraphLaunch failed");
      } else if (try_graph && full_bursts >= CG_GRAPH_AFTER && burst == CG_CHECK) {
        // every buffer exists and every layout decision has been taken by the first burst: capture
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          rc = run_burst(burst);
          hipGraph_t graph = nullptr;
          const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
          if (rc == KMVP_OK && ee == hipSuccess && graph &&
              hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) != hipSuccess)
            gexec = nullptr;
          if (graph) (void)hipGraphDestroy(graph);
          if (rc == KMVP_OK) {
            if (gexec) {
              if (hipGraphLaunch(gexec, c->stream) != hipSuccess) rc = fail(c, KMVP_E_DEVICE, "hipGraphLaunch failed");
            } else {
              (void)hipGetLastError();  // capture refused: nothing ran, go on launch by launch
              try_graph = false;
              rc = run_burst(burst);
            }
          }
        } else {
          (void)hipGetLastError();
          try_graph = false;
          rc = run_burst(burst);
        }
      } else {
        rc = run_burst(burst);
      }
      if (burst == CG_CHECK) ++full_bursts;
      if (rc) {
        c->async_product = false;
        if (gexec) (void)hipGraphExecDestroy(gexec);
        return rc;
      }
      hipError_t le = hipGetLastError();
      if (le == hipSuccess)
        le = hipMemcpyAsync(state.data(), scal, sizeof(double) * state.size(), hipMemcpyDeviceToHost, c->stream);
      if (le == hipSuccess) le = hipStreamSynchronize(c->stream);
      if (le != hipSuccess) {
        c->async_product = false;
        if (gexec) (void)hipGraphExecDestroy(gexec);
        HIP_TRY(c, le);
      }
      for (int e = 0; e < E; ++e) rs[e] = state[e];
      it = (int)state[(size_t)4 * E + 1];  // iterations that changed the iterate
      rel = worst(rs);
      if (state[(size_t)4 * E] != 0.0) break;  // the device met the tolerance inside the burst
    }
    c->async_product = false;

    // true residual ||a - K x|| / ||a|| with one more product: w.p = a (widened again) - K x
    rc = cg_apply(c, kernel, w.x, m, E);
    hipError_t he = hipSuccess;
    if (!rc) he = hipMemcpyAsync(c->b_raw.p, a_host, (size_t)m * E * elem_size(c->dtype), hipMemcpyHostToDevice, c->stream);
    if (!rc && he == hipSuccess) {
      if (c->dtype == KMVP_F64)
        hipLaunchKernelGGL((cg_widen_kernel<double>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream,
                           (const double*)c->b_raw.p, w.p, m * E);
      else
        hipLaunchKernelGGL((cg_widen_kernel<float>), dim3(blocks_for(m * E)), dim3(256), 0, c->stream,
                           (const float*)c->b_raw.p, w.p, m * E);
      he = hipGetLastError();
    }
    if (!rc && he == hipSuccess) {
      for (int e = 0; e < E; ++e) coef[e] = -1.0;
      rc = cg_axpy(c, w.p, w.p, (const double*)c->out.p, coef, m, E, w);
      if (!rc) rc = cg_dots(c, w.p, w.p, m, E, w, hp, rs_new);
    }
    if (rc || he != hipSuccess) {
      if (gexec) (void)hipGraphExecDestroy(gexec);
      if (rc) return rc;
      HIP_TRY(c, he);
    }
    true_rel = worst(rs_new);
    const bool met = std::isfinite(true_rel) && true_rel <= rtol * 1.5;
    if (met || !std::isfinite(true_rel) || it >= maxit || pass >= CG_MAX_RESTARTS || !(true_rel <= 0.5 * prev_true)) break;
    // restart from the true residual: r = p = a - K x, rs_old = |r|^2, stop flag cleared (the count goes on)
    prev_true = true_rel;
    HIP_TRY(c, hipMemcpyAsync(w.r, w.p, vec, hipMemcpyDeviceToDevice, c->stream));
    rs = rs_new;
    {
      std::vector<double> head((size_t)E);
      for (int e = 0; e < E; ++e) head[e] = rs[e];
      const double zero = 0.0;
      HIP_TRY(c, hipMemcpyAsync(scal, head.data(), sizeof(double) * E, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipMemcpyAsync(scal + 4 * (size_t)E, &zero, sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    rel = true_rel;
    c->async_product = true;
  }
  if (gexec) (void)hipGraphExecDestroy(gexec);

  HIP_TRY(c, hipMemcpyAsync(out_b, w.x, vec, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_signal = false;  // b_raw was used as scratch
  if (iters) *iters = it;
  if (resid) *resid = true_rel;
  // the verdict is on the TRUE residual (include/kmvp.h), with 1.5x slack for the rounding between it and the
  // recurrence; a non-finite residual (non-finite operator or right-hand side) is never a success
  if (!(std::isfinite(true_rel) && true_rel <= rtol * 1.5)) {
    c->err = std::isfinite(true_rel) ? "conjugate gradients stopped before the true residual reached the requested tolerance"
                                     : "conjugate gradients: the residual is not finite (non-finite operator or right-hand side)";
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

__global__ void vec_lin3_inline_kernel(double* __restrict__ out, const double* __restrict__ a,
                                       const double* __restrict__ b, const double* __restrict__ c,
                                       const Coef24 coef, int64_t m, int E) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  const int e = (int)(q % E);
  double v = coef.v[e] * a[q];
  if (b) v += coef.v[E + e] * b[q];
  if (c) v += coef.v[2 * E + e] * c[q];
  out[q] = v;
}

// ---- MINRES on the device.  State per column e (doubles): see MS_* below; five coefficient
// triples [3E] feed vec_lin3_dev_kernel.  Same scalar arithmetic, in the same order, as the
// textbook recurrence on the host would do.
enum : int { MS_BETA1 = 0, MS_BETA, MS_OLDB, MS_ALFA, MS_DBAR, MS_EPSLN, MS_CS, MS_SN, MS_PHIBAR, MS_DONE, MS_FIELDS };
// layout: state[MS_FIELDS][E] | T0..T4 [5][3E] | stop | iterations
__host__ __device__ inline size_t ms_triple(int E, int t) { return (size_t)MS_FIELDS * E + (size_t)t * 3 * E; }
__host__ __device__ inline size_t ms_stop(int E) { return (size_t)MS_FIELDS * E + 15 * (size_t)E; }

// out = c0 a + c1 b + c2 c with the coefficient triple in device memory; nothing once stopped.  out2 (or nullptr): a second
// copy of the result -- it may be `a` itself (each thread reads its element before it writes it)
__global__ void vec_lin3_dev_kernel(double* __restrict__ out, const double* a,
                                    const double* __restrict__ b, const double* __restrict__ c,
                                    const double* __restrict__ coef, const double* __restrict__ stop, int64_t m,
                                    int E, double* out2) {
  if (*stop != 0.0) return;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  const int e = (int)(q % E);
  double v = coef[e] * a[q];
  if (b) v += coef[E + e] * b[q];
  if (c) v += coef[2 * E + e] * c[q];
  out[q] = v;
  if (out2) out2[q] = v;
}

// The three vector updates that close a MINRES iteration, in one launch (each was a ~4.5 us kernel of its own in a solve whose
// iteration is a chain of such kernels): w = T3 . (v, w1, w2);  x = T4 . (x, w);  v = T0 y  (the NEXT iteration's Lanczos vector;
// v is read for w before it is overwritten, element by element).  Same expressions, same order as vec_lin3_dev_kernel.
__global__ void minres_tail_kernel(double* __restrict__ w, double* __restrict__ v, const double* __restrict__ w1,
                                   const double* __restrict__ w2, double* __restrict__ x, const double* __restrict__ y,
                                   const double* __restrict__ T3, const double* __restrict__ T4, const double* __restrict__ T0,
                                   const double* __restrict__ stop, int64_t m, int E) {
  if (*stop != 0.0) return;
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m * E) return;
  const int e = (int)(q % E);
  double wv = T3[e] * v[q];
  wv += T3[E + e] * w1[q];
  wv += T3[2 * E + e] * w2[q];
  w[q] = wv;
  double xv = T4[e] * x[q];
  xv += T4[E + e] * wv;
  x[q] = xv;
  v[q] = T0[e] * y[q];
}

// mode 1 (after v.y): alfa, T2 = (1, -alfa/beta, 0).
// mode 2 (after r2.r2): the Givens step, T3 (w update), T4 (x update), the done flags, the stopping
// test max_e phibar/beta1 <= rtol, and the next iteration's T0 = (1/beta, 0, 0), T1 = (1, -beta/oldb, 0).
__global__ void __launch_bounds__(CG_BLOCKS) minres_scalars_kernel(const double* __restrict__ partial,
                                                                  double* __restrict__ st, int E, int mode,
                                                                  double rtol) {
#pragma clang fp contract(off)  // the rotation exactly as written (no fused multiply-adds)
  __shared__ double red[CG_BLOCKS];
  double* stop = st + ms_stop(E);
  if (*stop != 0.0) return;
  bool not_met = false;
  double* T0 = st + ms_triple(E, 0);
  double* T1 = st + ms_triple(E, 1);
  double* T2 = st + ms_triple(E, 2);
  double* T3 = st + ms_triple(E, 3);
  double* T4 = st + ms_triple(E, 4);
  for (int e = 0; e < E; ++e) {
    const double dot = block_sum_partials(partial, E, e, red);
    if (threadIdx.x != 0) continue;
    auto S = [&](int f) -> double& { return st[(size_t)f * E + e]; };
    if (mode == 1) {
      S(MS_ALFA) = dot;
      T2[e] = 1.0;
      T2[E + e] = S(MS_BETA) > 0.0 ? -dot / S(MS_BETA) : 0.0;
      T2[2 * E + e] = 0.0;
    } else {
      const double alfa = S(MS_ALFA), cs0 = S(MS_CS), sn0 = S(MS_SN), dbar0 = S(MS_DBAR);
      const double oldb = S(MS_BETA);
      const double beta = sqrt(fmax(dot, 0.0));
      const double oldeps = S(MS_EPSLN);
      const double delta = cs0 * dbar0 + sn0 * alfa;
      const double gbar = sn0 * dbar0 - cs0 * alfa;
      const double epsln = sn0 * beta;
      const double dbar = -cs0 * beta;
      const double gamma = fmax(sqrt(gbar * gbar + beta * beta), 1e-300);
      const double cs = gbar / gamma, sn = beta / gamma;
      const double phi = cs * S(MS_PHIBAR);
      const double phibar = sn * S(MS_PHIBAR);
      const bool done0 = S(MS_DONE) != 0.0;
      const double dn = done0 ? 0.0 : 1.0 / gamma;
      T3[e] = dn;
      T3[E + e] = -oldeps * dn;
      T3[2 * E + e] = -delta * dn;
      T4[e] = 1.0;
      T4[E + e] = done0 ? 0.0 : phi;
      T4[2 * E + e] = 0.0;
      S(MS_OLDB) = oldb;
      S(MS_BETA) = beta;
      S(MS_EPSLN) = epsln;
      S(MS_DBAR) = dbar;
      S(MS_CS) = cs;
      S(MS_SN) = sn;
      S(MS_PHIBAR) = phibar;
      const bool done1 = done0 || phibar <= rtol * S(MS_BETA1) || beta == 0.0;
      S(MS_DONE) = done1 ? 1.0 : 0.0;
      // next iteration
      T0[e] = (!done1 && beta > 0.0) ? 1.0 / beta : 0.0;
      T0[E + e] = 0.0;
      T0[2 * E + e] = 0.0;
      T1[e] = 1.0;
      T1[E + e] = oldb > 0.0 ? -beta / oldb : 0.0;
      T1[2 * E + e] = 0.0;
      if (S(MS_BETA1) > 0.0 && !(phibar / S(MS_BETA1) <= rtol)) not_met = true;  // NaN / inf: not met
    }
  }
  if (mode == 2 && threadIdx.x == 0) {
    stop[1] += 1.0;
    if (!not_met) *stop = 1.0;
  }
}

int minres_solve(kmvp_ctx* c, int kernel, const void* a_host, int E, double rtol, int maxit,
                 double* out_b, int* iters, double* resid) {
  if (!c) return KMVP_E_INVALID;
  if (!c->have_points) return fail(c, KMVP_E_INVALID, "kmvp_set_points has not been called");
  if (int rc0 = solver_shape(c)) return rc0;
  if (!a_host || !out_b || E < 1 || maxit < 0 || !(rtol > 0)) return fail(c, KMVP_E_INVALID, "bad solver arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const int64_t m = c->N;  // length of the Krylov vectors: all points
  const size_t n = (size_t)m * E;
  const size_t vec = n * sizeof(double);
  int rc = ensure(c, c->scratch, 8 * vec + sizeof(double) * ((CG_BLOCKS + 3) * (size_t)E + ms_stop(E) + 2));
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
    if (E <= COEF_INLINE_E) {
      Coef24 k;
      for (int q = 0; q < 3 * COEF_INLINE_E; ++q) k.v[q] = q < 3 * E ? coef[q] : 0.0;
      hipLaunchKernelGGL(vec_lin3_inline_kernel, dim3(blocks_for((int64_t)n)), dim3(256), 0, c->stream, out, pa, pb,
                         pc, k, m, E);
      HIP_TRY(c, hipGetLastError());
      return KMVP_OK;
    }
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

  // ---- device-resident recurrence (see minres_scalars_kernel); the host rotates buffer pointers and
  // looks at the residual every CG_CHECK iterations
  std::vector<double> beta1(E);
  double* st = wk.coef + 3 * (size_t)E;
  const size_t st_len = ms_stop(E) + 2;
  std::vector<double> state(st_len, 0.0);
  for (int e = 0; e < E; ++e) {
    beta1[e] = std::sqrt(dots[e]);
    state[(size_t)MS_BETA1 * E + e] = beta1[e];
    state[(size_t)MS_BETA * E + e] = beta1[e];
    state[(size_t)MS_CS * E + e] = -1.0;
    state[(size_t)MS_PHIBAR * E + e] = beta1[e];
    const bool zero_rhs = !(beta1[e] > 0.0);  // x = 0
    state[(size_t)MS_DONE * E + e] = zero_rhs ? 1.0 : 0.0;
    state[ms_triple(E, 0) + e] = (!zero_rhs) ? 1.0 / beta1[e] : 0.0;  // T0 = (1/beta, 0, 0)
    state[ms_triple(E, 1) + e] = 1.0;                                   // T1 = (1, 0, 0): no r1 term in iteration 1
  }
  HIP_TRY(c, hipMemcpyAsync(st, state.data(), sizeof(double) * st_len, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const double* stop = st + ms_stop(E);
  const unsigned vb = blocks_for((int64_t)n);
  auto worst = [&]() {
    double wv = 0.0;
    for (int e = 0; e < E; ++e) {
      if (beta1[e] == 0.0) continue;  // zero right-hand side: x = 0
      const double v = state[(size_t)MS_PHIBAR * E + e] / beta1[e];
      if (!(v <= wv)) wv = v;  // NaN propagates
    }
    return wv;
  };
  auto dlin3 = [&](double* out, const double* pa, const double* pb, const double* pc, int triple) {
    hipLaunchKernelGGL(vec_lin3_dev_kernel, dim3(vb), dim3(256), 0, c->stream, out, pa, pb, pc,
                       st + ms_triple(E, triple), stop, m, E, (double*)nullptr);
  };

  int it = 0;
  double rel = worst();
  c->async_product = true;
  dlin3(v, y, nullptr, nullptr, 0);  // v = y / beta of the first iteration
  while (it < maxit && rel > rtol) {
    const int burst = std::min(CG_CHECK, maxit - it);
    for (int k = 0; k < burst; ++k) {
      // (v = y / beta was written by the previous iteration's tail, or before the loop)
      if ((rc = cg_apply(c, kernel, v, m, E))) {
        c->async_product = false;
        return rc;
      }
      dlin3(y, (const double*)c->out.p, r1, nullptr, 1);  // y = K v - (beta / oldb) r1
      hipLaunchKernelGGL(cg_dot_kernel, dim3(CG_BLOCKS), dim3(256), 0, c->stream, v, y, m, E, wk.partial);
      hipLaunchKernelGGL(minres_scalars_kernel, dim3(1), dim3(CG_BLOCKS), 0, c->stream, wk.partial, st, E, 1, rtol);
      // y - (alfa / beta) r2, written into the old r1 buffer AND back into y
      hipLaunchKernelGGL(vec_lin3_dev_kernel, dim3(vb), dim3(256), 0, c->stream, r1, (const double*)y, (const double*)r2,
                         (const double*)nullptr, st + ms_triple(E, 2), stop, m, E, y);
      std::swap(r1, r2);             // r1 <- r2, r2 <- the new vector
      hipLaunchKernelGGL(cg_dot_kernel, dim3(CG_BLOCKS), dim3(256), 0, c->stream, r2, r2, m, E, wk.partial);
      hipLaunchKernelGGL(minres_scalars_kernel, dim3(1), dim3(CG_BLOCKS), 0, c->stream, wk.partial, st, E, 2, rtol);
      {  // w_new = (v - oldeps w1 - delta w2) / gamma with w1 <- w2, w2 <- w
        double* t = w1;
        w1 = w2;
        w2 = w;
        w = t;
      }
      // w = T3 . (v, w1, w2);  x = x + phi w;  v = y / beta for the next iteration
      hipLaunchKernelGGL(minres_tail_kernel, dim3(vb), dim3(256), 0, c->stream, w, v, (const double*)w1, (const double*)w2, x,
                         (const double*)y, st + ms_triple(E, 3), st + ms_triple(E, 4), st + ms_triple(E, 0), stop, m, E);
    }
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipMemcpyAsync(state.data(), st, sizeof(double) * st_len, hipMemcpyDeviceToHost, c->stream);
    if (le == hipSuccess) le = hipStreamSynchronize(c->stream);
    if (le != hipSuccess) {
      c->async_product = false;
      HIP_TRY(c, le);
    }
    it = (int)state[ms_stop(E) + 1];
    rel = worst();
    if (state[ms_stop(E)] != 0.0) break;
  }
  c->async_product = false;

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
  for (int e = 0; e < E; ++e) {
    if (beta1[e] == 0.0) continue;
    const double v = std::sqrt(dots[e]) / beta1[e];
    if (!(v <= true_rel)) true_rel = v;  // NaN propagates
  }

  HIP_TRY(c, hipMemcpyAsync(out_b, x, vec, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_signal = false;
  if (iters) *iters = it;
  if (resid) *resid = true_rel;
  // the verdict is on the TRUE residual ||a - K x|| / ||a|| (include/kmvp.h), with 1.5x slack for the rounding
  // between it and the recurrence the iteration stops on; a non-finite residual is never a success
  if (!(std::isfinite(true_rel) && true_rel <= rtol * 1.5)) {
    c->err = std::isfinite(true_rel) ? "MINRES stopped before the true residual reached the requested tolerance"
                                     : "MINRES: the residual is not finite (non-finite operator or right-hand side)";
    return KMVP_E_NOT_CONVERGED;
  }
  return KMVP_OK;
}

}  // namespace kmvp
