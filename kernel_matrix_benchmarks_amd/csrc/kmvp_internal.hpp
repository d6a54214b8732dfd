// Host-side declarations shared by the translation units of libkmvp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kmvp_lowd.hpp"

namespace kmvp {

// Tuning of a specialised low-D launch (host-side choice, compile-time in the kernel).
struct LowdTuning {
  int targets_per_lane;  // T
  int feed;              // 0 scalar-cache stream, 1 LDS tiles
};

// Largest shapes the specialised kernels are instantiated for; anything else goes
// to lowd_mid_kernel / lowd_big_kernel (or to the MFMA path for bf16).
constexpr int LOWD_MAX_D = 8;
constexpr int LOWD_MAX_E = 4;

// defaults, from the sweep on MI355X (profiles/): one target per lane, LDS-staged tiles
constexpr int DEFAULT_TARGETS_PER_LANE = 1;
constexpr int DEFAULT_FEED = 1;

// Returns hipErrorInvalidValue when (D, E, sig, tuning) has no instantiation.
// One function per kernel x precision: each lives in its own translation unit
// (kmvp_lowd_inst.hip compiled with -DKMVP_KERNEL / -DKMVP_REAL) and contains no
// kernel-selection branch in device code.
#define KMVP_DECLARE_LOWD(NAME, REAL)                                                           \
  hipError_t NAME(int D, int E, int sig, LowdTuning tune, const LowdArgs<REAL>& args, dim3 grid, \
                  hipStream_t stream, const char** kernel_name);                                \
  hipError_t NAME##_generic(int sig, const REAL* x, const REAL* y, const REAL* b, double* part,  \
                            int64_t n, int64_t n_pad, int64_t m, int D, int E, int NE,           \
                            int segments, int64_t seg_len, int64_t j_offset, int64_t m_total,    \
                            hipStream_t stream, const char** kernel_name);

KMVP_DECLARE_LOWD(launch_lowd_gaussian_f32, float)
KMVP_DECLARE_LOWD(launch_lowd_absexp_f32, float)
KMVP_DECLARE_LOWD(launch_lowd_invdist_f32, float)
KMVP_DECLARE_LOWD(launch_lowd_gaussian_f64, double)
KMVP_DECLARE_LOWD(launch_lowd_absexp_f64, double)
KMVP_DECLARE_LOWD(launch_lowd_invdist_f64, double)

// bf16 MFMA path (kmvp_mfma.hpp): largest shapes instantiated
constexpr int MFMA_MAX_KS = 9;  // D <= 16*9 - 6 = 138
constexpr int MFMA_MAX_NT = 4;  // E <= 128
struct MfmaArgs;
hipError_t launch_mfma_gaussian(int KS, int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream,
                                const char** kernel_name);
hipError_t launch_mfma_absexp(int KS, int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream,
                              const char** kernel_name);
hipError_t launch_mfma_invdist(int KS, int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream,
                               const char** kernel_name);
// exp(<x,y>) with the per-target running shift (kmvp_mfma.hpp; D <= 16 KS - 3)
hipError_t launch_mfma_expdot(int KS, int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream,
                              const char** kernel_name);
// the Gaussian with the same shift (targets != sources; D <= 16 KS - 9)
hipError_t launch_mfma_gaussian_shifted(int KS, int NT, int TW, const MfmaArgs& args, dim3 grid, hipStream_t stream,
                                        const char** kernel_name);


// split-bf16 MFMA low-D path (kmvp_fast.hpp): 6 D + 6 <= 48, E == 1
constexpr int FAST_MAX_D = 39;             // K = 6 D + 6 <= 240 = 15 k-steps of 16
constexpr int FAST_MAX_D_FOUR_TILES = 7;  // four target tiles per wave up to K = 48
constexpr int FAST_MAX_D_TWO_TILES = 23;  // two up to K = 144, one beyond
constexpr int FAST_DEFAULT_TT = 4;
// auto mode: scaled squared radius of the clouds below which the expansion is used
constexpr float FAST_AUTO_RADIUS2 = 8.0f;
struct FastArgs;
hipError_t launch_fast_gaussian(int D, int sig, int TT, const FastArgs& args, dim3 grid,
                                hipStream_t stream, const char** kernel_name);
hipError_t launch_fast_absexp(int D, int sig, int TT, const FastArgs& args, dim3 grid,
                              hipStream_t stream, const char** kernel_name);
hipError_t launch_fast_invdist(int D, int sig, int TT, const FastArgs& args, dim3 grid,
                               hipStream_t stream, const char** kernel_name);


// both products on the matrix cores (kmvp_fastmm.hpp): Gaussian, float32, D <= 64, several signal columns (one column beyond D = 39)
// cost model of the auto choice between the two (ps per 32 x 32 tile of pairs, whole chip; tools/fmm_probe.py)
constexpr double FMM_PS_PER_TILE_16 = 180.0, FMM_PS_PER_TILE_32 = 225.0;
constexpr double CMM_PS_PER_TILE_MAIN = 26.0, CMM_PS_PER_TILE_TT4 = 30.0, CMM_PS_PER_TILE_REST = 48.0;
struct FastmmArgs;
// online != 0: per-target running shift (kmvp_fastmm.hpp "The shift")
hipError_t launch_fastmm_gaussian(int KS, int mode, int TT, int online, const FastmmArgs& args, dim3 grid, hipStream_t stream,
                                  const char** kernel_name);
// exp(-r) on the same expansion, closest pairs recomputed exactly (5 <= D <= 64; D <= 4: the centred forms)
hipError_t launch_fastmm_absexp(int KS, int mode, int TT, int online, const FastmmArgs& args, dim3 grid, hipStream_t stream,
                                const char** kernel_name);
// tau = FMM_ABSEXP_KAPPA R^4 (R^2: scaled squared radius of the clouds): below it the error ~1e-7 R^2 of s would show
// in 2^-sqrt(s) beyond 1e-6 -- sqrt(s) >= ln2 * 1e-7 R^2 / 2e-6 = 0.035 R^2; twice that for margin
constexpr float FMM_ABSEXP_KAPPA = 5.0e-3f;

// the same second product on cfast_kernel's distances (kmvp_cfastmm.hpp): Gaussian and exp(-r), float32, D <= 4
struct CfastmmArgs;
hipError_t launch_cfastmm(int kernel, int mode, int TT, int online, const CfastmmArgs& args, dim3 grid, hipStream_t stream,
                          const char** kernel_name);
constexpr int CFMM_AUTO_MIN_COLS = 4;  // auto: from four columns on (1e5 points, four columns: 3.0 ms against 3.7 ms of the difference form)

// centred split-bf16 MFMA path (kmvp_cfast.hpp): D <= 4, E == 1, every kernel
constexpr int CFAST_MAX_D = 4;
constexpr int CFAST_DEFAULT_TT = 2;
constexpr int LOWD_MID_MAX_D = 128;  // lowd_mid_kernel: coordinates in registers up to here (16 chunks of 8)
constexpr int64_t SMALL_PROBLEM_TARGETS = 32768;  // below: one target tile per wave, one stage per segment
struct CfastArgs;
hipError_t launch_cfast_gaussian(int sig, int TT, const CfastArgs& args, dim3 grid, hipStream_t stream,
                                 const char** kernel_name);
hipError_t launch_cfast_absexp(int sig, int TT, const CfastArgs& args, dim3 grid, hipStream_t stream,
                               const char** kernel_name);
hipError_t launch_cfast_invdist(int sig, int TT, const CfastArgs& args, dim3 grid, hipStream_t stream,
                                const char** kernel_name);
// cell-reduced Gaussian path (kmvp_cell.hpp): float32, D <= 3, E == 1
constexpr int CELL_MAX_D = 3;
// auto mode: the path is taken when padding the cells' last tiles adds at most this share of slots
constexpr double CELL_AUTO_MAX_PAD = 1.30;
struct CellArgs;
hipError_t launch_cell_gaussian(int sig, int TT, const CellArgs& args, dim3 grid, hipStream_t stream,
                                const char** kernel_name);
struct CellmmArgs;
hipError_t launch_cellmm_gaussian(int TT, int shape, const CellmmArgs& args, dim3 grid, hipStream_t stream, const char** kernel_name);
struct Cell64Args;
hipError_t launch_cell64_gaussian(int sig, const Cell64Args& args, dim3 grid, hipStream_t stream, const char** kernel_name);
// kmvp_sort.hip: hipcub radix sort of (key, value) pairs; tmp == nullptr queries the scratch size
hipError_t sort_pairs_u32(void* tmp, size_t* tmp_bytes, const unsigned* keys_in, unsigned* keys_out,
                          const int* vals_in, int* vals_out, int64_t n, hipStream_t stream);

}  // namespace kmvp
