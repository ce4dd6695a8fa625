// Shared declarations for the CellRegMap score-test HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/crm_hip.h"
#include "../../include/crm_hip_test.h"

struct crm_ctx;

namespace crm {

void set_error(const char* fmt, ...);
const char* last_error_text();

#define CRM_HIP(call)                                                                 \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            crm::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call,              \
                           hipGetErrorString(e__));                                   \
            return CRM_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

#define CRM_TRY(call)                    \
    do {                                 \
        int rc__ = (call);               \
        if (rc__ != CRM_OK) return rc__; \
    } while (0)

inline long round_up(long x, long m) { return (x + m - 1) / m * m; }

// Kernel forms that exist beside the default one (a slower or older variant kept because the default falls back to it, or
// because a test holds the two against each other).  Process-wide switches set through crm_test_set_form
// (include/crm_hip_test.h) -- not environment variables: the suite flips every one of them.
//   "gram_staged"       1: score-statistic Gram through the register-staged kernel instead of the direct-to-LDS one
//   "kr_no_tail"        1: the Khatri-Rao contraction of a block in one launch of 128-column tiles whatever the spectrum
//   "nullfit_per_wave"  1: null fits with one independent wavefront per (variant, grid point) instead of the LDS-shared queue
//   "kin_fold"          0: never fold the donor-level factor into the mixing matrices; 2: fold also with few columns of us
//   "eigh_one_stage"    1: the constructor's eigen-solver tridiagonalises every grid point on its own (eigh_trd.hip)
//   "nullfit_exact"     1: null-fit likelihood with IEEE division and one log per spectrum entry
int form(const char* name, int otherwise);

// No C++ exception may cross the C-ABI (ctypes / cgo / JNI callers cannot unwind, the process would end in
// std::terminate): every extern "C" entry point runs its body through this guard, which turns std::bad_alloc,
// std::length_error and anything else into a status code with the text in crm_last_error().
template <class Body>
inline int guarded(const char* entry, Body&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory (std::bad_alloc)", entry);
    } catch (const std::exception& e) {
        set_error("%s: unexpected C++ exception: %s", entry, e.what());
    } catch (...) {
        set_error("%s: unexpected C++ exception", entry);
    }
    return CRM_ERR_INTERNAL;
}

// Named ranges for rocprofv3 --marker-trace (rocprofiler-sdk roctx), resolved at run time: without the library
// in the process they are no-ops.  Around the phases of the constructor and of every block of a scan.
void trace_push(const char* name);
void trace_pop();
struct TraceRange {
    explicit TraceRange(const char* name) { trace_push(name); }
    ~TraceRange() { trace_pop(); }
};

// ---- contraction kernel (gemm_tn.hip) ------------------------------------------
// C[z][M x N] = X[z]' * Y[z]  with the contraction over the cell axis (rows of X, Y).
// Khatri-Rao form: X[i, b*k0 + j] = Gs[i, b] * E[i, j] is formed on the fly.
struct GemmProblem {
    const double* X;  // plain: [cells x ldx]; KR: Gs [cells x ldx] (first variant of the group)
    const double* E;  // KR only: [cells x lde], lde >= round_up(k0, 32), zero padded
    const double* Y;  // [cells x ldy]
    double* C;        // [M x ldc]
    long ldx, lde, ldy, ldc;
    int M, N;         // logical extents (stores are predicated on them)
    int k0;           // KR only
    int flags;        // GEMM_SUBTRACT: C -= X'Y instead of C = X'Y (not for split launches)
    long cells;       // > 0: this problem's own contraction length (multiple of GEMM_BK, <= the launch's)
};
constexpr int GEMM_SUBTRACT = 1;

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an
// XCD and its L2), so the workgroups of one residue class take a contiguous run of tile ids --
// neighbouring row tiles (shared genotype cache lines) and one column tile of Q0 per XCD at a time.
// Bijective for any grid size; affects speed / L2 traffic only.
__device__ inline int xcd_tile_id(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 16;

// Launch nz problems (device array `probs`), each over `cells` (multiple of GEMM_BK)
// rows, optionally split into `ksplit` slices along the cell axis (slice s writes
// C + s * split_stride; reduce with reduce_splits).
// Per-context choice of the contraction kernel variant (defaults = what the scan uses; the unit tests
// and tools/gemm_bench force the others through include/crm_hip_test.h).
struct GemmTune {
    int bn = 0;         // output-tile width: 0 = chosen per launch, else 64 or 128
    int glds = 1;       // 128-wide tiles / Khatri-Rao launches through the LDS-DMA kernel (gemm_tn_glds.hip)
    int sync = 1;       // > 0: large Khatri-Rao launches as 8 x 64 persistent workgroups re-aligned every `sync` generations
                        // (default: 7.3x less L2-fabric traffic, L2 hit rate 67 % -> 96 %, for 0.6 % of the kernel's time)
    int shared_h = -1;  // multi-gene scan: -1 cost model, 0 never, 1 always contract once per variant against H
    int band = 8;       // plain products: > 1 walks the output tiles in bands of `band` column tiles (see gemm_tn_glds_kernel)
    int tag = 0;        // 1 around the scan's dominant plain product (Mix(rho*)' [H'(g o E0)] of the kinship-structure route): the
                        // launch then uses an instantiation of its own, gemm_tn_glds_kernel<false, 1, 0, false, 128, 1>, so that
                        // kernel traces and counter passes can tell it from the other plain products of a block
};
int launch_gemm_tn(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int max_m, int max_n,
                   long cells, bool khatri_rao, int k0, int ksplit, long split_stride);
// number of slices along the cell axis for a launch that would otherwise have `blocks_without_split` workgroups
int split_for(long cells_pad, long blocks_without_split);
int contraction_tile_width(const crm_ctx* ctx, int mt, int max_n, int nz, int ksplit, bool khatri_rao);
int kr_split_for(const crm_ctx* ctx, long row_tiles, int max_n, int nz, long cells_pad, int max_split);
int launch_gemm_tn_glds(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int mt, int nt, long cells,
                        bool khatri_rao, int k0, int ksplit, long split_stride, bool transposed_out = false,
                        int bn = 128);
// Khatri-Rao contraction storing C' (N x M, leading dimension ldc): always the LDS-DMA kernel
int launch_kr_transposed(crm_ctx* ctx, const GemmProblem* probs_dev, int nz, int max_m, int max_n, long cells,
                         int k0);
// C = X'Y for problems with N <= 16 columns (X with an even leading dimension, 16-byte aligned): one pass over X
int launch_skinny_tn(hipStream_t st, const GemmProblem* probs_dev, int nz, int max_m, long cells);
int launch_reduce_splits(hipStream_t st, double* C, long count, int ksplit, long split_stride);
int launch_reduce_splits_band(hipStream_t st, double* C, long rows, long ld, int col0, int ncols, int ksplit,
                              long split_stride);

}  // namespace crm
