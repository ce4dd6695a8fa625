// Batched symmetric eigen-solver of the background constructor (hand-written, gfx950): replaces the
// LAPACK calls behind numpy_sugar.economic_qs_linear (thin SVD via the Gram matrix, or eigh of the n x n
// covariance; in-tree twin cellregmap/_math.py:204-256), which the reference runs once per grid point.
//
//   1. eigh_trd.hip   blocked Householder tridiagonalisation A = Q T Q' of all grid points at once
//                     (panel of 32 columns: column kernel -> lower-triangle symv over every CU -> w kernel;
//                     rank-2k update of the trailing matrix on the FP64-MFMA contraction kernel)
//   2. eigh_dc.hip    divide & conquer on the tridiagonals: Jacobi leaves (<= 32), then per level the
//                     deflation on the host (O(dim) per merge), secular equation / Loewner vector /
//                     eigenvector blocks in kernels, and the merge products on the contraction kernel
//   3. eigh_bt.hip    back-transformation Z <- Q Z with compact-WY blocks of 128 reflectors (contractions)
#pragma once
#include "crm_internal.h"

namespace crm {

constexpr int TRD_NB = 32;     // panel width of the tridiagonalisation
constexpr int BT_NB = 128;     // reflectors per compact-WY block of the back-transformation
constexpr int DC_LEAF = 32;    // largest tridiagonal block solved directly (Jacobi)
constexpr int DC_ALIGN = 16;   // block boundaries of the D&C tree are multiples of this

// Work buffers of one batched solve; all device matrices are [batch] x (dimp x ld) row-major, ld = dimp =
// round_up(dim, 128), plus 256 doubles of slack behind each batch slab (tile over-reads of the contractions).
struct EighWork {
    int batch = 0;
    long dim = 0, dimp = 0, ld = 0, slab = 0;  // slab = dimp * ld + 256 (doubles per matrix)
    DevBuf A;        // in: the matrices (both triangles); destroyed.  Later: the merge blocks U of the D&C
    DevBuf Vt;       // row j = Householder vector v_j (zeros up to j, one at j + 1)
    DevBuf Vc;       // its transpose (column j = v_j); before that: scratch rows of the D&C
    DevBuf QA, QB;   // eigenvector rows of the tridiagonal (ping-pong over the D&C levels)
    DevBuf d, e, tau, lam;   // [batch][ld]
    DevBuf small;    // panels, partial sums, problem records, D&C descriptors
    // two-stage solver of a family D(rho) C D(rho) (eigh2_band.hip, eigh2_chase.hip, eigh2_back.hip)
    bool v_shared = false;   // the reflectors of the back-transformation (Vt, tau: slab 0) serve every matrix of the batch
    DevBuf AB;       // [batch][dimp + 128][128]  lower band storage, column c at offsets row - c (room for the chase's fill)
    DevBuf Vbc;      // [batch][positions][dimp][64]  reflectors of the chase, chain position major
    DevBuf taubc;    // [batch][positions][dimp]
    DevBuf Tbc;      // [batch][sweep blocks][positions][64 x 64]  T factors of the grouped reflectors
    DevBuf s1;       // stage-1 scratch (T factors, partial sums, panels, counters, problem records)
    DevBuf sync;     // progress counters of the chase, abort flag
};

// Eigen-decomposition of `batch` symmetric matrices held in w.A (dim x dim each, leading dimension w.ld,
// both triangles filled).  On return lam_host[b * dim + j] holds the eigenvalues of matrix b in ascending
// order and *Zt points at a [batch] x slab device array whose row j (of matrix b) is the eigenvector that
// belongs to lam[b][j] (i.e. column-major eigenvector matrices with leading dimension w.ld).
int eigh_alloc(EighWork& w, int batch, long dim);
void eigh_free(EighWork& w);
int eigh_batched(crm_ctx* ctx, EighWork& w, double* lam_host, double** Zt);

// phases (also reachable one by one through the test hooks)
int eigh_tridiagonalise(crm_ctx* ctx, EighWork& w);                       // A -> d, e, tau, Vt
int eigh_dc(crm_ctx* ctx, EighWork& w, double* lam_host, double** Qt);    // d, e -> lam (ascending), rows
// (z_ready: the eigenvectors already stand as COLUMNS in w.A -- eigh_rows_to_columns -- and Qt only names the free slabs)
int eigh_back_transform(crm_ctx* ctx, EighWork& w, double* Qt, double** Zt, bool z_ready = false);
int eigh_rows_to_columns(crm_ctx* ctx, EighWork& w, const double* Qt);    // w.A <- Qt' per matrix (zero padded)

// Two-stage solver for the constructor's family of grid points (eigh2_band.hip / eigh2_chase.hip):
//   A_q = D_q C D_q,  D_q = diag(wa[q] on the first E2_W coordinates, wb[q] on the rest),  q < w.batch,
// with C (dim x dim, both triangles, leading dimension w.ld) in slab 0 of w.A.  The first E2_W coordinates are the
// caller's leading block (the contexts' columns of the half factor, padded in front with zero rows / columns up to E2_W).
// Same outputs as eigh_batched.  CRM_ERR_UNSUPPORTED with nothing computed when the problem is outside what the
// two-stage form serves, CRM_ERR_BUSY-like CRM_ERR_HIP never: a chase that cannot keep its workgroups co-resident gives
// up after a bounded wait and returns CRM_ERR_UNSUPPORTED too -- the caller then runs eigh_batched.
// bound of every inter-workgroup wait of the two-stage solver, in ticks of the constant 100 MHz wall clock (wall_clock64): 2 s
constexpr unsigned long long E2_WAIT_TICKS = 200000000ull;
constexpr int E2_W = 64;       // panel width of stage 1 = half-bandwidth = reflector length of the chase = sweeps per group
int eigh2_family(crm_ctx* ctx, EighWork& w, const double* wa, const double* wb, double* lam_host, double** Zt);
bool eigh2_serves(long dim, int batch);
// phases
int eigh2_to_band(crm_ctx* ctx, EighWork& w);                                   // slab 0 of A -> band (lower), Vt / tau slab 0
int eigh2_scale_band(crm_ctx* ctx, EighWork& w, const double* wa, const double* wb);   // -> AB of every matrix
int eigh2_chase(crm_ctx* ctx, EighWork& w);                                     // AB -> d, e, Vbc, taubc
int eigh2_back_chase(crm_ctx* ctx, EighWork& w, double* Z);                     // columns of Z <- Q2 Z (in place; eigh2_back.hip)

int launch_transpose(hipStream_t st, const double* src, long ld_src, long rows, long cols, double* dst, long ld_dst);

}  // namespace crm
