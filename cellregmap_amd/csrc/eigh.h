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
int eigh_back_transform(crm_ctx* ctx, EighWork& w, double* Qt, double** Zt);

int launch_transpose(hipStream_t st, const double* src, long ld_src, long rows, long cols, double* dst, long ld_dst);

}  // namespace crm
