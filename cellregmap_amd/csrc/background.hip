// On-device economic eigendecomposition of the background covariances
//     Sigma(rho) = hS(rho) hS(rho)',  hS(rho) = [ sqrt(rho) E1 , sqrt(1-rho) B ]
// replacing the rho loop of CellRegMap.__init__ (cellregmap/_cellregmap.py:101-131) and
// numpy_sugar.economic_qs_linear (in-tree twin cellregmap/_math.py:238-256).
//
// cols < n  (the reference's thin-SVD branch): the cols x cols Gram matrix of [E1, B] is formed
//   once with the FP64-MFMA contraction kernel, rescaled per rho, diagonalised -- all grid points in
//   one batch by the hand-written solver of eigh*.hip (tridiagonalisation, divide & conquer,
//   back-transformation) -- and Q0 = hS V L^-1/2 is formed by the contraction kernel again; two Newton-Schulz steps
//   (contractions only) restore orthonormality of the columns that belong to small
//   eigenvalues.  Columns with eigenvalue <= rel_tol * max are dropped: they are inert in
//   every bilinear form of the path (weight (1-d) S + d == d, cancelled by the complement
//   term) whereas the reference's SVD keeps them with S0 ~ 1e-29.
// cols >= n (the reference's eigh branch): Sigma(rho) itself (n x n) is diagonalised and
//   eigenvalues below sqrt(machine eps) are dropped, exactly as _math.py:204-235 does.
#include <algorithm>
#include <chrono>

#include "eigh.h"
#include "nullfit.h"
#include "objects.h"

using namespace crm;

namespace crm {
namespace {

__global__ void transpose_kernel(const double* __restrict__ src, long ld_src, long rows, long cols,
                                 double* __restrict__ dst, long ld_dst) {
    __shared__ double tile[32][33];
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const long r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[r * ld_src + c] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const long c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows) dst[c * ld_dst + r] = tile[threadIdx.x][i];
    }
}

// out[i, j] = w(i) * w(j) * C[i, j];  w = sqrt(rho) for i < k1 else sqrt(1 - rho)
__global__ void scale_gram_kernel(const double* __restrict__ C, long ldc, int cols, int k1, double a,
                                  double b, double* __restrict__ out, long ldo) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= cols) return;
    const double wi = i < k1 ? a : b, wj = j < k1 ? a : b;
    out[(long)i * ldo + j] = wi * wj * C[(long)i * ldc + j];
}

// out = wa * A + wb * B (n x n)
__global__ void combine_kernel(const double* __restrict__ A, const double* __restrict__ B, long ld, int n,
                               double wa, double wb, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= n) return;
    out[i * ld + j] = wa * A[i * ld + j] + wb * B[i * ld + j];
}

// M[k, j] = w(k) * V[k, keep[j]] / sqrt(S[keep[j]]);  V column-major (eigenvector j = column j)
__global__ void build_mixing_kernel(const double* __restrict__ V, long ldv, const double* __restrict__ S,
                                    const int* __restrict__ keep, int r, int cols, int k1, double a,
                                    double b, double* __restrict__ M, long ldm) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (j >= r) return;
    const int src = keep[j];
    const double wk = k < k1 ? a : b;
    M[(long)k * ldm + j] = wk * V[(long)src * ldv + k] * rsqrt(S[src]);
}

// N = 1.5 I - 0.5 G ; err = max |G - I|
__global__ void newton_schulz_kernel(const double* __restrict__ G, long ldg, int r, double* __restrict__ N,
                                     long ldn, double* __restrict__ err_blocks) {
    __shared__ double red[256];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    double e = 0.0;
    if (j < r) {
        const double g = G[(long)i * ldg + j];
        const double id = (i == j) ? 1.0 : 0.0;
        e = fabs(g - id);
        N[(long)i * ldn + j] = 1.5 * id - 0.5 * g;
    }
    red[threadIdx.x] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) err_blocks[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// err = max |G[i, jj] - (i == off + jj)| over an r x ncols slab of Mix' C Mix (columns off .. off + ncols)
__global__ void defect_slab_kernel(const double* __restrict__ G, long ldg, int r, int ncols, int off,
                                   double* __restrict__ err_blocks) {
    __shared__ double red[256];
    const int jj = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    double e = 0.0;
    if (jj < ncols) e = fabs(G[(long)i * ldg + jj] - ((i == off + jj) ? 1.0 : 0.0));
    red[threadIdx.x] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) err_blocks[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

// Q0[i, j] = V[i, keep[j]] from the column-major eigenvector matrix
__global__ void gather_vectors_kernel(const double* __restrict__ V, long ldv, const int* __restrict__ keep,
                                      int r, long n, double* __restrict__ Q0, long ldq) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const long i = blockIdx.y;
    if (j >= r || i >= n) return;
    Q0[i * ldq + j] = V[(long)keep[j] * ldv + i];
}

struct Scratch {
    DevBuf bufs[13];
    ~Scratch() {
        for (auto& b : bufs) b.release();
    }
};

int contract(crm_ctx* ctx, const double* X, long ldx, const double* Y, long ldy, double* C, long ldc,
             int M, int N, long cells) {
    GemmProblem p{};
    p.X = X; p.ldx = ldx; p.Y = Y; p.ldy = ldy; p.C = C; p.ldc = ldc; p.M = M; p.N = N;
    CRM_TRY(ctx->ws_probs.ensure(sizeof(GemmProblem) * (CRM_MAX_RHO + 4)));
    CRM_HIP(hipMemcpyAsync(ctx->ws_probs.ptr, &p, sizeof p, hipMemcpyHostToDevice, ctx->stream));
    CRM_TRY(launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>(), 1, M, N, cells, false, 0, 1, 0));
    CRM_HIP(hipStreamSynchronize(ctx->stream));  // the problem record is reused by the next call
    return CRM_OK;
}

// C[j, i] = C[i, j] for i < j (one workgroup per 32 x 32 tile above the diagonal)
__global__ void mirror_upper_kernel(double* __restrict__ C, long ldc, int n) {
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bi > bj) return;
    const int i0 = bi * 32, j0 = bj * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + threadIdx.x;
        tile[r][threadIdx.x] = (i < n && j < n) ? C[(long)i * ldc + j] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + threadIdx.x;     // element (j, i) of the lower triangle
        if (j < n && i < n && j > i) C[(long)j * ldc + i] = tile[threadIdx.x][r];
    }
}

// C = X'X (n x n) for X: cells x n -- only the tiles on or above the diagonal are computed (one problem per column
// panel of 128, as many row tiles as reach the diagonal), the rest is their mirror image: half the flops of the product.
int gram_upper_then_mirror(crm_ctx* ctx, const double* X, long ldx, double* C, long ldc, int n, long cells) {
    const int panels = (n + 127) / 128;
    if (panels <= 2) return contract(ctx, X, ldx, X, ldx, C, ldc, n, n, cells);
    std::vector<GemmProblem> pr(panels);
    for (int j = 0; j < panels; j++) {
        GemmProblem p{};
        p.X = X; p.ldx = ldx;
        p.Y = X + (long)j * 128; p.ldy = ldx;
        p.C = C + (long)j * 128; p.ldc = ldc;
        p.M = std::min(n, (j + 1) * 128); p.N = std::min(128, n - j * 128);
        pr[j] = p;
    }
    ScopedBuf d;
    CRM_TRY(d.ensure(sizeof(GemmProblem) * pr.size()));
    CRM_HIP(hipMemcpyAsync(d.ptr, pr.data(), sizeof(GemmProblem) * pr.size(), hipMemcpyHostToDevice, ctx->stream));
    CRM_TRY(launch_gemm_tn(ctx, d.as<GemmProblem>(), panels, n, 128, cells, false, 0, 1, 0));
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32));
    hipLaunchKernelGGL(mirror_upper_kernel, grid, dim3(32, 8), 0, ctx->stream, C, ldc, n);
    CRM_HIP(hipGetLastError());
    CRM_HIP(hipStreamSynchronize(ctx->stream));
    return CRM_OK;
}

int transpose(hipStream_t st, const double* src, long ld_src, long rows, long cols, double* dst, long ld_dst) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, st, src, ld_src, rows, cols, dst, ld_dst);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace
}  // namespace crm

namespace crm {
namespace {
// B[i, j * m + d] = U[i, j] * hK[i, d]: the halves L_j = diag(U[:, j]) hK of K o E2E2' (proof.md,
// get_L_values _cellregmap.py:533-545) written straight into H = [E1, B]
__global__ void hadamard_halves_kernel(const double* __restrict__ U, int k2, const double* __restrict__ hK, int m,
                                       long n, double* __restrict__ H, long ldh, int col0) {
    const long i = blockIdx.x;
    if (i >= n) return;
    for (int e = threadIdx.x; e < k2 * m; e += blockDim.x) {
        const int j = e / m, d = e - j * m;
        H[i * ldh + col0 + e] = U[i * k2 + j] * hK[i * m + d];
    }
}
}  // namespace
}  // namespace crm

// ---- the constructor in three phases -----------------------------------------------------------------
// begin    : upload the half factor H = [E1, B], its Gram matrix (thin branch) or E1E1' and BB' (eigh
//            branch), eigen-decompositions of the grid points this process owns -> their ranks
// complete : given the ranks of ALL grid points (they fix the common leading dimension): buffers for every
//            grid point, Q0 = H Mix (+ orthonormality polish) for the owned ones
// seal     : spectra of all grid points known (owned ones computed, the others imported): decide whether the
//            rotations may go through the mixing matrices, drop the builder
// One process: begin(all) -> complete -> seal.  Several processes (one per GPU): every rank owns a share of the
// grid points, the ranks are all-gathered, and Q0 / S0 / Mix of each grid point are broadcast by its owner
// through crm_background_export / _import (cellregmap_amd/distributed.py).
struct crm_background_builder {
    Scratch S;
    std::vector<DevBuf> Mbuf;                  // per grid point: mixing matrix (thin) or eigenvectors (eigh)
    std::vector<std::vector<double>> S0_host;  // kept eigenvalues per grid point
    std::vector<int> mine;
    long n = 0, np = 0, cols = 0, cp = 0, dim = 0, dimp = 0;
    int k1 = 0;
    bool thin = true, completed = false;
    double rel_tol = 1e-12;
    ~crm_background_builder() {
        for (auto& b : Mbuf) b.release();
    }
};

namespace {
struct SetupTrace {
    hipStream_t st;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit SetupTrace(hipStream_t s) : st(s), on(getenv("CRM_TRACE_SETUP") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[crm background] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t).count());
        t = now;
    }
};
}  // namespace

// B given explicitly (kb columns), or as Hadamard factors U (n x k2) and hK (n x m) with kb = k2 * m
static int background_begin(crm_ctx* ctx, long n, const double* E1, int k1, const double* B, long kb, const double* U,
                            int k2, const double* hK, int m, int nrho, const double* rho, const int* mine_flags,
                            double rel_tol, crm_background** out) {
    if (!ctx || !out || n <= 0 || !E1 || k1 < 1 || kb < 0 || (kb > 0 && !B && !(U && hK)) || nrho < 1 || !rho)
        return CRM_ERR_ARG;
    if (nrho > CRM_MAX_RHO) {
        set_error("background: %d grid points (supported up to %d)", nrho, CRM_MAX_RHO);
        return CRM_ERR_UNSUPPORTED;
    }
    *out = nullptr;
    if (rel_tol <= 0.0) rel_tol = 1e-12;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SetupTrace trace(st);
    TraceRange range("crm background begin");
    const long cols = k1 + kb;
    const long np = round_up(n, CELL_PAD);
    const long cp = round_up(cols, 128);
    const bool thin = n > cols;  // economic_qs_linear: rows > cols -> SVD branch

    crm_background* bg = new crm_background();
    crm_background_builder* bb = new crm_background_builder();
    bg->builder = bb;
    bg->ctx = ctx;
    bg->n = n;
    bg->n_pad = np;
    bg->nrho = nrho;
    bb->n = n; bb->np = np; bb->cols = cols; bb->cp = cp; bb->k1 = k1; bb->thin = thin; bb->rel_tol = rel_tol;
    bb->Mbuf.resize(nrho);
    bb->S0_host.resize(nrho);
    for (int i = 0; i < nrho; i++) {
        bg->rho[i] = rho[i];
        bg->r[i] = -1;
        if (!mine_flags || mine_flags[i]) bb->mine.push_back(i);
    }
    auto fail = [&](int code) {
        crm_background_destroy(bg);
        return code;
    };
    int rc = CRM_OK;
#define CRM_BG(call)                              \
    do {                                          \
        if ((rc = (call)) != CRM_OK) return fail(rc); \
    } while (0)
#define CRM_BG_HIP(call)                                                                    \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return fail(CRM_ERR_HIP);                                                       \
        }                                                                                   \
    } while (0)
    DevBuf &dH = bb->S.bufs[0], &dHt = bb->S.bufs[1], &dC = bb->S.bufs[2], &dMt = bb->S.bufs[9], &dG = bb->S.bufs[10];
    // H = [E1, B] (cells x cols) and its transpose
    CRM_BG(dH.ensure(sizeof(double) * np * cp));
    CRM_BG_HIP(hipMemsetAsync(dH.ptr, 0, sizeof(double) * np * cp, st));
    CRM_BG_HIP(hipMemcpy2DAsync(dH.ptr, cp * sizeof(double), E1, k1 * sizeof(double), k1 * sizeof(double), n,
                                hipMemcpyHostToDevice, st));
    if (kb > 0 && B) {
        CRM_BG_HIP(hipMemcpy2DAsync(dH.as<double>() + k1, cp * sizeof(double), B, kb * sizeof(double),
                                    kb * sizeof(double), n, hipMemcpyHostToDevice, st));
    } else if (kb > 0) {
        ScopedBuf dU, dK;
        CRM_BG(dU.ensure(sizeof(double) * n * k2));
        CRM_BG(dK.ensure(sizeof(double) * n * m));
        CRM_BG_HIP(hipMemcpyAsync(dU.ptr, U, sizeof(double) * n * k2, hipMemcpyHostToDevice, st));
        CRM_BG_HIP(hipMemcpyAsync(dK.ptr, hK, sizeof(double) * n * m, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(hadamard_halves_kernel, dim3((unsigned)n), dim3(256), 0, st, dU.as<double>(), k2,
                           dK.as<double>(), m, n, dH.as<double>(), cp, k1);
        CRM_BG_HIP(hipGetLastError());
        CRM_BG_HIP(hipStreamSynchronize(st));
    }
    trace.lap("  upload of the half factor");
    CRM_BG(dHt.ensure(sizeof(double) * cp * np));
    CRM_BG_HIP(hipMemsetAsync(dHt.ptr, 0, sizeof(double) * cp * np, st));
    CRM_BG(transpose(st, dH.as<double>(), cp, n, cols, dHt.as<double>(), np));
    trace.lap("  transpose");

    const long dim = thin ? cols : n;       // order of the matrices that get diagonalised
    const long dimp = round_up(dim, 128);
    bb->dim = dim; bb->dimp = dimp;
    if (thin) {
        // Gram matrix of the unscaled half factor, once
        CRM_BG(dC.ensure(sizeof(double) * cp * cp));
        CRM_BG_HIP(hipMemsetAsync(dC.ptr, 0, sizeof(double) * cp * cp, st));   // (its padding is an operand later)
        CRM_BG(gram_upper_then_mirror(ctx, dH.as<double>(), cp, dC.as<double>(), cp, (int)cols, np));
    } else {
        // E1 E1' and B B' (n x n), contraction over the column axis = rows of Ht; row blocks are
        // copied into zero-padded scratch so that their counts are multiples of the stage depth
        const long k1p = round_up(k1, GEMM_BK), kbp = round_up(std::max<long>(kb, 1), GEMM_BK);
        CRM_BG(dG.ensure(sizeof(double) * dimp * dimp * 2));
        CRM_BG(dMt.ensure(sizeof(double) * (k1p + kbp) * np));
        double* S1 = dG.as<double>();
        double* S2 = S1 + dimp * dimp;
        double* E1t = dMt.as<double>();
        double* Bt = E1t + k1p * np;
        CRM_BG_HIP(hipMemsetAsync(dG.ptr, 0, sizeof(double) * dimp * dimp * 2, st));
        CRM_BG_HIP(hipMemsetAsync(E1t, 0, sizeof(double) * (k1p + kbp) * np, st));
        CRM_BG_HIP(hipMemcpyAsync(E1t, dHt.ptr, sizeof(double) * k1 * np, hipMemcpyDeviceToDevice, st));
        if (kb > 0)
            CRM_BG_HIP(hipMemcpyAsync(Bt, dHt.as<double>() + (long)k1 * np, sizeof(double) * kb * np,
                                      hipMemcpyDeviceToDevice, st));
        CRM_BG(contract(ctx, E1t, np, E1t, np, S1, dimp, (int)n, (int)n, k1p));
        if (kb > 0) CRM_BG(contract(ctx, Bt, np, Bt, np, S2, dimp, (int)n, (int)n, kbp));
    }
    CRM_BG_HIP(hipStreamSynchronize(st));
    trace.lap("half factor + Gram");
    // eigen-decompositions of the owned grid points (eigh*.hip), all at once -- except that a grid point whose second
    // weight vanishes (rho = 1 in the thin branch: hS = [E1, 0]) has a scaled Gram matrix that is zero outside its
    // leading k1 x k1 block: that block is decomposed on its own (k1 contexts against cols = k1 + kb columns: 50
    // against 5 050 at config 3, i.e. one of the eleven full-size decompositions less)
    std::vector<int> full_pts, lead_pts;
    for (int i : bb->mine) ((thin && kb > 0 && rho[i] >= 1.0) ? lead_pts : full_pts).push_back(i);
    auto decompose = [&](const std::vector<int>& pts, const long sub) -> int {   // sub: order of the matrices of this batch
        const int npts = (int)pts.size();
        if (npts == 0) return CRM_OK;
        const long subp = round_up(sub, 128);
        // the context's cached work buffers (grown on demand; see crm_ctx::eigh_ws)
        EighWork& ew = *acquire_eigh_workspace(ctx);
        struct EGuard { crm_ctx* c; ~EGuard() { release_eigh_workspace(c); } } eguard{ctx};
        // Two-stage form for the thin branch's family  A(rho) = D(rho) C D(rho)  (eigh2_band.hip): the leading block --
        // the k1 contexts' columns, padded in front with zero rows / columns to 64 coordinates -- is the first panel of a
        // dense -> band reduction done ONCE, each grid point then only chases its own rescaling of that band.  The padding
        // adds 64 - k1 null directions, which the rank rule below drops like any other.
        const long pad = (thin && sub == dim && k1 >= 1 && k1 <= E2_W && eigh2_serves(sub + (E2_W - k1), npts)) ? E2_W - k1 : -1;
        std::vector<double> lam;
        double* Zt = nullptr;   // rows = eigenvectors, leading dimension ew.ld
        long voff = 0, nlam = sub;
        bool solved = false;
        if (pad >= 0) {
            const long sub2 = sub + pad;
            CRM_TRY(eigh_alloc(ew, npts, sub2));
            trace.lap("  eigen workspace");
            CRM_HIP(hipMemsetAsync(ew.A.ptr, 0, sizeof(double) * ew.slab, st));
            dim3 grid((unsigned)((sub + 255) / 256), (unsigned)sub);
            hipLaunchKernelGGL(scale_gram_kernel, grid, dim3(256), 0, st, dC.as<double>(), cp, (int)sub, k1, 1.0, 1.0,
                               ew.A.as<double>() + pad * ew.ld + pad, ew.ld);
            CRM_HIP(hipGetLastError());
            std::vector<double> wa(npts), wb(npts);
            for (int q = 0; q < npts; q++) {
                wa[q] = std::sqrt(rho[pts[q]]);
                wb[q] = std::sqrt(1.0 - rho[pts[q]]);
            }
            lam.resize((size_t)npts * sub2);
            const int rc = eigh2_family(ctx, ew, wa.data(), wb.data(), lam.data(), &Zt);
            if (rc == CRM_OK) {
                solved = true;
                voff = pad;
                nlam = sub2;
                trace.lap("eigen-decompositions (two-stage)");
            } else if (rc != CRM_ERR_UNSUPPORTED) {
                return rc;
            }
        }
        if (!solved) {
        CRM_TRY(eigh_alloc(ew, npts, sub));
        trace.lap("  eigen workspace");
        CRM_HIP(hipMemsetAsync(ew.A.ptr, 0, sizeof(double) * (size_t)npts * ew.slab, st));
        for (int q = 0; q < npts; q++) {
            const int i = pts[q];
            const double a = std::sqrt(rho[i]), b = std::sqrt(1.0 - rho[i]);
            double* Ai = ew.A.as<double>() + (size_t)q * ew.slab;
            if (thin) {
                dim3 grid((unsigned)((sub + 255) / 256), (unsigned)sub);
                hipLaunchKernelGGL(scale_gram_kernel, grid, dim3(256), 0, st, dC.as<double>(), cp, (int)sub, k1, a, b,
                                   Ai, ew.ld);
            } else {
                // Sigma(rho) = rho E1 E1' + (1 - rho) B B'
                dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
                hipLaunchKernelGGL(combine_kernel, grid, dim3(256), 0, st, dG.as<double>(),
                                   dG.as<double>() + dimp * dimp, dimp, (int)n, rho[i], 1.0 - rho[i], Ai);
            }
        }
        CRM_HIP(hipGetLastError());
        lam.assign((size_t)npts * sub, 0.0);
        trace.lap("  scaled Gram matrices");
        CRM_TRY(eigh_batched(ctx, ew, lam.data(), &Zt));
        trace.lap("eigen-decompositions");
        }
        ScopedBuf wKeep, wLam;
        CRM_TRY(wKeep.ensure(sizeof(int) * (subp + 128)));
        CRM_TRY(wLam.ensure(sizeof(double) * (subp + 128)));
        for (int q = 0; q < npts; q++) {
            const int i = pts[q];
            const double* hW = &lam[(size_t)q * nlam];   // ascending
            const double a = std::sqrt(rho[i]), b = std::sqrt(1.0 - rho[i]);
            std::vector<int> keep;
            if (thin) {
                const double cut = rel_tol * std::max(hW[nlam - 1], 0.0);
                for (long j = nlam - 1; j >= 0; j--)  // descending, like singular values
                    if (hW[j] > cut && hW[j] > 0.0) keep.push_back((int)j);
            } else {
                const double eps_small = 1.4901161193847656e-08;  // sqrt(machine eps), _math.py:204
                for (long j = 0; j < sub; j++)                    // ascending, like eigh
                    if (hW[j] >= eps_small) keep.push_back((int)j);
            }
            const int r = (int)keep.size();
            bg->r[i] = r;
            bb->S0_host[i].resize(r);
            for (int j = 0; j < r; j++) bb->S0_host[i][j] = hW[keep[j]];
            // keep what `complete` needs: thin -> mixing matrix M (cols x r; rows beyond `sub` stay zero); else -> the
            // vectors themselves
            const long ldm = round_up(std::max(r, 1), 128);
            CRM_TRY(bb->Mbuf[i].ensure(sizeof(double) * (thin ? cp : np) * ldm));
            CRM_HIP(hipMemsetAsync(bb->Mbuf[i].ptr, 0, sizeof(double) * (thin ? cp : np) * ldm, st));
            if (r > 0) {
                const double* Vi = Zt + (size_t)q * ew.slab + voff;   // (two-stage form: past the padding coordinates)
                CRM_HIP(hipMemcpyAsync(wKeep.ptr, keep.data(), sizeof(int) * r, hipMemcpyHostToDevice, st));
                CRM_HIP(hipMemcpyAsync(wLam.ptr, hW, sizeof(double) * nlam, hipMemcpyHostToDevice, st));
                if (thin) {
                    dim3 grid((unsigned)((r + 255) / 256), (unsigned)sub);
                    hipLaunchKernelGGL(build_mixing_kernel, grid, dim3(256), 0, st, Vi, ew.ld, wLam.as<double>(),
                                       wKeep.as<int>(), r, (int)sub, k1, a, b, bb->Mbuf[i].as<double>(), ldm);
                } else {
                    dim3 grid((unsigned)((r + 255) / 256), (unsigned)n);
                    hipLaunchKernelGGL(gather_vectors_kernel, grid, dim3(256), 0, st, Vi, ew.ld, wKeep.as<int>(), r, n,
                                       bb->Mbuf[i].as<double>(), ldm);
                }
                CRM_HIP(hipGetLastError());
            }
            CRM_HIP(hipStreamSynchronize(st));   // keep / hW are reused by the next grid point
        }
        return CRM_OK;
    };
    CRM_BG(decompose(full_pts, dim));
    CRM_BG(decompose(lead_pts, k1));
    trace.lap("  mixing matrices");
    *out = bg;
    return CRM_OK;
}

// ranks of all grid points known: buffers for every grid point; Q0 = H Mix and the polish for the owned ones
static int background_complete(crm_background* bg, const int* r_all) {
    crm_background_builder* bb = bg->builder;
    if (!bb || bb->completed) {
        set_error("background: complete called out of order");
        return CRM_ERR_ARG;
    }
    crm_ctx* ctx = bg->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    SetupTrace trace(st);
    TraceRange range("crm background complete");
    const long n = bb->n, np = bb->np, cols = bb->cols, cp = bb->cp;
    const bool thin = bb->thin;
    const int nrho = bg->nrho;
    long rmax = 1;
    for (int i = 0; i < nrho; i++) {
        const int r = r_all ? r_all[i] : bg->r[i];
        if (r < 0 || (bg->r[i] >= 0 && bg->r[i] != r)) {
            set_error("background: rank of grid point %d missing or inconsistent (%d vs %d)", i, r, bg->r[i]);
            return CRM_ERR_ARG;
        }
        bg->r[i] = r;
        rmax = std::max<long>(rmax, r);
    }
    bg->ldq = round_up(rmax, 128);
    const long ldq = bg->ldq;
    DevBuf &dH = bb->S.bufs[0], &dHt = bb->S.bufs[1], &dC = bb->S.bufs[2], &dMt = bb->S.bufs[9], &dG = bb->S.bufs[10],
           &dErr = bb->S.bufs[11], &dQt = bb->S.bufs[12], &dT1 = bb->S.bufs[3];
    dG.release();
    dMt.release();
    if (thin) {
        // keep the half factor: T(rho) = Q0(rho)'G is later taken as Mix(rho)' (H'G), see scan.hip
        bg->H = dH;
        dH = DevBuf();
        bg->Ht = dHt;   // (its transpose: operand of Q0 = H Mix, formed on first use)
        dHt = DevBuf();
        bg->ldh = cp;
        bg->cols = cols;
    } else {
        dH.release();
    }
    CRM_TRY(dG.ensure(sizeof(double) * ldq * ldq * 2));
    CRM_TRY(dQt.ensure(sizeof(double) * ldq * std::max(np, cp)));
    CRM_TRY(dErr.ensure(sizeof(double) * ((ldq + 255) / 256) * ldq));
    if (thin) {
        CRM_TRY(dT1.ensure(sizeof(double) * cp * ldq));
        CRM_HIP(hipMemsetAsync(dT1.ptr, 0, sizeof(double) * cp * ldq, st));
    }
    CRM_HIP(hipMemsetAsync(dG.ptr, 0, sizeof(double) * ldq * ldq * 2, st));
    const bool eager = false;   // (Q0 = H Mix is formed the first time a scan selects the grid point)
    for (int i = 0; i < nrho; i++) {
        if (!thin) {   // (thin branch: Q0 = H Mix on first use, crm_background_require_q0)
            CRM_TRY(bg->Q0[i].ensure(sizeof(double) * np * ldq));
            CRM_HIP(hipMemsetAsync(bg->Q0[i].ptr, 0, sizeof(double) * np * ldq, st));
        }
        CRM_TRY(bg->S0[i].ensure(sizeof(double) * ldq));
        CRM_HIP(hipMemsetAsync(bg->S0[i].ptr, 0, sizeof(double) * ldq, st));
        if (thin) {
            CRM_TRY(bg->Mix[i].ensure(sizeof(double) * cp * ldq));
            CRM_HIP(hipMemsetAsync(bg->Mix[i].ptr, 0, sizeof(double) * cp * ldq, st));
        }
    }
    trace.lap("  buffers of the grid points");
    double* Gq = dG.as<double>();
    double* N = Gq + ldq * ldq;
    auto defect_and_correction = [&](int r, double* err_out) -> int {
        // N = 1.5 I - 0.5 G, err = max |G - I|
        dim3 grid((unsigned)((r + 255) / 256), (unsigned)r);
        CRM_HIP(hipMemsetAsync(N, 0, sizeof(double) * ldq * ldq, st));
        hipLaunchKernelGGL(newton_schulz_kernel, grid, dim3(256), 0, st, Gq, ldq, r, N, ldq, dErr.as<double>());
        CRM_HIP(hipGetLastError());
        std::vector<double> herr((size_t)grid.x * grid.y);
        CRM_HIP(hipMemcpyAsync(herr.data(), dErr.ptr, sizeof(double) * herr.size(), hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        double err = 0.0;
        for (double e : herr) err = std::max(err, e);
        *err_out = err;
        return CRM_OK;
    };
    for (int i : bb->mine) {
        const int r = bg->r[i];
        if (r == 0) continue;   // (an empty grid point: require_q0 just allocates zeros)
        CRM_HIP(hipMemcpyAsync(bg->S0[i].ptr, bb->S0_host[i].data(), sizeof(double) * r, hipMemcpyHostToDevice, st));
        const long ldm = round_up(r, 128);
        if (thin) {
            double* Mix = bg->Mix[i].as<double>();
            CRM_HIP(hipMemcpy2DAsync(Mix, ldq * sizeof(double), bb->Mbuf[i].ptr, ldm * sizeof(double), r * sizeof(double),
                                     cols, hipMemcpyDeviceToDevice, st));
            CRM_HIP(hipStreamSynchronize(st));
            bb->Mbuf[i].release();
            // Newton-Schulz polish of the orthonormality, in the space of the half factor's columns:
            // Q0'Q0 = Mix' (H'H) Mix with the Gram matrix C = H'H at hand, so a pass costs three products of
            // order cols^2 r instead of n r^2; Q0 = H Mix is then formed once.  The Gram route loses
            // orthonormality for small eigenvalues (defect ~ eps * S_max / S_j) and the path's complement
            // terms (u'v - (Q0'u)'(Q0'v)) / d see any defect directly.
            // First a look at the columns that belong to the 256 .. 383 smallest kept eigenvalues (the last ones: the defect of
            // the Gram route is ~ eps |C| / sqrt(lambda_i lambda_j), largest there): 256 / r of the cost of a pass.
            // Below 2e-13 the mixing matrix stays as the eigen-solver left it.
            {
                // (offset on a tile boundary: the operand tiles of the contraction then end with the row, ldq % 128 == 0)
                const int off = r > 256 ? (r - 256) / 128 * 128 : 0, ns = r - off;
                CRM_TRY(contract(ctx, dC.as<double>(), cp, Mix + off, ldq, dT1.as<double>(), ldq, (int)cols, ns, cp));
                CRM_TRY(contract(ctx, Mix, ldq, dT1.as<double>(), ldq, Gq, ldq, r, ns, cp));
                dim3 grid((unsigned)((ns + 255) / 256), (unsigned)r);
                hipLaunchKernelGGL(defect_slab_kernel, grid, dim3(256), 0, st, Gq, ldq, r, ns, off, dErr.as<double>());
                CRM_HIP(hipGetLastError());
                std::vector<double> herr((size_t)grid.x * grid.y);
                CRM_HIP(hipMemcpyAsync(herr.data(), dErr.ptr, sizeof(double) * herr.size(), hipMemcpyDeviceToHost, st));
                CRM_HIP(hipStreamSynchronize(st));
                double est = 0.0;
                for (double e : herr) est = std::max(est, e);
                if (trace.on) fprintf(stderr, "[crm background]     rho[%d] defect over the last %d columns: %.3g\n", i, ns, est);
                bg->ortho_defect[i] = est;
                if (est < 2e-13) continue;
                if (ns < r) CRM_HIP(hipMemsetAsync(dT1.ptr, 0, sizeof(double) * cp * ldq, st));
            }
            for (int pass = 0; pass < 3; pass++) {
                CRM_TRY(contract(ctx, dC.as<double>(), cp, Mix, ldq, dT1.as<double>(), ldq, (int)cols, r, cp));   // C Mix
                CRM_TRY(contract(ctx, Mix, ldq, dT1.as<double>(), ldq, Gq, ldq, r, r, cp));                         // Mix' C Mix
                double err = 0.0;
                CRM_TRY(defect_and_correction(r, &err));
                if (!(err < 0.5)) {
                    set_error("background: Q0 lost orthonormality at rho=%g (defect %g)", bg->rho[i], err);
                    return CRM_ERR_NUMERIC;
                }
                bg->ortho_defect[i] = err;
                if (trace.on) fprintf(stderr, "[crm background]     rho[%d] pass %d: |Mix' C Mix - I| = %.3g\n", i, pass, err);
                if (err < 2e-14 || pass == 2) break;
                // Mix <- Mix N : contraction over r with X = Mix' (r x cols)
                CRM_HIP(hipMemsetAsync(dQt.ptr, 0, sizeof(double) * ldq * cp, st));
                CRM_TRY(transpose(st, Mix, ldq, cols, r, dQt.as<double>(), cp));
                CRM_TRY(contract(ctx, dQt.as<double>(), cp, N, ldq, Mix, ldq, (int)cols, r, round_up(r, GEMM_BK)));
                // the step squares the defect: from below 1e-8 it lands at rounding level (measured 3e-14 ..
                // 7e-13 before, 1e-15 .. 6e-15 after at config 3) -- no second look needed
                if (err < 1e-8) {
                    bg->ortho_defect[i] = err * err + 8e-15;
                    break;
                }
            }
        } else {
            CRM_HIP(hipMemcpy2DAsync(bg->Q0[i].ptr, ldq * sizeof(double), bb->Mbuf[i].ptr, ldm * sizeof(double),
                                     r * sizeof(double), n, hipMemcpyDeviceToDevice, st));
            CRM_HIP(hipStreamSynchronize(st));
            bb->Mbuf[i].release();
            bg->q0_ready[i] = true;
            // eigenvectors of the n x n route: orthonormal to ~1e-14 sqrt(n) as they come; one check, and a
            // correction if ever needed
            for (int pass = 0; pass < 3; pass++) {
                CRM_TRY(contract(ctx, bg->Q0[i].as<double>(), ldq, bg->Q0[i].as<double>(), ldq, Gq, ldq, r, r, np));
                double err = 0.0;
                CRM_TRY(defect_and_correction(r, &err));
                if (!(err < 0.5)) {
                    set_error("background: Q0 lost orthonormality at rho=%g (defect %g)", bg->rho[i], err);
                    return CRM_ERR_NUMERIC;
                }
                bg->ortho_defect[i] = err;
                if (err < 2e-14 || pass == 2) break;
                // Q0 <- Q0 N : contraction over r with X = Q0' (r x cells)
                CRM_HIP(hipMemsetAsync(dQt.ptr, 0, sizeof(double) * ldq * np, st));
                CRM_TRY(transpose(st, bg->Q0[i].as<double>(), ldq, n, r, dQt.as<double>(), np));
                CRM_TRY(contract(ctx, dQt.as<double>(), np, N, ldq, bg->Q0[i].as<double>(), ldq, (int)n, r, round_up(r, GEMM_BK)));
            }
        }
    }
    CRM_HIP(hipStreamSynchronize(st));
    trace.lap(thin ? "polish of the mixing matrices" : "Q0 + polish");
    bb->completed = true;
    if (thin && eager) {
        bg->fast_T = true;   // (H and Mix are in place)
        for (int i : bb->mine) CRM_TRY(crm_background_require_q0(bg, i));
        bg->fast_T = false;
        trace.lap("Q0 = H Mix (eager)");
    }
    return CRM_OK;
}

static int background_seal(crm_background* bg) {
    crm_background_builder* bb = bg->builder;
    if (!bb || !bb->completed) {
        set_error("background: seal called out of order");
        return CRM_ERR_ARG;
    }
    CRM_HIP(hipSetDevice(bg->ctx->device));
    // the mixing-matrix route amplifies rounding by sqrt(S_max / S_min): use it only for spectra
    // whose kept part is well conditioned
    bg->fast_T = bb->thin;
    // (also kept: the largest entry of every spectrum -- scan.hip decides from it which fits have no kinship term to speak
    // of; filled here, once, before the background is shared between genes and threads)
    std::vector<double> s0;
    bg->s0_max.assign(bg->nrho, 0.0);
    for (int i = 0; i < bg->nrho; i++) {
        const int r = bg->r[i];
        if (r == 0) continue;
        s0.resize(r);
        CRM_HIP(hipMemcpy(s0.data(), bg->S0[i].ptr, sizeof(double) * r, hipMemcpyDeviceToHost));  // (imported ones too)
        double smax = 0.0, smin = 1e300;
        for (double v : s0) { smax = std::max(smax, v); smin = std::min(smin, v); }
        bg->s0_max[i] = smax;
        if (!(smax <= 1e6 * smin)) bg->fast_T = false;
    }
    const bool thin = bb->thin;
    delete bb;
    bg->builder = nullptr;
    if (!bg->fast_T) {
        if (thin) {   // the scan will rotate with Q0 itself: form all of them now, then drop H and the mixing matrices
            bg->fast_T = true;
            const int rc = crm_background_require_q0(bg, -1);
            bg->fast_T = false;
            CRM_TRY(rc);
        }
        bg->H.release();
        bg->Ht.release();
        for (int i = 0; i < bg->nrho; i++) bg->Mix[i].release();
    }
    return CRM_OK;
}
#undef CRM_BG
#undef CRM_BG_HIP

static int background_create_core(crm_ctx* ctx, long n, const double* E1, int k1, const double* B, long kb,
                                  const double* U, int k2, const double* hK, int m, int nrho,
                                  const double* rho, double rel_tol, crm_background** out) {
    crm_background* bg = nullptr;
    CRM_TRY(background_begin(ctx, n, E1, k1, B, kb, U, k2, hK, m, nrho, rho, nullptr, rel_tol, &bg));
    int rc = background_complete(bg, nullptr);
    if (rc == CRM_OK) rc = background_seal(bg);
    if (rc != CRM_OK) {
        crm_background_destroy(bg);
        return rc;
    }
    *out = bg;
    return CRM_OK;
}

extern "C" int crm_background_create(crm_ctx* ctx, long n, const double* E1, int k1, const double* B,
                                     long kb, int nrho, const double* rho, double rel_tol,
                                     crm_background** out) {
    return crm::guarded_on("crm_background_create", ctx, [&]() -> int {
    if (kb > 0 && !B) return CRM_ERR_ARG;
    return background_create_core(ctx, n, E1, k1, B, kb, nullptr, 0, nullptr, 0, nrho, rho, rel_tol, out);
    });
}

extern "C" int crm_background_create_hadamard(crm_ctx* ctx, long n, const double* E1, int k1, const double* U,
                                              int k2, const double* hK, int m, int nrho, const double* rho,
                                              double rel_tol, crm_background** out) {
    return crm::guarded_on("crm_background_create_hadamard", ctx, [&]() -> int {
    if (!U || !hK || k2 < 1 || m < 1) return CRM_ERR_ARG;
    return background_create_core(ctx, n, E1, k1, nullptr, (long)k2 * m, U, k2, hK, m, nrho, rho, rel_tol, out);
    });
}

// ---- the same constructor split over several processes (one per GPU) ------------------------------------------
extern "C" int crm_background_begin(crm_ctx* ctx, long n, const double* E1, int k1, const double* B, long kb,
                                    const double* U, int k2, const double* hK, int m, int nrho, const double* rho,
                                    const int* mine, double rel_tol, crm_background** out) {
    return crm::guarded_on("crm_background_begin", ctx, [&]() -> int {
    if (kb > 0 && !B && !(U && hK && k2 >= 1 && m >= 1 && (long)k2 * m == kb)) return CRM_ERR_ARG;
    return background_begin(ctx, n, E1, k1, B, kb, U, k2, hK, m, nrho, rho, mine, rel_tol, out);
    });
}

extern "C" int crm_background_complete(crm_background* bg, const int* ranks) {
    return crm::guarded_on("crm_background_complete", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || !ranks) return CRM_ERR_ARG;
    return background_complete(bg, ranks);
    });
}

extern "C" int crm_background_seal(crm_background* bg) {
    return crm::guarded_on("crm_background_seal", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg) return CRM_ERR_ARG;
    return background_seal(bg);
    });
}

extern "C" int crm_background_layout(const crm_background* bg, long* n_pad, long* ldq, long* ldh, int* has_mix) {
    return crm::guarded_on("crm_background_layout", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || !bg->builder || !bg->builder->completed) return CRM_ERR_ARG;
    if (n_pad) *n_pad = bg->n_pad;
    if (ldq) *ldq = bg->ldq;
    if (ldh) *ldh = bg->builder->thin ? bg->builder->cp : 0;   // > 0: exchange S0 and Mix only, Q0 = H Mix is formed locally
    if (has_mix) *has_mix = bg->builder->thin ? 1 : 0;
    return CRM_OK;
    });
}

// what: 0 = Q0 (n_pad x ldq), 1 = S0 (ldq), 2 = Mix (ldh x ldq); copies on the context's stream, synchronised before
// returning; the other side may be device memory of this GPU or host memory (the runtime tells them apart)
static int background_slot(const crm_background* bg, int i, int what, void** ptr, size_t* bytes) {
    if (!bg || !bg->builder || !bg->builder->completed || i < 0 || i >= bg->nrho) return CRM_ERR_ARG;
    const DevBuf* b = what == 0 ? &bg->Q0[i] : what == 1 ? &bg->S0[i] : what == 2 ? &bg->Mix[i] : nullptr;
    if (!b || !b->ptr) return CRM_ERR_ARG;
    *ptr = b->ptr;
    *bytes = what == 0 ? sizeof(double) * bg->n_pad * bg->ldq
                       : what == 1 ? sizeof(double) * bg->ldq : sizeof(double) * bg->builder->cp * bg->ldq;
    return CRM_OK;
}

extern "C" int crm_background_export(const crm_background* bg, int i, int what, void* dst_device) {
    return crm::guarded_on("crm_background_export", bg ? bg->ctx : nullptr, [&]() -> int {
    void* p = nullptr;
    size_t bytes = 0;
    if (bg && what == 0 && i >= 0 && i < bg->nrho && !bg->q0_ready[i] && bg->builder && bg->builder->thin) {
        set_error("background: Q0 of a thin-branch grid point is not exchanged (S0 and Mix are; Q0 = H Mix is formed locally)");
        return CRM_ERR_ARG;
    }
    if (!dst_device || background_slot(bg, i, what, &p, &bytes) != CRM_OK) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(bg->ctx->device));
    CRM_HIP(hipMemcpyAsync(dst_device, p, bytes, hipMemcpyDefault, bg->ctx->stream));   // (device or host destination)
    CRM_HIP(hipStreamSynchronize(bg->ctx->stream));
    return CRM_OK;
    });
}

extern "C" int crm_background_import(crm_background* bg, int i, int what, const void* src_device) {
    return crm::guarded_on("crm_background_import", bg ? bg->ctx : nullptr, [&]() -> int {
    void* p = nullptr;
    size_t bytes = 0;
    if (!src_device || background_slot(bg, i, what, &p, &bytes) != CRM_OK) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(bg->ctx->device));
    CRM_HIP(hipMemcpyAsync(p, src_device, bytes, hipMemcpyDefault, bg->ctx->stream));   // (device or host source)
    CRM_HIP(hipStreamSynchronize(bg->ctx->stream));
    if (what == 0) bg->q0_ready[i] = true;
    return CRM_OK;
    });
}

int crm_background_require_q0(crm_background* bg, int i) {
    if (!bg) return CRM_ERR_ARG;
    if (i < 0) {
        for (int q = 0; q < bg->nrho; q++) CRM_TRY(crm_background_require_q0(bg, q));
        return CRM_OK;
    }
    if (i >= bg->nrho) return CRM_ERR_ARG;
    if (bg->q0_ready[i]) return CRM_OK;
    crm_ctx* ctx = bg->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long np = bg->n_pad, ldq = bg->ldq;
    if (!bg->H.ptr || !bg->Mix[i].ptr) {
        set_error("background: Q0 of grid point %d is neither present nor derivable (no half factor / mixing matrix)", i);
        return CRM_ERR_ARG;
    }
    TraceRange range("crm Q0 = H Mix");
    CRM_TRY(bg->Q0[i].ensure(sizeof(double) * np * ldq));
    CRM_HIP(hipMemsetAsync(bg->Q0[i].ptr, 0, sizeof(double) * np * ldq, st));
    if (bg->r[i] > 0) {
        if (!bg->Ht.ptr) {
            CRM_TRY(bg->Ht.ensure(sizeof(double) * bg->ldh * np));
            CRM_HIP(hipMemsetAsync(bg->Ht.ptr, 0, sizeof(double) * bg->ldh * np, st));
            CRM_TRY(transpose(st, bg->H.as<double>(), bg->ldh, bg->n, bg->cols, bg->Ht.as<double>(), np));
        }
        // Q0 = H Mix  ==  Ht' Mix  (contraction over the cols axis)
        CRM_TRY(contract(ctx, bg->Ht.as<double>(), np, bg->Mix[i].as<double>(), ldq, bg->Q0[i].as<double>(), ldq,
                         (int)bg->n, bg->r[i], bg->ldh));
    }
    CRM_HIP(hipStreamSynchronize(st));
    bg->q0_ready[i] = true;
    return CRM_OK;
}

void crm_background_builder_free(crm_background_builder* b) { delete b; }
