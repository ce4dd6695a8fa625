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

// B given explicitly (kb columns), or as Hadamard factors U (n x k2) and hK (n x m) with kb = k2 * m
static int background_create_core(crm_ctx* ctx, long n, const double* E1, int k1, const double* B, long kb,
                                  const double* U, int k2, const double* hK, int m, int nrho,
                                  const double* rho, double rel_tol, crm_background** out) {
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
    const bool trace = getenv("CRM_TRACE_SETUP") != nullptr;
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[crm background] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_mark).count());
        t_mark = now;
    };
    const long cols = k1 + kb;
    const long np = round_up(n, CELL_PAD);
    const long cp = round_up(cols, 128);
    const bool thin = n > cols;  // economic_qs_linear: rows > cols -> SVD branch


    Scratch S;
    DevBuf &dH = S.bufs[0], &dHt = S.bufs[1], &dC = S.bufs[2], &dMt = S.bufs[9],
           &dG = S.bufs[10], &dErr = S.bufs[11], &dQt = S.bufs[12];
    // H = [E1, B] (cells x cols) and its transpose
    CRM_TRY(dH.ensure(sizeof(double) * np * cp));
    CRM_HIP(hipMemsetAsync(dH.ptr, 0, sizeof(double) * np * cp, st));
    CRM_HIP(hipMemcpy2DAsync(dH.ptr, cp * sizeof(double), E1, k1 * sizeof(double), k1 * sizeof(double), n,
                             hipMemcpyHostToDevice, st));
    if (kb > 0 && B) {
        CRM_HIP(hipMemcpy2DAsync(dH.as<double>() + k1, cp * sizeof(double), B, kb * sizeof(double),
                                 kb * sizeof(double), n, hipMemcpyHostToDevice, st));
    } else if (kb > 0) {
        ScopedBuf dU, dK;
        CRM_TRY(dU.ensure(sizeof(double) * n * k2));
        CRM_TRY(dK.ensure(sizeof(double) * n * m));
        CRM_HIP(hipMemcpyAsync(dU.ptr, U, sizeof(double) * n * k2, hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(dK.ptr, hK, sizeof(double) * n * m, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(hadamard_halves_kernel, dim3((unsigned)n), dim3(256), 0, st, dU.as<double>(), k2,
                           dK.as<double>(), m, n, dH.as<double>(), cp, k1);
        CRM_HIP(hipGetLastError());
        CRM_HIP(hipStreamSynchronize(st));
    }
    CRM_TRY(dHt.ensure(sizeof(double) * cp * np));
    CRM_HIP(hipMemsetAsync(dHt.ptr, 0, sizeof(double) * cp * np, st));
    CRM_TRY(transpose(st, dH.as<double>(), cp, n, cols, dHt.as<double>(), np));

    crm_background* bg = new crm_background();
    bg->ctx = ctx;
    bg->n = n;
    bg->n_pad = np;
    bg->nrho = nrho;
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

    const long dim = thin ? cols : n;       // order of the matrix that gets diagonalised
    const long dimp = round_up(dim, 128);
    std::vector<std::vector<double>> S0_host(nrho);

    if (thin) {
        // Gram matrix of the unscaled half factor, once
        CRM_BG(dC.ensure(sizeof(double) * cp * cp));
        CRM_BG(contract(ctx, dH.as<double>(), cp, dH.as<double>(), cp, dC.as<double>(), cp, (int)cols, (int)cols, np));
    }
    if (!thin) {
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
    // pass 1: eigen-decompositions of all grid points at once (eigh*.hip); ranks decide the common leading
    // dimension (Q0 buffers are allocated after all ranks are known)
    std::vector<DevBuf> Mbuf(nrho);
    struct MGuard {
        std::vector<DevBuf>& v;
        ~MGuard() { for (auto& b : v) b.release(); }
    } mguard{Mbuf};
    long rmax = 1;
    CRM_BG_HIP(hipStreamSynchronize(st));
    lap("half factor + Gram");
    {
        EighWork ew;
        struct EGuard { EighWork& w; ~EGuard() { eigh_free(w); } } eguard{ew};
        CRM_BG(eigh_alloc(ew, nrho, dim));
        CRM_BG_HIP(hipMemsetAsync(ew.A.ptr, 0, sizeof(double) * (size_t)nrho * ew.slab, st));
        for (int i = 0; i < nrho; i++) {
            bg->rho[i] = rho[i];
            const double a = std::sqrt(rho[i]), b = std::sqrt(1.0 - rho[i]);
            double* Ai = ew.A.as<double>() + (size_t)i * ew.slab;
            if (thin) {
                dim3 grid((unsigned)((cols + 255) / 256), (unsigned)cols);
                hipLaunchKernelGGL(scale_gram_kernel, grid, dim3(256), 0, st, dC.as<double>(), cp, (int)cols, k1, a, b,
                                   Ai, ew.ld);
            } else {
                // Sigma(rho) = rho E1 E1' + (1 - rho) B B'
                dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
                hipLaunchKernelGGL(combine_kernel, grid, dim3(256), 0, st, dG.as<double>(),
                                   dG.as<double>() + dimp * dimp, dimp, (int)n, rho[i], 1.0 - rho[i], Ai);
            }
        }
        CRM_BG_HIP(hipGetLastError());
        std::vector<double> lam((size_t)nrho * dim);
        double* Zt = nullptr;   // rows = eigenvectors, leading dimension ew.ld
        CRM_BG(eigh_batched(ctx, ew, lam.data(), &Zt));
        lap("eigen-decompositions");
        ScopedBuf wKeep, wLam;
        CRM_BG(wKeep.ensure(sizeof(int) * dimp));
        CRM_BG(wLam.ensure(sizeof(double) * dimp));
        for (int i = 0; i < nrho; i++) {
            const double* hW = &lam[(size_t)i * dim];   // ascending
            const double a = std::sqrt(rho[i]), b = std::sqrt(1.0 - rho[i]);
            std::vector<int> keep;
            if (thin) {
                const double cut = rel_tol * std::max(hW[dim - 1], 0.0);
                for (long j = dim - 1; j >= 0; j--)  // descending, like singular values
                    if (hW[j] > cut && hW[j] > 0.0) keep.push_back((int)j);
            } else {
                const double eps_small = 1.4901161193847656e-08;  // sqrt(machine eps), _math.py:204
                for (long j = 0; j < dim; j++)                    // ascending, like eigh
                    if (hW[j] >= eps_small) keep.push_back((int)j);
            }
            const int r = (int)keep.size();
            bg->r[i] = r;
            rmax = std::max<long>(rmax, r);
            S0_host[i].resize(r);
            for (int j = 0; j < r; j++) S0_host[i][j] = hW[keep[j]];
            // keep what pass 2 needs: thin -> mixing matrix M (cols x r); else -> the vectors themselves
            const long ldm = round_up(std::max(r, 1), 128);
            CRM_BG(Mbuf[i].ensure(sizeof(double) * (thin ? cp : np) * ldm));
            CRM_BG_HIP(hipMemsetAsync(Mbuf[i].ptr, 0, sizeof(double) * (thin ? cp : np) * ldm, st));
            if (r > 0) {
                const double* Vi = Zt + (size_t)i * ew.slab;
                CRM_BG_HIP(hipMemcpyAsync(wKeep.ptr, keep.data(), sizeof(int) * r, hipMemcpyHostToDevice, st));
                CRM_BG_HIP(hipMemcpyAsync(wLam.ptr, hW, sizeof(double) * dim, hipMemcpyHostToDevice, st));
                if (thin) {
                    dim3 grid((unsigned)((r + 255) / 256), (unsigned)cols);
                    hipLaunchKernelGGL(build_mixing_kernel, grid, dim3(256), 0, st, Vi, ew.ld, wLam.as<double>(),
                                       wKeep.as<int>(), r, (int)cols, k1, a, b, Mbuf[i].as<double>(), ldm);
                } else {
                    dim3 grid((unsigned)((r + 255) / 256), (unsigned)n);
                    hipLaunchKernelGGL(gather_vectors_kernel, grid, dim3(256), 0, st, Vi, ew.ld, wKeep.as<int>(), r, n,
                                       Mbuf[i].as<double>(), ldm);
                }
                CRM_BG_HIP(hipGetLastError());
            }
            CRM_BG_HIP(hipStreamSynchronize(st));   // keep / hW are reused by the next grid point
        }
    }
    // pass 2: Q0 buffers with the common leading dimension
    bg->ldq = round_up(rmax, 128);
    const long ldq = bg->ldq;
    dG.release();
    dMt.release();
    dC.release();
    if (thin) {
        // keep the half factor: T(rho) = Q0(rho)'G is later taken as Mix(rho)' (H'G), see scan.hip
        bg->H = dH;
        dH = DevBuf();
        bg->ldh = cp;
        bg->cols = cols;
    } else {
        dH.release();
    }
    CRM_BG(dG.ensure(sizeof(double) * ldq * ldq * 2));
    CRM_BG(dQt.ensure(sizeof(double) * ldq * np));
    CRM_BG(dErr.ensure(sizeof(double) * ((ldq + 255) / 256) * ldq));
    for (int i = 0; i < nrho; i++) {
        const int r = bg->r[i];
        CRM_BG(bg->Q0[i].ensure(sizeof(double) * np * ldq));
        CRM_BG(bg->S0[i].ensure(sizeof(double) * ldq));
        CRM_BG_HIP(hipMemsetAsync(bg->Q0[i].ptr, 0, sizeof(double) * np * ldq, st));
        CRM_BG_HIP(hipMemsetAsync(bg->S0[i].ptr, 0, sizeof(double) * ldq, st));
        if (r == 0) continue;
        CRM_BG_HIP(hipMemcpyAsync(bg->S0[i].ptr, S0_host[i].data(), sizeof(double) * r, hipMemcpyHostToDevice, st));
        const long ldm = round_up(r, 128);
        if (!thin) {
            CRM_BG_HIP(hipMemcpy2DAsync(bg->Q0[i].ptr, ldq * sizeof(double), Mbuf[i].ptr, ldm * sizeof(double),
                                        r * sizeof(double), n, hipMemcpyDeviceToDevice, st));
        } else {
            // Q0 = H M  ==  Ht' M  (contraction over the cols axis)
            CRM_BG(contract(ctx, dHt.as<double>(), np, Mbuf[i].as<double>(), ldm, bg->Q0[i].as<double>(), ldq,
                            (int)n, r, cp));
        }
        CRM_BG_HIP(hipStreamSynchronize(st));
        if (thin) {
            CRM_BG(bg->Mix[i].ensure(sizeof(double) * cp * ldq));
            CRM_BG_HIP(hipMemsetAsync(bg->Mix[i].ptr, 0, sizeof(double) * cp * ldq, st));
            CRM_BG_HIP(hipMemcpy2DAsync(bg->Mix[i].ptr, ldq * sizeof(double), Mbuf[i].ptr, ldm * sizeof(double),
                                        r * sizeof(double), cols, hipMemcpyDeviceToDevice, st));
            CRM_BG_HIP(hipStreamSynchronize(st));
        }
        Mbuf[i].release();
        // Newton-Schulz polish of the orthonormality: Q0 <- Q0 (1.5 I - 0.5 Q0'Q0).  The Gram route
        // loses it for small eigenvalues (defect ~ eps * S_max / S_j) and the library eigenvectors
        // of the n x n route carry ~1e-13; the path's complement terms (u'v - (Q0'u)'(Q0'v)) / d see
        // any defect directly.
        double* Gq = dG.as<double>();
        double* N = Gq + ldq * ldq;
        for (int pass = 0; pass < 4; pass++) {
            CRM_BG(contract(ctx, bg->Q0[i].as<double>(), ldq, bg->Q0[i].as<double>(), ldq, Gq, ldq, r, r, np));
            dim3 grid((unsigned)((r + 255) / 256), (unsigned)r);
            CRM_BG_HIP(hipMemsetAsync(N, 0, sizeof(double) * ldq * ldq, st));
            hipLaunchKernelGGL(newton_schulz_kernel, grid, dim3(256), 0, st, Gq, ldq, r, N, ldq, dErr.as<double>());
            CRM_BG_HIP(hipGetLastError());
            std::vector<double> herr((size_t)grid.x * grid.y);
            CRM_BG_HIP(hipMemcpyAsync(herr.data(), dErr.ptr, sizeof(double) * herr.size(), hipMemcpyDeviceToHost, st));
            CRM_BG_HIP(hipStreamSynchronize(st));
            double err = 0.0;
            for (double e : herr) err = std::max(err, e);
            if (!(err < 0.5)) {
                set_error("background: Q0 lost orthonormality at rho=%g (defect %g)", rho[i], err);
                return fail(CRM_ERR_NUMERIC);
            }
            bg->ortho_defect[i] = err;
            if (err < 2e-14 || pass == 2) break;
            // Q0 <- Q0 N : contraction over r with X = Q0' (r x cells)
            CRM_BG_HIP(hipMemsetAsync(dQt.ptr, 0, sizeof(double) * ldq * np, st));
            CRM_BG(transpose(st, bg->Q0[i].as<double>(), ldq, n, r, dQt.as<double>(), np));
            CRM_BG(contract(ctx, dQt.as<double>(), np, N, ldq, bg->Q0[i].as<double>(), ldq, (int)n, r, round_up(r, GEMM_BK)));
            if (thin) {
                // the same correction on the mixing matrix keeps Q0 == H Mix:  Mix <- Mix N
                CRM_BG_HIP(hipMemsetAsync(dQt.ptr, 0, sizeof(double) * ldq * std::min(np, cp), st));
                CRM_BG(transpose(st, bg->Mix[i].as<double>(), ldq, cols, r, dQt.as<double>(), cp));
                CRM_BG(contract(ctx, dQt.as<double>(), cp, N, ldq, bg->Mix[i].as<double>(), ldq, (int)cols, r,
                                round_up(r, GEMM_BK)));
            }
        }
    }
    CRM_BG_HIP(hipStreamSynchronize(st));
    lap("Q0 = H Mix + polish");
    // the mixing-matrix route amplifies rounding by sqrt(S_max / S_min): use it only for spectra
    // whose kept part is well conditioned
    bg->fast_T = thin;
    for (int i = 0; i < nrho && bg->fast_T; i++) {
        if (bg->r[i] == 0) continue;
        double smax = 0.0, smin = 1e300;
        for (double v : S0_host[i]) { smax = std::max(smax, v); smin = std::min(smin, v); }
        if (!(smax <= 1e6 * smin)) bg->fast_T = false;
    }
    if (!bg->fast_T) {
        bg->H.release();
        for (int i = 0; i < nrho; i++) bg->Mix[i].release();
    }
#undef CRM_BG
#undef CRM_BG_HIP
    *out = bg;
    return CRM_OK;
}

extern "C" int crm_background_create(crm_ctx* ctx, long n, const double* E1, int k1, const double* B,
                                     long kb, int nrho, const double* rho, double rel_tol,
                                     crm_background** out) {
    if (kb > 0 && !B) return CRM_ERR_ARG;
    return background_create_core(ctx, n, E1, k1, B, kb, nullptr, 0, nullptr, 0, nrho, rho, rel_tol, out);
}

extern "C" int crm_background_create_hadamard(crm_ctx* ctx, long n, const double* E1, int k1, const double* U,
                                              int k2, const double* hK, int m, int nrho, const double* rho,
                                              double rel_tol, crm_background** out) {
    if (!U || !hK || k2 < 1 || m < 1) return CRM_ERR_ARG;
    return background_create_core(ctx, n, E1, k1, nullptr, (long)k2 * m, U, k2, hK, m, nrho, rho, rel_tol, out);
}
