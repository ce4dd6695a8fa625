// Back-transformation of the tridiagonal eigenvectors, Z <- Q Z with Q = H_0 H_1 ... H_{dim-2}, in compact-WY
// blocks of BT_NB reflectors (LAPACK dormtr / dlarft): per block
//     Q_p = I - V T V',   Z <- Z - V (T (V' Z))
// three contractions on the FP64 matrix pipe (V' Z over the rows, T through its transpose, the update with
// GEMM_SUBTRACT), T from the block's Gram matrix V'V by the dlarft recurrence (one workgroup per block).
// Plus the driver of the whole eigen-solver and its allocation.
#include <chrono>

#include "eigh.h"

namespace crm {
namespace {

__global__ void bt_transpose_kernel(const double* __restrict__ src, long ld_src, long rows, long cols,
                                    double* __restrict__ dst, long ld_dst, long slab_src, long slab_dst) {
    __shared__ double tile[32][33];
    const double* S = src + (long)blockIdx.z * slab_src;
    double* D = dst + (long)blockIdx.z * slab_dst;
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const long r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? S[r * ld_src + c] : 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const long c = c0 + i, r = r0 + threadIdx.x;
        if (c < cols && r < rows) D[c * ld_dst + r] = tile[threadIdx.x][i];
    }
}

// T (upper triangular, nbk x nbk) of the block of reflectors j0 .. j0 + nbk - 1 from S = V'V and tau:
//   T[i][i] = tau_i,  T[0:i, i] = -tau_i T[0:i, 0:i] S[0:i, i]            (dlarft, forward, columnwise)
// written TRANSPOSED (Tt[m][l] = T[l][m], leading dimension BT_NB) for the contraction kernel.
__global__ __launch_bounds__(BT_NB) void bt_larft_kernel(const double* __restrict__ S_all, const double* __restrict__ tau_all,
                                                         long ld, int nblocks, long dim, double* __restrict__ Tt_all) {
    extern __shared__ double T[];   // [BT_NB][BT_NB + 1]
    __shared__ double col[BT_NB];
    const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const long j0 = (long)p * BT_NB;
    const int nbk = (int)((dim - 1 - j0) < BT_NB ? (dim - 1 - j0) : BT_NB);   // reflectors 0 .. dim - 2
    const double* S = S_all + ((long)b * nblocks + p) * BT_NB * BT_NB;
    const double* tau = tau_all + (long)b * ld + j0;
    double* Tt = Tt_all + ((long)b * nblocks + p) * BT_NB * BT_NB;
    constexpr int LT = BT_NB + 1;
    for (int idx = tid; idx < BT_NB * LT; idx += BT_NB) T[idx] = 0.0;
    __syncthreads();
    for (int i = 0; i < nbk; i++) {
        const double ti = tau[i];
        // col[l] = -tau_i * sum_{m < i} T[l][m] S[m][i]   for l < i   (T upper triangular: m >= l)
        double acc = 0.0;
        if (tid < i) {
            for (int m = tid; m < i; m++) acc += T[tid * LT + m] * S[(long)m * BT_NB + i];
            col[tid] = -ti * acc;
        }
        __syncthreads();
        if (tid < i) T[tid * LT + i] = col[tid];
        if (tid == i) T[i * LT + i] = ti;
        __syncthreads();
    }
    for (int idx = tid; idx < BT_NB * BT_NB; idx += BT_NB) {
        const int m = idx / BT_NB, l = idx - m * BT_NB;
        Tt[idx] = T[l * LT + m];
    }
}

// T of two consecutive blocks a, b of BT_NB reflectors as one block of 2 BT_NB:
//     (I - V_a T_a V_a')(I - V_b T_b V_b') = I - [V_a V_b] [[T_a, -T_a S_ab T_b], [0, T_b]] [V_a V_b]',  S_ab = V_a'V_b,
// written transposed with leading dimension 2 BT_NB (Ttw[m][l] = T[l][m]).  G: the pair's 2 BT_NB x 2 BT_NB Gram matrix,
// Tta / Ttb: the halves' transposed T, M1: BT_NB x BT_NB scratch.  One workgroup per pair.
__global__ __launch_bounds__(256) void bt_merge_kernel(const double* __restrict__ G_all, const double* __restrict__ Tt_all,
                                                       int nblocks, int npairs, double* __restrict__ M1_all,
                                                       double* __restrict__ Ttw_all) {
    constexpr int NB = BT_NB, W2 = 2 * BT_NB;
    const int q = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const double* G = G_all + ((size_t)b * npairs + q) * W2 * W2;
    const double* Tta = Tt_all + ((size_t)b * nblocks + 2 * q) * NB * NB;
    const double* Ttb = Tta + (size_t)NB * NB;
    double* M1 = M1_all + ((size_t)b * npairs + q) * NB * NB;
    double* Ttw = Ttw_all + ((size_t)b * npairs + q) * W2 * W2;
    // M1 = S_ab T_b  (T_b upper triangular: T_b[k][j] = Ttb[j][k], k <= j)
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e - i * NB;
        double acc = 0.0;
        for (int k = 0; k <= j; k++) acc += G[(size_t)i * W2 + NB + k] * Ttb[(size_t)j * NB + k];
        M1[e] = acc;
    }
    __threadfence_block();
    __syncthreads();
    for (int e = tid; e < W2 * W2; e += 256) {
        const int m = e / W2, l = e - m * W2;       // Ttw[m][l] = T[l][m]
        double v = 0.0;
        if (l < NB && m < NB) v = Tta[(size_t)m * NB + l];
        else if (l >= NB && m >= NB) v = Ttb[(size_t)(m - NB) * NB + (l - NB)];
        else if (l < NB && m >= NB) {               // X[l][m - NB] = -sum_k T_a[l][k] M1[k][m - NB],  T_a[l][k] = Tta[k][l], k >= l
            double acc = 0.0;
            for (int k = l; k < NB; k++) acc += Tta[(size_t)k * NB + l] * M1[(size_t)k * NB + (m - NB)];
            v = -acc;
        }
        Ttw[e] = v;
    }
}

}  // namespace

int launch_transpose(hipStream_t st, const double* src, long ld_src, long rows, long cols, double* dst, long ld_dst) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), 1);
    hipLaunchKernelGGL(bt_transpose_kernel, grid, dim3(32, 8), 0, st, src, ld_src, rows, cols, dst, ld_dst, 0L, 0L);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

static int transpose_batch(hipStream_t st, int B, const double* src, double* dst, long slab, long ld, long rows, long cols) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)B);
    hipLaunchKernelGGL(bt_transpose_kernel, grid, dim3(32, 8), 0, st, src, ld, rows, cols, dst, ld, slab, slab);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

int eigh_alloc(EighWork& w, int batch, long dim) {
    w.batch = batch;
    w.dim = dim;
    w.dimp = round_up(dim, 128);
    w.ld = w.dimp;
    w.slab = w.dimp * w.ld + 256;
    const size_t mat = sizeof(double) * (size_t)batch * w.slab;
    for (DevBuf* b : {&w.A, &w.Vt, &w.Vc, &w.QA, &w.QB}) CRM_TRY(b->ensure(mat));
    for (DevBuf* b : {&w.d, &w.e, &w.tau, &w.lam}) CRM_TRY(b->ensure(sizeof(double) * (size_t)batch * w.ld));
    return CRM_OK;
}

void eigh_free(EighWork& w) {
    for (DevBuf* b : {&w.A, &w.Vt, &w.Vc, &w.QA, &w.QB, &w.d, &w.e, &w.tau, &w.lam, &w.small, &w.AB, &w.Vbc, &w.taubc, &w.Tbc, &w.s1,
                      &w.sync})
        b->release();
}

// Qt: rows = eigenvectors of the tridiagonal (sorted); returns *Zt: rows = eigenvectors of A.
int eigh_rows_to_columns(crm_ctx* ctx, EighWork& w, const double* Qt) {
    hipStream_t st = ctx->stream;
    CRM_HIP(hipMemsetAsync(w.A.ptr, 0, sizeof(double) * (size_t)w.batch * w.slab, st));
    return transpose_batch(st, w.batch, Qt, w.A.as<double>(), w.slab, w.ld, w.dim, w.dim);
}

int eigh_back_transform(crm_ctx* ctx, EighWork& w, double* Qt, double** Zt_out, bool z_ready) {
    hipStream_t st = ctx->stream;
    const long dim = w.dim, ld = w.ld, dimp = w.dimp, slab = w.slab;
    const int B = w.batch;
    // (two-stage solver: one set of reflectors -- slab 0 of Vt, row 0 of tau -- serves every matrix of the batch)
    const int VB = w.v_shared ? 1 : B;
    auto vb = [&](int b) { return w.v_shared ? 0 : b; };
    double* other = Qt == w.QA.as<double>() ? w.QB.as<double>() : w.QA.as<double>();
    if (dim < 2) {   // no reflectors
        *Zt_out = Qt;
        return CRM_OK;
    }
    // Z (column j = eigenvector j) in the A slab, Vc = Vt'
    double* Z = w.A.as<double>();
    double* Vc = w.Vc.as<double>();
    const double* Vt = w.Vt.as<double>();
    if (!z_ready) CRM_TRY(eigh_rows_to_columns(ctx, w, Qt));
    CRM_HIP(hipMemsetAsync(Vc, 0, sizeof(double) * (size_t)VB * slab, st));
    CRM_TRY(transpose_batch(st, VB, Vt, Vc, slab, ld, dim, dim));
    const int nblocks = (int)((dim - 1 + BT_NB - 1) / BT_NB);
    // Blocks of 2 BT_NB reflectors where the matrices are large: the update Z -= V (T (V'Z)) is a product over the
    // block's reflectors only, i.e. bound by the traffic of Z -- twice the reflectors per pass, half the passes.  The last
    // block of an odd count stays BT_NB wide.
    const int npairs = dim >= 2048 ? nblocks / 2 : 0;
    constexpr int WB = 2 * BT_NB;
    // scratch: S and Tt per (matrix, block), pair Gram matrices, merge scratch and wide T, W1 / W2 per matrix, problem records
    const size_t tt = (size_t)B * nblocks * BT_NB * BT_NB;
    const size_t tw = (size_t)B * std::max(npairs, 1) * WB * WB, tm = (size_t)B * std::max(npairs, 1) * BT_NB * BT_NB;
    const size_t wsz = (size_t)WB * ld + 256;
    const size_t need = sizeof(double) * (2 * tt + 2 * tw + tm + 2 * (size_t)B * wsz) + sizeof(GemmProblem) * (size_t)B * (nblocks + 3);
    CRM_TRY(w.small.ensure(need));
    double* S = w.small.as<double>();
    double* Tt = S + tt;
    double* Gw = Tt + tt;
    double* Ttw = Gw + tw;
    double* M1 = Ttw + tw;
    double* W1 = M1 + tm;
    double* W2 = W1 + (size_t)B * wsz;
    GemmProblem* d_probs = reinterpret_cast<GemmProblem*>(W2 + (size_t)B * wsz);
    std::vector<GemmProblem> probs((size_t)VB * nblocks);
    // S_p = V_p' V_p for every block (one 128 x 128 tile each, split over the rows)
    for (int b = 0; b < VB; b++)
        for (int p = 0; p < nblocks; p++) {
            GemmProblem g{};
            const long j0 = (long)p * BT_NB;
            g.X = Vc + (size_t)b * slab + j0; g.ldx = ld;
            g.Y = g.X; g.ldy = ld;
            g.C = S + ((size_t)b * nblocks + p) * BT_NB * BT_NB; g.ldc = BT_NB;
            g.M = BT_NB; g.N = BT_NB;
            probs[(size_t)b * nblocks + p] = g;
        }
    CRM_HIP(hipMemsetAsync(S, 0, sizeof(double) * (2 * tt + 2 * tw + tm), st));
    CRM_HIP(hipMemcpyAsync(d_probs, probs.data(), sizeof(GemmProblem) * probs.size(), hipMemcpyHostToDevice, st));
    CRM_TRY(launch_gemm_tn(ctx, d_probs, (int)probs.size(), BT_NB, BT_NB, dimp, false, 0, 1, 0));
    CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bt_larft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(double) * BT_NB * (BT_NB + 1))));
    hipLaunchKernelGGL(bt_larft_kernel, dim3(nblocks, VB), dim3(BT_NB), sizeof(double) * BT_NB * (BT_NB + 1), st, S,
                       w.tau.as<double>(), ld, nblocks, dim, Tt);
    CRM_HIP(hipGetLastError());
    CRM_HIP(hipStreamSynchronize(st));
    if (npairs > 0) {
        // pair Gram matrices [V_a V_b]'[V_a V_b] (their off-diagonal block is what the merge needs), then the wide T
        std::vector<GemmProblem> pg((size_t)VB * npairs);
        for (int b = 0; b < VB; b++)
            for (int q = 0; q < npairs; q++) {
                GemmProblem g{};
                g.X = Vc + (size_t)b * slab + (long)q * WB; g.ldx = ld;
                g.Y = g.X; g.ldy = ld;
                g.C = Gw + ((size_t)b * npairs + q) * WB * WB; g.ldc = WB;
                g.M = WB; g.N = WB;
                pg[(size_t)b * npairs + q] = g;
            }
        CRM_HIP(hipMemcpyAsync(d_probs, pg.data(), sizeof(GemmProblem) * pg.size(), hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_probs, (int)pg.size(), WB, WB, dimp, false, 0, 1, 0));
        hipLaunchKernelGGL(bt_merge_kernel, dim3(npairs, VB), dim3(256), 0, st, Gw, Tt, nblocks, npairs, M1, Ttw);
        CRM_HIP(hipGetLastError());
        CRM_HIP(hipStreamSynchronize(st));
    }
    // blocks from the last to the first: (a trailing narrow block,) then the pairs
    GemmProblem* d_p3 = d_probs + (size_t)B * nblocks;
    std::vector<GemmProblem> p3(3 * (size_t)B);
    auto apply = [&](long j0, int width, const double* Tblk, size_t t_stride) -> int {
        const long r0 = j0 / 16 * 16;              // the block's vectors vanish above row j0 + 1
        for (int b = 0; b < B; b++) {
            GemmProblem g{};
            // W1 (width x dim) = V_p' Z   over the rows r0 .. dimp
            g.X = Vc + (size_t)vb(b) * slab + (size_t)r0 * ld + j0; g.ldx = ld;
            g.Y = Z + (size_t)b * slab + (size_t)r0 * ld; g.ldy = ld;
            g.C = W1 + (size_t)b * wsz; g.ldc = ld;
            g.M = width; g.N = (int)dim;
            p3[b] = g;
            // W2 = T W1  ==  Tt' W1
            GemmProblem h{};
            h.X = Tblk + (size_t)vb(b) * t_stride; h.ldx = width;
            h.Y = W1 + (size_t)b * wsz; h.ldy = ld;
            h.C = W2 + (size_t)b * wsz; h.ldc = ld;
            h.M = width; h.N = (int)dim;
            p3[B + b] = h;
            // Z[r0:, :] -= V_p W2  ==  (Vt rows j0 .., columns r0 ..)' W2
            GemmProblem u{};
            u.X = Vt + (size_t)vb(b) * slab + (size_t)j0 * ld + r0; u.ldx = ld;
            u.Y = W2 + (size_t)b * wsz; u.ldy = ld;
            u.C = Z + (size_t)b * slab + (size_t)r0 * ld; u.ldc = ld;
            u.M = (int)(dim - r0); u.N = (int)dim;
            u.flags = GEMM_SUBTRACT;
            p3[2 * B + b] = u;
        }
        CRM_HIP(hipMemcpyAsync(d_p3, p3.data(), sizeof(GemmProblem) * p3.size(), hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_p3, B, width, (int)dim, dimp - r0, false, 0, 1, 0));
        CRM_TRY(launch_gemm_tn(ctx, d_p3 + B, B, width, (int)dim, width, false, 0, 1, 0));
        CRM_TRY(launch_gemm_tn(ctx, d_p3 + 2 * B, B, (int)(dim - r0), (int)dim, width, false, 0, 1, 0));
        CRM_HIP(hipStreamSynchronize(st));   // p3 is rewritten for the next block
        return CRM_OK;
    };
    for (int p = nblocks - 1; p >= 2 * npairs; p--)
        CRM_TRY(apply((long)p * BT_NB, BT_NB, Tt + (size_t)p * BT_NB * BT_NB, (size_t)nblocks * BT_NB * BT_NB));
    for (int q = npairs - 1; q >= 0; q--)
        CRM_TRY(apply((long)q * WB, WB, Ttw + (size_t)q * WB * WB, (size_t)npairs * WB * WB));
    // rows = eigenvectors again
    CRM_HIP(hipMemsetAsync(other, 0, sizeof(double) * (size_t)B * slab, st));
    CRM_TRY(transpose_batch(st, B, Z, other, slab, ld, dim, dim));
    CRM_HIP(hipStreamSynchronize(st));
    *Zt_out = other;
    return CRM_OK;
}

int eigh_batched(crm_ctx* ctx, EighWork& w, double* lam_host, double** Zt) {
    const bool trace = getenv("CRM_TRACE_SETUP") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(ctx->stream);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[crm eigh %d x %ld] %-24s %.3f s\n", w.batch, w.dim, what, std::chrono::duration<double>(now - t0).count());
        t0 = now;
    };
    {
        TraceRange r("crm eigh tridiagonalisation");
        CRM_TRY(eigh_tridiagonalise(ctx, w));
    }
    lap("tridiagonalisation");
    double* Qt = nullptr;
    {
        TraceRange r("crm eigh divide & conquer");
        CRM_TRY(eigh_dc(ctx, w, lam_host, &Qt));
    }
    lap("divide & conquer");
    {
        TraceRange r("crm eigh back-transformation");
        CRM_TRY(eigh_back_transform(ctx, w, Qt, Zt));
    }
    lap("back-transformation");
    return CRM_OK;
}

}  // namespace crm

// ---- test hook: the solver on host matrices ---------------------------------------------------------------------
extern "C" int crm_test_eigh(crm_ctx* ctx, int batch, int dim, const double* A, double* lam, double* Z, int stage,
                             double* d_out, double* e_out) {
    return crm::guarded_on("crm_test_eigh", ctx, [&]() -> int {
    using namespace crm;
    if (!ctx || batch < 1 || dim < 1 || !A || !lam) return CRM_ERR_ARG;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    EighWork w;
    struct Guard { EighWork& w; ~Guard() { eigh_free(w); } } guard{w};
    CRM_TRY(eigh_alloc(w, batch, dim));
    CRM_HIP(hipMemsetAsync(w.A.ptr, 0, sizeof(double) * (size_t)batch * w.slab, st));
    for (int b = 0; b < batch; b++)
        CRM_HIP(hipMemcpy2DAsync(w.A.as<double>() + (size_t)b * w.slab, w.ld * sizeof(double), A + (size_t)b * dim * dim,
                                 dim * sizeof(double), dim * sizeof(double), dim, hipMemcpyHostToDevice, st));
    CRM_TRY(eigh_tridiagonalise(ctx, w));
    if (d_out) CRM_HIP(hipMemcpy2DAsync(d_out, dim * sizeof(double), w.d.ptr, w.ld * sizeof(double), dim * sizeof(double), batch,
                                        hipMemcpyDeviceToHost, st));
    if (e_out) CRM_HIP(hipMemcpy2DAsync(e_out, dim * sizeof(double), w.e.ptr, w.ld * sizeof(double), dim * sizeof(double), batch,
                                        hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    if (stage == 1) return CRM_OK;
    double* Qt = nullptr;
    CRM_TRY(eigh_dc(ctx, w, lam, &Qt));
    double* Zt = Qt;
    if (stage != 2) CRM_TRY(eigh_back_transform(ctx, w, Qt, &Zt));
    if (Z) {
        // Z[b] (dim x dim, row-major) with COLUMN j = eigenvector j: transpose of the row storage
        std::vector<double> rows((size_t)dim * dim);
        for (int b = 0; b < batch; b++) {
            CRM_HIP(hipMemcpy2D(rows.data(), dim * sizeof(double), Zt + (size_t)b * w.slab, w.ld * sizeof(double),
                                dim * sizeof(double), dim, hipMemcpyDeviceToHost));
            double* out = Z + (size_t)b * dim * dim;
            for (int j = 0; j < dim; j++)
                for (int r = 0; r < dim; r++) out[(size_t)r * dim + j] = rows[(size_t)j * dim + r];
        }
    }
    return CRM_OK;
    });
}
