// Effect-size building blocks: the two device operations behind
//   CellRegMap.predict_interaction            cellregmap/_cellregmap.py:137-205
//   CellRegMap.estimate_aggregate_environment cellregmap/_cellregmap.py:207-244
// Both fit LMM(y, M = [W, g, E0], QS(rho), restricted=True) over the rho grid, keep the best fit
// and apply cov(y)^-1 = (v0 Q0 S0 Q0' + v1 I)^-1 to the residual y - M beta (QSCov.solve,
// cellregmap/_math.py:40-67).  The per-SNP covariance halves [sqrt(rho) g o E0, sqrt(1-rho) L..]
// are ordinary backgrounds (crm_background_create*), so the pieces exposed here are
//   crm_lmm_fit    best fit over the rho grid of a gene's background + fixed effects beta
//   crm_cov_solve  K^-1 rhs for one grid point of a background
#include <algorithm>

#include "nullfit.h"
#include "objects.h"

using namespace crm;

namespace {

constexpr int CMAX = CRM_MAX_COV_XWIDE;  // layout constant of the fastscan_prep record (assoc.hip)

// w_j = d_j t_j,  d_j = v0 S0_j / (v0 S0_j + v1), for m right-hand sides (rows of T)
__global__ void shrink_rotation_kernel(double* __restrict__ T, long ldT, int m, const double* __restrict__ S0,
                                       int r, double v0, double v1) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= r) return;
    const double s = v0 * S0[j];
    const double d = s / (s + v1);
    for (int k = 0; k < m; k++) T[(long)k * ldT + j] *= d;
}

// out[i, k] = (rhs[i, k] - Q0[i, :] . w_k) / v1 ; one wavefront per cell, lanes stride the spectrum
__global__ __launch_bounds__(256) void cov_solve_kernel(const double* __restrict__ Q0, long ldq, int r,
                                                         const double* __restrict__ Wt, long ldT,
                                                         const double* __restrict__ rhs, long ldr, int m,
                                                         long n, double inv_v1, double* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    const double* __restrict__ q = Q0 + row * ldq;
    for (int k = 0; k < m; k++) {
        const double* __restrict__ w = Wt + (long)k * ldT;
        double s = 0.0;
        for (int j = lane; j < r; j += 64) s += q[j] * w[j];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) out[row * m + k] = (rhs[row * ldr + k] - s) * inv_v1;
    }
}

}  // namespace

extern "C" int crm_lmm_fit(crm_gene* gene, int restricted, double* out_fit, double* out_beta) {
    return crm::guarded_on("crm_lmm_fit", gene ? gene->ctx : nullptr, [&]() -> int {
    if (!gene || !out_fit) return CRM_ERR_ARG;
    crm_background* bg = gene->bg;
    crm_ctx* ctx = bg->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long n = bg->n, ldq = bg->ldq;
    const int nrho = bg->nrho, c = gene->c;
    const long slab = (long)(1 + c) * ldq;
    const long ld_gW = round_up(std::max(c, 8), 8);

    ScopedBuf small;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_zero = carve(sizeof(double) * ldq), o_g3 = carve(sizeof(double) * (2 + ld_gW)),
                 o_trial = carve(sizeof(NullFitTrial) * nrho), o_fit = carve(sizeof(NullFitOut)),
                 o_prep = carve(sizeof(double) * fastscan_prep_doubles()), o_wts = carve(sizeof(double) * ldq);
    CRM_TRY(small.ensure(off));
    char* sm = small.as<char>();
    double* d_zero = (double*)(sm + o_zero);
    double* d_g3 = (double*)(sm + o_g3);
    CRM_HIP(hipMemsetAsync(sm, 0, off, st));

    // a zero "variant" is dropped by the fit kernels (rank-deficient [M, 0]): exactly LMM(y, M)
    NullFitArgs fa{};
    fa.nrho = nrho; fa.c = c; fa.restricted = restricted ? 1 : 0;
    fa.polish = (ctx->polish && c <= CRM_MAX_COV) ? 1 : 0; fa.exact = (ctx->nullfit_exact || form("nullfit_exact", 0)) ? 1 : 0;
    fa.n = n;
    for (int i = 0; i < nrho; i++) {
        NullFitRho& R = fa.rho[i];
        R.T = d_zero; R.ldT = 0;
        R.ty = gene->rot.as<double>() + (long)i * slab;
        R.tW = R.ty + ldq; R.ldW = ldq;
        R.S0 = bg->S0[i].as<double>();
        R.r = bg->r[i];
    }
    fa.WW = gene->WW.as<double>(); fa.Wy = gene->Wy.as<double>(); fa.yy = gene->yy;
    fa.gg = d_g3; fa.gy = d_g3 + 1; fa.gW = d_g3 + 2; fa.ld_gW = ld_gW;
    fa.trial = (NullFitTrial*)(sm + o_trial);
    fa.out = (NullFitOut*)(sm + o_fit);
    ScopedBuf xwide;
    if (c > CRM_MAX_COV_WIDE) {   // 63 .. 128 columns: the slower kernel with its scratch in global memory
        CRM_TRY(xwide.ensure(sizeof(double) * nullfit_xwide_scratch_doubles(1, nrho, c)));
        fa.xwide = xwide.as<double>();
    }
    CRM_TRY(launch_nullfit(st, fa, 1));
    NullFitOut fit{};
    CRM_HIP(hipMemcpyAsync(&fit, fa.out, sizeof fit, hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    if (!std::isfinite(fit.lml) || fit.rho_index < 0 || fit.rho_index >= bg->nrho) {
        set_error("lmm_fit: the model could not be fitted (non-finite log-likelihood)");
        return CRM_ERR_NUMERIC;
    }
    const int ri = fit.rho_index;
    out_fit[0] = bg->rho[ri];
    out_fit[1] = fit.v0;
    out_fit[2] = fit.v1;
    out_fit[3] = fit.lml;
    out_fit[4] = fit.delta;
    out_fit[5] = (double)ri;
    if (!out_beta) return CRM_OK;

    // beta = (M'K^-1M)^-1 M'K^-1 y at the optimum (glimix-core LMM.beta): the Cholesky factor and
    // L^-1 M'K^-1y of the frozen-delta record, back-substituted here
    AssocArgs aa{};
    aa.ty = fa.rho[ri].ty; aa.tW = fa.rho[ri].tW; aa.ldW = ldq; aa.S0 = fa.rho[ri].S0;
    aa.r = bg->r[ri]; aa.c = c; aa.n = n; aa.delta0 = fit.delta;
    aa.WW = fa.WW; aa.Wy = fa.Wy; aa.yy = fa.yy;
    double* d_prep = (double*)(sm + o_prep);
    CRM_TRY(launch_fastscan_prep(st, aa, d_prep, (double*)(sm + o_wts)));
    std::vector<double> prep(fastscan_prep_doubles());
    CRM_HIP(hipMemcpyAsync(prep.data(), d_prep, sizeof(double) * prep.size(), hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    if (prep[3] == 0.0) {
        set_error("lmm_fit: M'K^-1M is not positive definite (rank-deficient fixed effects)");
        return CRM_ERR_NUMERIC;
    }
    const double* zy = prep.data() + 8;
    const double* L = prep.data() + 8 + CMAX;
    for (int i = c - 1; i >= 0; i--) {
        double s = zy[i];
        for (int k = i + 1; k < c; k++) s -= L[k * CMAX + i] * out_beta[k];
        out_beta[i] = s / L[i * CMAX + i];
    }
    if (!gene->W_basis.empty()) {
        // crm_gene_create replaced correlated columns by W V (mutually orthogonal): M b' = (M V) b  <=>  b' = V b
        std::vector<double> b(out_beta, out_beta + c);
        for (int a = 0; a < c; a++) {
            double acc = 0.0;
            for (int k = 0; k < c; k++) acc += gene->W_basis[(size_t)a * c + k] * b[k];
            out_beta[a] = acc;
        }
    }
    return CRM_OK;
    });
}

extern "C" int crm_cov_solve(crm_background* bg, int rho_index, double v0, double v1, const double* rhs,
                             int m, double* out) {
    return crm::guarded_on("crm_cov_solve", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || !rhs || !out) return CRM_ERR_ARG;
    if (rho_index < 0 || rho_index >= bg->nrho || m < 1 || !(v1 > 0.0)) {
        set_error("cov_solve: rho_index=%d (grid of %d), m=%d, v1=%g", rho_index, bg->nrho, m, v1);
        return CRM_ERR_ARG;
    }
    crm_ctx* ctx = bg->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long n = bg->n, np = bg->n_pad, ldq = bg->ldq;
    const int r = bg->r[rho_index];
    const long ldr = round_up(m, 128);
    ScopedBuf d_rhs, d_T, d_out, d_prob;
    CRM_TRY(d_rhs.ensure(sizeof(double) * np * ldr));
    CRM_TRY(d_T.ensure(sizeof(double) * ldr * ldq));
    CRM_TRY(d_out.ensure(sizeof(double) * n * m));
    CRM_TRY(d_prob.ensure(sizeof(GemmProblem)));
    CRM_TRY(upload_padded(st, d_rhs.as<double>(), ldr, np, rhs, m, n, m));
    CRM_TRY(crm_background_require_q0(bg, rho_index));
    if (r > 0) {
        GemmProblem p{};
        p.X = d_rhs.as<double>(); p.ldx = ldr;
        p.Y = bg->Q0[rho_index].as<double>(); p.ldy = ldq;
        p.C = d_T.as<double>(); p.ldc = ldq; p.M = m; p.N = r;
        CRM_HIP(hipMemcpyAsync(d_prob.ptr, &p, sizeof p, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_prob.as<GemmProblem>(), 1, m, (int)ldq, np, false, 0, 1, 0));
        hipLaunchKernelGGL(shrink_rotation_kernel, dim3((r + 255) / 256), dim3(256), 0, st, d_T.as<double>(), ldq,
                           m, bg->S0[rho_index].as<double>(), r, v0, v1);
        CRM_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(cov_solve_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st,
                       bg->Q0[rho_index].as<double>(), ldq, r, d_T.as<double>(), ldq, d_rhs.as<double>(), ldr, m, n,
                       1.0 / v1, d_out.as<double>());
    CRM_HIP(hipGetLastError());
    CRM_HIP(hipMemcpyAsync(out, d_out.ptr, sizeof(double) * n * m, hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    return CRM_OK;
    });
}
