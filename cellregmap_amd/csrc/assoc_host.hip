// Host orchestration of the association scans (C-ABI crm_scan_association).
// Reference: cellregmap/_cellregmap.py:246-314 and :443-469.
#include <algorithm>

#include "nullfit.h"
#include "objects.h"

using namespace crm;

extern "C" int crm_scan_association(crm_gene* gene, crm_panel* panel, long first, long count, int fast,
                                    double* out_pvalue, double* out_alt_lml, double* out_null) {
    return crm::guarded_on("crm_scan_association", gene ? gene->ctx : nullptr, [&]() -> int {
    if (!gene || !panel) return CRM_ERR_ARG;
    crm_background* bg = gene->bg;
    crm_ctx* ctx = bg->ctx;
    if (panel->ctx != ctx || panel->n != bg->n) {
        set_error("association: gene and panel do not match (context / cell count)");
        return CRM_ERR_ARG;
    }
    if (first < 0 || count < 0 || first + count > panel->p) {
        set_error("association: variants [%ld, %ld) outside the panel (p = %ld)", first, first + count, panel->p);
        return CRM_ERR_ARG;
    }
    if (ctx->in_scan) {
        set_error("association: another scan is running on this context (started from a progress callback?)");
        return CRM_ERR_UNSUPPORTED;
    }
    struct InScan { crm_ctx* c; explicit InScan(crm_ctx* c_) : c(c_) { c->in_scan = true; } ~InScan() { c->in_scan = false; } } in_scan(ctx);
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long n = bg->n, np = bg->n_pad, ldq = bg->ldq;
    const int nrho = bg->nrho, c = gene->c;
    const long slab = (long)(1 + c) * ldq;
    const int BLK = (int)std::min<long>(ctx->block_variants > 0 ? ctx->block_variants : CRM_DEFAULT_BLOCK,
                                        round_up(std::max<long>(count, 1), 128));
    const long ldb = BLK + 128, ldT = ldq;
    const long ld_gW = round_up(c, 8);

    CRM_TRY(ctx->ws_T.ensure(sizeof(double) * (size_t)BLK * ldT));
    CRM_TRY(ctx->ws_Gb.ensure(sizeof(double) * (size_t)np * ldb));
    if (!fast) CRM_TRY(ctx->ws_Gx.ensure(sizeof(double) * (size_t)np * ldb));
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_gg = carve(sizeof(double) * BLK), o_gy = carve(sizeof(double) * BLK),
                 o_gW = carve(sizeof(double) * BLK * ld_gW),
                 o_trial = carve(sizeof(NullFitTrial) * std::max(BLK, nrho) * nrho),
                 o_fit = carve(sizeof(NullFitOut) * BLK), o_lml = carve(sizeof(double) * BLK),
                 o_pv = carve(sizeof(double) * BLK), o_prep = carve(sizeof(double) * fastscan_prep_doubles()),
                 o_wts = carve(sizeof(double) * ldq), o_zero = carve(sizeof(double) * ldq),
                 o_part = carve(variant_stats_workspace(BLK, std::min(c, CRM_MAX_COV))),
                 o_coef = carve(sizeof(double) * (size_t)c * ldb), o_thr = carve(sizeof(double) * BLK),
                 o_drop = carve(sizeof(int) * BLK);
    CRM_TRY(ctx->ws_small.ensure(off));
    char* sm = ctx->ws_small.as<char>();
    double* d_gg = (double*)(sm + o_gg);
    double* d_gy = (double*)(sm + o_gy);
    double* d_gW = (double*)(sm + o_gW);
    NullFitTrial* d_trial = (NullFitTrial*)(sm + o_trial);
    NullFitOut* d_fit = (NullFitOut*)(sm + o_fit);
    double* d_lml = (double*)(sm + o_lml);
    double* d_pv = (double*)(sm + o_pv);
    double* d_prep = (double*)(sm + o_prep);
    double* d_wts = (double*)(sm + o_wts);
    double* d_zero = (double*)(sm + o_zero);
    double* d_part = (double*)(sm + o_part);
    double* d_coef = (double*)(sm + o_coef);
    double* d_thr = (double*)(sm + o_thr);
    int* d_drop = (int*)(sm + o_drop);
    CRM_TRY(ctx->ws_probs.ensure(sizeof(GemmProblem) * (CRM_MAX_RHO + 4)));
    const double* d_y = gene->yW.as<double>();
    const double* d_W = gene->yW.as<double>() + 1;

    // ---- null model: ML fit with X = W over the rho grid (_cellregmap.py:250-266) ------------------
    // A zero "variant" makes [W, g] rank deficient, so the fit kernel drops g: exactly LMM(y, W).
    CRM_HIP(hipMemsetAsync(d_zero, 0, sizeof(double) * ldq, st));
    CRM_HIP(hipMemsetAsync(d_gg, 0, sizeof(double), st));
    CRM_HIP(hipMemsetAsync(d_gy, 0, sizeof(double), st));
    CRM_HIP(hipMemsetAsync(d_gW, 0, sizeof(double) * ld_gW, st));
    NullFitArgs fa{};
    fa.nrho = nrho; fa.c = c; fa.restricted = 0; fa.polish = (ctx->polish && c <= CRM_MAX_COV) ? 1 : 0; fa.exact = (ctx->nullfit_exact || form("nullfit_exact", 0)) ? 1 : 0; fa.n = n;
    for (int i = 0; i < nrho; i++) {
        NullFitRho& R = fa.rho[i];
        R.T = d_zero; R.ldT = 0;
        R.ty = gene->rot.as<double>() + (long)i * slab;
        R.tW = R.ty + ldq; R.ldW = ldq;
        R.S0 = bg->S0[i].as<double>();
        R.r = bg->r[i];
    }
    fa.WW = gene->WW.as<double>(); fa.Wy = gene->Wy.as<double>(); fa.yy = gene->yy;
    fa.gg = d_gg; fa.gy = d_gy; fa.gW = d_gW; fa.ld_gW = ld_gW;
    fa.trial = d_trial; fa.out = d_fit;
    ScopedBuf xwide;
    if (c > CRM_MAX_COV_WIDE) {   // 63 .. 128 columns (run_association binds the contexts here): nullfit_xwide.hip
        CRM_TRY(xwide.ensure(sizeof(double) * nullfit_xwide_scratch_doubles(std::max(BLK, 1), nrho, c)));
        fa.xwide = xwide.as<double>();
    }
    CRM_TRY(launch_nullfit(st, fa, 1));
    NullFitOut null{};
    CRM_HIP(hipMemcpyAsync(&null, d_fit, sizeof null, hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    const int ri = null.rho_index;
    if (ri < 0 || ri >= nrho) {
        set_error("association: the null model's fit did not run (grid index %d)", ri);
        return CRM_ERR_NUMERIC;
    }
    const double rho = bg->rho[ri];
    if (out_null) {
        out_null[0] = rho;
        out_null[1] = null.v0 * rho;
        out_null[2] = null.v0 * (1 - rho);
        out_null[3] = null.v1;
        out_null[4] = null.lml;
        out_null[5] = null.delta;
    }
    if (count == 0) return CRM_OK;
    if (!std::isfinite(null.lml)) {
        set_error("association: the null model could not be fitted");
        return CRM_ERR_NUMERIC;
    }

    // per-SNP arguments at the null's rho
    NullFitArgs alt = fa;
    alt.nrho = 1;
    alt.rho[0] = fa.rho[ri];
    alt.rho[0].T = ctx->ws_T.as<double>();
    alt.rho[0].ldT = ldT;
    alt.g_drop = d_drop;   // (full refit: LMM(y, [W, g]) reduces [W, g] by economic_svd, see launch_ortho_block)
    AssocArgs aa{};
    aa.T = ctx->ws_T.as<double>(); aa.ldT = ldT;
    aa.ty = fa.rho[ri].ty; aa.tW = fa.rho[ri].tW; aa.ldW = ldq; aa.S0 = fa.rho[ri].S0;
    aa.r = bg->r[ri]; aa.c = c; aa.n = n; aa.delta0 = null.delta;
    aa.WW = fa.WW; aa.Wy = fa.Wy; aa.yy = fa.yy; aa.gg = d_gg; aa.gy = d_gy; aa.gW = d_gW; aa.ld_gW = ld_gW;
    if (fast) CRM_TRY(launch_fastscan_prep(st, aa, d_prep, d_wts));

    for (long done = 0; done < count; done += BLK) {
        const int nb = (int)std::min<long>(BLK, count - done);
        double* Gb = ctx->ws_Gb.as<double>();
        if (panel->grouped)
            CRM_TRY(launch_expand_block(st, panel->Gd.as<double>() + first + done, panel->ld, panel->group.as<int>(),
                                        np, n, nullptr, nb, Gb, ldb, (int)ldb));
        else
            CRM_TRY(launch_gather_block(st, panel->G.as<double>() + first + done, panel->ld, np, n, nullptr, nullptr,
                                        nb, Gb, ldb, (int)ldb));
        CRM_TRY(launch_variant_stats(st, Gb, ldb, np, nb, d_y, d_W, gene->ld_yw, c, d_part, d_gg, d_gy, d_gW, ld_gW));
        if (!fast) {
            // the per-SNP refit works in the basis of economic_svd([W, g]) (glimix-core LMM): the block orthogonalised
            // against W in the cell axis, as in the interaction scan; the FastScanner's lstsq keeps the raw columns
            double* Gx = ctx->ws_Gx.as<double>();
            CRM_TRY(launch_ortho_block(st, Gb, ldb, np, nb, (int)ldb, d_W, gene->ld_yw, c, gene->Wproj.as<double>(), d_gW, ld_gW,
                                       d_coef, ldb, d_thr, Gx, ldb));
            CRM_TRY(launch_variant_stats(st, Gx, ldb, np, nb, d_y, d_W, gene->ld_yw, c, d_part, d_gg, d_gy, d_gW, ld_gW));
            CRM_TRY(launch_ortho_rank(st, d_gg, d_thr, nb, d_drop));
            Gb = Gx;
        }
        GemmProblem p{};
        CRM_TRY(crm_background_require_q0(bg, ri));
        p.X = Gb; p.ldx = ldb; p.Y = bg->Q0[ri].as<double>(); p.ldy = ldq;
        p.C = ctx->ws_T.as<double>(); p.ldc = ldT; p.M = nb; p.N = bg->r[ri] > 0 ? bg->r[ri] : 1;
        CRM_HIP(hipMemcpyAsync(ctx->ws_probs.ptr, &p, sizeof p, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>(), 1, nb, (int)ldq, np, false, 0, 1, 0));
        if (fast) {
            CRM_TRY(launch_fastscan(st, aa, d_prep, d_wts, nb, d_lml));
        } else {
            CRM_TRY(launch_nullfit(st, alt, nb));
            CRM_TRY(launch_gather_trial_lml(st, d_trial, nb, d_lml));
        }
        CRM_TRY(launch_lrt(st, d_lml, null.lml, nb, d_pv));
        if (out_pvalue) CRM_HIP(hipMemcpyAsync(out_pvalue + done, d_pv, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
        if (out_alt_lml) CRM_HIP(hipMemcpyAsync(out_alt_lml + done, d_lml, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        ctx->report(done + nb, count);   // (the reference's tqdm, :270)
    }
    return CRM_OK;
    });
}
