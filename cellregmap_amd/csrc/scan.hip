// Host orchestration of the interaction scan behind the C-ABI: device-resident background,
// gene and genotype-panel objects, and the per-block kernel pipeline
//   stats -> T(rho) = G' Q0(rho) -> null fits + rho* -> sort by rho* -> Khatri-Rao contraction
//   -> side contractions -> assemble (Q, F) -> eigenvalues + Davies.
// Reference loop being replaced: cellregmap/_cellregmap.py:340-436.
#include <algorithm>
#include <atomic>

#include "nullfit.h"
#include "objects.h"

using namespace crm;

namespace crm {

static unsigned long next_panel_uid() {
    static std::atomic<unsigned long> counter{0};
    return ++counter;
}

// content hash (64-bit words mixed splitmix-style): keys of the shared donor tables
static unsigned long content_key(const void* data, size_t bytes, unsigned long seed) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    unsigned long h = seed ^ (0x9E3779B97F4A7C15ul * (bytes + 1));
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        unsigned long w;
        memcpy(&w, p + i, 8);
        h ^= w + 0x9E3779B97F4A7C15ul + (h << 6) + (h >> 2);
        h *= 0xBF58476D1CE4E5B9ul;
        h ^= h >> 29;
    }
    for (; i < bytes; i++) h = (h ^ p[i]) * 0x100000001B3ul;
    return h ? h : 1;
}

static int pick_split(long cells_pad, long blocks_without_split) { return split_for(cells_pad, blocks_without_split); }

static int ctx_cus(const crm_ctx* ctx) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
    return cus;
}

// flags[0] |= any non-finite entry; flags[1] |= any cell differing from its group's representative
__global__ void verify_panel_kernel(const double* __restrict__ G, long ld, long p, const int* __restrict__ group,
                                    const long* __restrict__ rep, int* __restrict__ flags) {
    const long i = blockIdx.x;
    const long j = (long)blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= p) return;
    const double v = G[i * ld + j];
    if (!(fabs(v) < INFINITY)) atomicOr(&flags[0], 1);
    if (group) {
        const double r = G[rep[group[i]] * ld + j];
        if (__double_as_longlong(v) != __double_as_longlong(r)) atomicOr(&flags[1], 1);
    }
}

__global__ void gather_rows_kernel(const double* __restrict__ G, long ld, const long* __restrict__ rep,
                                   double* __restrict__ Gd) {
    const long d = blockIdx.x;
    const long j = (long)blockIdx.y * blockDim.x + threadIdx.x;
    if (j < ld) Gd[d * ld + j] = G[rep[d] * ld + j];
}

}  // namespace crm

extern "C" {

// ---- background ---------------------------------------------------------------------------
int crm_background_create_qs(crm_ctx* ctx, long n, int nrho, const double* rho, const int* r,
                             const double* const* Q0, const double* const* S0,
                             crm_background** out) {
    return crm::guarded_on("crm_background_create_qs", ctx, [&]() -> int {
    if (!ctx || !out || n <= 0 || nrho < 1 || !rho || !r || !Q0 || !S0) return CRM_ERR_ARG;
    if (nrho > CRM_MAX_RHO) {
        set_error("background: %d grid points (supported up to %d)", nrho, CRM_MAX_RHO);
        return CRM_ERR_UNSUPPORTED;
    }
    *out = nullptr;
    CRM_HIP(hipSetDevice(ctx->device));
    crm_background* bg = new crm_background();
    bg->ctx = ctx;
    bg->n = n;
    bg->n_pad = round_up(n, CELL_PAD);
    bg->nrho = nrho;
    long rmax = 1;
    for (int i = 0; i < nrho; i++) {
        if (r[i] < 0) { delete bg; return CRM_ERR_ARG; }
        bg->rho[i] = rho[i];
        bg->r[i] = r[i];
        rmax = std::max<long>(rmax, r[i]);
    }
    bg->ldq = round_up(rmax, 128);
    for (int i = 0; i < nrho; i++) {
        int rc = bg->Q0[i].ensure(sizeof(double) * bg->n_pad * bg->ldq);
        if (rc == CRM_OK) rc = bg->S0[i].ensure(sizeof(double) * bg->ldq);
        if (rc == CRM_OK)
            rc = upload_padded(ctx->stream, bg->Q0[i].as<double>(), bg->ldq, bg->n_pad, Q0[i], r[i], n, r[i]);
        if (rc == CRM_OK)
            rc = upload_padded(ctx->stream, bg->S0[i].as<double>(), bg->ldq, 1, S0[i], r[i], 1, r[i]);
        if (rc != CRM_OK) { crm_background_destroy(bg); return rc; }
        bg->q0_ready[i] = true;
    }
    bg->s0_max.assign(nrho, 0.0);   // (as background_seal leaves it for the backgrounds the library decomposes itself)
    for (int i = 0; i < nrho; i++)
        for (int j = 0; j < r[i]; j++) bg->s0_max[i] = std::max(bg->s0_max[i], S0[i][j]);
    CRM_HIP(hipStreamSynchronize(ctx->stream));
    *out = bg;
    return CRM_OK;
    });
}

void crm_background_destroy(crm_background* bg) {
    try {
    if (!bg) return;
    std::lock_guard<std::recursive_mutex> lock(bg->ctx->mu);   // (reachable from a finalizer on any thread)
    (void)hipSetDevice(bg->ctx->device);
    (void)hipStreamSynchronize(bg->ctx->stream);
    for (int i = 0; i < CRM_MAX_RHO; i++) {
        bg->Q0[i].release();
        bg->S0[i].release();
        bg->Mix[i].release();
    }
    bg->H.release();
    bg->Ht.release();
    bg->kin_map.release();
    bg->kin_Y.release();
    bg->kin_hKd.release();
    for (int i = 0; i < CRM_MAX_RHO; i++) bg->MixK[i].release();
    for (crm_donor_tables* t : bg->dt_cache) {
        t->release();
        delete t;
    }
    if (bg->builder) crm_background_builder_free(bg->builder);
    delete bg;
    } catch (...) {  // (nothing may unwind into the caller; a destroy has no status to return)
    }
}

int crm_background_set_kinship_groups(crm_background* bg, const int* group, long groups, const double* hKd, long m,
                                      const double* U, int k2) {
    return crm::guarded_on("crm_background_set_kinship_groups", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || !group || !hKd || !U || groups < 1 || m < 1 || k2 < 1) return CRM_ERR_ARG;
    if (bg->builder) {
        set_error("kinship groups: the background is still under construction");
        return CRM_ERR_ARG;
    }
    const long n = bg->n;
    const int k1 = (int)(bg->cols - (long)k2 * m);
    // only backgrounds that kept their half factor H = [E1, L_1 .. L_k2] (thin branch, well-conditioned spectrum) can use it
    if (!bg->fast_T || !bg->H.ptr || k1 < 1 || k1 + k2 > 2 * CRM_MAX_K0 || groups > 4096) return CRM_OK;   // (kin_operand: one thread per column of [us | E1], <= 1024)
    for (long i = 0; i < n; i++)
        if (group[i] < 0 || group[i] >= groups) {
            set_error("kinship groups: group index %d at cell %ld outside [0, %ld)", group[i], i, groups);
            return CRM_ERR_ARG;
        }
    crm_ctx* ctx = bg->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    bg->kin = false;   // (a second announcement that fails must not leave the first one's buffers in use)
    struct Undo {      // ... nor keep its own: whatever it allocated goes back unless it ends with bg->kin set
        crm_background* b;
        ~Undo() {
            if (b->kin) return;
            b->kin_fold = false;
            for (DevBuf* x : {&b->kin_map, &b->kin_Y, &b->kin_hKd}) x->release();
            for (int i = 0; i < b->nrho; i++) b->MixK[i].release();
        }
    } undo{bg};
    // cells in donor order, every donor's run padded to whole stages of the contraction
    std::vector<long> count(groups, 0);
    for (long i = 0; i < n; i++) count[group[i]]++;
    bg->kin_row0.assign(groups, 0);
    bg->kin_len.assign(groups, 0);
    long rows = 0;
    for (long d = 0; d < groups; d++) {
        bg->kin_row0[d] = rows;
        bg->kin_len[d] = round_up(std::max<long>(count[d], 1), GEMM_BK);
        rows += bg->kin_len[d];
    }
    std::vector<int> map(rows, -1);
    std::vector<long> fill(groups, 0);
    for (long i = 0; i < n; i++) {
        const long d = group[i];
        map[bg->kin_row0[d] + fill[d]++] = (int)i;
    }
    bg->kin_rows = rows;
    bg->kin_groups = groups;
    bg->kin_groups_pad = round_up(groups, GEMM_BK);
    bg->kin_cols = m;
    bg->kin_k1 = k1;
    bg->kin_k2 = k2;
    bg->kin_ldh = round_up(m, 128);
    bg->kin_ldy = round_up(k1 + k2, 128);
    CRM_TRY(bg->kin_map.ensure(sizeof(int) * rows));
    CRM_TRY(bg->kin_Y.ensure(sizeof(double) * rows * bg->kin_ldy));
    CRM_TRY(bg->kin_hKd.ensure(sizeof(double) * bg->kin_groups_pad * bg->kin_ldh));
    ScopedBuf dU;
    CRM_TRY(dU.ensure(sizeof(double) * n * k2));
    CRM_HIP(hipMemcpyAsync(bg->kin_map.ptr, map.data(), sizeof(int) * rows, hipMemcpyHostToDevice, st));
    CRM_HIP(hipMemcpyAsync(dU.ptr, U, sizeof(double) * n * k2, hipMemcpyHostToDevice, st));
    CRM_TRY(upload_padded(st, bg->kin_hKd.as<double>(), bg->kin_ldh, bg->kin_groups_pad, hKd, m, groups, m));
    CRM_TRY(launch_kin_operand(st, dU.as<double>(), k2, bg->H.as<double>(), bg->ldh, k1, bg->kin_map.as<int>(), rows,
                               bg->kin_Y.as<double>(), bg->kin_ldy));
    // the route rests on H[c, k1 + j m + d] = U[c, j] hKd[group(c), d] entry by entry: check it here, once, instead of
    // returning the results of another model when a caller announces a structure its half factor does not have
    ScopedBuf dgroup, dcheck;
    CRM_TRY(dgroup.ensure(sizeof(int) * n));
    CRM_TRY(dcheck.ensure(2 * sizeof(unsigned long long)));
    unsigned long long check[2] = {0, 0};
    CRM_HIP(hipMemcpyAsync(dgroup.ptr, group, sizeof(int) * n, hipMemcpyHostToDevice, st));
    CRM_HIP(hipMemsetAsync(dcheck.ptr, 0, sizeof check, st));
    CRM_TRY(launch_kin_verify(st, bg->H.as<double>(), bg->ldh, k1, dU.as<double>(), k2, dgroup.as<int>(),
                              bg->kin_hKd.as<double>(), bg->kin_ldh, m, n, dcheck.as<unsigned long long>()));
    CRM_HIP(hipMemcpyAsync(check, dcheck.ptr, sizeof check, hipMemcpyDeviceToHost, st));
    CRM_HIP(hipStreamSynchronize(st));
    double dmax, hmax;
    memcpy(&dmax, &check[0], sizeof dmax);
    memcpy(&hmax, &check[1], sizeof hmax);
    if (!(dmax <= 1e-12 * hmax)) {
        set_error("kinship groups: the half factor of this background is not U[c, j] * hKd[group(c), d] (largest difference "
                  "%.3g against entries up to %.3g)", dmax, hmax);
        return CRM_ERR_ARG;
    }
    // Fold the donor-level factor into the mixing matrices (objects.h: kin_fold) -- one small product per (grid point, j):
    // MixK[k1 + d' k2 + j, :] = sum_d hKd[d', d] Mix[k1 + j m + d, :], the contraction over d in stages of 16 rows (the
    // rows of hKd' beyond m are zero; the rows of Mix they meet belong to the next j or to Mix's own zero padding, which
    // must exist: cols + padding <= ldh).
    bg->kin_fold = false;
    const long kfold = k1 + groups * (long)k2, m_pad = round_up(m, GEMM_BK);
    const int fold_form = form("kin_fold", 1);   // 0 never, 1 where it pays, 2 also with few columns of us
    // (k2 >= 32: the folded form launches the per-donor sums for the us columns alone, 64 columns wide -- with few of them,
    // config 2's 20, one launch over [us | E1] together and the small contraction over the donors per block is the better
    // form: config 2 463 000 against 451 000 variant-tests/s.  k2 == 1 -- mode B, us a single column -- folds too: its us rows
    // are per-donor sums of the Khatri-Rao rows themselves, a plain batched product, see scan_pass step 6)
    if ((double)kfold <= 1.25 * (double)bg->cols && bg->cols + (m_pad - m) <= bg->ldh && (k2 >= 32 || k2 == 1 || fold_form > 1) && fold_form != 0) {
        const long kdim = round_up(kfold, GEMM_BK), ldq = bg->ldq, ld_t = round_up(groups, 128);
        ScopedBuf hKdT, probs_dev;
        CRM_TRY(hKdT.ensure(sizeof(double) * m_pad * ld_t));
        {
            std::vector<double> t((size_t)m_pad * ld_t, 0.0);
            for (long dd = 0; dd < groups; dd++)
                for (long d = 0; d < m; d++) t[(size_t)d * ld_t + dd] = hKd[dd * m + d];
            CRM_HIP(hipMemcpyAsync(hKdT.ptr, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, st));
            CRM_HIP(hipStreamSynchronize(st));
        }
        std::vector<GemmProblem> pr((size_t)bg->nrho * k2);
        for (int i = 0; i < bg->nrho; i++) {
            CRM_TRY(bg->MixK[i].ensure(sizeof(double) * kdim * ldq));
            CRM_HIP(hipMemsetAsync(bg->MixK[i].ptr, 0, sizeof(double) * kdim * ldq, st));
            CRM_HIP(hipMemcpyAsync(bg->MixK[i].ptr, bg->Mix[i].ptr, sizeof(double) * (size_t)k1 * ldq, hipMemcpyDeviceToDevice, st));
            for (int j = 0; j < k2; j++) {
                GemmProblem p{};
                p.X = hKdT.as<double>(); p.ldx = ld_t;
                p.Y = bg->Mix[i].as<double>() + (size_t)(k1 + (long)j * m) * ldq; p.ldy = ldq;
                p.C = bg->MixK[i].as<double>() + (size_t)(k1 + j) * ldq; p.ldc = (long)k2 * ldq;
                p.M = (int)groups; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
                pr[(size_t)i * k2 + j] = p;
            }
        }
        CRM_TRY(probs_dev.ensure(sizeof(GemmProblem) * pr.size()));
        CRM_HIP(hipMemcpyAsync(probs_dev.ptr, pr.data(), sizeof(GemmProblem) * pr.size(), hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, probs_dev.as<GemmProblem>(), (int)pr.size(), (int)groups, (int)ldq, m_pad, false, 0, 1, 0));
        CRM_HIP(hipStreamSynchronize(st));
        bg->kin_kdim = kdim;
        bg->kin_fold = true;
    }
    bg->kin = true;
    return CRM_OK;
    });
}

int crm_background_kinship_groups(const crm_background* bg) {
    if (!bg) return 0;
    try {
        std::lock_guard<std::recursive_mutex> lock(bg->ctx->mu);
        return bg->kin ? (int)bg->kin_groups : 0;
    } catch (...) {
        return 0;
    }
}

long crm_background_kinship_folded(const crm_background* bg) {
    if (!bg) return 0;
    try {
        std::lock_guard<std::recursive_mutex> lock(bg->ctx->mu);
        return bg->kin && bg->kin_fold ? bg->kin_kdim : 0;
    } catch (...) {
        return 0;
    }
}

int crm_background_rank(const crm_background* bg, int i) {
    return crm::guarded_on("crm_background_rank", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || i < 0 || i >= bg->nrho) return -1;
    return bg->r[i];
    });
}

int crm_background_read(const crm_background* bg, int i, double* Q0, double* S0) {
    return crm::guarded_on("crm_background_read", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || i < 0 || i >= bg->nrho) return CRM_ERR_ARG;
    if (Q0) CRM_TRY(crm_background_require_q0(const_cast<crm_background*>(bg), i));
    CRM_HIP(hipSetDevice(bg->ctx->device));
    const int r = bg->r[i];
    if (Q0 && r > 0)
        CRM_HIP(hipMemcpy2D(Q0, r * sizeof(double), bg->Q0[i].ptr, bg->ldq * sizeof(double),
                            r * sizeof(double), bg->n, hipMemcpyDeviceToHost));
    if (S0 && r > 0) CRM_HIP(hipMemcpy(S0, bg->S0[i].ptr, r * sizeof(double), hipMemcpyDeviceToHost));
    return CRM_OK;
    });
}

// ---- gene -----------------------------------------------------------------------------------
extern "C++" {
// rotations t = Q0(rho)' [y, W] of a gene for every grid point: rows of a [(1+c) x ldq] matrix per grid point
static int gene_rotations(crm_gene* g) {
    crm_background* bg = g->bg;
    crm_ctx* ctx = g->ctx;
    const int c = g->c;
    const long np = bg->n_pad, ldyw = g->ld_yw;
    int rc = CRM_OK;
    auto fail = [&](int code) { return code; };
    const int nrho = bg->nrho;
    const long ldq = bg->ldq;
    const long slab = (long)(1 + c) * ldq;
    if (bg->fast_T && ctx->fast_gene_rot) {
        // Q0(rho) = H Mix(rho):  t = Mix(rho)' (H'[y, W]) -- no Q0 needed
        ScopedBuf thw;
        const long ldh = bg->ldh;
        if ((rc = thw.ensure(sizeof(double) * ldh * 128)) != CRM_OK) return fail(rc);
        if ((rc = g->rot.ensure(sizeof(double) * slab * nrho)) != CRM_OK) return fail(rc);
        if ((rc = ctx->ws_probs.ensure(sizeof(GemmProblem) * (CRM_MAX_RHO + 4))) != CRM_OK) return fail(rc);
        CRM_HIP(hipMemsetAsync(thw.ptr, 0, sizeof(double) * ldh * 128, ctx->stream));
        CRM_HIP(hipMemsetAsync(g->rot.ptr, 0, sizeof(double) * slab * nrho, ctx->stream));
        std::vector<GemmProblem> pr(nrho + 1);
        GemmProblem p0{};
        p0.X = bg->H.as<double>(); p0.ldx = ldh; p0.Y = g->yW.as<double>(); p0.ldy = ldyw;
        p0.C = thw.as<double>(); p0.ldc = 128; p0.M = (int)bg->cols; p0.N = 1 + c;
        pr[0] = p0;
        for (int i = 0; i < nrho; i++) {
            GemmProblem p{};
            p.X = thw.as<double>(); p.ldx = 128; p.Y = bg->Mix[i].as<double>(); p.ldy = ldq;
            p.C = g->rot.as<double>() + (long)i * slab; p.ldc = ldq;
            p.M = 1 + c; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
            pr[1 + i] = p;
        }
        CRM_HIP(hipMemcpyAsync(ctx->ws_probs.ptr, pr.data(), sizeof(GemmProblem) * (nrho + 1), hipMemcpyHostToDevice, ctx->stream));
        if ((rc = launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>(), 1, (int)bg->cols, 1 + c, np, false, 0, 1, 0)) != CRM_OK) return fail(rc);
        if ((rc = launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>() + 1, nrho, 1 + c, (int)ldq, ldh, false, 0, 1, 0)) != CRM_OK) return fail(rc);
        CRM_HIP(hipStreamSynchronize(ctx->stream));
        return CRM_OK;
    }
    if ((rc = crm_background_require_q0(bg, -1)) != CRM_OK) return fail(rc);
    const int ks = pick_split(np, (ldq / GEMM_BN) * nrho);
    if ((rc = g->rot.ensure(sizeof(double) * slab * nrho * ks)) != CRM_OK) return fail(rc);
    std::vector<GemmProblem> probs(nrho);
    for (int i = 0; i < nrho; i++) {
        GemmProblem p{};
        p.X = g->yW.as<double>(); p.ldx = ldyw;
        p.Y = bg->Q0[i].as<double>(); p.ldy = ldq;
        p.C = g->rot.as<double>() + (long)i * slab; p.ldc = ldq;
        p.M = 1 + c; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
        probs[i] = p;
    }
    if ((rc = ctx->ws_probs.ensure(sizeof(GemmProblem) * CRM_MAX_RHO)) != CRM_OK) return fail(rc);
    CRM_HIP(hipMemcpyAsync(ctx->ws_probs.ptr, probs.data(), sizeof(GemmProblem) * nrho, hipMemcpyHostToDevice, ctx->stream));
    // splits write slabs nrho*slab apart
    CRM_HIP(hipMemsetAsync(g->rot.ptr, 0, sizeof(double) * slab * nrho * ks, ctx->stream));
    if ((rc = launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>(), nrho, 1 + c, (int)ldq, np, false, 0, ks, slab * nrho)) != CRM_OK) return fail(rc);
    if ((rc = launch_reduce_splits(ctx->stream, g->rot.as<double>(), slab * nrho, ks, slab * nrho)) != CRM_OK) return fail(rc);
    CRM_HIP(hipStreamSynchronize(ctx->stream));
    return CRM_OK;
}

}  // extern "C++"

int crm_gene_create(crm_background* bg, const double* y, const double* W, int c, const double* E0,
                    int k0, crm_gene** out) {
    return crm::guarded_on("crm_gene_create", bg ? bg->ctx : nullptr, [&]() -> int {
    if (!bg || !y || !W || !E0 || !out) return CRM_ERR_ARG;
    *out = nullptr;
    if (bg->builder) {
        set_error("gene: the background is still under construction (crm_background_seal not called)");
        return CRM_ERR_ARG;
    }
    if (c < 1 || c > CRM_MAX_COV_XWIDE) {
        set_error("gene: %d covariate columns (supported 1..%d; the interaction scan up to %d)", c, CRM_MAX_COV_XWIDE, CRM_MAX_COV_WIDE);
        return CRM_ERR_UNSUPPORTED;
    }
    if (k0 < 1 || k0 > CRM_MAX_K0) {
        set_error("gene: %d contexts (supported 1..%d)", k0, CRM_MAX_K0);
        return CRM_ERR_UNSUPPORTED;
    }
    crm_ctx* ctx = bg->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    const long n = bg->n, np = bg->n_pad;
    for (long i = 0; i < n; i++) {
        bool fin = std::isfinite(y[i]);
        for (int j = 0; j < c && fin; j++) fin = std::isfinite(W[i * c + j]);
        if (!fin) {
            set_error("gene: non-finite values in the outcome or the covariates");
            return CRM_ERR_NUMERIC;
        }
    }
    crm_gene* g = new crm_gene();
    g->bg = bg;
    g->ctx = bg->ctx;
    g->c = c;
    g->k0 = k0;
    g->e0_key = content_key(E0, sizeof(double) * (size_t)bg->n * k0, (unsigned long)k0);
    g->w_key = content_key(W, sizeof(double) * (size_t)bg->n * c, (unsigned long)c);
    g->ldw = 16;
    g->lde = round_up(k0, 16);
    int rc = CRM_OK;
    auto fail = [&](int code) { crm_gene_destroy(g); return code; };
    // host-side inner products
    std::vector<double> WW((size_t)c * c, 0.0), Wy(c, 0.0);
    auto inner_products = [&](const double* Wm) {
        std::fill(WW.begin(), WW.end(), 0.0);
        std::fill(Wy.begin(), Wy.end(), 0.0);
        for (long i = 0; i < n; i++)
            for (int a = 0; a < c; a++) {
                Wy[a] += Wm[i * c + a] * y[i];
                for (int b = a; b < c; b++) WW[a * c + b] += Wm[i * c + a] * Wm[i * c + b];
            }
        for (int a = 0; a < c; a++)
            for (int b = 0; b < a; b++) WW[a * c + b] = WW[b * c + a];
    };
    auto is_diagonal = [&]() {
        for (int a = 0; a < c; a++)
            for (int b = a + 1; b < c; b++)
                if (std::fabs(WW[a * c + b]) > 1e-13 * std::sqrt(WW[a * c + a] * WW[b * c + b])) return false;
        return true;
    };
    g->yy = 0.0;
    for (long i = 0; i < n; i++) g->yy += y[i] * y[i];
    inner_products(W);
    // The orthogonalisation of the variants against W (blockops.hip: launch_ortho_block) and the null fits work in a basis
    // of span(W) with mutually orthogonal columns -- U diag(s) of the thin SVD, the basis glimix-core's LMM holds its
    // covariates in, which is what the Python host passes.  Columns that are not orthogonal are brought there here: the
    // scans depend on W through its column space only.  W <- W V with V the eigenvectors of W'W by a cyclic Jacobi
    // iteration, repeated on the result: a pass leaves the columns orthogonal to ~eps cond(W)^2, and on a nearly diagonal
    // Gram matrix Jacobi resolves the small singular values to high relative accuracy, so two or three passes reach the
    // 1e-13 the diagonal test asks for up to cond(W) ~ 1e7 -- beyond which the reference's own rank rule
    // (numpy_sugar.economic_svd: singular values below sqrt(eps)) is what decides.
    std::vector<double> Wo;
    std::vector<double> Vtot;   // product of the passes' V (empty: W was orthogonal as passed)
    const double* Wuse = W;
    for (int pass = 0; pass < 4 && !is_diagonal(); pass++) {
        std::vector<double> A(WW), V((size_t)c * c, 0.0);
        for (int a = 0; a < c; a++) V[a * c + a] = 1.0;
        for (int sweep = 0; sweep < 60; sweep++) {
            double offd = 0.0, diag = 0.0;
            for (int a = 0; a < c; a++)
                for (int b = 0; b < c; b++) (a == b ? diag : offd) += A[a * c + b] * A[a * c + b];
            if (offd <= 1e-32 * diag) break;
            for (int pi = 0; pi < c - 1; pi++)
                for (int qi = pi + 1; qi < c; qi++) {
                    const double apq = A[pi * c + qi];
                    if (apq == 0.0) continue;
                    const double theta = (A[qi * c + qi] - A[pi * c + pi]) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                    for (int k = 0; k < c; k++) {
                        const double akp = A[k * c + pi], akq = A[k * c + qi];
                        A[k * c + pi] = cs * akp - sn * akq;
                        A[k * c + qi] = sn * akp + cs * akq;
                    }
                    for (int k = 0; k < c; k++) {
                        const double apk = A[pi * c + k], aqk = A[qi * c + k];
                        A[pi * c + k] = cs * apk - sn * aqk;
                        A[qi * c + k] = sn * apk + cs * aqk;
                    }
                    for (int k = 0; k < c; k++) {
                        const double vkp = V[k * c + pi], vkq = V[k * c + qi];
                        V[k * c + pi] = cs * vkp - sn * vkq;
                        V[k * c + qi] = sn * vkp + cs * vkq;
                    }
                }
        }
        std::vector<double> Wn((size_t)n * c);
        std::vector<double> row(c);
        for (long i = 0; i < n; i++) {
            for (int a = 0; a < c; a++) row[a] = Wuse[i * c + a];
            for (int b = 0; b < c; b++) {
                double acc = 0.0;
                for (int a = 0; a < c; a++) acc += row[a] * V[a * c + b];
                Wn[i * c + b] = acc;
            }
        }
        Wo.swap(Wn);
        Wuse = Wo.data();
        inner_products(Wuse);
        if (Vtot.empty()) Vtot = V;
        else {
            std::vector<double> T((size_t)c * c, 0.0);
            for (int a = 0; a < c; a++)
                for (int k = 0; k < c; k++)
                    for (int b = 0; b < c; b++) T[a * c + b] += Vtot[a * c + k] * V[k * c + b];
            Vtot.swap(T);
        }
    }
    if (!is_diagonal()) {
        set_error("gene: the covariates could not be brought to mutually orthogonal columns (W'W stays coupled beyond 1e-13 "
                  "after four passes): pass an orthogonal basis of span(W), e.g. U diag(s) of its thin SVD");
        return fail(CRM_ERR_NUMERIC);
    }
    // [y | W] packed as one operand (column 0 = y) for the rotations, plus separate views
    const long ldyw = 128;
    if ((rc = g->yW.ensure(sizeof(double) * np * ldyw)) != CRM_OK) return fail(rc);
    if ((rc = g->E0.ensure(sizeof(double) * np * g->lde)) != CRM_OK) return fail(rc);
    {
        std::vector<double> pack((size_t)n * (1 + c));
        for (long i = 0; i < n; i++) {
            pack[i * (1 + c)] = y[i];
            for (int j = 0; j < c; j++) pack[i * (1 + c) + 1 + j] = Wuse[i * c + j];
        }
        if ((rc = upload_padded(ctx->stream, g->yW.as<double>(), ldyw, np, pack.data(), 1 + c, n, 1 + c)) != CRM_OK)
            return fail(rc);
        CRM_HIP(hipStreamSynchronize(ctx->stream));
    }
    g->ld_yw = ldyw;
    if ((rc = upload_padded(ctx->stream, g->E0.as<double>(), g->lde, np, E0, k0, n, k0)) != CRM_OK) return fail(rc);
    // What the orthogonalisation of the variants against W needs: (W'W)^-1 = diag(1 / s^2), V = I, d^2 = s^2
    {
        constexpr double EPS = 2.220446049250313e-16;
        std::vector<double> proj((size_t)2 * c * c + c, 0.0);
        double* inv = proj.data();
        double* V = inv + (size_t)c * c;
        double* d2 = V + (size_t)c * c;
        for (int a = 0; a < c; a++) {
            if (!(WW[a * c + a] >= EPS)) {
                set_error("gene: the covariates are rank deficient by the reference's rule (a singular value of %.3g, below "
                          "sqrt(eps)): pass a basis of span(W)", std::sqrt(std::max(WW[a * c + a], 0.0)));
                return fail(CRM_ERR_NUMERIC);
            }
            V[a * c + a] = 1.0;
            d2[a] = WW[a * c + a];
            inv[a * c + a] = 1.0 / WW[a * c + a];
        }
        if ((rc = g->Wproj.ensure(sizeof(double) * proj.size())) != CRM_OK) return fail(rc);
        CRM_HIP(hipMemcpyAsync(g->Wproj.ptr, proj.data(), sizeof(double) * proj.size(), hipMemcpyHostToDevice, ctx->stream));
        CRM_HIP(hipStreamSynchronize(ctx->stream));   // (proj lives on this stack frame)
    }
    if ((rc = g->WW.ensure(sizeof(double) * c * c)) != CRM_OK) return fail(rc);
    if ((rc = g->Wy.ensure(sizeof(double) * c)) != CRM_OK) return fail(rc);
    CRM_HIP(hipMemcpyAsync(g->WW.ptr, WW.data(), sizeof(double) * c * c, hipMemcpyHostToDevice, ctx->stream));
    CRM_HIP(hipMemcpyAsync(g->Wy.ptr, Wy.data(), sizeof(double) * c, hipMemcpyHostToDevice, ctx->stream));
    CRM_HIP(hipStreamSynchronize(ctx->stream));
    g->W_host.assign(Wuse, Wuse + (size_t)n * c);
    g->W_basis = Vtot;
    if ((rc = gene_rotations(g)) != CRM_OK) return fail(rc);
    *out = g;
    return CRM_OK;
    });
}

// Another phenotype on the cohort of `like`: same background, covariates and contexts -- only y differs.  What a gene
// keeps of W and E0 (device copies, W'W, the projection onto span(W), the content keys that let a multi-phenotype pass check
// that its genes agree) is copied on the device instead of being checked, hashed, orthogonalised and uploaded again: binding
// a phenotype costs its own upload and rotations only (per-gene run_interaction calls of the reference, _cellregmap.py:547-587,
// over many genes of one cohort).  The results are bit for bit those of crm_gene_create with the same W and E0.
int crm_gene_create_like(const crm_gene* like, const double* y, crm_gene** out) {
    return crm::guarded_on("crm_gene_create_like", like ? like->ctx : nullptr, [&]() -> int {
    if (!like || !y || !out) return CRM_ERR_ARG;
    *out = nullptr;
    crm_background* bg = like->bg;
    crm_ctx* ctx = like->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    const long n = bg->n;
    const int c = like->c;
    if (like->W_host.size() != (size_t)n * c) {
        set_error("gene: the template gene holds no covariates");
        return CRM_ERR_ARG;
    }
    for (long i = 0; i < n; i++)
        if (!std::isfinite(y[i])) {
            set_error("gene: non-finite values in the outcome or the covariates");
            return CRM_ERR_NUMERIC;
        }
    crm_gene* g = new crm_gene();
    g->bg = bg; g->ctx = ctx; g->c = c; g->k0 = like->k0;
    g->e0_key = like->e0_key; g->w_key = like->w_key;
    g->ldw = like->ldw; g->lde = like->lde; g->ld_yw = like->ld_yw;
    g->W_host = like->W_host;
    g->W_basis = like->W_basis;
    int rc = CRM_OK;
    auto fail = [&](int code) { crm_gene_destroy(g); return code; };
    hipStream_t st = ctx->stream;
    const DevBuf* src[] = {&like->yW, &like->E0, &like->WW, &like->Wproj};
    DevBuf* dst[] = {&g->yW, &g->E0, &g->WW, &g->Wproj};
    for (int q = 0; q < 4; q++) {
        if ((rc = dst[q]->ensure(src[q]->bytes)) != CRM_OK) return fail(rc);
        CRM_HIP(hipMemcpyAsync(dst[q]->ptr, src[q]->ptr, src[q]->bytes, hipMemcpyDeviceToDevice, st));
    }
    // column 0 of [y | W]
    CRM_HIP(hipMemcpy2DAsync(g->yW.ptr, sizeof(double) * g->ld_yw, y, sizeof(double), sizeof(double), n, hipMemcpyHostToDevice, st));
    g->yy = 0.0;
    std::vector<double> Wy(c, 0.0);
    const double* Wm = g->W_host.data();
    for (long i = 0; i < n; i++) g->yy += y[i] * y[i];
    for (long i = 0; i < n; i++)
        for (int a = 0; a < c; a++) Wy[a] += Wm[i * c + a] * y[i];
    if ((rc = g->Wy.ensure(sizeof(double) * c)) != CRM_OK) return fail(rc);
    CRM_HIP(hipMemcpyAsync(g->Wy.ptr, Wy.data(), sizeof(double) * c, hipMemcpyHostToDevice, st));
    CRM_HIP(hipStreamSynchronize(st));
    if ((rc = gene_rotations(g)) != CRM_OK) return fail(rc);
    *out = g;
    return CRM_OK;
    });
}

extern "C++" {
__global__ void scatter_column_kernel(const double* __restrict__ src, long lds, int col, double* __restrict__ dst, long ldd, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i * ldd] = src[i * lds + col];
}
}  // extern "C++"

// ngenes phenotypes (the columns of Y: n x ngenes, row-major, leading dimension ldy) on the cohort of `like`, bound in one
// call: what crm_gene_create_like does per phenotype, with the rotations Q0(rho)'y of all of them as ONE product against the
// half factor and one against every mixing matrix -- those operands (0.8 GB + 11 x 0.2 GB at BASELINE config 3) are read once
// per batch instead of once per phenotype.  out: ngenes handles; on failure none is left behind.
int crm_gene_create_batch(const crm_gene* like, const double* Y, long ldy, int ngenes, crm_gene** out) {
    return crm::guarded_on("crm_gene_create_batch", like ? like->ctx : nullptr, [&]() -> int {
    if (!like || !Y || !out || ngenes < 1 || ldy < ngenes) return CRM_ERR_ARG;
    for (int j = 0; j < ngenes; j++) out[j] = nullptr;
    crm_background* bg = like->bg;
    crm_ctx* ctx = like->ctx;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long n = bg->n, np = bg->n_pad, ldq = bg->ldq;
    const int c = like->c, nrho = bg->nrho;
    if (like->W_host.size() != (size_t)n * c) {
        set_error("gene: the template gene holds no covariates");
        return CRM_ERR_ARG;
    }
    for (long i = 0; i < n; i++)
        for (int j = 0; j < ngenes; j++)
            if (!std::isfinite(Y[i * ldy + j])) {
                set_error("gene: non-finite values in the outcome or the covariates");
                return CRM_ERR_NUMERIC;
            }
    std::vector<crm_gene*> made;
    auto fail = [&](int code) {
        for (crm_gene* g : made) crm_gene_destroy(g);
        for (int j = 0; j < ngenes; j++) out[j] = nullptr;
        return code;
    };
    int rc = CRM_OK;
    const long ldY = round_up(ngenes, 128);
    ScopedBuf dY;
    if ((rc = dY.ensure(sizeof(double) * np * ldY)) != CRM_OK) return rc;
    if ((rc = upload_padded(st, dY.as<double>(), ldY, np, Y, ldy, n, ngenes)) != CRM_OK) return rc;
    const double* Wm = like->W_host.data();
    for (int j = 0; j < ngenes; j++) {
        crm_gene* g = new crm_gene();
        made.push_back(g);
        g->bg = bg; g->ctx = ctx; g->c = c; g->k0 = like->k0;
        g->e0_key = like->e0_key; g->w_key = like->w_key;
        g->ldw = like->ldw; g->lde = like->lde; g->ld_yw = like->ld_yw;
        g->W_host = like->W_host;
        g->W_basis = like->W_basis;
        const DevBuf* src[] = {&like->yW, &like->E0, &like->WW, &like->Wproj, &like->rot};
        DevBuf* dst[] = {&g->yW, &g->E0, &g->WW, &g->Wproj, &g->rot};
        for (int q = 0; q < 5; q++) {
            if ((rc = dst[q]->ensure(src[q]->bytes)) != CRM_OK) return fail(rc);
            CRM_HIP(hipMemcpyAsync(dst[q]->ptr, src[q]->ptr, src[q]->bytes, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL(scatter_column_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dY.as<double>(), ldY, j,
                           g->yW.as<double>(), g->ld_yw, n);
        g->yy = 0.0;
        std::vector<double> Wy(c, 0.0);
        for (long i = 0; i < n; i++) g->yy += Y[i * ldy + j] * Y[i * ldy + j];
        for (long i = 0; i < n; i++)
            for (int a = 0; a < c; a++) Wy[a] += Wm[i * c + a] * Y[i * ldy + j];
        if ((rc = g->Wy.ensure(sizeof(double) * c)) != CRM_OK) return fail(rc);
        CRM_HIP(hipMemcpyAsync(g->Wy.ptr, Wy.data(), sizeof(double) * c, hipMemcpyHostToDevice, st));
        CRM_HIP(hipStreamSynchronize(st));     // (Wy lives on this stack frame)
    }
    CRM_HIP(hipGetLastError());
    const long slab = (long)(1 + c) * ldq;
    if (bg->fast_T && ctx->fast_gene_rot) {
        // t_y = Mix(rho)' (H'y) for all phenotypes at once; the rows Q0(rho)'W came over with the copy of `like`'s rotations
        const long ldh = bg->ldh;
        ScopedBuf thw, rt;
        if ((rc = thw.ensure(sizeof(double) * ldh * ldY)) != CRM_OK) return fail(rc);
        if ((rc = rt.ensure(sizeof(double) * (size_t)nrho * ngenes * ldq)) != CRM_OK) return fail(rc);
        if ((rc = ctx->ws_probs.ensure(sizeof(GemmProblem) * (CRM_MAX_RHO + 4))) != CRM_OK) return fail(rc);
        CRM_HIP(hipMemsetAsync(thw.ptr, 0, sizeof(double) * ldh * ldY, st));
        CRM_HIP(hipMemsetAsync(rt.ptr, 0, sizeof(double) * (size_t)nrho * ngenes * ldq, st));
        std::vector<GemmProblem> pr(nrho + 1);
        GemmProblem p0{};
        p0.X = bg->H.as<double>(); p0.ldx = ldh; p0.Y = dY.as<double>(); p0.ldy = ldY;
        p0.C = thw.as<double>(); p0.ldc = ldY; p0.M = (int)bg->cols; p0.N = ngenes;
        pr[0] = p0;
        for (int i = 0; i < nrho; i++) {
            GemmProblem p{};
            p.X = thw.as<double>(); p.ldx = ldY; p.Y = bg->Mix[i].as<double>(); p.ldy = ldq;
            p.C = rt.as<double>() + (size_t)i * ngenes * ldq; p.ldc = ldq;
            p.M = ngenes; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
            pr[1 + i] = p;
        }
        CRM_HIP(hipMemcpyAsync(ctx->ws_probs.ptr, pr.data(), sizeof(GemmProblem) * (nrho + 1), hipMemcpyHostToDevice, st));
        if ((rc = launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>(), 1, (int)bg->cols, ngenes, np, false, 0, 1, 0)) != CRM_OK) return fail(rc);
        if ((rc = launch_gemm_tn(ctx, ctx->ws_probs.as<GemmProblem>() + 1, nrho, ngenes, (int)ldq, ldh, false, 0, 1, 0)) != CRM_OK) return fail(rc);
        for (int j = 0; j < ngenes; j++)     // row 0 of every grid point's [(1 + c) x ldq] block
            CRM_HIP(hipMemcpy2DAsync(made[j]->rot.ptr, sizeof(double) * slab, rt.as<double>() + (size_t)j * ldq,
                                     sizeof(double) * (size_t)ngenes * ldq, sizeof(double) * ldq, nrho, hipMemcpyDeviceToDevice, st));
        CRM_HIP(hipStreamSynchronize(st));
    } else {
        for (crm_gene* g : made)
            if ((rc = gene_rotations(g)) != CRM_OK) return fail(rc);
    }
    for (int j = 0; j < ngenes; j++) out[j] = made[j];
    return CRM_OK;
    });
}

void crm_gene_destroy(crm_gene* g) {
    try {
    if (!g) return;
    std::lock_guard<std::recursive_mutex> lock(g->ctx->mu);   // (reachable from a finalizer on any thread)
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    g->dt_own.release();
    for (auto* b : {&g->yW, &g->E0, &g->WW, &g->Wy, &g->Wproj, &g->rot, &g->Ep, &g->YE, &g->EE, &g->idx, &g->dt_Z1, &g->dt_sums,
                    &g->dt_Zt, &g->kinEp, &g->kinP, &g->kinUE, &g->kinEE})
        b->release();
    delete g;
    } catch (...) {  // (nothing may unwind into the caller; a destroy has no status to return)
    }
}

// ---- panel ----------------------------------------------------------------------------------
// (No lock of the context: the upload touches none of its work buffers and runs on a stream of its own, so that a panel
// can go to the device from one thread while another thread's constructor holds the context -- 8 GB over PCIe beside
// the eleven decompositions.  The copy is complete when the call returns.)
int crm_panel_create(crm_ctx* ctx, long n, const double* G, long ldg, long p, crm_panel** out) {
    return crm::guarded("crm_panel_create", [&]() -> int {
    if (!ctx || !G || !out || n <= 0 || p <= 0 || ldg < p) return CRM_ERR_ARG;
    *out = nullptr;
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->upload_stream ? ctx->upload_stream : ctx->stream;
    crm_panel* P = new crm_panel();
    P->ctx = ctx;
    P->uid = next_panel_uid();
    P->n = n;
    P->n_pad = round_up(n, CELL_PAD);
    P->p = p;
    P->ld = round_up(p, 128);
    int rc = P->G.ensure(sizeof(double) * P->n_pad * P->ld);
    if (rc == CRM_OK) rc = upload_padded(st, P->G.as<double>(), P->ld, P->n_pad, G, ldg, n, p);
    if (rc != CRM_OK) { crm_panel_destroy(P); return rc; }
    CRM_HIP(hipStreamSynchronize(st));
    *out = P;
    return CRM_OK;
    });
}

void crm_panel_destroy(crm_panel* P) {
    try {
    if (!P) return;
    std::lock_guard<std::recursive_mutex> lock(P->ctx->mu);   // (reachable from a finalizer on any thread)
    (void)hipSetDevice(P->ctx->device);
    (void)hipStreamSynchronize(P->ctx->stream);
    P->G.release();
    P->Gd.release();
    P->group.release();
    P->Z.release();
    delete P;
    } catch (...) {  // (nothing may unwind into the caller; a destroy has no status to return)
    }
}

// ---- grouped (donor-constant) panel --------------------------------------------------------------
int crm_panel_create_grouped(crm_ctx* ctx, long n, const int* group, long m, const double* Gd, long ldg,
                             long p, crm_panel** out) {
    return crm::guarded_on("crm_panel_create_grouped", ctx, [&]() -> int {
    if (!ctx || !group || !Gd || !out || n <= 0 || m <= 0 || p <= 0 || ldg < p) return CRM_ERR_ARG;
    *out = nullptr;
    for (long i = 0; i < n; i++) {
        if (group[i] < 0 || group[i] >= m) {
            set_error("grouped panel: group index %d at cell %ld outside [0, %ld)", group[i], i, m);
            return CRM_ERR_ARG;
        }
    }
    CRM_HIP(hipSetDevice(ctx->device));
    crm_panel* P = new crm_panel();
    P->ctx = ctx;
    P->uid = next_panel_uid();
    P->n = n;
    P->n_pad = round_up(n, CELL_PAD);
    P->p = p;
    P->ld = round_up(p, 128);
    P->grouped = true;
    P->m = m;
    P->m_pad = round_up(m, GEMM_BK);
    P->ldz = round_up(m, 128) + 128;
    int rc = P->Gd.ensure(sizeof(double) * P->m_pad * P->ld);
    if (rc == CRM_OK) rc = upload_padded(ctx->stream, P->Gd.as<double>(), P->ld, P->m_pad, Gd, ldg, m, p);
    if (rc == CRM_OK) rc = P->group.ensure(sizeof(int) * n);
    if (rc == CRM_OK) rc = P->Z.ensure(sizeof(double) * P->n_pad * P->ldz);
    if (rc != CRM_OK) { crm_panel_destroy(P); return rc; }
    CRM_HIP(hipMemcpyAsync(P->group.ptr, group, sizeof(int) * n, hipMemcpyHostToDevice, ctx->stream));
    P->group_key = content_key(group, sizeof(int) * n, (unsigned long)m);
    rc = launch_indicator(ctx->stream, P->group.as<int>(), n, P->n_pad, (int)m, P->Z.as<double>(), P->ldz);
    if (rc != CRM_OK) { crm_panel_destroy(P); return rc; }
    CRM_HIP(hipStreamSynchronize(ctx->stream));
    *out = P;
    return CRM_OK;
    });
}

}  // extern "C"

namespace crm {
// One workgroup per variant: mean and standard deviation of the EXPANDED column (every donor's dosage weighted by
// its number of cells; population variance, as numpy's std / the reference simulator's column_normalize,
// cellregmap/_simulate.py:50-54), then the standardised donor-level column in float64.
__global__ __launch_bounds__(256) void standardise_dosages_kernel(const signed char* __restrict__ D, long ldd, long m, long p,
                                                                  const double* __restrict__ cells_of, double n, int standardise,
                                                                  double* __restrict__ Gd, long ldg, int* __restrict__ flags) {
    __shared__ double red[2][4];
    const long j = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s1 = 0.0, s2 = 0.0;
    for (long d = tid; d < m; d += blockDim.x) {
        const double g = (double)D[d * ldd + j], w = cells_of[d];
        s1 += w * g;
        s2 += w * g * g;
    }
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double mean = s1 / n;
    double var = 0.0;   // second pass in the centred form (the raw-moment difference cancels for rare alleles)
    for (long d = tid; d < m; d += blockDim.x) {
        const double c = (double)D[d * ldd + j] - mean;
        var += cells_of[d] * c * c;
    }
    for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
    __syncthreads();
    if (lane == 0) red[0][wave] = var;
    __syncthreads();
    var = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / n;
    const double sd = sqrt(var);
    if (standardise && !(sd > 0.0)) {
        if (tid == 0) atomicOr(&flags[0], 1);   // monomorphic column: the reference's normalisation divides by zero
        return;
    }
    for (long d = tid; d < m; d += blockDim.x) {
        const double g = (double)D[d * ldd + j];
        Gd[d * ldg + j] = standardise ? (g - mean) / sd : g;
    }
}
}  // namespace crm

extern "C" int crm_panel_create_grouped_i8(crm_ctx* ctx, long n, const int* group, long m, const signed char* dosage,
                                           long ldd, long p, int standardise, crm_panel** out) {
    return crm::guarded_on("crm_panel_create_grouped_i8", ctx, [&]() -> int {
    if (!ctx || !group || !dosage || !out || n <= 0 || m <= 0 || p <= 0 || ldd < p) return CRM_ERR_ARG;
    *out = nullptr;
    std::vector<double> cells_of(m, 0.0);
    for (long i = 0; i < n; i++) {
        if (group[i] < 0 || group[i] >= m) {
            set_error("grouped panel: group index %d at cell %ld outside [0, %ld)", group[i], i, m);
            return CRM_ERR_ARG;
        }
        cells_of[group[i]] += 1.0;
    }
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    crm_panel* P = new crm_panel();
    auto fail = [&](int code) { crm_panel_destroy(P); return code; };
    P->ctx = ctx;
    P->uid = next_panel_uid();
    P->n = n;
    P->n_pad = round_up(n, CELL_PAD);
    P->p = p;
    P->ld = round_up(p, 128);
    P->grouped = true;
    P->m = m;
    P->m_pad = round_up(m, GEMM_BK);
    P->ldz = round_up(m, 128) + 128;
    ScopedBuf dD, dCells, dFlags;
    int rc = P->Gd.ensure(sizeof(double) * P->m_pad * P->ld);
    if (rc == CRM_OK) rc = P->group.ensure(sizeof(int) * n);
    if (rc == CRM_OK) rc = P->Z.ensure(sizeof(double) * P->n_pad * P->ldz);
    if (rc == CRM_OK) rc = dD.ensure((size_t)m * p);
    if (rc == CRM_OK) rc = dCells.ensure(sizeof(double) * m);
    if (rc == CRM_OK) rc = dFlags.ensure(sizeof(int));
    if (rc != CRM_OK) return fail(rc);
    int h_flag = 0;
    if (hipMemsetAsync(P->Gd.ptr, 0, sizeof(double) * P->m_pad * P->ld, st) != hipSuccess ||
        hipMemsetAsync(dFlags.ptr, 0, sizeof(int), st) != hipSuccess ||
        hipMemcpy2DAsync(dD.ptr, p, dosage, ldd, p, m, hipMemcpyHostToDevice, st) != hipSuccess ||   // 1 byte per dosage over PCIe
        hipMemcpyAsync(dCells.ptr, cells_of.data(), sizeof(double) * m, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(P->group.ptr, group, sizeof(int) * n, hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(CRM_ERR_HIP);
    hipLaunchKernelGGL(standardise_dosages_kernel, dim3((unsigned)p), dim3(256), 0, st, dD.as<signed char>(), p, m, p,
                       dCells.as<double>(), (double)n, standardise ? 1 : 0, P->Gd.as<double>(), P->ld, dFlags.as<int>());
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(&h_flag, dFlags.ptr, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess)
        return fail(CRM_ERR_HIP);
    P->group_key = content_key(group, sizeof(int) * n, (unsigned long)m);
    rc = launch_indicator(st, P->group.as<int>(), n, P->n_pad, (int)m, P->Z.as<double>(), P->ldz);
    if (rc != CRM_OK) return fail(rc);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(CRM_ERR_HIP);
    if (h_flag) {
        set_error("panel: a monomorphic variant cannot be standardised (zero variance)");
        return fail(CRM_ERR_NUMERIC);
    }
    *out = P;
    return CRM_OK;
    });
}

extern "C" {
// Upload an expanded genotype matrix, check it for non-finite entries and -- when a candidate
// grouping is given (group_hint[i] in [0, m_hint), rep_rows[d] = a row carrying group d) -- verify ON THE
// DEVICE that every cell equals its group's representative in every variant; if so the panel is stored
// donor-level (grouped), otherwise dense.  Replaces the host-side isfinite pass and the host-side
// verification of detect_groups (both O(n p) memory passes that dominated short scans).
int crm_panel_create_auto(crm_ctx* ctx, long n, const double* G, long ldg, long p, const int* group_hint,
                          long m_hint, const long* rep_rows, crm_panel** out, int* out_grouped) {
    return crm::guarded("crm_panel_create_auto", [&]() -> int {   // (no lock of the context, like crm_panel_create)
    if (!ctx || !G || !out || n <= 0 || p <= 0 || ldg < p) return CRM_ERR_ARG;
    if (out_grouped) *out_grouped = 0;
    crm_panel* P = nullptr;
    CRM_TRY(crm_panel_create(ctx, n, G, ldg, p, &P));
    hipStream_t st = ctx->upload_stream ? ctx->upload_stream : ctx->stream;
    auto fail = [&](int code) { crm_panel_destroy(P); return code; };
    ScopedBuf flags, dgroup, drep;
    int rc;
    if ((rc = flags.ensure(sizeof(int) * 2)) != CRM_OK) return fail(rc);
    const bool hinted = group_hint && rep_rows && m_hint > 0 && m_hint < BLOCK_SLACK_MAX - 1;
    if (hinted) {
        for (long i = 0; i < n; i++)
            if (group_hint[i] < 0 || group_hint[i] >= m_hint) return fail(CRM_ERR_ARG);
        for (long d = 0; d < m_hint; d++)
            if (rep_rows[d] < 0 || rep_rows[d] >= n) return fail(CRM_ERR_ARG);
        if ((rc = dgroup.ensure(sizeof(int) * n)) != CRM_OK) return fail(rc);
        if ((rc = drep.ensure(sizeof(long) * m_hint)) != CRM_OK) return fail(rc);
        if (hipMemcpyAsync(dgroup.ptr, group_hint, sizeof(int) * n, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(drep.ptr, rep_rows, sizeof(long) * m_hint, hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(CRM_ERR_HIP);
    }
    if (hipMemsetAsync(flags.ptr, 0, sizeof(int) * 2, st) != hipSuccess) return fail(CRM_ERR_HIP);
    hipLaunchKernelGGL(verify_panel_kernel, dim3((unsigned)n, (unsigned)((p + 255) / 256)), dim3(256), 0, st,
                       P->G.as<double>(), P->ld, p, hinted ? dgroup.as<int>() : nullptr,
                       hinted ? drep.as<long>() : nullptr, flags.as<int>());
    int h_flags[2] = {0, 0};
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(h_flags, flags.ptr, sizeof h_flags, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return fail(CRM_ERR_HIP);
    if (h_flags[0]) {
        set_error("panel: non-finite values in the genotype matrix");
        return fail(CRM_ERR_NUMERIC);
    }
    if (hinted && !h_flags[1]) {
        // collapse: gather the representative rows, drop the dense copy
        P->grouped = true;
        P->m = m_hint;
        P->m_pad = round_up(m_hint, GEMM_BK);
        P->ldz = round_up(m_hint, 128) + 128;
        if ((rc = P->Gd.ensure(sizeof(double) * P->m_pad * P->ld)) != CRM_OK) return fail(rc);
        if (hipMemsetAsync(P->Gd.ptr, 0, sizeof(double) * P->m_pad * P->ld, st) != hipSuccess) return fail(CRM_ERR_HIP);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)m_hint, (unsigned)((P->ld + 255) / 256)), dim3(256), 0,
                           st, P->G.as<double>(), P->ld, drep.as<long>(), P->Gd.as<double>());
        if ((rc = P->group.ensure(sizeof(int) * n)) != CRM_OK) return fail(rc);
        if ((rc = P->Z.ensure(sizeof(double) * P->n_pad * P->ldz)) != CRM_OK) return fail(rc);
        if (hipMemcpyAsync(P->group.ptr, dgroup.ptr, sizeof(int) * n, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(CRM_ERR_HIP);
        if ((rc = launch_indicator(st, P->group.as<int>(), n, P->n_pad, (int)m_hint, P->Z.as<double>(), P->ldz)) != CRM_OK)
            return fail(rc);
        if (hipStreamSynchronize(st) != hipSuccess) return fail(CRM_ERR_HIP);
        P->G.release();
        P->group_key = content_key(group_hint, sizeof(int) * n, (unsigned long)m_hint);
        if (out_grouped) *out_grouped = 1;
    }
    *out = P;
    return CRM_OK;
    });
}

int crm_set_donor_collapse(crm_ctx* ctx, int on) {
    return crm::guarded_on("crm_set_donor_collapse", ctx, [&]() -> int {
    if (!ctx) return CRM_ERR_ARG;
    ctx->collapse = on != 0;
    return CRM_OK;
    });
}

}  // extern "C"

namespace crm {

// sums[d][q] over the cells of donor d: q = 0 count, 1 y, 2.. the covariate columns.  One workgroup per (donor, q):
// its threads stride over the cells and meet in a fixed order (the same bits whatever the launch).
__global__ __launch_bounds__(256) void donor_sums_kernel(const int* __restrict__ group, long cells, int m,
                                                         const double* __restrict__ yW, long ldw, int c,
                                                         double* __restrict__ sums) {
    __shared__ double part[256];
    const int d = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    double acc = 0.0;
    for (long i = tid; i < cells; i += 256)
        if (group[i] == d) acc += q == 0 ? 1.0 : yW[i * ldw + (q - 1)];
    part[tid] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) part[tid] += part[tid + w];
        __syncthreads();
    }
    if (tid == 0) sums[d * DT_SUMS_LD + q] = part[0];
}

__global__ void permute_group_kernel(const int* __restrict__ group, const int* __restrict__ idx, long n,
                                     int* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = group[idx[i]];
}

// Z2[b, j] = sum_{d, d'} gamma_{d,b} gamma_{d',b} C[(d, d'), j]  with  C[(d, d'), :] = sum_i z'_d[i] z_d'[i] E[i, :]
// (test direction carried by the permuted indicators z', fixed effect by the unpermuted ones)
__global__ __launch_bounds__(128) void donor_cross_kernel(const double* __restrict__ Gam, long ld_gam, int m,
                                                           const double* __restrict__ C, long ldc, int k0,
                                                           double* __restrict__ Z2, long ldz2) {
    __shared__ double gam[BLOCK_SLACK_MAX > 256 ? 256 : BLOCK_SLACK_MAX];
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < m; d += blockDim.x) gam[d] = Gam[(long)d * ld_gam + b];
    __syncthreads();
    for (int j = threadIdx.x; j < k0; j += blockDim.x) {
        double acc = 0.0;
        for (int d = 0; d < m; d++) {
            const double gd = gam[d];
            if (gd == 0.0) continue;
            double inner = 0.0;
            const double* row = C + (long)d * m * ldc + j;
            for (int e = 0; e < m; e++) inner += gam[e] * row[(long)e * ldc];
            acc += gd * inner;
        }
        Z2[(long)b * ldz2 + j] = acc;
    }
}

// Per-donor tables of the collapsed path: every n-length contraction of the scan is linear in
// diag(g) (or diag(g)^2 = sum_d gamma_d^2 diag(z_d) for donor-constant g), so it is taken once per donor
// indicator z_d with the same kernels and afterwards combined with the donor dosages gamma.
static int build_donor_tables(crm_gene* gene, const crm_panel* panel, crm_donor_tables* shared, const double* d_Ep,
                              const double* d_EE, const double* Zt, bool cross) {
    // shared != nullptr: also (re)build the phenotype-free tables (TZ, Bd, Z2, Z3) into *shared.
    // Zt: indicators of the test direction (rows permuted by idx_G, else the panel's own);
    // cross: Z2 becomes the m*m-row table of the mixed products z'_d o z_d'.
    crm_background* bg = gene->bg;
    crm_ctx* ctx = bg->ctx;
    hipStream_t st = ctx->stream;
    const long n = bg->n, np = bg->n_pad, ldq = bg->ldq;
    const int nrho = bg->nrho, c = gene->c, k0 = gene->k0;
    const long m = panel->m, mp = panel->m_pad;
    const int npair = k0 * (k0 + 1) / 2;
    const long ldZ1 = gene->ld_ye, ldZ2 = gene->ld_ep, ldZ3 = gene->ld_ee;
    const bool full = shared != nullptr;
    CRM_TRY(ctx->ws_probs.ensure(sizeof(GemmProblem) * (CRM_MAX_RHO + 4)));
    GemmProblem* d_probs = ctx->ws_probs.as<GemmProblem>();
    std::vector<GemmProblem> probs(CRM_MAX_RHO + 4);
    const double* Z = panel->Z.as<double>();
    if (full) CRM_TRY(crm_background_require_q0(bg, -1));   // the tables are contractions against every Q0(rho)
    if (full) {
        CRM_TRY(shared->TZ.ensure(sizeof(double) * (size_t)nrho * mp * ldq));
        CRM_TRY(shared->Bd.ensure(sizeof(double) * (size_t)nrho * mp * k0 * ldq));
        CRM_HIP(hipMemsetAsync(shared->TZ.ptr, 0, sizeof(double) * (size_t)nrho * mp * ldq, st));
        CRM_HIP(hipMemsetAsync(shared->Bd.ptr, 0, sizeof(double) * (size_t)nrho * mp * k0 * ldq, st));
        for (int i = 0; i < nrho; i++) {
            GemmProblem p{};
            p.X = Z; p.ldx = panel->ldz; p.Y = bg->Q0[i].as<double>(); p.ldy = ldq;
            p.C = shared->TZ.as<double>() + (size_t)i * mp * ldq; p.ldc = ldq;
            p.M = (int)m; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
            probs[i] = p;
        }
        CRM_HIP(hipMemcpyAsync(d_probs, probs.data(), sizeof(GemmProblem) * nrho, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_probs, nrho, (int)m, (int)ldq, np, false, 0, 1, 0));
        CRM_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < nrho; i++) {
            GemmProblem p{};
            p.X = Zt; p.ldx = panel->ldz; p.E = d_Ep; p.lde = gene->ld_ep; p.k0 = k0;
            p.Y = bg->Q0[i].as<double>(); p.ldy = ldq;
            p.C = shared->Bd.as<double>() + (size_t)i * mp * k0 * ldq; p.ldc = ldq;
            p.M = (int)m * k0; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
            probs[i] = p;
        }
        CRM_HIP(hipMemcpyAsync(d_probs, probs.data(), sizeof(GemmProblem) * nrho, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_probs, nrho, (int)m * k0, (int)ldq, np, true, k0, 1, 0));
        CRM_HIP(hipStreamSynchronize(st));
    }
    // side tables (split over the cell axis: only one M tile)
    DevBuf unused;
    struct Side { DevBuf* buf; const double* Y; long ldy; int N; long ld; bool needed; } side[3] = {
        {&gene->dt_Z1, gene->YE.as<double>(), gene->ld_ye, k0 * (1 + c), ldZ1, true},
        {full ? &shared->Z2 : &unused, d_Ep, gene->ld_ep, k0, ldZ2, full},
        {full ? &shared->Z3 : &unused, d_EE, gene->ld_ee, npair, ldZ3, full}};
    if (full && cross) {
        // C[(d*m + d'), :] = KR(Zt, Z)' Ep : the Khatri-Rao contraction with the indicators as "contexts"
        side[1].needed = false;
        const long rows = m * m;
        CRM_TRY(shared->Z2.ensure(sizeof(double) * (size_t)rows * ldZ2));
        CRM_HIP(hipMemsetAsync(shared->Z2.ptr, 0, sizeof(double) * (size_t)rows * ldZ2, st));
        GemmProblem p{};
        p.X = Zt; p.ldx = panel->ldz; p.E = Z; p.lde = panel->ldz; p.k0 = (int)m;
        p.Y = d_Ep; p.ldy = gene->ld_ep; p.C = shared->Z2.as<double>(); p.ldc = ldZ2;
        p.M = (int)rows; p.N = k0;
        CRM_HIP(hipMemcpyAsync(d_probs, &p, sizeof p, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_probs, 1, (int)rows, k0, np, true, (int)m, 1, 0));
        CRM_HIP(hipStreamSynchronize(st));
    }
    for (auto& sd : side) {
        if (!sd.needed) continue;
        const int ks = pick_split(np, sd.ld / GEMM_BN);
        const long sz = mp * sd.ld;
        CRM_TRY(sd.buf->ensure(sizeof(double) * (size_t)sz * ks));
        CRM_HIP(hipMemsetAsync(sd.buf->ptr, 0, sizeof(double) * (size_t)sz * ks, st));
        GemmProblem p{};
        p.X = Zt; p.ldx = panel->ldz; p.Y = sd.Y; p.ldy = sd.ldy; p.C = sd.buf->as<double>(); p.ldc = sd.ld;
        p.M = (int)m; p.N = sd.N;
        CRM_HIP(hipMemcpyAsync(d_probs, &p, sizeof p, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_probs, 1, (int)m, sd.N, np, false, 0, ks, sz));
        CRM_TRY(launch_reduce_splits(st, sd.buf->as<double>(), sz, ks, sz));
        CRM_HIP(hipStreamSynchronize(st));
    }
    CRM_TRY(gene->dt_sums.ensure(sizeof(double) * mp * DT_SUMS_LD));
    CRM_HIP(hipMemsetAsync(gene->dt_sums.ptr, 0, sizeof(double) * mp * DT_SUMS_LD, st));
    hipLaunchKernelGGL(donor_sums_kernel, dim3((unsigned)m, c + 2), dim3(256), 0, st,
                       panel->group.as<int>(), n, (int)m, gene->yW.as<double>(), gene->ld_yw, c,
                       gene->dt_sums.as<double>());
    CRM_HIP(hipGetLastError());
    CRM_HIP(hipStreamSynchronize(st));
    return CRM_OK;
}

}  // namespace crm

extern "C" {

// ---- the scan ---------------------------------------------------------------------------------
}  // extern "C"

namespace crm {

// Collapsed path: a variant that keeps less than this share of its squared norm outside span(W) is repeated on the dense
// path (the donor-level sums can only form [W, g]'K^-1[W, g] in the raw basis: eps / share instead of eps / sqrt(share))
constexpr double COLLINEAR_TAU = 1e-2;

// Fit records of the flat-optimum probes (include/crm_hip.h: CRM_MODEL_FLAT_OPTIMUM): delta moved by one stopping
// tolerance of the reference's search on x = logit(delta) (brent-search: tol = 1e-6 |x| + 1e-6), unit scale -- the
// assembly derives the scale at that delta itself (assemble.hip: fit.scale < 0).
__global__ void flat_probe_fit_kernel(const crm::NullFitOut* __restrict__ fit, int count, double sign,
                                      crm::NullFitOut* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    crm::NullFitOut f = fit[b];
    const double tiny = 2.220446049250313e-16;
    const double d = fmin(fmax(f.delta, tiny), 1.0 - tiny);
    const double x = log(d) - log1p(-d);
    const double tol = 1e-6 * fabs(x) + 1e-6;
    const double dp = fmin(fmax(1.0 / (1.0 + exp(-(x + sign * tol))), tiny), 1.0 - tiny);
    f.delta = dp;
    f.v0 = 1.0 - dp;
    f.v1 = dp;
    f.scale = -1.0;
    out[b] = f;
}

// crm_scan_interaction_permuted (crm_ctx::ReplayBlock): the rows T(rho*(b)) of a block out of / back into the per-grid-point
// slabs of the rotations, T[(rho * blk + b) * ldT + j]
__global__ void replay_rows_kernel(double* __restrict__ T, long blk, long ldT, const crm::NullFitOut* __restrict__ fit, int nb,
                                   int cols, double* __restrict__ rows, int restore) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb || j >= cols) return;
    const int ri = fit[b].rho_index;
    if (ri < 0) return;
    double* slab = T + ((size_t)ri * blk + b) * ldT;
    if (restore) slab[j] = rows[(size_t)b * ldT + j];
    else rows[(size_t)b * ldT + j] = slab[j];
}

// How far apart may two faithful runs of the reference's null fit stop (include/crm_hip.h: crm_scan_interaction_bounds)?
// Measured on device-vs-oracle streams of 71 000 scans (tools/diag/flat_flag_study.py, profiles/r06_flat_flag_*): the
// distance of the two stopping points in units of the search's tolerance, times the relative gain of the objective over one
// tolerance at the stopping point (NullFitTrial::curv / |lml|), never exceeded 2.2e-13 (99.9 %: 1.2e-13, 99 %: 6e-14,
// median 4e-19): the flatter the likelihood, the further rounding noise moves the last parabolic steps -- up to one whole
// tolerance, where the search's last comparison f(x +- tol) <= f(x) itself falls the other way.
constexpr double STOP_SHIFT_C = 2.5e-13;
// ... and a decision of the search counts as open to rounding outright (shift = one tolerance) when its margin is within
// this many times the first-order noise bound of the objective (nullfit.hip: objective_noise_bound); the choice of rho*
// likewise (CRM_MODEL_RHO_TIE)
constexpr double FLAT_KAPPA = 1.0;
constexpr double RHO_KAPPA = 1.0;
constexpr int FLAT_REC = 10;   // doubles per variant of the diagnostics record (crm_test_null_fit_probe_read)

struct ScanOut {  // per-gene output bases (host), each `count` long (lambda: count*k0, F: count*k0*k0)
    double *pv, *rho1, *e2, *g2, *eps2, *Q, *lml, *delta, *scale, *lambda, *F;
    int* ifault = nullptr;   // Davies' fault code per variant (0 ok; 1, 2, 4 as AS 155; < 0: no usable eigenvalues)
    double* liu = nullptr;   // the modified-Liu p-value (chiscore's info["liu_pval"])
    int* flags = nullptr;    // CRM_MODEL_* bits per variant (include/crm_hip.h)
    double* bound_Q = nullptr;   // crm_scan_interaction_bounds: how far Q / p of two faithful runs may differ (relative)
    double* bound_p = nullptr;
};

// Variants per block of a scan of `count` variants.  Automatic: as many as keep the A~ buffer (block x k0 x ldq doubles)
// within 16 GB, at most 4096 -- fixed per-block costs (host round trip for the rho* groups, small launches, the last, partly
// filled round of workgroups) then weigh 2-3 % less than at 1024.
static bool scan_slow_forms(const crm_gene* g0) {
    // the slower per-variant kernels (more than 144 Gram rows or 128 contexts): their global-memory work space
    return g0->k0 + g0->c + 2 > 144 || g0->k0 > 128 || assemble_rows_scratch_doubles(1, g0->k0, g0->c) > 0;
}
static int scan_block_variants(const crm_ctx* ctx, const crm_gene* g0, long count) {
    long auto_blk = (long)(16.0 * (1ull << 30) / (sizeof(double) * (double)g0->k0 * (double)g0->bg->ldq)) / 128 * 128;
    auto_blk = std::max<long>(256, std::min<long>(auto_blk, CRM_MAX_AUTO_BLOCK));
    int BLK = (int)std::min<long>(ctx->block_variants > 0 ? ctx->block_variants : auto_blk, round_up(count, 128));
    if (g0->c > CRM_MAX_COV_WIDE) BLK = std::min(BLK, 512);   // (63 .. 128 covariate columns: the slow null-fit kernel's scratch)
    if (scan_slow_forms(g0)) BLK = std::min(BLK, 512);
    return BLK;
}

// One pass over variants [first, first + count) for one or several genes that share the background,
// the covariates W and the contexts E0 (several phenotypes against one panel).  What does not depend
// on the phenotype is done once per block: the block copies, T(rho) = G'Q0(rho), the Khatri-Rao
// contraction per (variant, rho) pair that at least one gene selected, and the y-free side
// contractions.  Per gene: g'y, the null fits, E'(g o y), assembly, eigenvalues and Davies.
//
// allow_collapse = false keeps a grouped panel on the dense path; near_out (collapsed passes only) receives the positions
// (relative to `first`) of the variants that are nearly collinear with the covariates -- scan_core repeats those on the
// dense path, where the block is orthogonalised against W in the cell axis (blockops.hip: launch_ortho_block).
static int scan_pass(const std::vector<crm_gene*>& genes, crm_panel* panel, long first, long count,
                     const int* idx_E, const int* idx_G, const std::vector<ScanOut>& outs, bool allow_collapse,
                     std::vector<long>* near_out) {
    const int ng = (int)genes.size();
    crm_gene* g0 = genes[0];
    crm_background* bg = g0->bg;
    crm_ctx* ctx = bg->ctx;
    if (panel->ctx != ctx) {
        set_error("scan: gene and panel live on different contexts");
        return CRM_ERR_ARG;
    }
    if (panel->n != bg->n) {
        set_error("scan: panel has %ld cells, background has %ld", panel->n, bg->n);
        return CRM_ERR_ARG;
    }
    if (panel->grouped && panel->m + 1 > BLOCK_SLACK_MAX) {
        set_error("scan: grouped panel with %ld groups (supported up to %d)", panel->m, BLOCK_SLACK_MAX - 1);
        return CRM_ERR_UNSUPPORTED;
    }
    if (first < 0 || count < 0 || first + count > panel->p) {
        set_error("scan: variants [%ld, %ld) outside the panel (p = %ld)", first, first + count, panel->p);
        return CRM_ERR_ARG;
    }
    for (crm_gene* g : genes) {
        // the shared pass computes g'W, the context features and the donor tables once, from the first
        // gene's W and E0: the others must hold the same values, not just the same shapes
        if (g->bg != bg || g->c != g0->c || g->k0 != g0->k0 || g->w_key != g0->w_key || g->e0_key != g0->e0_key) {
            set_error("scan: genes of one call must share the background, W and E0 (contents, not only shapes)");
            return CRM_ERR_ARG;
        }
    }
    if (count == 0) return CRM_OK;
    if (ctx->in_scan) {
        set_error("scan: another scan is running on this context (started from a progress callback?); its work buffers are in use");
        return CRM_ERR_UNSUPPORTED;
    }
    struct InScan { crm_ctx* c; explicit InScan(crm_ctx* c_) : c(c_) { c->in_scan = true; } ~InScan() { c->in_scan = false; } } in_scan(ctx);
    {
        // (the Gram kernel stages all k0 + c + 2 rows of a variant in LDS: refused here, before anything is launched, with
        // the limit named; past 144 rows / 128 contexts the scan runs through the slower forms of its per-variant kernels)
        if (g0->k0 + g0->c + 2 > CRM_MAX_GRAM_ROWS) {
            set_error("interaction scan: %d contexts with %d covariate columns (supported: contexts + covariates + 2 <= %d; "
                      "the association scans take up to %d covariate columns)", g0->k0, g0->c, CRM_MAX_GRAM_ROWS,
                      CRM_MAX_COV_XWIDE);
            return CRM_ERR_UNSUPPORTED;
        }
    }
    if (ctx->polish && g0->c > CRM_MAX_COV) {
        set_error("interaction scan: the null-fit polish is only built for up to %d covariate columns", CRM_MAX_COV);
        return CRM_ERR_UNSUPPORTED;
    }
    CRM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const long n = bg->n, np = bg->n_pad, ldq = bg->ldq;
    const int nrho = bg->nrho, c = g0->c, k0 = g0->k0;
    for (long i = 0; i < n; i++) {
        if ((idx_E && (idx_E[i] < 0 || idx_E[i] >= n)) || (idx_G && (idx_G[i] < 0 || idx_G[i] >= n))) {
            set_error("scan: permutation index out of range at position %ld", i);
            return CRM_ERR_ARG;
        }
    }
    // several genes may ask for several rho* per variant: keep the (variant, rho) pair list bounded
    const bool slow_forms = scan_slow_forms(g0);
    int BLK = scan_block_variants(ctx, g0, count);
    // Several phenotypes: the pair-ordered buffers (A~ and, on the routes through H, its gathered operand) grow with the
    // number of distinct (variant, rho*) pairs, up to min(nrho, ng) per variant.  They are kept within 128 GB (under half of the
    // device) by running the pair stage of a block -- steps 5 to 11 -- over sub-ranges of its variants, while the stages
    // before it (block copies, rotations and, above all, the per-phenotype null fits, which run twice as fast per variant in
    // launches of 4096 variants as in launches of 2048) keep the full block.
    int pair_cap = BLK;
    if (ng > 1) {
        const char* cap_env = getenv("CRM_PAIR_BUFFER_GB");
        const double cap_gb = cap_env && atof(cap_env) > 0 ? atof(cap_env) : 128.0;
        const double per_pair = 2.0 * sizeof(double) * g0->k0 * (double)bg->ldq;
        const long most = (long)std::min(nrho, ng) * BLK, least = (long)std::min(nrho, ng) * std::min(BLK, 128);
        pair_cap = (int)std::max<long>(least, std::min<long>(most, (long)(cap_gb * (1ull << 30) / per_pair)));
    }
    const int max_pairs = pair_cap;
    const long ldb = BLK + 128;              // slack columns for the Khatri-Rao tile over-read
    const long ldp = max_pairs + 128;        // pair-ordered copy of the block
    const long ldA = ldq, ldT = ldq;
    const int npair = k0 * (k0 + 1) / 2;
    const long ldZ1 = round_up((long)k0 * (1 + c), 128), ldZ2 = round_up(k0, 128), ldZ3 = round_up(npair, 128);
    const int KT = k0 + c + 2;
    const long ld_gW = round_up(std::max(c, CRM_MAX_COV), 8);

    // ---- context features for this permutation (E, E (x) E shared; y o E per gene) --------------
    CRM_TRY(g0->idx.ensure(sizeof(int) * 2 * n));
    int* d_idxE = nullptr;
    int* d_idxG = nullptr;
    if (idx_E) {
        d_idxE = g0->idx.as<int>();
        CRM_HIP(hipMemcpyAsync(d_idxE, idx_E, sizeof(int) * n, hipMemcpyHostToDevice, st));
    }
    if (idx_G) {
        d_idxG = g0->idx.as<int>() + n;
        CRM_HIP(hipMemcpyAsync(d_idxG, idx_G, sizeof(int) * n, hipMemcpyHostToDevice, st));
    }
    for (crm_gene* g : genes) {
        g->ld_ep = round_up(k0, 128);
        g->ld_ye = ldZ1;
        g->ld_ee = ldZ3;
    }
    // y o E, W o E per gene; the permuted contexts and their pair products E (x) E once -- unless the
    // scan runs collapsed on donor tables the background already holds (then nothing reads them)
    const double* d_Ep = nullptr;
    const double* d_EE = nullptr;
    auto context_features = [&](bool shared_too) -> int {
        for (int gi = 0; gi < ng; gi++) {
            crm_gene* g = genes[gi];
            const bool both = gi == 0 && shared_too;
            CRM_TRY(g->YE.ensure(sizeof(double) * np * g->ld_ye));
            if (both) {
                CRM_TRY(g->Ep.ensure(sizeof(double) * np * g->ld_ep));
                CRM_TRY(g->EE.ensure(sizeof(double) * np * g->ld_ee));
            }
            CRM_TRY(launch_context_features(st, g->E0.as<double>(), g->lde, d_idxE, n, np, k0, g->yW.as<double>(),
                                            g->yW.as<double>() + 1, g->ld_yw, c,
                                            both ? g->Ep.as<double>() : nullptr, g->ld_ep, g->YE.as<double>(),
                                            g->ld_ye, both ? g->EE.as<double>() : nullptr, g->ld_ee));
        }
        if (shared_too) {
            d_Ep = g0->Ep.as<double>();
            d_EE = g0->EE.as<double>();
        }
        return CRM_OK;
    };

    // ---- workspaces ----------------------------------------------------------------------------
    CRM_TRY(ctx->ws_T.ensure(sizeof(double) * (size_t)nrho * BLK * ldT));
    CRM_TRY(ctx->ws_A.ensure(sizeof(double) * (size_t)max_pairs * k0 * ldA));
    CRM_TRY(ctx->ws_Anone.ensure(sizeof(double) * (size_t)ldA));
    CRM_HIP(hipMemsetAsync(ctx->ws_Anone.ptr, 0, sizeof(double) * (size_t)ldA, st));
    // A fit that ends with (practically) no kinship term -- delta at the upper clamp, v0 = 2.2e-16 scale: a phenotype without
    // a random effect, half of an eQTL run -- has K0 = v1 (I + (v0 / v1) Q0 S0 Q0'): where (v0 / v1) max S0 <= 1e-10 the
    // rotated test direction A~ enters Q and F with weights d_j <= 1e-10, below the tolerance of the test by four orders of
    // magnitude, while its product is most of a step.  Such tests get no (variant, rho*) pair: their Gram reads rows of zeros
    // (AssembleArgs::A_none).  rho* of such a fit is decided by rounding (the likelihood is flat in rho), so over many
    // phenotypes these are also the fits that would scatter a variant's pairs over the whole grid.
    if ((int)bg->s0_max.size() != nrho) {   // (filled when the background was sealed / created)
        set_error("scan: the background was not sealed");
        return CRM_ERR_INTERNAL;
    }
    const bool skip_pairs = form("pairs_without_kinship_term", 1) == 0 ? false : true;
    auto no_kinship_term = [&](const NullFitOut& f) {
        return skip_pairs && f.v1 > 0.0 && f.v0 >= 0.0 && f.v0 * bg->s0_max[f.rho_index] <= 1e-10 * f.v1;
    };
    CRM_TRY(ctx->ws_Gb.ensure(sizeof(double) * (size_t)np * ldb));
    CRM_TRY(ctx->ws_Gs.ensure(sizeof(double) * (size_t)np * ldp));
    CRM_TRY(ctx->ws_G2.ensure(sizeof(double) * (size_t)np * ldb));
    if (idx_G)   // (unused when the scan ends up on the collapsed path)
        CRM_TRY(ctx->ws_Gt.ensure(sizeof(double) * (size_t)np * ldb));
    const int mt_blk = (BLK + GEMM_BM - 1) / GEMM_BM;
    const int ks1 = pick_split(np, (long)mt_blk * (ldZ1 / GEMM_BN));
    const int ks2 = pick_split(np, (long)mt_blk * (ldZ2 / GEMM_BN));
    const int ks3 = pick_split(np, (long)mt_blk * (ldZ3 / GEMM_BN));
    const long z1_sz = (long)BLK * ldZ1, z2_sz = (long)BLK * ldZ2, z3_sz = (long)BLK * ldZ3;
    // several phenotypes: Z1 = Gt' [y o E, W o E] of all of them in ONE batched launch per block (a problem per phenotype,
    // each with its own output region) instead of a skinny launch + reduction per phenotype -- at config 4 those 64 pairs of
    // launches were an eighth of the scan.  The slices along the cell axis shrink with the number of problems.
    const bool z1_batched = ng > 1;
    const int ks1b = z1_batched ? pick_split(np, (long)mt_blk * (ldZ1 / GEMM_BN) * ng) : ks1;
    const long z1_all = z1_batched ? z1_sz * ks1b * ng : z1_sz * ks1;
    CRM_TRY(ctx->ws_Z.ensure(sizeof(double) * (size_t)(z1_all + z2_sz * ks2 + z3_sz * ks3)));
    double* dZ1 = ctx->ws_Z.as<double>();
    double* dZ2 = dZ1 + z1_all;
    double* dZ3 = dZ2 + z2_sz * ks2;
    // H'G of step 3: few output tiles (cols x block) against a long contraction (cells) -- slices along the cell axis
    // until the launch fills the chip twice with 128-wide tiles (mode B at config 3: 64 tiles, cfg3 mode C: 320)
    int ks_h = 1;
    // (rows of the operand of the rotations' Mix products: the half factor's columns, or -- folded kinship structure,
    // objects.h: kin_fold -- k1 + donors k2)
    const long th_slab = std::max<long>(bg->ldh, bg->kin && bg->kin_fold ? bg->kin_kdim : 0) * ldb;
    if (bg->fast_T) {
        const long tiles_h = (long)((bg->cols + GEMM_BM - 1) / GEMM_BM) * ((BLK + 127) / 128);
        while (tiles_h * ks_h < 1024 && ks_h < 16 && np / GEMM_BK / (ks_h + 1) >= 16) ks_h++;
        CRM_TRY(ctx->ws_TH.ensure(sizeof(double) * (size_t)th_slab * ks_h));
        CRM_HIP(hipMemsetAsync(ctx->ws_TH.ptr, 0, sizeof(double) * (size_t)th_slab, st));
    }
    CRM_TRY(ctx->ws_F.ensure(sizeof(double) * (size_t)BLK * k0 * k0));
    CRM_TRY(ctx->ws_Gext.ensure(sizeof(double) * (size_t)BLK * KT * KT));
    const size_t stats_ws = variant_stats_workspace(BLK, std::min(c, CRM_MAX_COV));
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_gg = carve(sizeof(double) * BLK), o_gy = carve(sizeof(double) * BLK * ng),
                 o_gW = carve(sizeof(double) * BLK * ld_gW), o_trial = carve(sizeof(NullFitTrial) * BLK * nrho),
                 o_fit = carve(sizeof(NullFitOut) * BLK * ng), o_pos = carve(sizeof(int) * BLK * ng),
                 o_ord = carve(sizeof(int) * max_pairs), o_Q = carve(sizeof(double) * BLK),
                 o_pv = carve(sizeof(double) * BLK), o_lam = carve(sizeof(double) * BLK * k0),
                 o_if = carve(sizeof(int) * BLK), o_liu = carve(sizeof(double) * BLK),
                 o_part = carve(stats_ws), o_queue = carve(sizeof(unsigned) * CRM_MAX_RHO),
                 o_coef = carve(sizeof(double) * (size_t)c * ldb), o_thr = carve(sizeof(double) * BLK),
                 o_drop = carve(sizeof(int) * BLK), o_near = carve(sizeof(int) * BLK);
    CRM_TRY(ctx->ws_small.ensure(off));
    char* sm = ctx->ws_small.as<char>();
    double* d_gg = (double*)(sm + o_gg);
    double* d_gy = (double*)(sm + o_gy);          // [ng][BLK]
    double* d_gW = (double*)(sm + o_gW);
    NullFitTrial* d_trial = (NullFitTrial*)(sm + o_trial);
    NullFitOut* d_fit = (NullFitOut*)(sm + o_fit);  // [ng][BLK]
    int* d_pos = (int*)(sm + o_pos);               // [ng][BLK]
    int* d_ord = (int*)(sm + o_ord);
    double* d_Q = (double*)(sm + o_Q);
    double* d_pv = (double*)(sm + o_pv);
    double* d_lam = (double*)(sm + o_lam);
    int* d_if = (int*)(sm + o_if);
    double* d_liu = (double*)(sm + o_liu);
    double* d_part = (double*)(sm + o_part);
    unsigned* d_queue = (unsigned*)(sm + o_queue);   // work queue of the null fits (one counter per grid point)
    double* d_coef = (double*)(sm + o_coef);         // [c][ldb] projection coefficients of the block onto W
    double* d_thr = (double*)(sm + o_thr);           // the reference's rank rule as a bound on |gx|^2
    int* d_drop = (int*)(sm + o_drop);               // 1: the variant's direction is dropped from [W, g]
    int* d_near = (int*)(sm + o_near);               // collapsed path: 1 = repeat this variant on the dense path
    const int kin_probs = bg->kin ? bg->kin_groups + bg->kin_k2 + 16 : 0;
    CRM_TRY(ctx->ws_probs.ensure(sizeof(GemmProblem) * (2 * CRM_MAX_RHO + 4 + kin_probs + ng)));
    GemmProblem* d_probs = ctx->ws_probs.as<GemmProblem>();

    const long slab = (long)(1 + c) * ldq;  // rotations of [y, W] per grid point
    // donor-collapsed mode: exact when every variant is constant within the panel's groups and the
    // genotype permutation hook is not in use
    const bool grouped = panel->grouped;
    const size_t bd_bytes = grouped ? sizeof(double) * (size_t)nrho * panel->m_pad * k0 * ldq : 0;
    // (with the genotype permutation hook the test direction is constant within the permuted groups;
    // its mixed table needs the indicators as Khatri-Rao "contexts": m <= 128)
    const bool collapsed = grouped && ctx->collapse && allow_collapse && bd_bytes <= ((size_t)48 << 30) &&
                           (!idx_G || panel->m <= 128);
    if (!collapsed) {   // the block in the fixed effects' own basis, and its product with the test direction
        CRM_TRY(ctx->ws_Gx.ensure(sizeof(double) * (size_t)np * ldb));
        CRM_TRY(ctx->ws_GG.ensure(sizeof(double) * (size_t)np * ldb));
    }
    const bool cross = collapsed && idx_G;
    const long ld_ah = round_up((long)BLK * k0, 128) + 128, ld_xg = round_up((long)max_pairs * k0, 128) + 128;
    // Kinship-structure route (objects.h, crm_background::kin): H'(g o E0) donor by donor, then Mix(rho*)' -- the dense
    // scan's default whenever the background knows the donor structure of its kinship factor.  S: per-donor sums.
    const int KK = bg->kin ? bg->kin_k1 + bg->kin_k2 : 0;   // rows of S per donor: [us | E1]
    // folded form (objects.h: kin_fold): S holds [E1 rows ; (donor, us_j) rows] and is the operand of the Mix product itself
    const bool fold = bg->kin && bg->kin_fold;
    const size_t s_bytes = !bg->kin ? 0 : sizeof(double) * (fold ? (size_t)bg->kin_kdim : (size_t)bg->kin_groups_pad * KK) * ld_ah;
    // The route pays when its flops per variant -- per-donor sums over runs padded to whole 16-cell stages, the E1 rows /
    // the contraction over the donors, and the product with the mixing matrix -- stay under the direct contraction's
    // 2 n r k0 (thousands of tiny donors: every run is mostly padding); a multi-gene test that forces one of the other
    // two routes (crm_test_set_shared_h 0 / 1) gets that route.
    bool kin_pays = false;
    if (bg->kin) {
        double rbar = 1.0;
        for (int i = 0; i < nrho; i++) rbar = std::max(rbar, (double)bg->r[i]);
        const double kk = fold ? (double)bg->kin_kdim : (double)bg->ldh;
        const double prep = fold ? 2.0 * bg->kin_rows * bg->kin_k2 + 2.0 * (double)np * bg->kin_k1
                                 : 2.0 * bg->kin_rows * KK + 2.0 * (double)bg->kin_groups_pad * bg->kin_cols * bg->kin_k2;
        kin_pays = prep + 2.0 * kk * rbar < 0.9 * 2.0 * (double)n * rbar || ctx->kin_route >= 2;
    }
    const bool kin_route = bg->kin && bg->fast_T && ctx->fast_T && !collapsed && ctx->kin_route > 0 && kin_pays &&
                           !(ng > 1 && ctx->tune.shared_h >= 0) && s_bytes <= ((size_t)48 << 30);
    const bool kfold = kin_route && fold;
    const long kdim = kfold ? bg->kin_kdim : bg->ldh;       // contraction length of the products with the mixing matrices
    // cell-axis slices of the folded form's all-cells launches for the E1 rows (few output tiles, long contraction)
    int fold_split6 = 1, fold_split3 = 1;
    // E1 rows of step 6: as a plain product G'P with the pair products P = E1_a o E0_i (n x k1 k0; the contraction kernel's
    // best form) followed by a re-ordering of its rows, unless P would be large (> 8 GB): then as a Khatri-Rao contraction
    // over all cells with the transposed store (64-wide tiles when k1 <= 64: 50 of 64 columns at config 3)
    const long ldP = round_up((long)(bg->kin ? bg->kin_k1 : 0) * k0, 128);
    const bool e1_pairs = kfold && sizeof(double) * (double)np * (double)ldP <= 8.0 * (1ull << 30);
    if (kfold) {
        const long tiles6 = e1_pairs ? ((long)BLK + GEMM_BM - 1) / GEMM_BM * (ldP / 128) : ((long)BLK * k0 + GEMM_BM - 1) / GEMM_BM;
        const long slots6 = e1_pairs || bg->kin_k1 > 64 ? 512 : 768;
        double best = 0.0;
        for (int sps = 1; sps <= 8 && np / GEMM_BK / sps >= 64; sps++) {
            const double rounds = (double)(tiles6 * sps) / (double)slots6, eff = rounds / std::ceil(rounds);
            if (eff > best + 0.02) { best = eff; fold_split6 = sps; }
        }
        const long tiles3 = (long)((bg->kin_k1 + GEMM_BM - 1) / GEMM_BM) * ((BLK + 127) / 128);
        while (tiles3 * fold_split3 < 1024 && fold_split3 < 16 && np / GEMM_BK / (fold_split3 + 1) >= 16) fold_split3++;
    }
    if (bg->fast_T && ctx->fast_T && (ng > 1 || kin_route) && !collapsed) {  // operands of the routes through H (step 6)
        if (kfold) {
            // (scratch of the sliced all-cells launch for the E1 rows)
            CRM_TRY(ctx->ws_AH.ensure(sizeof(double) * (size_t)fold_split6 *
                                      (e1_pairs ? (size_t)(std::max<long>(BLK, max_pairs) + 128) * ldP : (size_t)bg->kin_k1 * ld_ah)));
        } else {
            CRM_TRY(ctx->ws_AH.ensure(sizeof(double) * (size_t)bg->ldh * ld_ah));
            // (zeroed further down, once the form of the per-donor sums is known: all of it, or its padding rows alone)
        }
        if (ng > 1) CRM_TRY(ctx->ws_XG.ensure(sizeof(double) * (size_t)kdim * ld_xg));
    }
    if (kfold) {
        CRM_TRY(ctx->ws_S.ensure(s_bytes));
        CRM_TRY(ctx->ws_Gk.ensure(sizeof(double) * (size_t)bg->kin_rows * std::max(ldb, ldp)));
        CRM_TRY(ctx->ws_S2.ensure(sizeof(double) * (size_t)fold_split3 * bg->kin_k1 * ldb));
        // rows between k1 + donors k2 and the padded contraction length stay zero
        const long used = bg->kin_k1 + bg->kin_groups * (long)bg->kin_k2;
        if (kdim > used)
            CRM_HIP(hipMemsetAsync(ctx->ws_S.as<double>() + (size_t)used * ld_ah, 0, sizeof(double) * (size_t)(kdim - used) * ld_ah, st));
    } else if (kin_route) {
        CRM_TRY(ctx->ws_S.ensure(s_bytes));
        CRM_TRY(ctx->ws_Gk.ensure(sizeof(double) * (size_t)bg->kin_rows * std::max(ldb, ldp)));
        CRM_TRY(ctx->ws_S2.ensure(sizeof(double) * (size_t)bg->kin_groups_pad * KK * ldb));
        if (bg->kin_groups_pad > bg->kin_groups)
            CRM_HIP(hipMemsetAsync(ctx->ws_S2.as<double>() + (size_t)bg->kin_groups * KK * ldb, 0,
                                   sizeof(double) * (size_t)(bg->kin_groups_pad - bg->kin_groups) * KK * ldb, st));
        // rows of the padding donors (kin_groups .. kin_groups_pad) are operands of the contraction over the donors
        if (bg->kin_groups_pad > bg->kin_groups)
            CRM_HIP(hipMemsetAsync(ctx->ws_S.as<double>() + (size_t)bg->kin_groups * KK * ld_ah, 0,
                                   sizeof(double) * (size_t)(bg->kin_groups_pad - bg->kin_groups) * KK * ld_ah, st));
    }
    const long mp = grouped ? panel->m_pad : 0;
    crm_donor_tables* tab = nullptr;  // phenotype-free donor tables of this call (collapsed mode)
    if (collapsed) {
        const double* Zt = panel->Z.as<double>();
        if (cross) {
            CRM_TRY(g0->dt_Zt.ensure(sizeof(double) * (size_t)np * panel->ldz + sizeof(int) * n));
            int* gperm = reinterpret_cast<int*>(g0->dt_Zt.as<double>() + (size_t)np * panel->ldz);
            hipLaunchKernelGGL(permute_group_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               panel->group.as<int>(), d_idxG, n, gperm);
            CRM_HIP(hipGetLastError());
            CRM_TRY(launch_indicator(st, gperm, n, np, (int)panel->m, g0->dt_Zt.as<double>(), panel->ldz));
            Zt = g0->dt_Zt.as<double>();
        }
        // phenotype-free tables: shared through the background when no permutation hook is in use
        // (key: contents of E0 and of the donor index), else private to this call's first gene
        const bool reusable = !idx_E && !idx_G;
        bool build_shared = false;
        if (reusable) {
            for (crm_donor_tables* t : bg->dt_cache)
                if (t->e0_key == g0->e0_key && t->group_key == panel->group_key) tab = t;
            if (!tab) {
                if ((int)bg->dt_cache.size() >= crm_background::DT_CACHE) {  // drop the least recently used
                    size_t lru = 0;
                    for (size_t i = 1; i < bg->dt_cache.size(); i++)
                        if (bg->dt_cache[i]->stamp < bg->dt_cache[lru]->stamp) lru = i;
                    tab = bg->dt_cache[lru];
                } else {
                    tab = new crm_donor_tables();
                    bg->dt_cache.push_back(tab);
                }
                tab->e0_key = 0;  // invalid until built
                build_shared = true;
            }
            tab->stamp = ++bg->dt_clock;
        } else {
            tab = &g0->dt_own;
            build_shared = true;
        }
        CRM_TRY(context_features(build_shared));
        for (int gi = 0; gi < ng; gi++) {
            crm_gene* g = genes[gi];
            const bool have = reusable && g->dt_group == panel->group_key && !(gi == 0 && build_shared);
            if (!have) {
                g->dt_group = 0;
                CRM_TRY(build_donor_tables(g, panel, (gi == 0 && build_shared) ? tab : nullptr, d_Ep, d_EE, Zt, cross));
                if (reusable) g->dt_group = panel->group_key;
            }
        }
        if (build_shared && reusable) {
            tab->e0_key = g0->e0_key;
            tab->group_key = panel->group_key;
        }
    }
    if (!collapsed) CRM_TRY(context_features(true));
    if (kin_route) {   // the (permuted) contexts in donor order
        CRM_TRY(g0->kinEp.ensure(sizeof(double) * (size_t)bg->kin_rows * g0->ld_ep));
        CRM_TRY(launch_gather_rows(st, d_Ep, g0->ld_ep, bg->kin_map.as<int>(), bg->kin_rows, (int)g0->ld_ep,
                                   g0->kinEp.as<double>(), g0->ld_ep));
    }
    if (kfold && bg->kin_k2 == 1) {
        // one column of us: S[(k1 + d'), (b, i)] = sum over the cells of donor d' of us(c) g_b(c) E0(c, i) is the plain product
        // G_d'' (us o E0)_d' of the donor's own cells -- its (b, i) layout is the row of S as it stands.  kinUE = us o E0 in
        // donor order.
        CRM_TRY(g0->kinUE.ensure(sizeof(double) * (size_t)bg->kin_rows * g0->ld_ep));
        CRM_TRY(launch_scale_rows(st, g0->kinEp.as<double>(), g0->ld_ep, bg->kin_Y.as<double>(), bg->kin_ldy, bg->kin_rows,
                                  (int)g0->ld_ep, g0->kinUE.as<double>(), g0->ld_ep));
    }
    // E1 = E, the reference's default (and no context permutation): the pair features E1_a o E0_i are the symmetric
    // E_a E_i that the scan holds anyway for E0'diag(g^2)E0 (EE: k0 (k0 + 1) / 2 columns) -- half the product
    bool e1_sym = false;
    if (e1_pairs && bg->kin_k1 == k0 && d_EE) {
        int h_flag = 0;
        int* d_flag = reinterpret_cast<int*>(d_near);
        CRM_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
        CRM_TRY(launch_same_columns(st, bg->H.as<double>(), bg->ldh, d_Ep, g0->ld_ep, n, k0, d_flag));
        CRM_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        e1_sym = h_flag == 0;
    }
    if (e1_pairs && !e1_sym) {
        CRM_TRY(g0->kinP.ensure(sizeof(double) * (size_t)np * ldP));
        CRM_TRY(launch_pair_features(st, bg->H.as<double>(), bg->ldh, bg->kin_k1, d_Ep, g0->ld_ep, k0, np, g0->kinP.as<double>(), ldP));
    }
    // The kinship term's contexts are E as well (the reference's default E2 = E): the per-donor sums S_d = sum_c g_c e_c e_c'
    // are symmetric -- one batched product per donor against E (x) E in donor order, half the flops of the Khatri-Rao form and
    // a plain product, then a pass that writes the rows of S (blockops.hip: donor_pairs_expand_kernel); the E1 rows are the
    // sum of those products over the donors, so their product over all cells goes as well.  Taken where its time is the
    // smaller one (many tiny donors: the pass over S costs more than the products save).
    bool donor_pairs = false;
    const long ldPd = round_up((long)npair, 128);
    const long pd_rows = std::max<long>(BLK, max_pairs) + 128, pd_slab = pd_rows * ldPd;
    int donor_pair_splits = 1;
    if (kfold && e1_sym && bg->kin_k2 == k0 && donor_pairs_serves(k0) && form("donor_pairs", 1)) {
        const double peak = 78.6e12, hbm = 4.0e12;
        const double t_kr = 2.0 * bg->kin_rows * (double)k0 * k0 / (0.6 * peak) + 2.0 * (double)np * npair / (0.92 * peak);
        const double t_pairs = 2.0 * bg->kin_rows * (double)npair / (0.8 * peak) +
                               (double)bg->kin_groups * (2.0 * npair + (double)k0 * k0) * sizeof(double) / hbm;
        const bool fits = sizeof(double) * (double)bg->kin_groups * (double)pd_slab <= 8.0 * (1ull << 30);
        if (fits && (t_pairs < t_kr || form("donor_pairs", 1) >= 2)) {
            int h_flag = 0;
            int* d_flag = reinterpret_cast<int*>(d_near);
            CRM_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
            CRM_TRY(launch_same_columns(st, bg->kin_Y.as<double>(), bg->kin_ldy, g0->kinEp.as<double>(), g0->ld_ep, bg->kin_rows, k0, d_flag));
            CRM_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
            CRM_HIP(hipStreamSynchronize(st));
            donor_pairs = h_flag == 0;
        }
        if (donor_pairs) {
            CRM_TRY(g0->kinEE.ensure(sizeof(double) * (size_t)bg->kin_rows * g0->ld_ee));
            CRM_TRY(launch_gather_rows(st, d_EE, g0->ld_ee, bg->kin_map.as<int>(), bg->kin_rows, (int)g0->ld_ee, g0->kinEE.as<double>(),
                                       g0->ld_ee));
            CRM_TRY(ctx->ws_Pd.ensure(sizeof(double) * (size_t)bg->kin_groups * pd_slab));
            // the expansion pass: four variants per workgroup, three workgroups per CU -- donor ranges fill its rounds
            const long wgs = (std::max<long>(BLK, 1) + 3) / 4;
            while ((wgs * donor_pair_splits) % 768 != 0 && wgs * donor_pair_splits < 4 * 768 && donor_pair_splits < 8 &&
                   donor_pair_splits < bg->kin_groups)
                donor_pair_splits++;
            CRM_TRY(ctx->ws_AH.ensure(sizeof(double) * (size_t)std::max<long>(donor_pair_splits, fold_split6) *
                                      (size_t)std::max<long>(pd_slab, (std::max<long>(BLK, max_pairs) + 128) * ldP)));
        }
    }
    // The same idea on the UNFOLDED kinship-structure route (few contexts: BASELINE config 2's 20): with E1 = E2 = E the
    // per-donor blocks [us | E1]'(g o E0) are one symmetric matrix S_d = sum_c g_c e_c e_c' twice over -- one batched plain
    // product per donor against E (x) E in donor order (P_d), the contraction over the donors with the kinship factor ON THE
    // PAIR PRODUCTS (Z_c = sum_d hKd[d, c] P_d: 210 columns per variant instead of 400, and an extra column of ones in hKd
    // gives the sum over the donors that the E1 rows are), then the rows of AH = H'(g o E0) written from Z in one pass --
    // instead of the Khatri-Rao launch per donor (64-wide tiles a third full), the contraction over the donors on k0 x k0
    // blocks and the E1 sums.  Config 2: 1.7 -> 0.7 ms of a 8.6 ms step.
    bool pairs_unfolded = false;
    if (kin_route && !kfold && d_EE && bg->kin_k1 == k0 && bg->kin_k2 == k0 && donor_pairs_serves(k0) && form("donor_pairs", 1) &&
        bg->kin_cols + 1 <= bg->kin_ldh) {
        const double cost_kr = 2.0 * bg->kin_rows * (double)KK * k0 + 2.0 * (double)bg->kin_groups_pad * bg->kin_cols * bg->kin_k2 * k0;
        const double cost_pairs = 2.0 * bg->kin_rows * (double)npair + 2.0 * (double)(bg->kin_cols + 1) * bg->kin_groups_pad * (double)ldPd;
        const bool fits = sizeof(double) * (double)bg->kin_groups_pad * (double)pd_slab <= 8.0 * (1ull << 30) &&
                          sizeof(double) * (double)(bg->kin_cols + 1) * (double)pd_slab <= (double)s_bytes;
        if (fits && (cost_pairs < 0.8 * cost_kr || form("donor_pairs", 1) >= 2)) {
            int h_flag = 0;
            int* d_flag = reinterpret_cast<int*>(d_near);
            CRM_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
            CRM_TRY(launch_same_columns(st, bg->H.as<double>(), bg->ldh, d_Ep, g0->ld_ep, n, k0, d_flag));
            CRM_TRY(launch_same_columns(st, bg->kin_Y.as<double>(), bg->kin_ldy, g0->kinEp.as<double>(), g0->ld_ep, bg->kin_rows, k0, d_flag));
            CRM_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
            CRM_HIP(hipStreamSynchronize(st));
            pairs_unfolded = h_flag == 0;
        }
        if (pairs_unfolded) {
            CRM_TRY(g0->kinEE.ensure(sizeof(double) * (size_t)bg->kin_rows * g0->ld_ee));
            CRM_TRY(launch_gather_rows(st, d_EE, g0->ld_ee, bg->kin_map.as<int>(), bg->kin_rows, (int)g0->ld_ee, g0->kinEE.as<double>(),
                                       g0->ld_ee));
            // (the slabs of the padding donors meet zero rows of hKd in the contraction over the donors: they must be finite --
            // cleared here, not left to whatever the allocation or an earlier call put there)
            CRM_TRY(ctx->ws_Pd.ensure(sizeof(double) * (size_t)bg->kin_groups_pad * pd_slab));
            if (bg->kin_groups_pad > bg->kin_groups)
                CRM_HIP(hipMemsetAsync(ctx->ws_Pd.as<double>() + (size_t)bg->kin_groups * pd_slab, 0,
                                       sizeof(double) * (size_t)(bg->kin_groups_pad - bg->kin_groups) * pd_slab, st));
            // the column of ones behind the kinship factor's m columns (kin_hKd: kin_groups_pad x kin_ldh, zero beyond m)
            std::vector<double> ones((size_t)bg->kin_groups, 1.0);
            CRM_HIP(hipMemcpy2DAsync(bg->kin_hKd.as<double>() + bg->kin_cols, sizeof(double) * bg->kin_ldh, ones.data(), sizeof(double),
                                     sizeof(double), bg->kin_groups, hipMemcpyHostToDevice, st));
            CRM_HIP(hipStreamSynchronize(st));   // (ones lives on this stack frame)
        }
    }
    if (bg->fast_T && ctx->fast_T && (ng > 1 || kin_route) && !collapsed && !kfold) {
        // AH = H'(g o E0), the operand of the products with the mixing matrices: its rows beyond the half factor's columns meet
        // zero rows of Mix and must be finite -- zero.  The pair-feature form writes every row below them for every column it
        // is read at (columns beyond the block's only feed output rows that are never stored), so the padding rows are all
        // there is to clear: 4 rows instead of 0.67 GB per call at config 2.
        if (pairs_unfolded && bg->ldh > bg->cols)
            CRM_HIP(hipMemsetAsync(ctx->ws_AH.as<double>() + (size_t)bg->cols * ld_ah, 0, sizeof(double) * (size_t)(bg->ldh - bg->cols) * ld_ah, st));
        else if (!pairs_unfolded)
            CRM_HIP(hipMemsetAsync(ctx->ws_AH.ptr, 0, sizeof(double) * (size_t)bg->ldh * ld_ah, st));
    }
    const long xrows = collapsed ? mp : np;  // length of the contraction axis in this mode
    std::vector<NullFitOut> h_fit((size_t)BLK * ng);
    std::vector<int> h_pos((size_t)BLK * ng), h_ord(max_pairs);
    std::vector<GemmProblem> probs(CRM_MAX_RHO + 4);
    std::vector<int> pair_of((size_t)nrho * BLK);
    std::vector<int> h_near(BLK);

    for (long done = 0; done < count; done += BLK) {
        const int nb = (int)std::min<long>(BLK, count - done);
        const long col0 = first + done;
        const long done_blk = done;
        double* Gb = ctx->ws_Gb.as<double>();
        TraceRange range_block("crm scan block");
        // 1. aligned copy of the block (and its row-permuted twin for the test direction); in
        //    collapsed mode the "block" is the donor dosage slab (m_pad rows)
        double* Gt = Gb;
        if (collapsed) {
            CRM_TRY(launch_gather_block(st, panel->Gd.as<double>() + col0, panel->ld, mp, panel->m, nullptr, nullptr, nb, Gb, ldb, (int)ldb));
        } else if (grouped) {
            CRM_TRY(launch_expand_block(st, panel->Gd.as<double>() + col0, panel->ld, panel->group.as<int>(), np, n, nullptr, nb, Gb, ldb, (int)ldb));
            if (idx_G) {
                Gt = ctx->ws_Gt.as<double>();
                CRM_TRY(launch_expand_block(st, panel->Gd.as<double>() + col0, panel->ld, panel->group.as<int>(), np, n, d_idxG, nb, Gt, ldb, (int)ldb));
            }
        } else {
            CRM_TRY(launch_gather_block(st, panel->G.as<double>() + col0, panel->ld, np, n, nullptr, nullptr, nb, Gb, ldb, (int)ldb));
            if (idx_G) {
                Gt = ctx->ws_Gt.as<double>();
                CRM_TRY(launch_gather_block(st, panel->G.as<double>() + col0, panel->ld, np, n, d_idxG, nullptr, nb, Gt, ldb, (int)ldb));
            }
        }
        // 2. The fixed effects' role of the variants: Gx = G - W (W'W)^-1 W'G, orthogonalised against the covariates in
        //    the cell axis as the reference's economic_svd([W, g]) basis is (blockops.hip); the test direction keeps G.
        //    Then g'g, g'W (shared) and g'y per gene of that role.  The collapsed path works on donor-level sums and
        //    cannot do this: it marks the variants that are nearly collinear with W for a second, dense pass.
        double* Gx = Gb;
        if (!collapsed) {
            Gx = ctx->ws_Gx.as<double>();
            CRM_TRY(launch_variant_stats(st, Gb, ldb, np, nb, g0->yW.as<double>(), g0->yW.as<double>() + 1, g0->ld_yw, c, d_part, d_gg, d_gy, d_gW, ld_gW));
            CRM_TRY(launch_ortho_block(st, Gb, ldb, np, nb, (int)ldb, g0->yW.as<double>() + 1, g0->ld_yw, c, g0->Wproj.as<double>(),
                                       d_gW, ld_gW, d_coef, ldb, d_thr, Gx, ldb));
        }
        for (int gi = 0; gi < ng; gi++) {
            crm_gene* g = genes[gi];
            if (collapsed)
                CRM_TRY(launch_donor_stats(st, Gb, ldb, (int)panel->m, nb, g->dt_sums.as<double>(), c, d_gg, d_gy + (size_t)gi * BLK, d_gW, ld_gW));
            else
                CRM_TRY(launch_variant_stats(st, Gx, ldb, np, nb, g->yW.as<double>(), g->yW.as<double>() + 1, g->ld_yw, c, d_part, d_gg, d_gy + (size_t)gi * BLK, d_gW, ld_gW));
        }
        if (collapsed) {
            if (near_out) CRM_TRY(launch_collinear_flag(st, d_gg, d_gW, ld_gW, g0->Wproj.as<double>(), c, nb, COLLINEAR_TAU, d_near));
        } else
            CRM_TRY(launch_ortho_rank(st, d_gg, d_thr, nb, d_drop));
        // 3. T(rho) = G' Q0(rho) for all grid points.  With Q0(rho) = H Mix(rho) the n-length work is
        //    done once, (H'G), followed by eleven small products Mix(rho)'(H'G): 2 n cols + 2 cols sum r
        //    flops per variant instead of 2 n sum r.
        const bool fastT = !collapsed && bg->fast_T && ctx->fast_T;
        if (!fastT && !collapsed) CRM_TRY(crm_background_require_q0(bg, -1));
        // (crm_scan_interaction_permuted: the passes after the first take the rotations at rho* and the fits of this block
        // from the first one's record -- neither depends on the permutation hooks)
        const bool replaying = ctx->replay_mode == 2;
        std::vector<double> flat_obj;
        if (replaying) {
            if (ng != 1 || ctx->replay_cursor >= ctx->replay_blocks.size()) {
                set_error("scan: the replayed pass visits a block the recorded one did not");
                return CRM_ERR_INTERNAL;
            }
            crm_ctx::ReplayBlock* rb = ctx->replay_blocks[ctx->replay_cursor++];
            if (rb->col0 != col0 || rb->nb != nb || rb->collapsed != collapsed || rb->fit.size() != sizeof(NullFitOut) * (size_t)nb) {
                set_error("scan: the replayed pass visits its blocks in another order than the recorded one");
                return CRM_ERR_INTERNAL;
            }
            CRM_HIP(hipMemcpyAsync(d_fit, rb->fit.data(), rb->fit.size(), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(replay_rows_kernel, dim3((unsigned)((ldT + 255) / 256), nb), dim3(256), 0, st, ctx->ws_T.as<double>(),
                               (long)BLK, ldT, d_fit, nb, (int)ldT, rb->T.as<double>(), 1);
            CRM_HIP(hipGetLastError());
        } else {
        if (fastT && kfold) {
            // folded form: rows [0, k1) = E1'G over all cells (sliced along the cell axis), rows k1 + d' k2 + j = per-donor
            // us_j'G over the donor's own cells; the contraction over the donors sits in MixK (objects.h)
            double* Gk = ctx->ws_Gk.as<double>();
            double* TH = ctx->ws_TH.as<double>();
            const int k1 = bg->kin_k1, k2 = bg->kin_k2;
            const long groups = bg->kin_groups;
            CRM_TRY(launch_gather_rows(st, Gx, ldb, bg->kin_map.as<int>(), bg->kin_rows, (int)ldb, Gk, ldb));
            std::vector<GemmProblem> kp((size_t)groups + 1);
            long maxlen = GEMM_BK;
            for (long d = 0; d < groups; d++) {
                GemmProblem p{};
                p.X = bg->kin_Y.as<double>() + bg->kin_row0[d] * bg->kin_ldy; p.ldx = bg->kin_ldy;
                p.Y = Gk + bg->kin_row0[d] * ldb; p.ldy = ldb;
                p.C = TH + (size_t)(k1 + d * k2) * ldb; p.ldc = ldb;
                p.M = k2; p.N = nb; p.cells = bg->kin_len[d];
                maxlen = std::max(maxlen, bg->kin_len[d]);
                kp[d] = p;
            }
            {
                GemmProblem p{};
                p.X = bg->H.as<double>(); p.ldx = bg->ldh; p.Y = Gx; p.ldy = ldb;
                p.C = ctx->ws_S2.as<double>(); p.ldc = ldb; p.M = k1; p.N = nb;
                kp[groups] = p;
            }
            GemmProblem* d_kp = d_probs + 2 * CRM_MAX_RHO + 4;
            CRM_HIP(hipMemcpyAsync(d_kp, kp.data(), sizeof(GemmProblem) * kp.size(), hipMemcpyHostToDevice, st));
            CRM_TRY(launch_gemm_tn(ctx, d_kp, (int)groups, k2, nb, maxlen, false, 0, 1, 0));
            const long e1_slab = (long)k1 * ldb;
            CRM_TRY(launch_gemm_tn(ctx, d_kp + groups, 1, k1, nb, np, false, 0, fold_split3, e1_slab));
            CRM_TRY(launch_reduce_splits(st, ctx->ws_S2.as<double>(), e1_slab, fold_split3, e1_slab));
            CRM_HIP(hipMemcpyAsync(TH, ctx->ws_S2.ptr, sizeof(double) * (size_t)e1_slab, hipMemcpyDeviceToDevice, st));
            CRM_HIP(hipStreamSynchronize(st));   // (kp lives on this stack frame)
        } else if (fastT && kin_route) {
            // H'G donor by donor (as H'(g o E0) in step 6): per donor [us | E1]' G over its own cells, then the L rows by a
            // contraction over the donors with hKd and the E1 rows as sums over the donors
            double* Gk = ctx->ws_Gk.as<double>();
            double* S2 = ctx->ws_S2.as<double>();
            const int k1 = bg->kin_k1, k2 = bg->kin_k2;
            const long groups = bg->kin_groups, mk = bg->kin_cols;
            CRM_TRY(launch_gather_rows(st, Gx, ldb, bg->kin_map.as<int>(), bg->kin_rows, (int)ldb, Gk, ldb));
            std::vector<GemmProblem> kp((size_t)groups + k2);
            long maxlen = GEMM_BK;
            for (long d = 0; d < groups; d++) {
                GemmProblem p{};
                p.X = bg->kin_Y.as<double>() + bg->kin_row0[d] * bg->kin_ldy; p.ldx = bg->kin_ldy;
                p.Y = Gk + bg->kin_row0[d] * ldb; p.ldy = ldb;
                p.C = S2 + (size_t)d * KK * ldb; p.ldc = ldb;
                p.M = KK; p.N = nb; p.cells = bg->kin_len[d];
                maxlen = std::max(maxlen, bg->kin_len[d]);
                kp[d] = p;
            }
            for (int j = 0; j < k2; j++) {
                GemmProblem p{};
                p.X = bg->kin_hKd.as<double>(); p.ldx = bg->kin_ldh;
                p.Y = S2 + (size_t)j * ldb; p.ldy = (long)KK * ldb;
                p.C = ctx->ws_TH.as<double>() + (size_t)(k1 + (long)j * mk) * ldb; p.ldc = ldb;
                p.M = (int)mk; p.N = nb;
                kp[groups + j] = p;
            }
            GemmProblem* d_kp = d_probs + 2 * CRM_MAX_RHO + 4;
            CRM_HIP(hipMemcpyAsync(d_kp, kp.data(), sizeof(GemmProblem) * kp.size(), hipMemcpyHostToDevice, st));
            CRM_TRY(launch_gemm_tn(ctx, d_kp, (int)groups, KK, nb, maxlen, false, 0, 1, 0));
            CRM_TRY(launch_gemm_tn(ctx, d_kp + groups, k2, (int)mk, nb, bg->kin_groups_pad, false, 0, 1, 0));
            CRM_TRY(launch_kin_sum_e1(st, S2, ldb, KK, k2, k1, (int)groups, nb, ctx->ws_TH.as<double>(), ldb));
            CRM_HIP(hipStreamSynchronize(st));   // (kp lives on this stack frame)
        } else if (fastT) {
            GemmProblem p{};
            p.X = bg->H.as<double>(); p.ldx = bg->ldh; p.Y = Gx; p.ldy = ldb;
            p.C = ctx->ws_TH.as<double>(); p.ldc = ldb; p.M = (int)bg->cols; p.N = nb;
            CRM_HIP(hipMemcpyAsync(d_probs, &p, sizeof p, hipMemcpyHostToDevice, st));
            CRM_TRY(launch_gemm_tn(ctx, d_probs, 1, (int)bg->cols, nb, np, false, 0, ks_h, th_slab));
            CRM_TRY(launch_reduce_splits(st, ctx->ws_TH.as<double>(), (long)bg->cols * ldb, ks_h, th_slab));
        }
        for (int i = 0; i < nrho; i++) {
            GemmProblem p{};
            if (fastT) {
                p.X = ctx->ws_TH.as<double>(); p.ldx = ldb;
                p.Y = kfold ? bg->MixK[i].as<double>() : bg->Mix[i].as<double>(); p.ldy = ldq;
            } else {
                p.X = Gx; p.ldx = ldb;
                p.Y = collapsed ? tab->TZ.as<double>() + (size_t)i * mp * ldq : bg->Q0[i].as<double>(); p.ldy = ldq;
            }
            p.C = ctx->ws_T.as<double>() + (size_t)i * BLK * ldT; p.ldc = ldT;
            p.M = nb; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
            probs[i] = p;
        }
        // The eleven products run as one launch of equally long tiles, i.e. in rounds of as many tiles as the chip holds
        // workgroups (two per CU): at config 3, 12 832 tiles are 25.06 rounds of 512 and the last 0.06 costs a whole one.
        // The smallest problems that make up that remainder (there: rho = 1, r = 50, 32 tiles) are taken out and run cut
        // along the contraction axis instead -- a sixteenth of a round plus a reduction.
        int n_main = nrho, n_cut = 0;
        int cut_ks = 1;
        GemmProblem cut_probs[CRM_MAX_RHO];
        double* cut_dst[CRM_MAX_RHO];
        if (fastT) {
            const long slots = 2L * ctx_cus(ctx), mtl = (nb + GEMM_BM - 1) / GEMM_BM;
            long tiles[CRM_MAX_RHO], total = 0;
            int order[CRM_MAX_RHO];
            for (int i = 0; i < nrho; i++) { tiles[i] = mtl * ((probs[i].N + 127) / 128); total += tiles[i]; order[i] = i; }
            std::sort(order, order + nrho, [&](int a, int b) { return tiles[a] < tiles[b]; });
            const long need = total % slots;
            long acc = 0;
            int take = 0;
            while (take < nrho - 1 && acc < need) acc += tiles[order[take++]];
            if (total > slots && need > 0 && acc >= need && acc <= slots / 4) {
                bool is_cut[CRM_MAX_RHO] = {false};
                long cut_doubles = 0;
                for (int q = 0; q < take; q++) is_cut[order[q]] = true;
                while ((long)(cut_ks + 1) * acc <= slots && cut_ks < 16 && kdim / GEMM_BK / (cut_ks + 1) >= 8) cut_ks++;
                n_main = 0;
                for (int i = 0; i < nrho; i++) {
                    if (!is_cut[i]) { probs[n_main++] = probs[i]; continue; }
                    GemmProblem c = probs[i];
                    cut_dst[n_cut] = c.C;
                    c.ldc = round_up(c.N, 128);
                    cut_doubles += (long)BLK * c.ldc;
                    cut_probs[n_cut++] = c;
                }
                CRM_TRY(ctx->ws_Tcut.ensure(sizeof(double) * (size_t)cut_doubles * cut_ks));
                long at = 0;
                for (int q = 0; q < n_cut; q++) {
                    cut_probs[q].C = ctx->ws_Tcut.as<double>() + at;
                    at += (long)BLK * cut_probs[q].ldc;
                }
                CRM_HIP(hipMemcpyAsync(d_probs + 1 + nrho, cut_probs, sizeof(GemmProblem) * n_cut, hipMemcpyHostToDevice, st));
                int cut_maxn = 1;
                for (int q = 0; q < n_cut; q++) cut_maxn = std::max(cut_maxn, cut_probs[q].N);
                CRM_TRY(launch_gemm_tn(ctx, d_probs + 1 + nrho, n_cut, nb, cut_maxn, kdim, false, 0, cut_ks, cut_doubles));
                CRM_TRY(launch_reduce_splits(st, ctx->ws_Tcut.as<double>(), cut_doubles, cut_ks, cut_doubles));
                for (int q = 0; q < n_cut; q++)
                    CRM_HIP(hipMemcpy2DAsync(cut_dst[q], sizeof(double) * ldT, cut_probs[q].C, sizeof(double) * cut_probs[q].ldc,
                                             sizeof(double) * cut_probs[q].N, nb, hipMemcpyDeviceToDevice, st));
            }
        }
        CRM_HIP(hipMemcpyAsync(d_probs + 1, probs.data(), sizeof(GemmProblem) * n_main, hipMemcpyHostToDevice, st));
        CRM_TRY(launch_gemm_tn(ctx, d_probs + 1, n_main, nb, (int)ldq, fastT ? kdim : xrows, false, 0, 1, 0));
        // 4. null fits + rho* per gene
        trace_push("crm null fits");
        for (int gi = 0; gi < ng; gi++) {
            crm_gene* g = genes[gi];
            NullFitArgs fa{};
            fa.nrho = nrho; fa.c = c; fa.restricted = 1; fa.n = n; fa.polish = ctx->polish ? 1 : 0; fa.exact = (ctx->nullfit_exact || form("nullfit_exact", 0)) ? 1 : 0;
            for (int i = 0; i < nrho; i++) {
                NullFitRho& R = fa.rho[i];
                R.T = ctx->ws_T.as<double>() + (size_t)i * BLK * ldT; R.ldT = ldT;
                R.ty = g->rot.as<double>() + (long)i * slab;
                R.tW = R.ty + ldq; R.ldW = ldq;
                R.S0 = bg->S0[i].as<double>();
                R.r = bg->r[i];
            }
            fa.WW = g->WW.as<double>(); fa.Wy = g->Wy.as<double>(); fa.yy = g->yy;
            fa.gg = d_gg; fa.gy = d_gy + (size_t)gi * BLK; fa.gW = d_gW; fa.ld_gW = ld_gW;
            fa.g_drop = collapsed ? nullptr : d_drop;
            if (c > CRM_MAX_COV_WIDE) {
                CRM_TRY(ctx->ws_xwide.ensure(sizeof(double) * nullfit_xwide_scratch_doubles(BLK, nrho, c)));
                fa.xwide = ctx->ws_xwide.as<double>();
            }
            fa.trial = d_trial; fa.out = d_fit + (size_t)gi * BLK;
            fa.probe = ctx->probe_on ? 1 : 0; fa.probe_x = ctx->probe_x;
            fa.track = outs[gi].flags ? 1 : 0;
            CRM_TRY(launch_nullfit(st, fa, nb, false, d_queue));
        }
        trace_pop();
        if (ctx->probe_on) {   // test hook: keep the (variant, grid point) records of this block and stop here
            std::vector<NullFitTrial> h_trial((size_t)nb * nrho);
            CRM_HIP(hipMemcpyAsync(h_trial.data(), d_trial, sizeof(NullFitTrial) * h_trial.size(), hipMemcpyDeviceToHost, st));
            CRM_HIP(hipStreamSynchronize(st));
            ctx->probe_out.assign(2 * h_trial.size(), 0.0);
            for (size_t q = 0; q < h_trial.size(); q++) {
                ctx->probe_out[2 * q] = h_trial[q].lml;
                ctx->probe_out[2 * q + 1] = h_trial[q].scale;
            }
            return CRM_OK;
        }
        }   // (not replaying)
        // 5. the (rho, variant) pairs some gene selected, ordered by rho (host; nb*ng*48 bytes cross PCIe)
        CRM_HIP(hipMemcpyAsync(h_fit.data(), d_fit, sizeof(NullFitOut) * (size_t)BLK * ng, hipMemcpyDeviceToHost, st));
        if (collapsed && near_out) CRM_HIP(hipMemcpyAsync(h_near.data(), d_near, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
        CRM_HIP(hipStreamSynchronize(st));
        if (collapsed && near_out)
            for (int b = 0; b < nb; b++)
                if (h_near[b]) near_out->push_back(done + b);
        for (int gi = 0; gi < ng; gi++)
            for (int b = 0; b < nb; b++) {
                const int ri = h_fit[(size_t)gi * BLK + b].rho_index;
                if (ri < 0 || ri >= nrho) {   // (indexes host arrays below: never trust it unchecked)
                    set_error("scan: the null fit of variant %ld (phenotype %d) did not run (grid index %d)", col0 + b, gi, ri);
                    return CRM_ERR_NUMERIC;
                }
            }
        if (ctx->replay_mode == 1) {
            if (ng != 1) {
                set_error("scan: the permutation replay serves one phenotype per call");
                return CRM_ERR_INTERNAL;
            }
            crm_ctx::ReplayBlock* rb = new crm_ctx::ReplayBlock();
            ctx->replay_blocks.push_back(rb);
            rb->col0 = col0; rb->nb = nb; rb->collapsed = collapsed;
            rb->fit.resize(sizeof(NullFitOut) * (size_t)nb);
            memcpy(rb->fit.data(), h_fit.data(), rb->fit.size());
            CRM_TRY(rb->T.ensure(sizeof(double) * (size_t)nb * ldT));
            hipLaunchKernelGGL(replay_rows_kernel, dim3((unsigned)((ldT + 255) / 256), nb), dim3(256), 0, st, ctx->ws_T.as<double>(),
                               (long)BLK, ldT, d_fit, nb, (int)ldT, rb->T.as<double>(), 0);
            CRM_HIP(hipGetLastError());
        }
        // Flat-optimum flag, first half (info calls only; include/crm_hip.h: CRM_MODEL_FLAT_OPTIMUM): how far the search of the
        // selected fit was from taking another path -- the smallest margin of the decisions on objective values that steered
        // it (brent_search.h), in units of the first-order bound on the objective's rounding noise at the optimum
        // (nullfit.hip: cur_noise; select_rho_kernel: decision).  NaN: a fit whose kernel did not measure it.
        for (int gi = 0; gi < ng; gi++) {
            if (!outs[gi].flags) continue;
            if (flat_obj.empty()) flat_obj.assign((size_t)BLK * ng, -1.0);
            for (int b = 0; b < nb; b++) {
                const double dec = h_fit[(size_t)gi * BLK + b].decision;
                flat_obj[(size_t)gi * BLK + b] = dec == dec ? dec : -1.0;
            }
        }
        // ---- the pair stage, over sub-ranges [sb0, sb0 + nsb) of the block (one sub-range unless several phenotypes ask for
        //      more (variant, rho*) pairs than the pair-ordered buffers hold).  Inside, the block-order names below stand for
        //      the sub-range: the same code serves a whole block and a part of it.
        const int nb_blk = nb;
        auto& h_fit_blk = h_fit;
        double* const Gt_blk = Gt; double* const Gx_blk = Gx; double* const Gb_blk = Gb;
        double* const d_gg_blk = d_gg; double* const d_gy_blk = d_gy; double* const d_gW_blk = d_gW;
        double* const d_coef_blk = d_coef; NullFitOut* const d_fit_blk = d_fit;
        for (int sb0 = 0; sb0 < nb_blk;) {
        int nsb = nb_blk - sb0;
        if (ng > 1) {
            long pairs = 0;
            int take = 0;
            for (; sb0 + take < nb_blk; take++) {
                unsigned seen = 0;
                for (int gi = 0; gi < ng; gi++) {
                    const NullFitOut& f = h_fit_blk[(size_t)gi * BLK + sb0 + take];
                    if (!no_kinship_term(f)) seen |= 1u << f.rho_index;
                }
                const int here = __builtin_popcount(seen);
                if (take > 0 && pairs + here > pair_cap) break;
                pairs += here;
            }
            nsb = take;
        }
        const int nb = nsb;
        const long done = done_blk + sb0;
        struct FitView { const NullFitOut* p; const NullFitOut& operator[](size_t i) const { return p[i]; } } h_fit{h_fit_blk.data() + sb0};
        double* const Gt = Gt_blk + sb0; double* const Gx = Gx_blk + sb0; double* const Gb = Gb_blk + sb0;
        double* const d_gg = d_gg_blk + sb0; double* const d_gy = d_gy_blk + sb0;
        double* const d_gW = d_gW_blk + (size_t)sb0 * ld_gW; double* const d_coef = d_coef_blk + sb0;
        NullFitOut* const d_fit = d_fit_blk + sb0;
        const int blk_cols = (int)(ldb - sb0);      // columns of the block buffers from the sub-range's first one on
        std::fill(pair_of.begin(), pair_of.end(), -1);
        long with_pair = 0;
        for (int gi = 0; gi < ng; gi++)
            for (int b = 0; b < nb; b++) {
                const NullFitOut& f = h_fit[(size_t)gi * BLK + b];
                if (no_kinship_term(f)) continue;
                pair_of[(size_t)f.rho_index * BLK + b] = 0;
                with_pair++;
            }
        ctx->tests_without_pair += (long)ng * nb - with_pair;
        // (a sub-range without any pair keeps its first test's: the launches below always have something to do)
        if (with_pair == 0) pair_of[(size_t)h_fit[0].rho_index * BLK] = 0;
        int cnt[CRM_MAX_RHO] = {0}, start[CRM_MAX_RHO + 1] = {0};
        int npairs = 0;
        for (int i = 0; i < nrho; i++) {
            start[i] = npairs;
            for (int b = 0; b < nb; b++) {
                if (pair_of[(size_t)i * BLK + b] == 0) {
                    pair_of[(size_t)i * BLK + b] = npairs;
                    h_ord[npairs++] = b;
                }
            }
            cnt[i] = npairs - start[i];
        }
        start[nrho] = npairs;
        for (int gi = 0; gi < ng; gi++)
            for (int b = 0; b < nb; b++) {
                const NullFitOut& f = h_fit[(size_t)gi * BLK + b];
                h_pos[(size_t)gi * BLK + b] = no_kinship_term(f) ? -1 : pair_of[(size_t)f.rho_index * BLK + b];
            }
        CRM_HIP(hipMemcpyAsync(d_pos, h_pos.data(), sizeof(int) * (size_t)BLK * ng, hipMemcpyHostToDevice, st));
        CRM_HIP(hipMemcpyAsync(d_ord, h_ord.data(), sizeof(int) * npairs, hipMemcpyHostToDevice, st));
        double* Gs = ctx->ws_Gs.as<double>();
        CRM_TRY(launch_gather_block(st, Gt, ldb, xrows, xrows, nullptr, d_ord, npairs, Gs, ldp, (int)ldp));
        // 6. A~ = KR(Gs, Ep)' Q0(rho), one problem per non-empty rho group of pairs.
        //    Several genes can select several rho* for one variant; with Q0(rho) = H Mix(rho) the
        //    n-length Khatri-Rao contraction is then done once per variant against H (stored transposed)
        //    and every (variant, rho) pair costs a cols-length product with Mix(rho) instead.
        bool via_H = kin_route;
        if (fastT && ng > 1 && !kin_route) {
            double direct = 0.0, via = (double)nb * (double)bg->cols * (double)n;
            for (int i = 0; i < nrho; i++) {
                direct += (double)cnt[i] * bg->r[i] * (double)n;
                via += (double)cnt[i] * bg->r[i] * (double)bg->ldh;
            }
            via_H = ctx->tune.shared_h < 0 ? via < 0.9 * direct : ctx->tune.shared_h > 0;
        }
        int nz = 0, max_m = 0, max_n = 1;
        double kr_flops = 0.0;
        if (!collapsed && !via_H)
            for (int i = 0; i < nrho; i++)
                if (cnt[i] > 0) CRM_TRY(crm_background_require_q0(bg, i));
        // few-round launches of the direct Khatri-Rao route: slices along the cell axis (see kr_split_for)
        int kr_split = 1;
        const size_t a_slab = (size_t)max_pairs * k0 * ldA;
        if (!collapsed && !via_H) {
            long row_tiles = 0;
            int nzz = 0, mn = 1;
            for (int i = 0; i < nrho; i++)
                if (cnt[i] > 0) { row_tiles += ((long)cnt[i] * k0 + GEMM_BM - 1) / GEMM_BM; nzz++; mn = std::max(mn, bg->r[i]); }
            const int cap = (int)std::min<size_t>(8, ((size_t)16 << 30) / std::max<size_t>(sizeof(double) * a_slab, 1));
            kr_split = kr_split_for(ctx, row_tiles, mn, 1, np, std::max(cap, 1));
            (void)nzz;
            if (kr_split > 1) CRM_TRY(ctx->ws_A.ensure(sizeof(double) * a_slab * kr_split));
        }
        // A spectrum a little longer than a multiple of the 128-column tile (config 3: r = 5000 = 39 tiles + 8 columns)
        // would pay a whole last column of tiles -- 1 / 40 of the launch -- for those few columns: the last 128 + rem
        // columns (rem <= 32) go into a second launch of 160-column tiles instead, cut along the cell axis to fill its
        // rounds (38 x 128 + 160 = 5024 columns computed instead of 5120).
        std::vector<GemmProblem> tails, spectrum_tails;
        int tail_split = 1, tail_maxn = 0;
        bool tail_of[CRM_MAX_RHO] = {false};
        if (!collapsed && !via_H && kr_split == 1 && ctx->tune.glds && ctx->tune.bn != 64 && ctx->tune.bn != 160 &&
            !form("kr_no_tail", 0)) {
            long tail_row_tiles = 0, main_tiles = 0;
            for (int i = 0; i < nrho; i++) {
                if (cnt[i] == 0) continue;
                const int N = bg->r[i], rem = N % 128;
                const long rt = ((long)cnt[i] * k0 + GEMM_BM - 1) / GEMM_BM;
                main_tiles += rt * ((N + 127) / 128);
                if (N >= 1024 && rem > 0 && rem <= 32) {
                    tail_of[i] = true;
                    tail_row_tiles += rt;
                    tail_maxn = std::max(tail_maxn, 128 + rem);
                }
            }
            if (tail_row_tiles == 0 || main_tiles <= 1024) {
                std::fill(tail_of, tail_of + CRM_MAX_RHO, false);
            } else {
                const int saved_bn = ctx->tune.bn;
                ctx->tune.bn = 160;
                const int cap = (int)std::min<size_t>(8, ((size_t)16 << 30) / std::max<size_t>(sizeof(double) * a_slab, 1));
                tail_split = kr_split_for(ctx, tail_row_tiles, tail_maxn, 1, np, std::max(cap, 1));
                ctx->tune.bn = saved_bn;
                // (before the problem records take addresses inside ws_A: growing the buffer does not keep its contents)
                if (tail_split > 1) CRM_TRY(ctx->ws_A.ensure(sizeof(double) * a_slab * tail_split));
            }
        }
        for (int i = 0; i < nrho; i++) {
            if (cnt[i] == 0) continue;
            GemmProblem p{};
            p.X = Gs + start[i]; p.ldx = ldp;
            p.C = ctx->ws_A.as<double>() + (size_t)start[i] * k0 * ldA;
            if (collapsed) {
                // A~(b) = sum_d gamma_d,b * Bd(rho)[d]: rows of Bd are (k0 x ldq) slabs per donor
                p.Y = tab->Bd.as<double>() + (size_t)i * mp * k0 * ldq; p.ldy = (long)k0 * ldq;
                p.ldc = (long)k0 * ldA;
                p.M = cnt[i]; p.N = (int)((long)k0 * ldq);
            } else if (via_H && kin_route && ng == 1) {   // (AH / S is in pair order already, see below)
                p.X = (kfold ? ctx->ws_S.as<double>() : ctx->ws_AH.as<double>()) + (size_t)start[i] * k0; p.ldx = ld_ah;
                p.Y = kfold ? bg->MixK[i].as<double>() : bg->Mix[i].as<double>(); p.ldy = ldq;
                p.ldc = ldA;
                p.M = cnt[i] * k0; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
                kr_flops += 2.0 * (double)(kfold ? bg->kin_k1 + bg->kin_groups * (long)bg->kin_k2 : bg->cols) * (double)bg->r[i] *
                            (double)k0 * (double)cnt[i];
            } else if (via_H) {
                p.X = ctx->ws_XG.as<double>() + (size_t)start[i] * k0; p.ldx = ld_xg;
                p.Y = kfold ? bg->MixK[i].as<double>() : bg->Mix[i].as<double>(); p.ldy = ldq;
                p.ldc = ldA;
                p.M = cnt[i] * k0; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
                kr_flops += 2.0 * (double)(kfold ? bg->kin_k1 + bg->kin_groups * (long)bg->kin_k2 : bg->cols) * (double)bg->r[i] *
                            (double)k0 * (double)cnt[i];
            } else {
                p.E = d_Ep; p.lde = g0->ld_ep; p.k0 = k0;
                p.Y = bg->Q0[i].as<double>(); p.ldy = ldq;
                p.ldc = ldA;
                p.M = cnt[i] * k0; p.N = bg->r[i] > 0 ? bg->r[i] : 1;
                kr_flops += 2.0 * (double)n * (double)bg->r[i] * (double)k0 * (double)cnt[i];
                if (tail_of[i]) {
                    GemmProblem t = p;
                    const int rem = 128 + p.N % 128;
                    p.N -= rem;
                    t.Y = p.Y + p.N; t.C = p.C + p.N; t.N = rem;
                    tails.push_back(t);
                }
            }
            // The mixing-matrix products of the kinship-structure route: a spectrum a little longer than a multiple of the
            // 128-column tile (config 3: r = 5000 = 39 tiles + 8 columns) would pay a whole last column of tiles -- 1 / 40 of
            // the launch -- for those few columns; they go through one pass over the operand instead (launch_skinny_tn).
            if (via_H && kin_route && !collapsed && p.N >= 1024 && p.N % 128 > 0 && p.N % 128 <= 16 && p.ldx % 2 == 0 &&
                (reinterpret_cast<uintptr_t>(p.X) & 15) == 0 && !form("kr_no_tail", 0)) {
                GemmProblem t = p;
                const int rem = p.N % 128;
                kr_flops -= 2.0 * (double)(kfold ? bg->kin_k1 + bg->kin_groups * (long)bg->kin_k2 : bg->cols) * (double)rem *
                            (double)k0 * (double)cnt[i];      // (the timed launch is the tiled one alone)
                p.N -= rem;
                t.Y = p.Y + p.N; t.C = p.C + p.N; t.N = rem;
                spectrum_tails.push_back(t);
            }
            max_m = std::max(max_m, p.M);
            max_n = std::max(max_n, p.N);
            probs[nz++] = p;
        }
        const bool timing = ctx->timing && ctx->timed_used < 65536;  // bounded: a forgotten timer cannot grow for ever
        if (timing) {
            if (ctx->timed_used == ctx->timed.size()) {
                hipEvent_t a, b;
                CRM_HIP(hipEventCreate(&a));
                CRM_HIP(hipEventCreate(&b));
                ctx->timed.emplace_back(a, b);
            }
            // (kinship-structure route: the pair brackets the dominant launch alone, the Mix(rho*)' product further down)
            if (!kin_route) CRM_HIP(hipEventRecord(ctx->timed[ctx->timed_used].first, st));
        }
        if (via_H && kfold) {
            // Folded form (objects.h: kin_fold): S = [E1 rows ; (donor, us_j) rows] of "H'(g o E0) before the contraction over
            // the donors", which MixK carries.  (a) the block in donor order; (b) per donor d' the Khatri-Rao contraction over
            // its own cells against us (transposed store into rows k1 + d' k2 + j); (c) the E1 rows by one Khatri-Rao
            // contraction over ALL cells against the E1 columns of the half factor, cut into slices along the cell axis so
            // that its few output tiles fill the chip, summed, and copied into rows [0, k1).
            double* Gk = ctx->ws_Gk.as<double>();
            double* S = ctx->ws_S.as<double>();
            const int k1 = bg->kin_k1, k2 = bg->kin_k2;
            const long groups = bg->kin_groups;
            const bool in_pair_order = ng == 1;
            const double* Gsrc = in_pair_order ? Gs : Gt;
            const long ldg_k = in_pair_order ? ldp : ldb;
            const int ncol = in_pair_order ? npairs : nb;
            CRM_TRY(launch_gather_rows(st, Gsrc, ldg_k, bg->kin_map.as<int>(), bg->kin_rows, in_pair_order ? (int)ldg_k : blk_cols, Gk, ldg_k));
            std::vector<GemmProblem> kp((size_t)groups + fold_split6);
            long maxlen = GEMM_BK;
            for (long d = 0; d < groups; d++) {
                GemmProblem p{};
                p.X = Gk + bg->kin_row0[d] * ldg_k; p.ldx = ldg_k;
                p.E = g0->kinEp.as<double>() + bg->kin_row0[d] * g0->ld_ep; p.lde = g0->ld_ep; p.k0 = k0;
                p.Y = bg->kin_Y.as<double>() + bg->kin_row0[d] * bg->kin_ldy; p.ldy = bg->kin_ldy;
                p.C = S + (size_t)(k1 + d * k2) * ld_ah; p.ldc = ld_ah;
                p.M = ncol * k0; p.N = k2; p.cells = bg->kin_len[d];
                if (k2 == 1) {   // plain product G_d'' (us o E0)_d': C[b, i] = row k1 + d' of S at column b k0 + i
                    p.E = nullptr; p.lde = 0; p.k0 = 0;
                    p.Y = g0->kinUE.as<double>() + bg->kin_row0[d] * g0->ld_ep; p.ldy = g0->ld_ep;
                    p.ldc = k0; p.M = ncol; p.N = k0;
                }
                maxlen = std::max(maxlen, bg->kin_len[d]);
                kp[d] = p;
            }
            // slices of whole stages along the cell axis, the last one shorter
            const long stages_all = np / GEMM_BK, per = (stages_all + fold_split6 - 1) / fold_split6;
            const long e1_slab = (long)k1 * ld_ah;
            int slices = 0;
            long chunk_max = GEMM_BK;
            if (e1_pairs) {
                GemmProblem p{};
                p.X = Gsrc; p.ldx = ldg_k; p.Y = g0->kinP.as<double>(); p.ldy = ldP;
                p.C = ctx->ws_AH.as<double>(); p.ldc = ldP; p.M = ncol; p.N = k1 * k0;
                if (e1_sym) { p.Y = d_EE; p.ldy = g0->ld_ee; p.N = npair; }
                kp[groups] = p;
            }
            for (int sps = 0; sps < fold_split6 && !e1_pairs; sps++) {
                const long s0 = sps * per, s1 = std::min(stages_all, s0 + per);
                if (s1 <= s0) break;
                GemmProblem p{};
                p.X = Gsrc + s0 * GEMM_BK * ldg_k; p.ldx = ldg_k;
                p.E = d_Ep + s0 * GEMM_BK * g0->ld_ep; p.lde = g0->ld_ep; p.k0 = k0;
                p.Y = bg->H.as<double>() + s0 * GEMM_BK * bg->ldh; p.ldy = bg->ldh;
                p.C = ctx->ws_AH.as<double>() + (size_t)sps * e1_slab; p.ldc = ld_ah;
                p.M = ncol * k0; p.N = k1; p.cells = (s1 - s0) * GEMM_BK;
                chunk_max = std::max(chunk_max, p.cells);
                kp[groups + slices++] = p;
            }
            GemmProblem* d_kp = d_probs + 2 * CRM_MAX_RHO + 4;
            if (donor_pairs) {     // per donor G_d' (E (x) E)_d, then the rows of S and the E1 rows from it
                for (long d = 0; d < groups; d++) {
                    GemmProblem& p = kp[d];
                    p.E = nullptr; p.lde = 0; p.k0 = 0;
                    p.Y = g0->kinEE.as<double>() + bg->kin_row0[d] * g0->ld_ee; p.ldy = g0->ld_ee;
                    p.C = ctx->ws_Pd.as<double>() + (size_t)d * pd_slab; p.ldc = ldPd;
                    p.M = ncol; p.N = npair;
                }
            }
            CRM_HIP(hipMemcpyAsync(d_kp, kp.data(), sizeof(GemmProblem) * (size_t)(groups + std::max(slices, 1)), hipMemcpyHostToDevice, st));
            if (donor_pairs) {
                CRM_TRY(launch_gemm_tn(ctx, d_kp, (int)groups, ncol, npair, maxlen, false, 0, 1, 0));
                CRM_TRY(launch_donor_pairs_expand(st, ctx->ws_Pd.as<double>(), pd_slab, ldPd, (int)groups, ncol, k0, k1, S, ld_ah,
                                                  ctx->ws_AH.as<double>(), pd_slab, donor_pair_splits));
                CRM_TRY(launch_reduce_splits(st, ctx->ws_AH.as<double>(), (long)ncol * ldPd, donor_pair_splits, pd_slab));
                CRM_TRY(launch_pair_rows_sym(st, ctx->ws_AH.as<double>(), ldPd, ncol, k0, S, ld_ah));
                ctx->donor_pair_blocks++;
            } else if (k2 == 1) CRM_TRY(launch_gemm_tn(ctx, d_kp, (int)groups, ncol, k0, maxlen, false, 0, 1, 0));
            else CRM_TRY(launch_kr_transposed(ctx, d_kp, (int)groups, ncol * k0, k2, maxlen, k0));
            if (donor_pairs) {
            } else if (e1_pairs) {
                const long p_slab = (long)(std::max<long>(BLK, max_pairs) + 128) * ldP;
                CRM_TRY(launch_gemm_tn(ctx, d_kp + groups, 1, ncol, e1_sym ? npair : k1 * k0, np, false, 0, fold_split6, p_slab));
                CRM_TRY(launch_reduce_splits(st, ctx->ws_AH.as<double>(), (long)ncol * ldP, fold_split6, p_slab));
                if (e1_sym) CRM_TRY(launch_pair_rows_sym(st, ctx->ws_AH.as<double>(), ldP, ncol, k0, S, ld_ah));
                else CRM_TRY(launch_pair_rows(st, ctx->ws_AH.as<double>(), ldP, ncol, k1, k0, S, ld_ah));
            } else {
                CRM_TRY(launch_kr_transposed(ctx, d_kp + groups, slices, ncol * k0, k1, chunk_max, k0));
                CRM_TRY(launch_reduce_splits(st, ctx->ws_AH.as<double>(), e1_slab, slices, e1_slab));
                CRM_HIP(hipMemcpyAsync(S, ctx->ws_AH.ptr, sizeof(double) * (size_t)e1_slab, hipMemcpyDeviceToDevice, st));
            }
            // (2 kin_rows k2 k0 + 2 n k1 k0 flops per variant, outside the timed pair: the roofline figure is the MixK product's own;
            // bench.py's whole_path counts them)
            if (!in_pair_order) {
                const int xg_cols = (int)std::min<long>(ld_xg, round_up((long)npairs * k0, 128) + 128);
                CRM_TRY(launch_gather_slabs(st, S, ld_ah, kdim, d_ord, npairs, k0, ctx->ws_XG.as<double>(), ld_xg, xg_cols));
            }
            CRM_HIP(hipStreamSynchronize(st));   // (kp lives on this stack frame)
        } else if (via_H && kin_route) {
            // AH = H'(g o E0) without an n-length contraction against the cols columns of H:
            // (a) the block in donor order; (b) per donor d' the Khatri-Rao contraction over its own cells against
            // [us | E1] (transposed store: S[(d' KK + q), (b, i)]); (c) the L rows: for every j a contraction over the
            // donors with hKd, AH[(k1 + j m + d), .] = sum_d' hKd[d', d] S[(d' KK + j), .]; (d) the E1 rows: sums over d'
            double* Gk = ctx->ws_Gk.as<double>();
            double* S = ctx->ws_S.as<double>();
            const int k1 = bg->kin_k1, k2 = bg->kin_k2;
            const long groups = bg->kin_groups, mk = bg->kin_cols;
            // one phenotype: the columns are taken in the rho*-sorted pair order (Gs) straight away, so that AH is the
            // operand of the Mix products as it stands; several phenotypes share a variant between pairs: block order, then
            // the pair gather below
            const bool in_pair_order = ng == 1;
            const double* Gsrc = in_pair_order ? Gs : Gt;
            const long ldg_k = in_pair_order ? ldp : ldb;
            const int ncol = in_pair_order ? npairs : nb;
            CRM_TRY(launch_gather_rows(st, Gsrc, ldg_k, bg->kin_map.as<int>(), bg->kin_rows, in_pair_order ? (int)ldg_k : blk_cols, Gk, ldg_k));
            std::vector<GemmProblem> kp((size_t)groups + k2);
            long maxlen = GEMM_BK;
            if (pairs_unfolded) {
                // P_d = G_d'(E (x) E)_d per donor; Z = [hKd | 1]' P over the donors (in ws_S: the per-donor blocks are not formed);
                // rows k1 + j m + c of AH from Z_c, rows [0, k1) from the sums over the donors Z_m
                double* Pd = ctx->ws_Pd.as<double>();
                double* Z = S;
                for (long d = 0; d < groups; d++) {
                    GemmProblem p{};
                    p.X = Gk + bg->kin_row0[d] * ldg_k; p.ldx = ldg_k;
                    p.Y = g0->kinEE.as<double>() + bg->kin_row0[d] * g0->ld_ee; p.ldy = g0->ld_ee;
                    p.C = Pd + (size_t)d * pd_slab; p.ldc = ldPd;
                    p.M = ncol; p.N = npair; p.cells = bg->kin_len[d];
                    maxlen = std::max(maxlen, bg->kin_len[d]);
                    kp[d] = p;
                }
                {
                    GemmProblem p{};
                    p.X = bg->kin_hKd.as<double>(); p.ldx = bg->kin_ldh;
                    p.Y = Pd; p.ldy = pd_slab;
                    p.C = Z; p.ldc = pd_slab;
                    p.M = (int)mk + 1; p.N = (int)((long)ncol * ldPd);
                    kp[groups] = p;
                }
                GemmProblem* d_kp = d_probs + 2 * CRM_MAX_RHO + 4;
                CRM_HIP(hipMemcpyAsync(d_kp, kp.data(), sizeof(GemmProblem) * (size_t)(groups + 1), hipMemcpyHostToDevice, st));
                CRM_TRY(launch_gemm_tn(ctx, d_kp, (int)groups, ncol, npair, maxlen, false, 0, 1, 0));
                CRM_TRY(launch_gemm_tn(ctx, d_kp + groups, 1, (int)mk + 1, (int)((long)ncol * ldPd), bg->kin_groups_pad, false, 0, 1, 0));
                CRM_TRY(launch_donor_pairs_expand(st, Z, pd_slab, ldPd, (int)mk, ncol, k0, k1, ctx->ws_AH.as<double>(), ld_ah, Pd, pd_slab, 1,
                                                  1, mk));
                CRM_TRY(launch_pair_rows_sym(st, Z + (size_t)mk * pd_slab, ldPd, ncol, k0, ctx->ws_AH.as<double>(), ld_ah));
                ctx->donor_pair_blocks++;
                if (!in_pair_order) {
                    const int xg_cols = (int)std::min<long>(ld_xg, round_up((long)npairs * k0, 128) + 128);
                    CRM_TRY(launch_gather_slabs(st, ctx->ws_AH.as<double>(), ld_ah, bg->ldh, d_ord, npairs, k0,
                                                ctx->ws_XG.as<double>(), ld_xg, xg_cols));
                }
                CRM_HIP(hipStreamSynchronize(st));   // (kp lives on this stack frame)
            } else {
            for (long d = 0; d < groups; d++) {
                GemmProblem p{};
                p.X = Gk + bg->kin_row0[d] * ldg_k; p.ldx = ldg_k;
                p.E = g0->kinEp.as<double>() + bg->kin_row0[d] * g0->ld_ep; p.lde = g0->ld_ep; p.k0 = k0;
                p.Y = bg->kin_Y.as<double>() + bg->kin_row0[d] * bg->kin_ldy; p.ldy = bg->kin_ldy;
                p.C = S + (size_t)d * KK * ld_ah; p.ldc = ld_ah;
                p.M = ncol * k0; p.N = KK; p.cells = bg->kin_len[d];
                maxlen = std::max(maxlen, bg->kin_len[d]);
                kp[d] = p;
            }
            for (int j = 0; j < k2; j++) {
                GemmProblem p{};
                p.X = bg->kin_hKd.as<double>(); p.ldx = bg->kin_ldh;
                p.Y = S + (size_t)j * ld_ah; p.ldy = (long)KK * ld_ah;
                p.C = ctx->ws_AH.as<double>() + (size_t)(k1 + (long)j * mk) * ld_ah; p.ldc = ld_ah;
                p.M = (int)mk; p.N = ncol * k0;
                kp[groups + j] = p;
            }
            GemmProblem* d_kp = d_probs + 2 * CRM_MAX_RHO + 4;
            CRM_HIP(hipMemcpyAsync(d_kp, kp.data(), sizeof(GemmProblem) * kp.size(), hipMemcpyHostToDevice, st));
            CRM_TRY(launch_kr_transposed(ctx, d_kp, (int)groups, ncol * k0, KK, maxlen, k0));
            CRM_TRY(launch_gemm_tn(ctx, d_kp + groups, k2, (int)mk, ncol * k0, bg->kin_groups_pad, false, 0, 1, 0));
            CRM_TRY(launch_kin_sum_e1(st, S, ld_ah, KK, k2, k1, (int)groups, (long)ncol * k0, ctx->ws_AH.as<double>(), ld_ah));
            // (2 kin_rows KK k0 + 2 groups_pad m k2 k0 flops per variant, outside the timed pair: the roofline figure is the Mix
            // product's own; bench.py's whole_path counts them)
            if (!in_pair_order) {
                const int xg_cols = (int)std::min<long>(ld_xg, round_up((long)npairs * k0, 128) + 128);
                CRM_TRY(launch_gather_slabs(st, ctx->ws_AH.as<double>(), ld_ah, bg->ldh, d_ord, npairs, k0,
                                            ctx->ws_XG.as<double>(), ld_xg, xg_cols));
            }
            CRM_HIP(hipStreamSynchronize(st));   // (kp lives on this stack frame)
            }   // (the Khatri-Rao form of the per-donor blocks)
        } else if (via_H) {
            GemmProblem p{};
            p.X = Gt; p.ldx = ldb; p.E = d_Ep; p.lde = g0->ld_ep; p.k0 = k0;
            p.Y = bg->H.as<double>(); p.ldy = bg->ldh;
            p.C = ctx->ws_AH.as<double>(); p.ldc = ld_ah;
            p.M = nb * k0; p.N = (int)bg->cols;
            CRM_HIP(hipMemcpyAsync(d_probs, &p, sizeof p, hipMemcpyHostToDevice, st));
            CRM_TRY(launch_kr_transposed(ctx, d_probs, 1, p.M, p.N, np, k0));
            kr_flops += 2.0 * (double)n * (double)bg->cols * (double)k0 * (double)nb;
            const int xg_cols = (int)std::min<long>(ld_xg, round_up((long)npairs * k0, 128) + 128);
            CRM_TRY(launch_gather_slabs(st, ctx->ws_AH.as<double>(), ld_ah, bg->ldh, d_ord, npairs, k0,
                                        ctx->ws_XG.as<double>(), ld_xg, xg_cols));
        }
        CRM_HIP(hipMemcpyAsync(d_probs, probs.data(), sizeof(GemmProblem) * nz, hipMemcpyHostToDevice, st));
        if (collapsed)
            CRM_TRY(launch_gemm_tn(ctx, d_probs, nz, max_m, (int)((long)k0 * ldq), mp, false, 0, 1, 0));
        else if (via_H && kin_route) {
            if (timing) CRM_HIP(hipEventRecord(ctx->timed[ctx->timed_used].first, st));
            struct Restore { crm_ctx* c; ~Restore() { c->tune.tag = 0; } } restore{ctx};
            ctx->tune.tag = 1;
            CRM_TRY(launch_gemm_tn(ctx, d_probs, nz, max_m, max_n, kdim, false, 0, 1, 0));
        } else if (via_H)
            CRM_TRY(launch_gemm_tn(ctx, d_probs, nz, max_m, max_n, kdim, false, 0, 1, 0));
        else {
            CRM_TRY(launch_gemm_tn(ctx, d_probs, nz, max_m, max_n, np, true, k0, kr_split, (long)a_slab));
            CRM_TRY(launch_reduce_splits(st, ctx->ws_A.as<double>(), (long)npairs * k0 * ldA, kr_split, (long)a_slab));
            if (!tails.empty()) {
                const int saved_bn = ctx->tune.bn;
                ctx->tune.bn = 160;
                struct Restore { crm_ctx* c; int bn; ~Restore() { c->tune.bn = bn; } } restore{ctx, saved_bn};
                CRM_HIP(hipMemcpyAsync(d_probs + nz, tails.data(), sizeof(GemmProblem) * tails.size(), hipMemcpyHostToDevice, st));
                CRM_TRY(launch_gemm_tn(ctx, d_probs + nz, (int)tails.size(), max_m, tail_maxn, np, true, k0, tail_split, (long)a_slab));
                ctx->tail_launches++;
                for (const GemmProblem& t : tails)
                    CRM_TRY(launch_reduce_splits_band(st, t.C, (long)t.M, t.ldc, 0, t.N, tail_split, (long)a_slab));
            }
        }
        if (timing) {
            CRM_HIP(hipEventRecord(ctx->timed[ctx->timed_used].second, st));
            ctx->timed_used++;
            ctx->kr_flops += kr_flops;
        }
        if (!spectrum_tails.empty()) {
            CRM_HIP(hipMemcpyAsync(d_probs + nz, spectrum_tails.data(), sizeof(GemmProblem) * spectrum_tails.size(), hipMemcpyHostToDevice, st));
            CRM_TRY(launch_skinny_tn(st, d_probs + nz, (int)spectrum_tails.size(), max_m, kdim));
            ctx->spectrum_tail_launches++;
            CRM_HIP(hipStreamSynchronize(st));   // (the records live on this stack frame)
        }
        // 7. elementwise products for the side contractions
        double* G2 = ctx->ws_G2.as<double>();
        double* GG = !collapsed ? ctx->ws_GG.as<double>() : nullptr;   // (test direction) o (fixed-effect role)
        CRM_TRY(launch_square_block(st, Gt, Gx, ldb, ldb, xrows, blk_cols, G2, GG, ldb));
        if (!GG) GG = G2;
        // 8. y-free side contractions: Z2 = (Gt o G)' E, Z3 = (Gt o Gt)' (E (x) E)
        const int s1 = collapsed ? 1 : ks1, s2 = collapsed ? 1 : ks2, s3 = collapsed ? 1 : ks3;
        {
            GemmProblem p{};
            p.ldx = ldb; p.M = nb;
            p.X = GG; p.Y = collapsed ? tab->Z2.as<double>() : d_Ep; p.ldy = g0->ld_ep; p.C = dZ2; p.ldc = ldZ2; p.N = k0;
            probs[1] = p;
            p.X = G2; p.Y = collapsed ? tab->Z3.as<double>() : d_EE; p.ldy = g0->ld_ee; p.C = dZ3; p.ldc = ldZ3; p.N = npair;
            probs[2] = p;
            CRM_HIP(hipMemcpyAsync(d_probs + 1, probs.data() + 1, sizeof(GemmProblem) * 2, hipMemcpyHostToDevice, st));
            if (cross) {
                hipLaunchKernelGGL(donor_cross_kernel, dim3(nb), dim3(128), 0, st, Gb, ldb, (int)panel->m,
                                   tab->Z2.as<double>(), ldZ2, k0, dZ2, ldZ2);
                CRM_HIP(hipGetLastError());
            } else {
                CRM_TRY(launch_gemm_tn(ctx, d_probs + 1, 1, nb, k0, xrows, false, 0, s2, z2_sz));
                CRM_TRY(launch_reduce_splits(st, dZ2, z2_sz, s2, z2_sz));
            }
            CRM_TRY(launch_gemm_tn(ctx, d_probs + 2, 1, nb, npair, xrows, false, 0, s3, z3_sz));
            CRM_TRY(launch_reduce_splits(st, dZ3, z3_sz, s3, z3_sz));
        }
        // 9.-11. per gene: Z1 = Gt' [y o E, W o E], Q and F, eigenvalues + Davies, results
        const int s1g = collapsed ? 1 : ks1b;
        if (z1_batched) {
            std::vector<GemmProblem> zp((size_t)ng);
            for (int gi = 0; gi < ng; gi++) {
                crm_gene* g = genes[gi];
                GemmProblem p{};
                p.X = Gt; p.ldx = ldb; p.Y = collapsed ? g->dt_Z1.as<double>() : g->YE.as<double>(); p.ldy = g->ld_ye;
                p.C = dZ1 + (size_t)gi * z1_sz * ks1b; p.ldc = ldZ1; p.M = nb; p.N = k0 * (1 + c);
                zp[(size_t)gi] = p;
            }
            GemmProblem* d_zp = d_probs + 2 * CRM_MAX_RHO + 4 + kin_probs;
            CRM_HIP(hipMemcpyAsync(d_zp, zp.data(), sizeof(GemmProblem) * zp.size(), hipMemcpyHostToDevice, st));
            CRM_TRY(launch_gemm_tn(ctx, d_zp, ng, nb, k0 * (1 + c), xrows, false, 0, s1g, z1_sz));
            if (s1g > 1)
                for (int gi = 0; gi < ng; gi++)
                    CRM_TRY(launch_reduce_splits(st, dZ1 + (size_t)gi * z1_sz * ks1b, z1_sz, s1g, z1_sz));
        }
        for (int gi = 0; gi < ng; gi++) {
            crm_gene* g = genes[gi];
            const ScanOut& o = outs[gi];
            double* const dZ1g = z1_batched ? dZ1 + (size_t)gi * z1_sz * ks1b : dZ1;
            if (!z1_batched) {
                GemmProblem p{};
                p.X = Gt; p.ldx = ldb; p.Y = collapsed ? g->dt_Z1.as<double>() : g->YE.as<double>(); p.ldy = g->ld_ye;
                p.C = dZ1; p.ldc = ldZ1; p.M = nb; p.N = k0 * (1 + c);
                CRM_HIP(hipMemcpyAsync(d_probs, &p, sizeof p, hipMemcpyHostToDevice, st));
                CRM_TRY(launch_gemm_tn(ctx, d_probs, 1, nb, k0 * (1 + c), xrows, false, 0, s1, z1_sz));
                CRM_TRY(launch_reduce_splits(st, dZ1, z1_sz, s1, z1_sz));
            }
            AssembleArgs aa{};
            for (int i = 0; i < nrho; i++) {
                AssembleRho& R = aa.rho[i];
                R.ty = g->rot.as<double>() + (long)i * slab;
                R.tW = R.ty + ldq; R.ldW = ldq;
                R.S0 = bg->S0[i].as<double>();
                R.T = ctx->ws_T.as<double>() + ((size_t)i * BLK + sb0) * ldT; R.ldT = ldT;
                R.r = bg->r[i];
            }
            aa.fit = d_fit + (size_t)gi * BLK; aa.sorted_pos = d_pos + (size_t)gi * BLK;
            aa.A = ctx->ws_A.as<double>(); aa.ldA = ldA; aa.k0 = k0; aa.c = c; aa.n = n;
            aa.A_none = ctx->ws_Anone.as<double>();
            aa.Z1 = dZ1g; aa.ldZ1 = ldZ1; aa.Z2 = dZ2; aa.ldZ2 = ldZ2; aa.Z3 = dZ3; aa.ldZ3 = ldZ3;
            aa.WW = g->WW.as<double>(); aa.Wy = g->Wy.as<double>(); aa.yy = g->yy;
            aa.gg = d_gg; aa.gy = d_gy + (size_t)gi * BLK; aa.gW = d_gW; aa.ld_gW = ld_gW;
            aa.coef = collapsed ? nullptr : d_coef; aa.ld_coef = ldb;
            aa.Q = d_Q; aa.F = ctx->ws_F.as<double>();
            double* slow_ws = nullptr;
            if (slow_forms) {
                CRM_TRY(ctx->ws_xwide.ensure(sizeof(double) * std::max(std::max(assemble_rows_scratch_doubles(BLK, k0, c), eig_scratch_doubles(BLK, k0)),
                                                                       c > CRM_MAX_COV_WIDE ? nullfit_xwide_scratch_doubles(BLK, nrho, c) : (size_t)0)));
                slow_ws = ctx->ws_xwide.as<double>();   // (the null fits of the block are done: their scratch is free)
            }
            CRM_TRY(launch_assemble(st, aa, nb, ctx->ws_Gext.as<double>(), slow_ws));
            CRM_TRY(launch_eig_davies(st, ctx->ws_F.as<double>(), d_Q, nb, k0, d_lam, d_pv, d_if, d_liu, true, slow_ws));
            if (o.pv) CRM_HIP(hipMemcpyAsync(o.pv + done, d_pv, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
            if (o.Q) CRM_HIP(hipMemcpyAsync(o.Q + done, d_Q, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
            if (o.lambda) CRM_HIP(hipMemcpyAsync(o.lambda + done * k0, d_lam, sizeof(double) * nb * k0, hipMemcpyDeviceToHost, st));
            if (o.F) CRM_HIP(hipMemcpyAsync(o.F + done * k0 * k0, ctx->ws_F.ptr, sizeof(double) * nb * k0 * k0, hipMemcpyDeviceToHost, st));
            if (o.ifault) CRM_HIP(hipMemcpyAsync(o.ifault + done, d_if, sizeof(int) * nb, hipMemcpyDeviceToHost, st));
            if (o.liu) CRM_HIP(hipMemcpyAsync(o.liu + done, d_liu, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
            // flat-optimum probes (info calls only): the score test again with delta one stopping tolerance of the
            // reference's search to either side; how far Q and p move says whether the search's last comparison matters
            std::vector<char> flat;
            std::vector<double> probe_rec;
            const double flat_kappa = FLAT_KAPPA * 1e-3 * form("flat_kappa_milli", 1000);
            const double rho_kappa = RHO_KAPPA * 1e-3 * form("flat_kappa_milli", 1000);
            if (o.flags) {
                probe_rec.assign((size_t)nb * FLAT_REC, 0.0);
                std::vector<double> q0(nb), p0(nb), q1(nb), p1(nb), lam0((size_t)nb * k0);
                CRM_HIP(hipMemcpyAsync(lam0.data(), d_lam, sizeof(double) * (size_t)nb * k0, hipMemcpyDeviceToHost, st));
                CRM_HIP(hipMemcpyAsync(q0.data(), d_Q, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
                CRM_HIP(hipMemcpyAsync(p0.data(), d_pv, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
                ScopedBuf probe;
                CRM_TRY(probe.ensure(sizeof(NullFitOut) * (size_t)nb + 64));
                flat.assign(nb, 0);
                for (int side = 0; side < 2; side++) {
                    hipLaunchKernelGGL(flat_probe_fit_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, aa.fit, nb,
                                       side == 0 ? 1.0 : -1.0, probe.as<NullFitOut>());
                    CRM_HIP(hipGetLastError());
                    AssembleArgs ap = aa;
                    ap.fit = probe.as<NullFitOut>();
                    CRM_TRY(launch_assemble(st, ap, nb, ctx->ws_Gext.as<double>(), slow_ws));
                    CRM_TRY(launch_eig_davies(st, ctx->ws_F.as<double>(), d_Q, nb, k0, d_lam, d_pv, d_if, d_liu, true, slow_ws));
                    CRM_HIP(hipMemcpyAsync(q1.data(), d_Q, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
                    CRM_HIP(hipMemcpyAsync(p1.data(), d_pv, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
                    CRM_HIP(hipStreamSynchronize(st));
                    for (int b = 0; b < nb; b++) {
                        // (Q against max(Q, its expectation under the null = tr F): a score vector that nearly vanishes,
                        // p ~ 1, leaves Q itself ill-conditioned)
                        double trace = 0.0;
                        for (int j = 0; j < k0; j++) trace += lam0[(size_t)b * k0 + j];
                        // (equal values -- a p-value that underflows to zero on both sides included -- have not moved)
                        const double mq = q1[b] == q0[b] ? 0.0 : std::fabs(q1[b] - q0[b]) / std::max(std::fabs(q0[b]), trace);
                        const double mp = p1[b] == p0[b] ? 0.0 : std::fabs(p1[b] - p0[b]) / std::fabs(p0[b]);
                        const NullFitOut& fo = h_fit[(size_t)gi * BLK + b];
                        double* rec = &probe_rec[(size_t)b * FLAT_REC];
                        // (NaN -- a probe that could not be evaluated -- must survive the maximum)
                        rec[1] = (mq == mq && rec[1] == rec[1]) ? std::max(rec[1], mq) : NAN;
                        rec[2] = (mp == mp && rec[2] == rec[2]) ? std::max(rec[2], mp) : NAN;
                        rec[0] = flat_obj.empty() ? -1.0 : flat_obj[(size_t)gi * BLK + sb0 + b];
                        rec[3] = fo.margin; rec[4] = fo.noise; rec[5] = fo.rho_decision; rec[6] = fo.gap; rec[7] = fo.lml;
                        rec[8] = fo.curv; rec[9] = fo.delta;
                    }
                }
                // the bounds: (movement of Q / p over one tolerance) x (the largest distance, in tolerances, at which two
                // faithful searches stop: STOP_SHIFT_C / relative gain of the objective over one tolerance, at most one --
                // and one outright where a decision of the search itself was within the objective's noise bound)
                for (int b = 0; b < nb; b++) {
                    const NullFitOut& fo = h_fit[(size_t)gi * BLK + b];
                    const double* rec = &probe_rec[(size_t)b * FLAT_REC];
                    const double gain = fo.curv / std::fabs(fo.lml);
                    double shift = (gain > 0.0 && gain == gain) ? std::min(1.0, STOP_SHIFT_C / gain) : 1.0;
                    if (!(rec[0] > flat_kappa)) shift = 1.0;
                    const double bq = rec[1] * shift, bp = rec[2] * shift;
                    if (o.bound_Q) o.bound_Q[done + b] = bq;
                    if (o.bound_p) o.bound_p[done + b] = bp;
                    if (!(bp <= 1e-5)) flat[b] |= 1;
                    if (!(bq <= 1e-6)) flat[b] |= 2;
                }
            }
            if (o.flags && ng == 1) {   // (diagnostics: what the probes measured, crm_test_null_fit_probe_read)
                if (done == 0) ctx->probe_out.clear();
                ctx->probe_out.insert(ctx->probe_out.end(), probe_rec.begin(), probe_rec.end());
            }
            int rmax = 0;
            for (int i = 0; i < nrho; i++) rmax = std::max(rmax, bg->r[i]);
            const bool saturated = (long)rmax + c + 1 >= n;
            for (int b = 0; b < nb; b++) {
                const NullFitOut& f = h_fit[(size_t)gi * BLK + b];
                const double rho = bg->rho[f.rho_index];
                if (o.flags) {
                    int fl = saturated ? CRM_MODEL_SATURATED : 0;
                    if (!(f.delta > 1e-8)) fl |= CRM_MODEL_DELTA_AT_ZERO;
                    if (!f.use_g) fl |= CRM_MODEL_G_IN_SPAN_W;
                    if (!flat.empty() && (flat[b] & 1)) fl |= CRM_MODEL_FLAT_OPTIMUM;
                    if (!flat.empty() && (flat[b] & 2)) fl |= CRM_MODEL_STATISTIC_AT_TOLERANCE;
                    if (f.rho_decision == f.rho_decision && !(f.rho_decision > rho_kappa)) fl |= CRM_MODEL_RHO_TIE;
                    o.flags[done + b] = fl;
                }
                if (o.rho1) o.rho1[done + b] = rho;
                if (o.e2) o.e2[done + b] = f.v0 * rho;
                if (o.g2) o.g2[done + b] = f.v0 * (1 - rho);
                if (o.eps2) o.eps2[done + b] = f.v1;
                if (o.lml) o.lml[done + b] = f.lml;
                if (o.delta) o.delta[done + b] = f.delta;
                if (o.scale) o.scale[done + b] = f.scale;
            }
            // the per-gene device buffers (Z1, Q, F, pv) are reused by the next gene
            CRM_HIP(hipStreamSynchronize(st));
        }
        sb0 += nsb;
        }   // pair stage over the sub-ranges of the block
        ctx->report(done + nb, count);   // (the reference's tqdm, :340)
    }
    return CRM_OK;
}

// The scan of [first, first + count): one pass, plus -- after a collapsed pass -- a dense pass over every run of variants
// the collapsed one marked as nearly collinear with the covariates (their results are overwritten).
static int scan_core(const std::vector<crm_gene*>& genes, crm_panel* panel, long first, long count,
                     const int* idx_E, const int* idx_G, const std::vector<ScanOut>& outs) {
    std::vector<long> near;
    CRM_TRY(scan_pass(genes, panel, first, count, idx_E, idx_G, outs, true, &near));
    crm_ctx* ctx = genes[0]->ctx;
    if (near.empty() || ctx->probe_on) return CRM_OK;
    const int k0 = genes[0]->k0;
    struct Quiet {   // (the repeated variants were reported as done by the first pass)
        crm_ctx* c;
        explicit Quiet(crm_ctx* c_) : c(c_) { c->progress_muted = true; }
        ~Quiet() { c->progress_muted = false; }
    } quiet(ctx);
    ctx->dense_repeats += (long)near.size();
    // runs of marked variants, neighbours closer than 32 variants merged (a dense pass has a fixed cost of a few
    // milliseconds whatever its length); a panel that is marked on more than a quarter of its variants is simply scanned
    // again as a whole
    if ((long)near.size() * 4 > count) {
        near.resize((size_t)count);
        for (long v = 0; v < count; v++) near[(size_t)v] = v;
    }
    for (size_t i = 0; i < near.size();) {
        size_t j = i + 1;
        while (j < near.size() && near[j] <= near[j - 1] + 32) j++;
        const long off = near[i], len = near[j - 1] - near[i] + 1;
        std::vector<ScanOut> shifted(outs);
        for (ScanOut& o : shifted) {
            auto at = [&](double* p, long stride) { return p ? p + off * stride : nullptr; };
            o.pv = at(o.pv, 1); o.rho1 = at(o.rho1, 1); o.e2 = at(o.e2, 1); o.g2 = at(o.g2, 1); o.eps2 = at(o.eps2, 1);
            o.Q = at(o.Q, 1); o.lml = at(o.lml, 1); o.delta = at(o.delta, 1); o.scale = at(o.scale, 1);
            o.lambda = at(o.lambda, k0); o.F = at(o.F, (long)k0 * k0); o.liu = at(o.liu, 1);
            if (o.ifault) o.ifault += off;
            if (o.flags) o.flags += off;
        }
        CRM_TRY(scan_pass(genes, panel, first + off, len, idx_E, idx_G, shifted, false, nullptr));
        i = j;
    }
    return CRM_OK;
}

}  // namespace crm

extern "C" {

int crm_scan_interaction(crm_gene* gene, crm_panel* panel, long first, long count, const int* idx_E,
                         const int* idx_G, double* out_pvalue, double* out_rho1, double* out_e2,
                         double* out_g2, double* out_eps2, double* out_Q, double* out_lml,
                         double* out_delta, double* out_scale, double* out_lambda, double* out_F) {
    return crm::guarded_on("crm_scan_interaction", gene ? gene->ctx : nullptr, [&]() -> int {
    if (!gene || !panel) return CRM_ERR_ARG;
    std::vector<crm_gene*> genes{gene};
    std::vector<ScanOut> outs{{out_pvalue, out_rho1, out_e2, out_g2, out_eps2, out_Q, out_lml, out_delta,
                               out_scale, out_lambda, out_F}};
    return scan_core(genes, panel, first, count, idx_E, idx_G, outs);
    });
}

int crm_scan_interaction_info(crm_gene* gene, crm_panel* panel, long first, long count, const int* idx_E,
                              const int* idx_G, double* out_pvalue, int* out_ifault, double* out_liu_pvalue,
                              int* out_model_flags) {
    return crm::guarded_on("crm_scan_interaction_info", gene ? gene->ctx : nullptr, [&]() -> int {
    if (!gene || !panel) return CRM_ERR_ARG;
    std::vector<crm_gene*> genes{gene};
    ScanOut o{out_pvalue, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    o.ifault = out_ifault;
    o.liu = out_liu_pvalue;
    o.flags = out_model_flags;
    std::vector<ScanOut> outs{o};
    return scan_core(genes, panel, first, count, idx_E, idx_G, outs);
    });
}

// B permutations of one scan (include/crm_hip.h): the first permutation's pass records, per block, the fits and the rows
// T(rho*); the others replay them.  The panel is walked in chunks that keep the record within REPLAY_CAP_BYTES.
int crm_scan_interaction_permuted(crm_gene* gene, crm_panel* panel, long first, long count, int nperm, const int* idx_E,
                                  const int* idx_G, double* out_pvalue, double* out_rho1, double* out_e2, double* out_g2,
                                  double* out_eps2, double* out_Q) {
    return crm::guarded_on("crm_scan_interaction_permuted", gene ? gene->ctx : nullptr, [&]() -> int {
    if (!gene || !panel || nperm < 1 || !out_pvalue) return CRM_ERR_ARG;
    if (first < 0 || count < 0 || first + count > panel->p) {
        set_error("scan: variants [%ld, %ld) outside the panel (p = %ld)", first, first + count, panel->p);
        return CRM_ERR_ARG;
    }
    crm_ctx* ctx = gene->ctx;
    if (ctx->replay_mode != 0 || ctx->in_scan) {
        set_error("scan: another scan is running on this context");
        return CRM_ERR_UNSUPPORTED;
    }
    const long n = gene->bg->n;
    constexpr double REPLAY_CAP_BYTES = 8.0 * (1ull << 30);
    // (whole blocks of the scan as one call over all `count` variants would cut them: the same launches, the same bits)
    const long blk = scan_block_variants(ctx, gene, count);
    const long chunk_cap = std::max<long>(blk, (long)(REPLAY_CAP_BYTES / (sizeof(double) * (double)gene->bg->ldq)) / blk * blk);
    struct Clear { crm_ctx* c; ~Clear() { c->replay_clear(); } } clear{ctx};
    std::vector<crm_gene*> genes{gene};
    for (long at = 0; at < count; at += chunk_cap) {
        const long len = std::min(chunk_cap, count - at);
        ctx->replay_clear();
        for (int q = 0; q < nperm; q++) {
            ctx->replay_mode = q == 0 ? 1 : 2;
            ctx->replay_cursor = 0;
            // (rho*, the variance components and the fit do not depend on the permutation: written by the first pass)
            ScanOut o{out_pvalue + (size_t)q * count + at, nullptr, nullptr, nullptr, nullptr,
                      out_Q ? out_Q + (size_t)q * count + at : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            if (q == 0) {
                o.rho1 = out_rho1 ? out_rho1 + at : nullptr; o.e2 = out_e2 ? out_e2 + at : nullptr;
                o.g2 = out_g2 ? out_g2 + at : nullptr; o.eps2 = out_eps2 ? out_eps2 + at : nullptr;
            }
            std::vector<ScanOut> outs{o};
            const int rc = scan_core(genes, panel, first + at, len, idx_E ? idx_E + (size_t)q * n : nullptr,
                                     idx_G ? idx_G + (size_t)q * n : nullptr, outs);
            if (rc != CRM_OK) return rc;
        }
    }
    return CRM_OK;
    });
}

int crm_scan_interaction_bounds(crm_gene* gene, crm_panel* panel, long first, long count, const int* idx_E, const int* idx_G,
                                double* out_pvalue, int* out_ifault, double* out_liu_pvalue, int* out_model_flags,
                                double* out_bound_Q, double* out_bound_p) {
    return crm::guarded_on("crm_scan_interaction_bounds", gene ? gene->ctx : nullptr, [&]() -> int {
    if (!gene || !panel || !out_model_flags) return CRM_ERR_ARG;
    std::vector<crm_gene*> genes{gene};
    ScanOut o{out_pvalue, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    o.ifault = out_ifault;
    o.liu = out_liu_pvalue;
    o.flags = out_model_flags;
    o.bound_Q = out_bound_Q;
    o.bound_p = out_bound_p;
    std::vector<ScanOut> outs{o};
    return scan_core(genes, panel, first, count, idx_E, idx_G, outs);
    });
}

long crm_test_tail_launches(const crm_ctx* ctx) { return ctx ? ctx->tail_launches : -1; }

long crm_test_spectrum_tail_launches(const crm_ctx* ctx) { return ctx ? ctx->spectrum_tail_launches : -1; }

long crm_test_dense_repeats(const crm_ctx* ctx) { return ctx ? ctx->dense_repeats : -1; }

long crm_test_donor_pair_blocks(const crm_ctx* ctx) { return ctx ? ctx->donor_pair_blocks : -1; }

long crm_test_tests_without_pair(const crm_ctx* ctx) { return ctx ? ctx->tests_without_pair : -1; }

int crm_test_set_shared_h(crm_ctx* ctx, int mode) {
    return crm::guarded_on("crm_test_set_shared_h", ctx, [&]() -> int {
    if (!ctx) return CRM_ERR_ARG;
    ctx->tune.shared_h = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
    return CRM_OK;
    });
}

int crm_scan_interaction_multi(crm_gene* const* genes, int ngenes, crm_panel* panel, long first, long count,
                               const int* idx_E, const int* idx_G, double* out_pvalue, double* out_rho1,
                               double* out_e2, double* out_g2, double* out_eps2, double* out_Q) {
    return crm::guarded_on("crm_scan_interaction_multi", (genes && ngenes > 0 && genes[0]) ? genes[0]->ctx : nullptr, [&]() -> int {
    if (!genes || ngenes < 1 || !panel) return CRM_ERR_ARG;
    std::vector<crm_gene*> gs(genes, genes + ngenes);
    for (crm_gene* g : gs)
        if (!g) return CRM_ERR_ARG;
    std::vector<ScanOut> outs(ngenes);
    for (int i = 0; i < ngenes; i++) {
        auto at = [&](double* base) { return base ? base + (size_t)i * count : nullptr; };
        outs[i] = ScanOut{at(out_pvalue), at(out_rho1), at(out_e2), at(out_g2), at(out_eps2), at(out_Q),
                          nullptr, nullptr, nullptr, nullptr, nullptr};
    }
    return scan_core(gs, panel, first, count, idx_E, idx_G, outs);
    });
}

}  // extern "C"
