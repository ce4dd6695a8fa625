// Argument records shared by the per-variant kernels and the host orchestration.
#pragma once
#include "crm_internal.h"

namespace crm {

// ---- null fits (nullfit.hip) -------------------------------------------------------------
struct NullFitRho {
    const double* T;   // [variants x ldT]   rows = Q0(rho)' g
    const double* ty;  // [r]                Q0(rho)' y
    const double* tW;  // [c x ldW]          rows = Q0(rho)' W_i
    const double* S0;  // [r]
    long ldT, ldW;
    int r;
    int pad_;
};

struct NullFitTrial {  // one (variant, rho) fit
    double lml, delta, scale;
    int use_g, nfev;
    double margin;  // brent_search.h: the smallest margin of the decisions that steered the search (units of the objective)
    double noise;   // first-order bound on the rounding noise of the objective at the optimum, in roundings (x 2^-53)
    double curv;    // (f(x + tol) + f(x - tol)) / 2 - f(x) at the stopping point x, tol = 1e-6 |x| + 1e-6: what the objective gains
                    // over one stopping tolerance (units of the objective)
};

struct NullFitOut {
    int rho_index;
    int use_g;  // 0 when g lies in span(W): X = [W, g] is rank deficient
    double lml, delta, scale, v0, v1;
    // How far the fit is from another outcome, in units of its own noise bound (select_rho_kernel):
    // decision = margin / (2^-53 noise) of the search at rho* (include/crm_hip.h: CRM_MODEL_FLAT_OPTIMUM);
    // rho_decision = min over the other grid points i of (lml(rho*) - lml(i)) / (2^-53 (noise(rho*) + noise(i))): how far
    // the choice of rho* itself is from another one (CRM_MODEL_RHO_TIE).  NaN where a kernel does not measure it.
    double decision, rho_decision;
    double margin, noise, gap, curv;   // (the raw figures of the winner: decision = margin / noise; curv: NullFitTrial)
};

struct NullFitArgs {
    NullFitRho rho[CRM_MAX_RHO];
    int nrho, c, restricted;
    int polish;        // secant refinement of the optimum on the analytic derivative
    int exact;         // spectrum pass with IEEE division and one log per entry (the reference's own operations)
    int track;         // 1: the searches leave their trace behind (brent_search.h: margins of their decisions, movement of the
                       // stopping point; the objective's noise bound) -- calls that ask for model flags
    int probe;         // 1: evaluate the objective at probe_x instead of searching (register kernels; test hook)
    double probe_x;
    long n;            // cells (unpadded)
    const double* WW;  // [c x c]
    const double* Wy;  // [c]
    double yy;
    const double* gg;  // [variants]
    const double* gy;  // [variants]
    const double* gW;  // [variants x ld_gW]
    long ld_gW;
    const int* g_drop; // [variants] 1: the variant's direction falls under the reference's rank rule (launch_ortho_rank);
                       // null: decided here from the last pivot of the Cholesky factor of X'X (relative 1e-12)
    double* xwide;        // scratch of the 63..128-column kernel: nullfit_xwide_scratch_doubles(variants, nrho, c) doubles
    NullFitTrial* trial;  // [variants x nrho]
    NullFitOut* out;      // [variants]
};

// c <= CRM_MAX_COV: register kernel (nullfit.hip); larger c (or force_wide): LDS kernel
// (nullfit_wide.hip).  Both end with the rho* selection into a.out.
// queue: CRM_MAX_RHO unsigned counters in device memory (optional; enables the LDS-shared form for c == 1)
int launch_nullfit(hipStream_t st, const NullFitArgs& a, int variants, bool force_wide = false, unsigned* queue = nullptr);
int launch_nullfit_wide(hipStream_t st, const NullFitArgs& a, int variants);
// 63 .. 128 covariate columns (nullfit_xwide.hip): needs NullFitArgs::xwide
int launch_nullfit_xwide(hipStream_t st, const NullFitArgs& a, int variants);
size_t nullfit_xwide_scratch_doubles(int variants, int nrho, int c);

// ---- association paths (assoc.hip) ---------------------------------------------------------------
struct AssocArgs {
    const double* T;   // [variants x ldT] rows Q0(rho*)' g
    const double* ty;  // [r]
    const double* tW;  // [c x ldW]
    const double* S0;  // [r]
    long ldT, ldW;
    int r, c;
    long n;
    double delta0;     // the null model's delta
    const double* WW; const double* Wy; double yy;
    const double* gg; const double* gy; const double* gW; long ld_gW;
};
size_t fastscan_prep_doubles();
int launch_fastscan_prep(hipStream_t st, const AssocArgs& a, double* prep, double* wts);
int launch_fastscan(hipStream_t st, const AssocArgs& a, const double* prep, const double* wts,
                    int variants, double* alt_lml);
int launch_lrt(hipStream_t st, const double* alt_lml, double null_lml, int count, double* pv);
int launch_gather_trial_lml(hipStream_t st, const NullFitTrial* trial, int count, double* out);

// ---- small per-block kernels (blockops.hip) ----------------------------------------------------
// gg, gy, gW for a block of variants: column reductions of G (cells x ldg), deterministic.
int launch_variant_stats(hipStream_t st, const double* G, long ldg, long cells, int variants,
                         const double* y, const double* W, long ldw, int c, double* partial,
                         double* gg, double* gy, double* gW, long ld_gW);
size_t variant_stats_workspace(int variants, int c);
// The block in the fixed effects' own basis (blockops.hip): Gx = G - W coef with coef = (W'W)^-1 W'g (gW = W'g of the
// block as launch_variant_stats leaves it; proj = crm_gene::Wproj), thr = the bound on |gx|^2 below which
// numpy_sugar.economic_svd drops the variant's direction from [W, g].  coef: [c x ld_coef], ld_coef >= cols.
int launch_ortho_block(hipStream_t st, const double* G, long ldg, long cells_pad, int variants, int cols,
                       const double* W, long ldw, int c, const double* proj, const double* gW, long ld_gW,
                       double* coef, long ld_coef, double* thr, double* Gx, long ldx);
int launch_ortho_rank(hipStream_t st, const double* gg, const double* thr, int variants, int* drop);
// near[b] = 1 when gg - gW'(W'W)^-1 gW <= tau gg (donor-level sums: the collapsed path's guard)
int launch_collinear_flag(hipStream_t st, const double* gg, const double* gW, long ld_gW, const double* proj, int c,
                          int variants, double tau, int* near);
// out[i, b] = src[row(i), col(b)]: optional row permutation (idx_G) and column order (sorted by rho*).
int launch_gather_block(hipStream_t st, const double* src, long ld_src, long cells_pad, long cells,
                        const int* row_index, const int* col_index, int variants, double* dst,
                        long ld_dst, int dst_cols);
// dst[h, p*k0 + j] = src[h, ord[p]*k0 + j] for p < npairs, zero in the remaining dst_cols
int launch_gather_slabs(hipStream_t st, const double* src, long ld_src, long rows, const int* ord, int npairs,
                        int k0, double* dst, long ld_dst, int dst_cols);
// rows in the donor order of a background's kinship structure: dst[k, :cols] = src[map[k], :cols], zeros where map[k] < 0
int launch_gather_rows(hipStream_t st, const double* src, long ld_src, const int* map, long rows, int cols, double* dst,
                       long ld_dst);
// the per-donor right-hand operand [us | E1] in donor order (crm_background::kin_Y)
int launch_kin_operand(hipStream_t st, const double* U, int k2, const double* H, long ldh, int k1, const int* map, long rows,
                       double* Y, long ldy);
// E1 rows of H'(g o E0): sums of the per-donor blocks S over the donors
int launch_kin_verify(hipStream_t st, const double* H, long ldh, int k1, const double* U, int k2, const int* group,
                      const double* hKd, long ldk, long m, long n, unsigned long long* out);
int launch_kin_sum_e1(hipStream_t st, const double* S, long ld_s, int KT, int k2, int k1, int groups, long cols, double* AH,
                      long ld_ah);
// pair products P[c, a k0 + i] = H[c, a] Ep[c, i] (a < k1), and the rows S[a, b k0 + i] = C[b, a k0 + i] of a product G'P
int launch_pair_features(hipStream_t st, const double* H, long ldh, int k1, const double* Ep, long ld_ep, int k0,
                         long cells_pad, double* P, long ldp);
int launch_pair_rows(hipStream_t st, const double* C, long ldc, int variants, int k1, int k0, double* S, long lds);
// are H[:, :k] and Ep[:, :k] the same numbers (flag[0] |= 1 if not)?  and S[a, b k0 + i] = C[b, pair(a, i)] for E1 = E
int launch_same_columns(hipStream_t st, const double* H, long ldh, const double* Ep, long ld_ep, long cells, int k, int* flag);
int launch_pair_rows_sym(hipStream_t st, const double* C, long ldc, int variants, int k0, double* S, long lds);
bool donor_pairs_serves(int k0);
// (row_d, row_j: row of (donor d, context j) = k1 + d row_d + j row_j; default (k0, 1): the folded form's layout)
int launch_donor_pairs_expand(hipStream_t st, const double* P, long p_slab, long ldp, int donors, int variants, int k0, int k1,
                              double* S, long lds, double* Psum, long psum_slab, int splits, long row_d = 0, long row_j = 0);
// dst[k, :cols] = src[k, :cols] * scale[k * ld_scale]
int launch_scale_rows(hipStream_t st, const double* src, long ld_src, const double* scale, long ld_scale, long rows, int cols,
                      double* dst, long ld_dst);
// dense block from a grouped panel: dst[i, b] = Gd[group[row(i)], b]
int launch_expand_block(hipStream_t st, const double* Gd, long ld_gd, const int* group, long cells_pad,
                        long cells, const int* row_index, int variants, double* dst, long ld_dst,
                        int dst_cols);
// Z[i, d] = (group[i] == d)
int launch_indicator(hipStream_t st, const int* group, long cells, long cells_pad, int m, double* Z, long ldz);
// collapsed statistics: gg = sum_d gamma^2 n_d, gy = sum_d gamma ysum_d, gW likewise (sums: [m_pad x 16])
int launch_donor_stats(hipStream_t st, const double* Gam, long ld_gam, int m, int variants,
                       const double* sums, int c, double* gg, double* gy, double* gW, long ld_gW);
// G2 = Gt o Gt, GG = Gt o G
int launch_square_block(hipStream_t st, const double* Gt, const double* G, long ldg, long ldg_t,
                        long cells_pad, int cols, double* G2, double* GG, long ld_out);
// rows of E permuted; YE = [y o E, W_1 o E, ...]; EE = pairwise products E_j E_j' (j <= j')
int launch_context_features(hipStream_t st, const double* E, long lde, const int* row_index,
                            long cells, long cells_pad, int k0, const double* y, const double* W,
                            long ldw, int c, double* Ep, long ld_ep, double* YE, long ld_ye,
                            double* EE, long ld_ee);

// ---- score statistic assembly (assemble.hip) ------------------------------------------------------
struct AssembleRho {
    const double* ty;
    const double* tW;
    const double* S0;
    const double* T;  // [variants x ldT] in block order
    long ldW, ldT;
    int r;
    int pad_;
};

struct AssembleArgs {
    AssembleRho rho[CRM_MAX_RHO];
    const NullFitOut* fit;    // [variants] block order
    const int* sorted_pos;    // [variants] position of variant b inside the rho*-sorted A~ buffer; < 0: none was formed
    const double* A;          // [variants*k0 x ldA]  rows (pos*k0 + j) = Q0(rho*)' (gtest o E_j)
    long ldA;
    const double* A_none;     // ldA zeros: the rows of a variant without a position (scan.hip: no kinship term in its fit)
    int k0, c;
    long n;
    // n-length reductions, block order
    const double* Z1; long ldZ1;   // [variants x k0*(1+c)]  E' (gt o y), E' (gt o W_i)
    const double* Z2; long ldZ2;   // [variants x k0]        E' (gt o g)
    const double* Z3; long ldZ3;   // [variants x k0(k0+1)/2] E' diag(gt^2) E (upper, row-major pairs)
    const double* WW; const double* Wy; double yy;
    const double* gg; const double* gy; const double* gW; long ld_gW;
    const double* coef; long ld_coef;   // [c x ld_coef] projection coefficients of the block's variants onto W
                                        // (launch_ortho_block; null where the block was not orthogonalised)
    double* Q;   // [variants]
    double* F;   // [variants x k0 x k0]
};

// Gext: workspace [variants x (k0+c+2)^2]; fin_rows: assemble_rows_scratch_doubles(...) doubles (0: not needed -- the
// per-variant rows D'K^-1X and their solves fit LDS)
int launch_assemble(hipStream_t st, const AssembleArgs& a, int variants, double* Gext, double* fin_rows = nullptr);
size_t assemble_rows_scratch_doubles(int variants, int k0, int c);

// ---- eigenvalues + Davies / Liu (davies.hip) -----------------------------------------------------
// lambda: ascending eigenvalues of the lower triangle of F (count x k x k); pvalue per SKAT rule.
// scratch: eig_scratch_doubles(count, k) doubles (0: F fits LDS, k <= 128)
int launch_eig_davies(hipStream_t st, const double* F, const double* Q, int count, int k,
                      double* lambda, double* pvalue, int* ifault, double* liu, bool do_eig, double* scratch = nullptr);
size_t eig_scratch_doubles(int count, int k);

}  // namespace crm
