/* Unit-test hooks of libcrm_hip.so: single kernels exercised through the same C-ABI, and the knobs
 * that force a kernel variant.  Not part of the drop-in boundary (include/crm_hip.h); every knob lives in
 * the context it is given, so distinct contexts stay independent. */
#ifndef CRM_HIP_TEST_H
#define CRM_HIP_TEST_H

#include "crm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel forms kept beside the default one (process-wide; reset != 0 returns the form to its default):
 *   "gram_staged" 1       score-statistic Gram through the register-staged kernel instead of the direct-to-LDS one
 *   "kr_no_tail" 1        the product over the spectrum (Khatri-Rao contraction, or the mixing-matrix product of the
 *                         kinship-structure route) of a block in one launch of 128-column tiles whatever the spectrum
 *   "nullfit_per_wave" 1  null fits with one independent wavefront per (variant, grid point), not the LDS-shared queue
 *   "kin_fold" 0 / 2      never fold the donor-level kinship factor into the mixing matrices / fold with few columns too
 *   "eigh_one_stage" 1    the constructor tridiagonalises every grid point on its own (eigh_trd.hip) instead of the
 *                         two-stage family solver (eigh2_band.hip, eigh2_chase.hip)
 *   "nullfit_exact" 1     null-fit likelihood with IEEE division and one log per spectrum entry
 *   "donor_pairs" 0 / 2   per-donor sums of the kinship-structure route never / always from the symmetric pair features
 *                         (default 1: where the kinship term's contexts are the scan's own and a cost model says so)
 *   "pairs_without_kinship_term" 0   form the rotated test direction for every test, also where the fit has no kinship
 *                         term to speak of (default: not formed where (v0 / v1) max S0 <= 1e-10)
 *   "nullfit_one_per_wave" 1   the LDS-shared null-fit kernel with one fit per wavefront (default: four, one per row of sixteen
 *                         lanes, nullfit.hip) -- the two forms give the same bits
 *   "chase_abort" 1       the bulge chase of the two-stage eigen-solver gives up at its first inter-workgroup wait (what a
 *                         device shared with another process does to it after 2 s): the constructor must come out through
 *                         the one-stage solver with the same spectra
 *   "flat_kappa_milli" v  factor (in thousandths) on the noise bound of CRM_MODEL_FLAT_OPTIMUM / CRM_MODEL_RHO_TIE (study tool)
 * These replace the environment switches of earlier versions; the GPU suite flips every one of them. */
int crm_test_set_form(const char* name, int value, int reset);
/* Contraction kernel variant for subsequent launches on this context: tile_width 0 = chosen per
 * launch, 64 or 128 forced; lds_dma = 1 lets 128-wide launches use the direct-to-LDS kernel. */
int crm_test_set_contraction(crm_ctx* ctx, int tile_width, int lds_dma);
/* every > 0 (default 1): Khatri-Rao launches of more than 1024 tiles run as 8 x 64 persistent workgroups that walk
 * contiguous tile runs per XCD and re-align (bounded wait) every `every` generations -- 7.5x less L2-fabric
 * traffic, 0.65 % slower; 0: one workgroup per tile. */
int crm_test_set_contraction_sync(crm_ctx* ctx, int every);
/* Times this context gave the persistent form up because its bounded waits ran out (the 512 workgroups were not
 * co-resident: a shared GPU); from then on its launches use one workgroup per tile. */
long crm_test_sync_fallbacks(const crm_ctx* ctx);
/* CRM_POISON=1 only (else always 0): number of device buffers of this process whose red zone -- 4 KiB of 0xFF behind every
 * allocation -- was found overwritten when the buffer was released, i.e. kernels that wrote past the end of a buffer. */
long crm_test_overruns(void);
/* ... and the same inspection, now, of the context's own work buffers (which live as long as the context). */
int crm_test_check_context(crm_ctx* ctx);
/* Self-test of that detector: writes eight bytes past the end of a scratch buffer on purpose (CRM_POISON=1 only) and
 * returns by how much the overrun count grew (1 with the poison fill, 0 without). */
int crm_test_overrun_selftest(crm_ctx* ctx);
/* C (M x N, ld N) = X' Y with X: cells x M, Y: cells x N (row-major, tight). */
int crm_test_contract(crm_ctx* ctx, long cells, int M, int N, const double* X, const double* Y,
                      double* C, int ksplit);
/* C ((B*k0) x N) = KR(G, E)' Y with G: cells x B, E: cells x k0, Y: cells x N. */
int crm_test_contract_kr(crm_ctx* ctx, long cells, int B, int k0, int N, const double* G,
                         const double* E, const double* Y, double* C);
/* Route of the multi-gene scan's contraction: -1 (default) cost model, 0 always per (variant, rho)
 * pair against Q0(rho), 1 once per variant against H followed by Mix(rho) per pair (when the
 * background keeps H). */
int crm_test_set_shared_h(crm_ctx* ctx, int mode);
/* on = 0: scans on this context contract against Q0(rho*) over all cells even when the background knows the donor
 * structure of its kinship factor (crm_background_set_kinship_groups); 1 (default): the kinship-structure route where its
 * flop count is below the direct contraction's; 2: always (environment: CRM_KIN_ROUTE=0 / 1 / 2). */
int crm_test_set_kinship_route(crm_ctx* ctx, int on);
/* Number of Khatri-Rao blocks of this context's scans whose last columns went through the 160-column-tile launch
 * (scan.hip: spectra with r mod 128 <= 32); tests use it to know which form they exercised. */
long crm_test_tail_launches(const crm_ctx* ctx);
/* ... and whose last few columns of the spectrum (r mod 128 <= 16 on the kinship-structure route) went through the skinny
 * one-pass kernel instead of another 128-column tile of the product with the mixing matrix. */
long crm_test_spectrum_tail_launches(const crm_ctx* ctx);
/* Variants that scans on the donor-collapsed path repeated on the dense path because they were nearly collinear with the
 * covariates (scan.hip: COLLINEAR_TAU; the dense path orthogonalises the block against W in the cell axis). */
long crm_test_dense_repeats(const crm_ctx* ctx);
/* Blocks of this context's scans whose per-donor sums H'(g o E0) came from one batched product against the symmetric pair
 * features E (x) E (scan.hip: donor pairs -- the kinship term's contexts are the scan's own; form "donor_pairs": 0 never,
 * 1 where its estimated time is the smaller one, 2 always). */
long crm_test_donor_pair_blocks(const crm_ctx* ctx);
/* (phenotype, variant) tests of this context's scans whose selected fit has no kinship term to speak of --
 * (v0 / v1) max S0(rho*) <= 1e-10: delta at its upper clamp -- and for which no rotated test direction A~ was formed
 * (scan.hip; form "pairs_without_kinship_term" = 0 forms it for every test). */
long crm_test_tests_without_pair(const crm_ctx* ctx);
/* The same product stored transposed: CT (N x (B*k0)). */
int crm_test_contract_kr_t(crm_ctx* ctx, long cells, int B, int k0, int N, const double* G,
                           const double* E, const double* Y, double* CT);
/* Null-fit objective at a given point instead of the search (c <= 8 covariate columns): while on, crm_scan_interaction
 * stops after the null-fit kernels of its first block (outputs untouched) and keeps, per (variant, grid point), the
 * log-likelihood and the scale at x = logit(delta); crm_test_null_fit_probe_read copies them out as
 * [variants][grid points][2] doubles and returns their number (> capacity: CRM_ERR_ARG).  Lets a test compare the
 * likelihood itself with the oracle's at the same point, apart from where the two searches stop. */
int crm_test_null_fit_probe(crm_ctx* ctx, int on, double x);
int crm_test_null_fit_probe_read(crm_ctx* ctx, double* out, long capacity);
/* Eigenvalues (ascending) of `count` symmetric k x k matrices (lower triangle read). */
int crm_test_eigvalsh(crm_ctx* ctx, int count, int k, const double* F, double* lambda);
/* Davies/Liu p-values for `count` (Q, lambda[k]) pairs after the eigenvalue filter. */
int crm_test_davies(crm_ctx* ctx, int count, int k, const double* Q, const double* lambda,
                    double* pvalue, int* ifault, double* liu);

/* The background constructor's symmetric eigen-solver on `batch` host matrices A (dim x dim each, row-major,
 * tight): lam (batch x dim, ascending), Z (batch x dim x dim, column j = eigenvector j; may be NULL).
 * stage 0: the whole solver; 1: tridiagonalisation only (d_out / e_out: batch x dim diagonals and
 * sub-diagonals, e[dim - 1] unused); 2: tridiagonal eigenproblem without the back-transformation (Z then
 * holds the eigenvectors of the tridiagonal matrix). */
int crm_test_eigh(crm_ctx* ctx, int batch, int dim, const double* A, double* lam, double* Z, int stage,
                  double* d_out, double* e_out);

/* The constructor's two-stage solver for a family of grid points A_q = D_q C D_q, D_q = diag(wa[q] on the first 64
 * coordinates, wb[q] on the rest) (cellregmap_amd/csrc/eigh2_band.hip, eigh2_chase.hip): C dim x dim host matrix
 * (row-major, symmetric).  stage 0: everything -- lam (nq x dim, ascending), Z (nq x dim x dim, column j = eigenvector j;
 * may be NULL); stage 1: the dense -> band reduction only (band_out: dim x dim, its lower triangle holds Q1' C Q1);
 * stage 2: ... and the chase of every scaled band (d_out / e_out: nq x dim diagonals and sub-diagonals). */
int crm_test_eigh2(crm_ctx* ctx, int nq, int dim, const double* C, const double* wa, const double* wb, double* lam, double* Z,
                   int stage, double* d_out, double* e_out, double* band_out);

/* Host-only (no GPU needed): the deflation plan of one divide-and-conquer merge of the eigen-solver -- eigenvalues
 * lam[n] of the two halves (n1 + n2), z[n] = last / first components of their eigenvectors, coupling beta.  Returns
 * k surviving poles, the rho of the normalised secular problem, rows[n] (local source rows: k survivors in pole order,
 * then the deflated ones), dl / w (poles and normalised update vector for the first k; deflated eigenvalues behind),
 * and the Givens rotations (a, b, c, s) x nrot that were applied. */
int crm_test_dc_plan(const double* lam, const double* z, int n1, int n, double beta, int* k, double* rho, int* rows,
                     double* dl, double* w, int* nrot, double* rots);

/* Host-only: the task table of the back-transformation through the chase's reflectors (cellregmap_amd/csrc/eigh2_back.hip)
 * for `batch` matrices of order `dim` on a device of `cus` compute units: count triples (matrix, first tile of sixteen
 * eigenvectors, tiles <= 8) in launch order.  CRM_ERR_ARG when capacity (ints) is too small; *count is set either way. */
int crm_test_back_tasks(int batch, long dim, int cus, int* tasks, int capacity, int* count);

#ifdef __cplusplus
}
#endif
#endif /* CRM_HIP_TEST_H */
