/* C-ABI of the MI355X score-test engine (libcrm_hip.so).
 *
 * Status: the reference (limix/CellRegMap, pure Python) has no FFI seam of its
 * own; its boundary for this path is the Python API
 *     CellRegMap.__init__            cellregmap/_cellregmap.py:63-131
 *     CellRegMap.scan_interaction    cellregmap/_cellregmap.py:317-440
 *     CellRegMap.scan_association*   cellregmap/_cellregmap.py:246-314
 *     run_interaction / run_association(_fast)   :471-587
 * The entry points below are what a ctypes binding inside those methods binds
 * (see INTEGRATION.md for the stub).  Conventions: plain pointers and sizes,
 * float64 host buffers owned by the caller, int status (0 = OK, < 0 = error,
 * text via crm_last_error()), no exceptions across the boundary (every entry
 * point catches them: CRM_ERR_INTERNAL), one context = one device + one HIP
 * stream + one set of work buffers.  Calls that touch one context -- directly or
 * through a background, gene or panel created on it -- are serialised by the
 * library (a lock per context), so handles may be used from several threads;
 * distinct contexts run concurrently (the destroy entry points take the same
 * lock).  Settings (crm_set_*) belong to the context, not to the calling thread --
 * except the progress callback, which is kept per calling thread.
 */
#ifndef CRM_HIP_H
#define CRM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CRM_OK 0
#define CRM_ERR_HIP (-1)
#define CRM_ERR_ARG (-2)
#define CRM_ERR_UNSUPPORTED (-3)
#define CRM_ERR_NUMERIC (-4)
#define CRM_ERR_INTERNAL (-5) /* host side: out of memory or an unexpected C++ exception, stopped at the boundary */

typedef struct crm_ctx crm_ctx;
typedef struct crm_background crm_background;
typedef struct crm_gene crm_gene;
typedef struct crm_panel crm_panel;

/* Thread-local text of the last error raised on this thread. */
const char* crm_last_error(void);
/* Library version, "major.minor.patch". */
const char* crm_version(void);

/* ---- context ------------------------------------------------------------------- */
int crm_ctx_create(int device, crm_ctx** out);
void crm_ctx_destroy(crm_ctx* ctx);
/* Block until all work queued on the context's stream has finished. */
int crm_ctx_synchronize(crm_ctx* ctx);
/* Hand the context's cached work buffers back to the device (the constructor's eigen-solver keeps its
 * workspace between calls -- up to five matrices per owned grid point; it is also released on its own when another
 * allocation of this library would otherwise run out of memory). */
int crm_ctx_trim(crm_ctx* ctx);

/* ---- background covariance: replaces CellRegMap.__init__'s rho loop ---------------
 * (_cellregmap.py:101-131 + numpy_sugar.economic_qs_linear, twin _math.py:238-256).
 *
 * From precomputed economic decompositions: for each of `nrho` grid points,
 * Q0[i] is n x r[i] (row-major, leading dimension r[i]) and S0[i] has r[i] entries. */
int crm_background_create_qs(crm_ctx* ctx, long n, int nrho, const double* rho, const int* r,
                             const double* const* Q0, const double* const* S0,
                             crm_background** out);
/* On-device decomposition of the half covariances
 *     hS(rho) = [ sqrt(rho) * E1 , sqrt(1-rho) * B ]        B: n x kb (row-major, ld kb)
 * (mode B: B = hK; mode C: B = [L_1 ... L_q]; mode A: kb = 0 and rho = {1}).
 * Columns whose squared singular value is below `rel_tol` * max are dropped (they are
 * inert in every quantity of the path); rel_tol <= 0 selects the default 1e-12. */
int crm_background_create(crm_ctx* ctx, long n, const double* E1, int k1, const double* B, long kb,
                          int nrho, const double* rho, double rel_tol, crm_background** out);
/* Mode C without materialising the halves on the host: B = [L_1 ... L_k2], L_j = diag(U[:, j]) hK
 * (get_L_values, _cellregmap.py:533-545; U = left singular vectors of E2 times the singular values,
 * n x k2; hK n x m) is formed on the device.  Same result as crm_background_create with B given. */
int crm_background_create_hadamard(crm_ctx* ctx, long n, const double* E1, int k1, const double* U, int k2,
                                   const double* hK, int m, int nrho, const double* rho, double rel_tol,
                                   crm_background** out);
/* Donor structure of the kinship factor of a background built by crm_background_create_hadamard (or _begin with U / hK):
 * the rows of hK are constant within a donor -- hK[c, :] = hKd[group[c], :], group[c] in [0, groups), hKd groups x m
 * row-major (the reference's callers pass the "expanded" factor of a donor-level relatedness matrix, _cellregmap.py:559).
 * U is the n x k2 matrix given to the constructor.  With it the interaction scan of GENERAL genotypes takes
 * Q0(rho*)'(g o E0) as Mix(rho*)'[H'(g o E0)] and forms H'(g o E0) donor by donor: 2 n (k1 + k2) k0 + 2 cols r k0 flops per
 * variant instead of 2 n r k0 (3.6 times fewer at 20 000 cells x 50 contexts x 100 donors); results agree with the direct
 * route to rounding.  Optional: without it the scan contracts against Q0 itself.  The announcement is verified against the
 * background's own half factor entry by entry (H[c, k1 + j m + d] = U[c, j] hKd[group[c], d]): CRM_ERR_ARG when it does not
 * hold, and the background then keeps scanning by the direct route. */
int crm_background_set_kinship_groups(crm_background* bg, const int* group, long groups, const double* hKd, long m,
                                      const double* U, int k2);
/* Number of donors of the kinship structure in use by this background (0: none announced, or not usable). */
int crm_background_kinship_groups(const crm_background* bg);
/* > 0 when the donor-level factor has been folded into the mixing matrices (MixK(rho)[k1 + d' k2 + j, :] =
 * sum_d hKd[d', d] Mix(rho)[k1 + j m + d, :], formed once when the structure is announced): the per-donor sums are then
 * the operand of the product with the mixing matrix as they stand and no contraction over the donors runs per block of
 * variants.  The value is the length of that product's contraction, k1 + donors k2 padded to whole stages.  Taken when it
 * is at most a quarter longer than cols = k1 + m k2 (donor-level factors of full rank). */
long crm_background_kinship_folded(const crm_background* bg);
/* The same constructor split over several processes, one per GPU (the grid points are decomposed
 * by different ranks, the results exchanged over RCCL; cellregmap_amd/distributed.py drives it):
 *   begin    -- H = [E1, B] (B explicit, or U / hK as in crm_background_create_hadamard when B == NULL), Gram
 *               matrix, eigen-decompositions of the grid points with mine[i] != 0 (NULL: all); afterwards
 *               crm_background_rank(bg, i) is the rank of an owned grid point, -1 otherwise
 *   complete -- ranks[i] of ALL grid points (they fix the common leading dimension): buffers for every grid
 *               point, Q0 / S0 / Mix computed for the owned ones
 *   layout   -- n_pad, ldq, ldh (0 when there are no mixing matrices): slot sizes in doubles are
 *               Q0 n_pad * ldq, S0 ldq, Mix ldh * ldq.  With mixing matrices (cols < n) only S0 and Mix are
 *               exchanged: every rank holds H and forms Q0(rho) = H Mix(rho) itself when a scan first needs it
 *   export / import -- copy slot `what` (0 Q0, 1 S0, 2 Mix) of grid point i to / from a buffer of the caller:
 *               device memory of the same GPU (e.g. a torch tensor's data_ptr(), what RCCL broadcasts) or host
 *               memory (a CPU tensor under the gloo backend) -- the owner exports and broadcasts, the others import
 *   seal     -- all slots filled: finish (the object is then an ordinary background)
 * crm_background_create* == begin(all) + complete + seal. */
int crm_background_begin(crm_ctx* ctx, long n, const double* E1, int k1, const double* B, long kb, const double* U,
                         int k2, const double* hK, int m, int nrho, const double* rho, const int* mine,
                         double rel_tol, crm_background** out);
int crm_background_complete(crm_background* bg, const int* ranks);
int crm_background_layout(const crm_background* bg, long* n_pad, long* ldq, long* ldh, int* has_mix);
int crm_background_export(const crm_background* bg, int i, int what, void* dst_device);
int crm_background_import(crm_background* bg, int i, int what, const void* src_device);
int crm_background_seal(crm_background* bg);
void crm_background_destroy(crm_background* bg);
/* Introspection / read-back (tests): rank at grid point i; copy of S0 / Q0 (n x r, ld r). */
int crm_background_rank(const crm_background* bg, int i);
int crm_background_read(const crm_background* bg, int i, double* Q0, double* S0);

/* ---- one phenotype bound to a background: y, W (n x c), E0 (n x k0) ------------------
 * (the per-object state of CellRegMap: _y, _W, _E0; _cellregmap.py:64-79).
 * W: best passed as U diag(s) of its thin SVD with the singular values below sqrt(eps) dropped (numpy_sugar.economic_svd,
 * the basis glimix-core's LMM holds its covariates in; the scans depend on W through its column space only) -- mutually
 * orthogonal columns need no arithmetic here.  Other W is brought to that form inside the call (W <- W V, V from repeated
 * Jacobi passes on W'W; exact to the working precision up to cond(W) ~ 1e7, where the reference's rank rule takes over);
 * only W that is rank deficient by that rule is refused with CRM_ERR_NUMERIC.  The scans then orthogonalise every block of variants against W in the cell axis and
 * apply the reference's rank rules to [W, g] (economic_svd in the null fits, lstsq in the projection of the score test).
 * c <= 128, k0 <= 256, and in the interaction scan k0 + c + 2 <= 288 (past 128 contexts or 144 rows: slower kernel forms). */
int crm_gene_create(crm_background* bg, const double* y, const double* W, int c, const double* E0,
                    int k0, crm_gene** out);
/* Another phenotype y on the cohort of `like` (same background, covariates and contexts): what a gene keeps of W and E0 is
 * copied on the device instead of being checked, hashed and uploaded again -- the per-gene calls of the reference's
 * run_interaction over many genes of one cohort (_cellregmap.py:547-587) then cost one upload of y and its rotations each.
 * Results are bit for bit those of crm_gene_create(bg, y, W, c, E0, k0). */
int crm_gene_create_like(const crm_gene* like, const double* y, crm_gene** out);
/* ... and `ngenes` of them in one call: Y is n x ngenes (row-major, leading dimension ldy, column j = phenotype j), out
 * receives ngenes handles (none on failure).  The rotations Q0(rho)'y of the whole batch are one product against the
 * background's half factor and one against every mixing matrix, so those operands are read once per batch. */
int crm_gene_create_batch(const crm_gene* like, const double* Y, long ldy, int ngenes, crm_gene** out);
void crm_gene_destroy(crm_gene* gene);

/* ---- genotype panel resident in HBM: G is n x p, row-major, leading dimension ldg ------ */
int crm_panel_create(crm_ctx* ctx, long n, const double* G, long ldg, long p, crm_panel** out);
/* Donor-constant panel ("Genotypes (expanded)", _cellregmap.py:488,561): cell i carries the genotypes
 * of group[i] in [0, m); Gd is m x p (row-major, leading dimension ldg), one row per donor.  Scans of
 * such a panel are exact collapses of the dense computation onto per-donor tables (every n-length
 * contraction is linear in diag(g) or diag(g)^2); results agree with the dense path to rounding.  The
 * genotype permutation hook (idx_G) falls back to the dense path by expanding blocks on the fly. */
int crm_panel_create_grouped(crm_ctx* ctx, long n, const int* group, long m, const double* Gd, long ldg,
                             long p, crm_panel** out);
/* Compact ingest of a donor-constant panel: `dosage` holds one signed byte per (donor, variant) -- allele counts
 * 0 / 1 / 2, m x p row-major with leading dimension ldd -- i.e. n / m x 8 times less data over PCIe than the
 * expanded float64 matrix "Genotypes (expanded)" of _cellregmap.py:488,561.  standardise != 0: every variant is
 * centred and scaled on the device by the mean and (population) standard deviation of its EXPANDED column, donors
 * weighted by their cell counts (what the reference's callers do on the host before expanding); a monomorphic
 * variant is then an error (CRM_ERR_NUMERIC).  The panel behaves like one from crm_panel_create_grouped. */
int crm_panel_create_grouped_i8(crm_ctx* ctx, long n, const int* group, long m, const signed char* dosage, long ldd,
                                long p, int standardise, crm_panel** out);
/* Upload an expanded n x p matrix; reject non-finite entries (CRM_ERR_NUMERIC: the reference's LMM
 * raises ValueError on them) and, given a candidate grouping (group_hint[i] in [0, m_hint),
 * rep_rows[d] = index of a cell of group d), verify on the device that every cell equals its group's
 * representative in every variant.  *out_grouped = 1 when the panel was stored donor-level. */
int crm_panel_create_auto(crm_ctx* ctx, long n, const double* G, long ldg, long p, const int* group_hint,
                          long m_hint, const long* rep_rows, crm_panel** out, int* out_grouped);
/* on = 0 forces the dense path for grouped panels (default 1). */
int crm_set_donor_collapse(crm_ctx* ctx, int on);
void crm_panel_destroy(crm_panel* panel);

/* ---- interaction scan: replaces the loop body of scan_interaction (_cellregmap.py:340-436)
 * for variants [first, first + count) of the panel.  idx_E / idx_G are the permutation
 * hooks of :398-413 (NULL = identity; n entries each).  Outputs have `count` entries;
 * optional outputs may be NULL.  out_lambda (count x k0) receives the eigenvalues of F in
 * ascending order, out_F (count x k0 x k0) the matrix itself.
 * A variant whose selected null fit has no kinship term to speak of -- (v0 / v1) max S0(rho*) <= 1e-10: delta at its
 * upper clamp, a phenotype without a random effect -- is tested with K0 = v1 I: the rotated test direction
 * Q0(rho*)'(g o E0), which enters Q and F of such a fit through weights <= 1e-10, is not formed. */
int crm_scan_interaction(crm_gene* gene, crm_panel* panel, long first, long count, const int* idx_E,
                         const int* idx_G, double* out_pvalue, double* out_rho1, double* out_e2,
                         double* out_g2, double* out_eps2, double* out_Q, double* out_lml,
                         double* out_delta, double* out_scale, double* out_lambda, double* out_F);

/* The same scan returning what chiscore.davies_pvalue(Q, F, True) reports beside the p-value (_cellregmap.py:435 asks
 * for it and discards it): out_ifault = Davies' fault code per variant (0: converged = Is_Converged 1; 1 / 4: the
 * integration gave up and the p-value IS the modified-Liu one, 2: round-off flagged, the integral kept; -2: no
 * eigenvalue above the SKAT threshold, p = NaN where the reference raises), out_liu_pvalue = info["liu_pval"]. */
int crm_scan_interaction_info(crm_gene* gene, crm_panel* panel, long first, long count, const int* idx_E,
                              const int* idx_G, double* out_pvalue, int* out_ifault, double* out_liu_pvalue,
                              int* out_model_flags);
/* out_model_flags (may be NULL): per variant, where the reference's own answer is decided by rounding noise rather than by
 * the data, so that two faithful implementations (or two BLAS builds under the reference) may report different numbers:
 *   SATURATED      the background's columns and the fixed effects [W, g] together span all n cells (rank + c + 1 >= n):
 *                  the complement terms (u'v - (Q0'u)'(Q0'v)) / delta of glimix-core's likelihood are rounding noise
 *                  divided by delta (mode A with at least as many contexts as cells is the extreme case);
 *   DELTA_AT_ZERO  the null fit at rho* ended at delta <= 1e-8: the likelihood was flat or still rising towards delta = 0
 *                  (no residual variance left), where the same noise / delta terms decide the reported optimum;
 *   G_IN_SPAN_W    the variant lies in the span of the covariates: the fit dropped it (glimix-core's SVD-reduced X);
 *   FLAT_OPTIMUM   the P-VALUE of this variant may differ by more than 1e-5 (relative) between two faithful runs of the
 *                  reference's procedure: bound_p of crm_scan_interaction_bounds (below) exceeds the tolerance p-values are
 *                  held to.  1.6 - 2.0 % of the scans of the measured streams; none of the others was found beyond;
 *   STATISTIC_AT_TOLERANCE   the same for the score statistic Q and its tolerance of 1e-6 (bound_Q > 1e-6).  The reference
 *                  stops its search at rtol = atol = 1e-6 on logit(delta), so Q -- which moves by ~1e-6 of its value per
 *                  tolerance -- is reproducible to about that and no better on more than a third of all scans (36 - 38 %);
 *   RHO_TIE        the likelihoods of rho* and of another grid point differ by less than the first-order bound on the
 *                  rounding noise of the objective (nullfit.hip): which of the two the reference's strict `>`
 *                  (_cellregmap.py:354-357) keeps is decided by rounding, and info["rho1"] with it (phenotypes without a
 *                  kinship term tie on the whole grid: their p-values do not depend on rho). */
#define CRM_MODEL_SATURATED 1
#define CRM_MODEL_DELTA_AT_ZERO 2
#define CRM_MODEL_G_IN_SPAN_W 4
#define CRM_MODEL_FLAT_OPTIMUM 8
#define CRM_MODEL_RHO_TIE 16
#define CRM_MODEL_STATISTIC_AT_TOLERANCE 32

/* How reproducible is each variant's result?  The reference stops its null fit with Brent's search at rtol = atol = 1e-6
 * on x = logit(delta) (_cellregmap.py:351-352, glimix-core LMM.fit).  Two faithful implementations (this library and the
 * reference's numpy; two BLAS builds under the reference) evaluate the same likelihood to ~1e-15 of its value, and that is
 * enough to move where the search stops: the last parabolic steps are quotients of differences of nearly equal values, and
 * the final comparisons f(x +- tol) <= f(x) can fall either way -- by up to one whole tolerance.  The library measures, per
 * variant, (i) how far Q and p move when delta moves by one tolerance (the score test re-evaluated to either side: scale
 * re-estimated, Q, F, eigenvalues and p recomputed) and (ii) how flat the likelihood is at the stopping point (its gain over
 * one tolerance, from the search's own kernel), and returns
 *   out_bound_Q    bound on |Q - Q'| / max(Q, tr F) between two faithful runs  = (i) x min(1, 2.5e-13 / (ii))
 *   out_bound_p    the same for |p - p'| / p
 *   out_model_flags   CRM_MODEL_* bits (above); FLAT_OPTIMUM = bound_p > 1e-5, STATISTIC_AT_TOLERANCE = bound_Q > 1e-6.
 * On the device-vs-oracle streams the constant was read from (71 000 scans) no scan exceeded its
 * bounds (p: plus the ~1e-6 that two roundings of Davies' integration differ by); held-out streams: profiles/r06_*.
 * out_ifault / out_liu_pvalue: as crm_scan_interaction_info.  Every output but out_model_flags may be NULL. */
int crm_scan_interaction_bounds(crm_gene* gene, crm_panel* panel, long first, long count, const int* idx_E, const int* idx_G,
                                double* out_pvalue, int* out_ifault, double* out_liu_pvalue, int* out_model_flags,
                                double* out_bound_Q, double* out_bound_p);

/* nperm permutations of one scan in one call -- the reference's use of its permutation hooks (_cellregmap.py:398-413; its
 * calibration test cellregmap/test/test_struct_lmm2.py:208-209 calls scan_interaction(G, idx_E=perm) in a loop).  The hooks
 * enter only the test direction sqrt(dK) = diag(g[idx_G]) E0[idx_E]: the eleven null fits, rho*, the variance components and
 * the rotations of the variants are the same for every permutation and are computed once per block of variants; each
 * permutation then runs the score test proper (its contraction, Gram, eigenvalues, Davies).
 *   idx_E, idx_G   nperm x n int (row q = the permutation of call q; either may be NULL = identity for every call)
 *   out_pvalue     nperm x count; out_Q the same (may be NULL)
 *   out_rho1 / out_e2 / out_g2 / out_eps2   count each (may be NULL): permutation-independent
 * Row q of the outputs is bit for bit what crm_scan_interaction(gene, panel, first, count, idx_E + q n, idx_G + q n, ...)
 * returns. */
int crm_scan_interaction_permuted(crm_gene* gene, crm_panel* panel, long first, long count, int nperm, const int* idx_E,
                                  const int* idx_G, double* out_pvalue, double* out_rho1, double* out_e2, double* out_g2,
                                  double* out_eps2, double* out_Q);

/* Several phenotypes against one panel in one pass ("genes" that share the background, W and E0):
 * everything that does not depend on y -- G'Q0(rho), the Khatri-Rao contraction per (variant, rho)
 * pair selected by at least one gene, the y-free side contractions -- is computed once per block.
 * Outputs are ngenes x count, row-major (gene-major); optional ones may be NULL.  Results per gene
 * are those of crm_scan_interaction. */
int crm_scan_interaction_multi(crm_gene* const* genes, int ngenes, crm_panel* panel, long first, long count,
                               const int* idx_E, const int* idx_G, double* out_pvalue, double* out_rho1,
                               double* out_e2, double* out_g2, double* out_eps2, double* out_Q);

/* ---- association scans: replace scan_association (fast = 0, _cellregmap.py:246-281: ML refit per
 * SNP at the null model's rho) and scan_association_fast (fast = 1, :284-314: glimix-core
 * FastScanner, delta frozen at the null) including lrt_pvalues (:443-469).  The null model (ML,
 * X = W, first strictly larger lml over the rho grid) is refitted on every call.
 * out_pvalue / out_alt_lml: `count` entries (may be NULL); out_null: 6 doubles
 * {rho1, e2, g2, eps2, null lml, null delta} (may be NULL). */
int crm_scan_association(crm_gene* gene, crm_panel* panel, long first, long count, int fast,
                         double* out_pvalue, double* out_alt_lml, double* out_null);

/* ---- effect sizes: the device operations behind predict_interaction (_cellregmap.py:137-205) and
 * estimate_aggregate_environment (:207-244).
 * crm_lmm_fit: LMM(y, M, QS(rho), restricted).fit() for every grid point of the gene's background
 * (M = the gene's covariate matrix), keeping the first strictly larger lml.
 *   out_fit  6 doubles {rho, v0, v1, lml, delta, grid index}
 *   out_beta c doubles (may be NULL): the fixed effects of the kept fit (LMM.beta), as coefficients of the columns of M
 *            the caller passed to crm_gene_create (columns the library orthogonalised are mapped back)
 * crm_cov_solve: out = (v0 Q0 S0 Q0' + v1 I)^-1 rhs for grid point rho_index of `bg` -- QSCov.solve
 * (_math.py:40-67); rhs and out are n x m row-major host arrays. */
int crm_lmm_fit(crm_gene* gene, int restricted, double* out_fit, double* out_beta);
int crm_cov_solve(crm_background* bg, int rho_index, double v0, double v1, const double* rhs, int m,
                  double* out);

/* Block size (variants per internal batch); 0 restores the default (automatic: up to 4096 variants of
 * the interaction scan while its largest work buffer stays within 16 GB; 1024 for the association scans). */
int crm_set_block_variants(crm_ctx* ctx, int variants);
/* Progress of the scans the CALLING THREAD runs on this context: `callback(done, total, user)` is called on that thread
 * after every internal block of variants (the reference shows a tqdm bar over variants, _cellregmap.py:270,340); NULL
 * switches it off.  The callback is kept per calling thread: two threads that scan on one context each install and see
 * their own. */
int crm_set_progress_callback(crm_ctx* ctx, void (*callback)(long done, long total, void* user), void* user);
/* on = 1 (default): for backgrounds built on the device with a well-conditioned kept spectrum
 * (S_max <= 1e6 S_min), the rotations G'Q0(rho) of the dense scan are taken as Mix(rho)'(H'G) with
 * Q0(rho) = H Mix(rho) -- one n-length product instead of one per grid point.  on = 0: always the
 * direct products. */
int crm_set_fast_rotation(crm_ctx* ctx, int on);
/* Null-fit convergence.  on = 0 (default): the reference's procedure verbatim (Brent on
 * logit(delta), rtol = atol = 1e-6, glimix-core LMM.fit as called at _cellregmap.py:352).
 * on = 1: followed by secant steps on the analytic derivative, which pins the optimum to ~1e-12
 * and makes Q and the p-value insensitive to summation-order noise in the likelihood; the result
 * then differs from the reference's by the reference's own optimiser tolerance (~1e-6 on Q). */
int crm_set_null_fit_polish(crm_ctx* ctx, int on);

/* ---- instrumentation ----------------------------------------------------------------
 * Sum of HIP-event durations (ms) and launch count of the dominant kernel (the Khatri-Rao
 * contraction) since the last reset, measured on the context's stream. */
int crm_kernel_timer_reset(crm_ctx* ctx);
int crm_kernel_timer_read(crm_ctx* ctx, double* kr_ms, long* kr_launches, double* kr_flops,
                          double* total_ms);
/* Stop recording (scans after this call create no further events; what was recorded stays readable
 * until the next reset).  Recording also stops by itself after 65536 launches. */
int crm_kernel_timer_stop(crm_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* CRM_HIP_H */
