// Device-resident objects behind the opaque handles of include/crm_hip.h.
#pragma once
#include <cmath>

#include "crm_internal.h"

namespace crm {
constexpr long CELL_PAD = 128;  // cell axis padded (zero rows) to a multiple of this
}

// Phenotype-free per-donor tables of the collapsed path.  They depend on the background, on the
// contexts E0 and on the donor structure of the panel (which cells belong to which donor), not on
// the phenotype nor on the genotypes, so a background keeps the most recent ones for all its genes
// and panels (an eQTL run scans thousands of genes, each against its own cis window, on one cohort).
struct crm_donor_tables {
    unsigned long e0_key = 0, group_key = 0;  // content hashes of E0 and of the donor index
    unsigned long stamp = 0;                  // recency
    crm::DevBuf TZ;   // [nrho][m_pad x ldq]          Z' Q0(rho)
    crm::DevBuf Bd;   // [nrho][(m_pad*k0) x ldq]     KR(Z, E0)' Q0(rho)
    crm::DevBuf Z2;   // [m_pad x ld]  Z'E   (or the m*m-row mixed table under idx_G)
    crm::DevBuf Z3;   // [m_pad x ld]  Z'(E (x) E)
    void release() { TZ.release(); Bd.release(); Z2.release(); Z3.release(); }
};

struct crm_background_builder;   // state of a constructor in progress (background.hip)
void crm_background_builder_free(crm_background_builder*);

// Sigma(rho) = Q0 diag(S0) Q0' for every grid point (cellregmap/_cellregmap.py:95-131).
struct crm_background {
    crm_background_builder* builder = nullptr;   // non-null between `begin` and `seal`
    crm_ctx* ctx = nullptr;
    long n = 0, n_pad = 0;
    int nrho = 0;
    double rho[crm::CRM_MAX_RHO] = {0};
    int r[crm::CRM_MAX_RHO] = {0};
    double ortho_defect[crm::CRM_MAX_RHO] = {0};  // max |Q0'Q0 - I| after the polish
    long ldq = 0;                        // common leading dimension (multiple of 128)
    crm::DevBuf Q0[crm::CRM_MAX_RHO];    // [n_pad x ldq], zero padded; thin branch: formed on first use (q0_ready)
    bool q0_ready[crm::CRM_MAX_RHO] = {false};
    crm::DevBuf Ht;                      // H' (ldh x n_pad), kept once a lazy Q0 has been formed
    crm::DevBuf S0[crm::CRM_MAX_RHO];    // [ldq]
    std::vector<double> s0_max;          // largest entry of every spectrum (host; filled when the background is sealed / created from given spectra)
    // thin branch with a well-conditioned kept spectrum: Q0(rho) = H Mix(rho), H = [E1, B]
    bool fast_T = false;
    long ldh = 0, cols = 0;
    crm::DevBuf H;                        // [n_pad x ldh]
    crm::DevBuf Mix[crm::CRM_MAX_RHO];   // [ldh x ldq]
    // Donor structure of the kinship factor (crm_background_set_kinship_groups): H = [E1, L_1 .. L_k2] with
    // L_j = diag(us[:, j]) hK and hK[c, :] = hKd[group(c), :] -- rows of hK constant within a donor, as in the reference's
    // "expanded" kinship factor.  Then  H'(g o E0)  needs no n-length contraction against the cols = k1 + k2 m columns of H:
    //   S[d', (us_j | E1_a), (b, i)] = sum over the cells c of donor d' of  [us(c, j) | E1(c, a)] g_b(c) E0(c, i)
    //   (H'(g o E0))[(j, d), .] = sum_d' hKd[d', d] S[d', us_j, .],   (H'(g o E0))[a, .] = sum_d' S[d', E1_a, .]
    // i.e. 2 n (k1 + k2) k0 flops per variant for S instead of 2 n cols k0 (40 times fewer at config 3).
    bool kin = false;
    long kin_groups = 0, kin_groups_pad = 0;   // m' donors (padded to the contraction's stage depth)
    long kin_cols = 0;                          // m: columns of hK
    int kin_k1 = 0, kin_k2 = 0;
    long kin_rows = 0;                          // cells in donor order, every donor's run padded to a multiple of 16 rows
    std::vector<long> kin_row0, kin_len;        // first row / padded length of each donor's run
    crm::DevBuf kin_map;                        // int[kin_rows]: cell of a sorted row, -1 for padding
    crm::DevBuf kin_Y;                          // [kin_rows x kin_ldy]: columns 0..k2-1 us, k2..k2+k1-1 E1, donor order
    long kin_ldy = 128;
    crm::DevBuf kin_hKd;                        // [kin_groups_pad x kin_ldh]
    long kin_ldh = 0;
    // The donor-level factor folded into the mixing matrices: with rows ordered [E1_a ; (d', us_j)],
    //   Q0(rho)'(g o E0) = MixK(rho)' [sum over all cells for E1_a ; S[d', us_j, .]],
    //   MixK(rho)[a, :] = Mix(rho)[a, :],   MixK(rho)[k1 + d' k2 + j, :] = sum_d hKd[d', d] Mix(rho)[k1 + j m + d, :]
    // -- the contraction over the donors is done once per grid point, when the structure is announced, instead of once
    // per block of variants; the per-donor sums S are then the operand of the Mix product as they stand.  Taken when
    // k1 + donors k2 is no more than a quarter longer than cols = k1 + m k2 (donor-level factors of full rank).
    bool kin_fold = false;
    long kin_kdim = 0;                          // k1 + kin_groups * k2, padded to whole stages of the contraction
    crm::DevBuf MixK[crm::CRM_MAX_RHO];        // [kin_kdim x ldq]
    // shared donor tables, most recently used first (at most DT_CACHE entries)
    static constexpr int DT_CACHE = 2;
    std::vector<crm_donor_tables*> dt_cache;
    unsigned long dt_clock = 0;
};

// Q0(rho_i) = H Mix(rho_i) in device memory (i < 0: every grid point).  Backgrounds of the thin branch keep the
// half factor H and the mixing matrices; the rotations of the scan go through them, and most scans select only a
// few grid points as rho*, so Q0 -- 2 n cols r flops and n x r doubles per grid point -- is formed on first use.
int crm_background_require_q0(crm_background* bg, int i);

// One phenotype: y, W, E0 and what only depends on them.
struct crm_gene {
    crm_background* bg = nullptr;
    crm_ctx* ctx = nullptr;  // (kept separately: destruction must not depend on the background's lifetime)
    int c = 0, k0 = 0;
    long ld_yw = 0, ldw = 0, lde = 0;
    std::vector<double> W_host;   // the covariates as used (orthogonal columns), n x c: crm_gene_create_like sums W'y from it
    std::vector<double> W_basis;  // c x c, empty = identity: W_host = W V for the caller's W (crm_gene_create brought correlated
                                  // columns to orthogonal ones); crm_lmm_fit returns V beta, the coefficients of the caller's W
    crm::DevBuf yW;   // [n_pad x ld_yw]: column 0 = y, columns 1..c = W
    crm::DevBuf E0;   // [n_pad x lde]
    crm::DevBuf WW, Wy;
    crm::DevBuf Wproj;   // (W'W)^-1 [c x c], eigenvectors V of W'W [c x c], eigenvalues d^2 [c]: launch_ortho_block
    double yy = 0.0;
    crm::DevBuf rot;  // [nrho][(1+c) x ldq]: rows Q0(rho)'y, Q0(rho)'W_i
    // features of the (possibly row-permuted) contexts, rebuilt per scan call
    crm::DevBuf Ep, YE, EE, idx;
    crm::DevBuf kinEp;    // the (permuted) contexts in the donor order of the background's kinship structure
    crm::DevBuf kinP;     // pair products E1_a o E0_i of the folded kinship-structure form (E1 rows of step 6)
    crm::DevBuf kinUE;    // us o E0 in donor order (folded form with a single column of us: mode B)
    crm::DevBuf kinEE;    // E (x) E (pairs j <= j') in donor order (folded form whose kinship contexts are the scan's own)
    long ld_ep = 0, ld_ye = 0, ld_ee = 0;
    // donor tables of the collapsed path (valid for one grouped panel and the identity permutation)
    unsigned long e0_key = 0;    // content hash of E0 (key of the background's shared donor tables)
    unsigned long w_key = 0;     // content hash of W (genes of one multi-gene pass must agree on it)
    unsigned long dt_group = 0;  // donor structure (panel group_key) dt_Z1 / dt_sums were built for (0 = none)
    crm_donor_tables dt_own;     // phenotype-free tables under a permutation hook (not shareable)
    crm::DevBuf dt_Z1;    // [m_pad x ld]     Z'[y o E, W o E]
    crm::DevBuf dt_sums;  // [m_pad x DT_SUMS_LD]: column 0 group size, 1 sum y, 2.. sum W_i
    crm::DevBuf dt_Zt;    // indicators of the permuted groups (idx_G) + the permuted group index
};

struct crm_panel {
    crm_ctx* ctx = nullptr;
    unsigned long uid = 0;  // process-unique (addresses get recycled)
    long n = 0, n_pad = 0, p = 0, ld = 0;
    crm::DevBuf G;  // dense: [n_pad x ld]
    // grouped (donor-constant) panel: cell i carries the genotypes of group[i]
    bool grouped = false;
    long m = 0, m_pad = 0;
    crm::DevBuf Gd;     // [m_pad x ld] one row per donor
    crm::DevBuf group;  // int[n]
    unsigned long group_key = 0;  // content hash of (group, m)
    crm::DevBuf Z;      // [n_pad x ldz] 0/1 indicator of the groups (operand of the table builds)
    long ldz = 0;
};
