"""The reference's scan path restated on the CPU (oracle; test infrastructure only).

Follows /root/reference cellregmap/_cellregmap.py:
  * background construction         :63-131   (modes A / B / C)
  * interaction scan                :317-440  (11 REML fits -> rho* -> score test -> Davies)
  * association scans + LRT         :246-314, :443-469
  * functional wrappers             :471-587  (including their positional quirks)

This is the "reference-shaped" path: per variant it builds 11 LMM objects, each
redoing the Q0'[y, X] rotations, then evaluates P via three products with Q0
per solve.  It is what ``bench.py`` times as ``cpu_baseline`` (kind "port").
"""
import numpy as np

from .davies import davies_pvalue
from .lmm import LMM
from .scoretest import LowRankCov, Projection, score_F, score_Q
from .sugar import ddot, economic_qs_linear, economic_svd, epsilon

RHO_GRID = np.linspace(0, 1, 11)


def khatri_rao_halves(hK, E2):
    """get_L_values (_cellregmap.py:533-545): L_i = diag((U S)[:, i]) hK."""
    U, S, _ = economic_svd(E2)
    us = U * S
    return [ddot(us[:, i], hK) for i in range(us.shape[1])]


class OracleCellRegMap:
    """CellRegMap(y, E, W=None, Ls=None, E1=None, hK=None)  (_cellregmap.py:63)."""

    def __init__(self, y, E, W=None, Ls=None, E1=None, hK=None, polish=False):
        # polish=False (default) is the reference procedure verbatim (Brent, 1e-6); polish=True adds
        # the derivative-based refinement the HIP engine offers as an option (oracle/lmm.py)
        self._polish = bool(polish)
        self._y = np.asarray(y, float).flatten()
        self._E0 = np.asarray(E, float)
        Ls = [] if Ls is None else Ls
        self._W = np.asarray(W, float) if W is not None else np.ones((self._y.shape[0], 1))
        self._E1 = np.asarray(E1, float) if E1 is not None else np.asarray(E, float)
        self._Ls = [np.asarray(L, float) for L in Ls]
        n = self._y.shape[0]
        assert self._W.ndim == 2 and self._E0.ndim == 2 and self._E1.ndim == 2
        assert n == self._W.shape[0] == self._E0.shape[0] == self._E1.shape[0]
        for L in self._Ls:
            assert L.ndim == 2 and L.shape[0] == n

        self._half = {}
        self._qs = {}
        if len(self._Ls) == 0 and hK is None:  # mode A (:101-106)
            self._rho = [1.0]
            blocks = lambda rho: [self._E1]
        elif len(self._Ls) == 0:  # mode B (:107-116)
            hK = np.asarray(hK, float)
            self._rho = RHO_GRID
            blocks = lambda rho: [np.sqrt(rho) * self._E1, np.sqrt(1 - rho) * hK]
        else:  # mode C (:117-131); hK ignored
            self._rho = RHO_GRID
            blocks = lambda rho: [np.sqrt(rho) * self._E1] + [np.sqrt(1 - rho) * L for L in self._Ls]
        for rho in self._rho:
            hS = np.concatenate(blocks(rho), axis=1)
            self._half[rho] = hS
            self._qs[rho] = economic_qs_linear(hS, return_q1=False)

    @property
    def n_samples(self):
        return self._y.shape[0]

    # -- interaction (:317-440) ----------------------------------------------------
    def null_fit(self, X, restricted=True):
        """The rho loop (:345-357): first strictly-greater lml wins."""
        best_lml, best_rho, best = -np.inf, 0, None
        for rho in self._rho:
            lmm = LMM(self._y, X, self._qs[rho], restricted=restricted)
            lmm.fit(verbose=False, polish=self._polish)
            val = lmm.lml()
            if val > best_lml:
                best_lml, best_rho, best = val, rho, lmm
        return best_rho, best, best_lml

    def scan_interaction(self, G, idx_E=None, idx_G=None, return_stats=False):
        G = np.asarray(G, float)
        p = G.shape[1]
        pv = np.empty(p)
        info = {k: np.empty(p) for k in ("rho1", "e2", "g2", "eps2")}
        stats = {"Q": np.empty(p), "F": [], "delta": np.empty(p), "scale": np.empty(p), "lml": np.empty(p)}
        for i in range(p):
            g = G[:, [i]]
            X = np.concatenate((self._W, g), axis=1)
            rho, lmm, lml = self.null_fit(X, restricted=True)
            info["rho1"][i] = rho
            info["e2"][i] = lmm.v0 * rho
            info["g2"][i] = lmm.v0 * (1 - rho)
            info["eps2"][i] = lmm.v1
            Q0, S0 = self._qs[rho][0][0], self._qs[rho][1]
            cov = LowRankCov(Q0, S0, lmm.v0, lmm.v1)
            P = Projection(cov, X)
            E0 = self._E0 if idx_E is None else self._E0[idx_E, :]
            gtest = g.ravel() if idx_G is None else g.ravel()[idx_G]
            half_dK = ddot(gtest, E0)
            Q = score_Q(P, half_dK, self._y)
            F = score_F(P, half_dK)
            pv[i] = davies_pvalue(Q, F, True)[0]
            stats["Q"][i] = Q
            stats["F"].append(F)
            stats["delta"][i] = lmm.delta
            stats["scale"][i] = lmm.scale
            stats["lml"][i] = lml
        if return_stats:
            return pv, info, stats
        return pv, info

    # -- association (:246-314) --------------------------------------------------------
    def _assoc_null(self):
        rho, lmm, lml = self.null_fit(self._W, restricted=False)
        info = {
            "rho1": np.asarray([rho], float),
            "e2": np.asarray([lmm.v0 * rho], float),
            "g2": np.asarray([lmm.v0 * (1 - rho)], float),
            "eps2": np.asarray([lmm.v1], float),
        }
        return rho, lmm, lml, info

    def scan_association(self, G):
        G = np.asarray(G, float)
        rho, null, null_lml, info = self._assoc_null()
        alt = np.empty(G.shape[1])
        for i in range(G.shape[1]):
            X = np.concatenate((self._W, G[:, [i]]), axis=1)
            lmm = LMM(self._y, X, self._qs[rho], restricted=False)
            lmm.fit(verbose=False, polish=self._polish)
            alt[i] = lmm.lml()
        return lrt_pvalues(null_lml, alt, dof=1), info

    def scan_association_fast(self, G):
        G = np.asarray(G, float)
        rho, null, null_lml, info = self._assoc_null()
        alt = null.get_fast_scanner().fast_scan(G, verbose=False)["lml"]
        return lrt_pvalues(null_lml, alt, dof=1), info


    # -- effect sizes (:137-244) --------------------------------------------------------------------
    def predict_interaction(self, G, MAF):
        """Per-SNP persistent effect and cell-level GxC effects (BLUP), _cellregmap.py:137-205."""
        from .scoretest import cov_solve

        G = np.asarray(G, float)
        E0, W = self._E0, self._W
        maf = np.asarray(np.atleast_1d(MAF), float)
        norm = 1 / np.sqrt(2 * maf * (1 - maf))
        beta_g_s, beta_gxe_s = [], []
        for i in range(G.shape[1]):
            g = G[:, [i]]
            M = np.concatenate((W, g, E0), axis=1)
            gE = g * E0
            best_lml, best_rho, best, best_qs = -np.inf, 0, None, None
            for rho in self._rho:
                hS = np.concatenate([np.sqrt(rho) * gE] + [np.sqrt(1 - rho) * L for L in self._Ls], axis=1)
                qs = economic_qs_linear(hS, return_q1=False)
                lmm = LMM(self._y, M, qs, restricted=True)
                lmm.fit(verbose=False, polish=self._polish)
                if lmm.lml() > best_lml:
                    best_lml, best_rho, best, best_qs = lmm.lml(), rho, lmm, qs
            beta_g = best.beta[W.shape[1]]
            yadj = (self._y - best.mean()).reshape(-1, 1)
            cov = LowRankCov(best_qs[0][0], best_qs[1], best.v0, best.v1)
            v = cov_solve(cov, yadj)
            sigma2_gxe = best.v0 * best_rho
            beta_gxe = sigma2_gxe * E0 @ (gE.T @ v) * norm[i]
            beta_g_s.append(beta_g)
            beta_gxe_s.append(beta_gxe)
        return np.asarray(beta_g_s), np.stack(beta_gxe_s).T

    def estimate_aggregate_environment(self, g):
        """_cellregmap.py:207-244 (the LMM runs on the object's own background decompositions; only
        the final solve uses the per-SNP one)."""
        from .scoretest import cov_solve

        g = np.atleast_2d(g).reshape((np.asarray(g).size, 1))
        E0, W = self._E0, self._W
        gE = g * E0
        M = np.concatenate((W, g, E0), axis=1)
        best_lml, best_rho, best = -np.inf, 0, None
        half = {}
        for rho in self._rho:
            half[rho] = np.concatenate([np.sqrt(rho) * gE] + [np.sqrt(1 - rho) * L for L in self._Ls], axis=1)
            lmm = LMM(self._y, M, self._qs[rho], restricted=True)
            lmm.fit(verbose=False, polish=self._polish)
            if lmm.lml() > best_lml:
                best_lml, best_rho, best = lmm.lml(), rho, lmm
        yadj = self._y - best.mean()
        qs = economic_qs_linear(half[best_rho], return_q1=False)
        v = cov_solve(LowRankCov(qs[0][0], qs[1], best.v0, best.v1), yadj)
        return E0 @ (best_rho * best.v0 * gE.T @ v)


def compute_maf(X):
    """Plain-array branch of _cellregmap.py:589-638 (the reference imports dask / xarray first)."""
    X = np.asarray(X, float)
    s0 = np.nansum(X, axis=0) / (2 * np.logical_not(np.isnan(X)).sum(axis=0))
    return np.minimum(s0, 1 - s0)


def estimate_betas(y, W, E, G, maf=None, E1=None, E2=None, hK=None, polish=False):
    """_cellregmap.py:640-682."""
    E1 = E if E1 is None else E1
    E2 = E if E2 is None else E2
    Ls = None if hK is None else khatri_rao_halves(hK, E2)
    crm = OracleCellRegMap(y=y, E=E, W=W, E1=E1, Ls=Ls, polish=polish)
    if maf is None:
        maf = compute_maf(G)
    return crm.predict_interaction(G, maf)


def lrt_pvalues(null_lml, alt_lmls, dof=1):
    """_cellregmap.py:443-469."""
    from scipy.stats import chi2

    lrs = np.clip(-2 * null_lml + 2 * np.asarray(alt_lmls, float), epsilon.super_tiny, np.inf)
    pv = chi2(df=dof).sf(lrs)
    return np.clip(pv, epsilon.super_tiny, 1 - epsilon.tiny)


def run_interaction(y, E, G, W=None, E1=None, E2=None, hK=None, idx_G=None, polish=False):
    """_cellregmap.py:547-587.  NB ``idx_G`` is handed over positionally and so
    lands in ``scan_interaction``'s ``idx_E`` slot (:586 vs :318)."""
    E1 = E if E1 is None else E1
    E2 = E if E2 is None else E2
    Ls = None if hK is None else khatri_rao_halves(hK, E2)
    crm = OracleCellRegMap(y=y, E=E, W=W, E1=E1, Ls=Ls, polish=polish)
    return crm.scan_interaction(G, idx_G)


def run_association(y, W, E, G, hK=None):
    """_cellregmap.py:471-500.  The positional constructor call (:498) binds the
    covariates ``W`` to the ctor's ``E`` and the contexts ``E`` to its ``W``."""
    crm = OracleCellRegMap(y, W, E, hK=hK)
    return crm.scan_association(G)


def run_association_fast(y, W, E, G, hK=None):
    """_cellregmap.py:502-531 (same positional binding, :529)."""
    crm = OracleCellRegMap(y, W, E, hK=hK)
    return crm.scan_association_fast(G)
