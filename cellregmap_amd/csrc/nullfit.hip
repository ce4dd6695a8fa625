// Batched null-model fits of the interaction scan.
//
// Reference: for every variant, `for rho1 in self._rho1: LMM(y, [W, g], QS[rho1],
// restricted=True).fit(); keep the first strictly larger lml`
// (cellregmap/_cellregmap.py:345-357), then rho1 / e2 / g2 / eps2 (:366-369).
//
// One wavefront per (variant, rho grid point).  A wavefront holds the
// rotated vectors t_u = Q0(rho)' u  (u in {W columns, g, y}) only as streams from L2:
// each likelihood evaluation is one pass over the r spectrum entries with lanes striding
// j, accumulating  sum_j t_u[j] t_v[j] / ((1-d) S0[j] + d)  for all pairs and
// sum_j log((1-d) S0[j] + d), followed by a 64-lane butterfly so that every lane holds
// bitwise identical totals.  The scalar logic on top of the totals -- closed-form beta
// and scale, REML log-likelihood, and the bracket + Brent search over x = logit(d) with
// rtol = atol = 1e-6 -- is executed redundantly by all lanes (wave-uniform control flow)
// and follows oracle/brent.py statement by statement.
#include "crm_internal.h"
#include "nullfit.h"
#include "brent_search.h"
#include "wave_ops.h"
#include <type_traits>

namespace crm {

namespace {

constexpr double LOG2PI = 1.8378770664093453;
constexpr double EPS_TINY = 2.220446049250313e-16;    // numpy_sugar.epsilon.tiny
constexpr double EPS_SMALL = 1.4901161193847656e-08;  // numpy_sugar.epsilon.small

// FROM = 32: all six levels (one fit per wavefront); FROM = 8: levels 8 .. 1 (a row of sixteen lanes)
template <int N, int FROM>
__device__ inline void butterfly_sums(double (&v)[N]) {
#ifdef CRM_NF_BUTTERFLY_SUM   // (diagnostic builds, tools/diag/compare_builds.py: the round-5 form, value by value)
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int off = FROM; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
    return;
#endif
    double t[N];
    if constexpr (FROM >= 32) {
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = __shfl_xor(v[i], 32, 64);
#pragma unroll
        for (int i = 0; i < N; i++) v[i] += t[i];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = lane_xor_swizzle(v[i], 16);
#pragma unroll
        for (int i = 0; i < N; i++) v[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = lane_xor_swizzle(v[i], 8);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += t[i];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = lane_xor_swizzle(v[i], 4);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += t[i];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = lane_xor_quad(v[i], 2);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += t[i];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = lane_xor_quad(v[i], 1);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += t[i];
}

// 1 / D for D > 0 (normal range): hardware reciprocal + two Newton steps -- full double precision to
// the last bit or so, without the scaling / fix-up sequence of an IEEE division.
__device__ inline double fast_rcp(double D) {
    double x = __builtin_amdgcn_rcp(D);
    double e = fma(-D, x, 1.0);
    x = fma(x, e, x);
    e = fma(-D, x, 1.0);
    return fma(x, e, x);
}

// sum_j log D_j is accumulated as a product of mantissas and a sum of exponents (two bit-field
// operations, one multiply and one integer add per entry) and turned into a logarithm once per pass.
struct LogProduct {
    double mant = 1.0;
    int expo = 0;
    __device__ inline void mul(double D) {
        mant *= __builtin_amdgcn_frexp_mant(D);
        expo += __builtin_amdgcn_frexp_exp(D);
    }
    __device__ inline void renorm() {  // mantissas lie in [0.5, 1): call at least every ~900 entries
        expo += __builtin_amdgcn_frexp_exp(mant);
        mant = __builtin_amdgcn_frexp_mant(mant);
    }
    __device__ inline double log_value() const { return log(mant) + (double)expo * 0.6931471805599453; }
};

__device__ inline double logistic_clamped(double x) {
    double v;
    if (x > 0.0) {
        v = 1.0 / (1.0 + exp(-x));
    } else {
        v = exp(x);
        v = v / (v + 1.0);
    }
    return fmin(fmax(v, EPS_TINY), 1.0 - EPS_TINY);
}

__host__ __device__ constexpr int pair_index(int u, int v, int U) {
    // u <= v, row-major upper triangle
    return u * U - u * (u - 1) / 2 + (v - u);
}

// In-place Cholesky of the leading P x P block; returns false on a non-positive pivot.
template <int P>
__device__ inline bool cholesky(double (&A)[P][P], double& logdet) {
    logdet = 0.0;
#pragma unroll
    for (int j = 0; j < P; j++) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j][k] * A[j][k];
        if (!(d > 0.0)) return false;
        const double l = sqrt(d);
        A[j][j] = l;
        logdet += 2.0 * log(l);
#pragma unroll
        for (int i = j + 1; i < P; i++) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k];
            A[i][j] = s / l;
        }
    }
    return true;
}

// The same factorisation with the logarithms of its pivots left to the caller (piv[j] = L_jj): the objective takes them,
// together with log(delta) and log(s), in ONE pass of the log routine with one argument per lane -- every lane of the
// wavefront executes the scalar part of an evaluation redundantly, so five logarithms one after the other cost five times
// what five logarithms side by side do; the routine and its arguments are the same, and so is every bit of the results.
template <int P>
__device__ inline bool cholesky_pivots(double (&A)[P][P], double (&piv)[P]) {
#pragma unroll
    for (int j = 0; j < P; j++) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j][k] * A[j][k];
        if (!(d > 0.0)) return false;
        const double l = sqrt(d);
        A[j][j] = l;
        piv[j] = l;
#pragma unroll
        for (int i = j + 1; i < P; i++) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k];
            A[i][j] = s / l;
        }
    }
    return true;
}

template <int P>
__device__ inline void cholesky_solve(const double (&L)[P][P], double (&b)[P]) {
#pragma unroll
    for (int i = 0; i < P; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i][k] * b[k];
        b[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < P; k++) s -= L[k][i] * b[k];
        b[i] = s / L[i][i];
    }
}

// First-order bound, in units of one rounding (2^-53), on what rounding can do to a value of the objective:
//   rss = b' K b with b = (-beta, 1): every entry K_uv = sum_j t_u t_v / D_j + (u'v - t_u't_v) / delta carries the
//   roundings of its terms' magnitudes, sum_j |t_u t_v| / D_j <= sqrt(K~_uu K~_vv) (spectrum part) and
//   (|u'v| + |t_u't_v|) / delta (the complement is a difference: this is where a small delta amplifies);
//   the value takes rss through (df / 2) log rss;
//   the other terms -- log-determinants, n log s, the constant -- through their own magnitudes.
// Kept out of line: the objective's own expression tree (and with it every fused multiply-add the compiler forms there)
// must not depend on whether a kernel also asks for this bound.
template <int U>
__device__ __noinline__ double objective_noise_bound(const double (&bb)[U], const double (&sd)[U], const double (&pl)[U * (U + 1) / 2],
                                                     double inv_d, double rss, double df, double lsum, double n_minus_r, double log_delta,
                                                     double n, double log_s, double logdetXX, double logdetH, double p_eff) {
    constexpr double LOG2PI_ = 1.8378770664093453;
    double mag = 0.0;
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int v = u; v < U; v++) {
            const double spec = sqrt(fabs(sd[u] * sd[v]));
            const double m_uv = spec + pl[pair_index(u, v, U)] * inv_d;
            mag += (u == v ? 1.0 : 2.0) * bb[u] * bb[v] * m_uv;
        }
    const double logs = fabs(lsum) + fabs(n_minus_r) * fabs(log_delta) + n * fabs(log_s) + df * (LOG2PI_ + 1.0) + fabs(logdetXX) +
                        fabs(logdetH) + p_eff * fabs(log_s);
    return 0.5 * (df * mag / fabs(rss) + logs);
}

// One fit: variant b at grid point w, by one wavefront.  SH: the vectors every variant of a grid point shares --
// Q0'W, Q0'y, S0 -- are read from LDS (sW [C][sld], sy, sS) instead of global memory.
// EX: the spectrum pass in the reference's own operations -- an IEEE division and one log per entry -- instead of the
// hardware reciprocal + Newton steps and the mantissa-product log-determinant (NullFitArgs::exact).
// TR: the search also leaves its trace behind (brent_search.h; NullFitTrial::margin / noise / xunc) -- the kernels of the calls
// that ask for model flags.  The kernels without it are the scan's.
// G: fits per wavefront.  1: the wavefront's 64 lanes stride the spectrum of one fit.  4 (LDS-shared kernel): every row of
// sixteen lanes runs a fit of its own (four variants of one grid point), each lane standing for the four lanes
// sub, sub + 16, sub + 32, sub + 48 of the one-fit form -- the same entries in the same order into four separate
// accumulators, reduced with the same pairings -- so that every sum is the one-fit form's to the last bit, while the part
// of an evaluation that every lane executes redundantly (logistic, Cholesky, solves, logarithms, the search's own
// arithmetic: most of an evaluation at a short spectrum) is executed once for four fits.  Rows diverge as their searches
// do; nothing crosses a row (DPP row operations and in-row shuffles only).
template <int C, bool SH, bool EX, bool TR, int G = 1>
__device__ __forceinline__ void nullfit_fit(const NullFitArgs& a, const int b, const int w, const int lane,
                                            const double* sW, const double* sy, const double* sS, const int sld) {
    static_assert(G == 1 || (G == 4 && C == 1 && !EX), "four fits per wavefront: one covariate column, default arithmetic");
    const int sub = G == 1 ? lane : (lane & 15);   // lane within its fit
    constexpr int P = C + 1;  // columns of X = [W, g]
    constexpr int U = C + 2;  // ... plus y
    constexpr int NP = U * (U + 1) / 2;
    const NullFitRho R = a.rho[w];
    const double* __restrict__ tg = R.T + (long)b * R.ldT;
    const int r = R.r;
    const double n = (double)a.n;

    // plain inner products u'v:  W'W, W'y, y'y per gene; W'g, g'g, g'y per variant
    double uv[NP];
#pragma unroll
    for (int i = 0; i < C; i++) {
#pragma unroll
        for (int j = i; j < C; j++) uv[pair_index(i, j, U)] = a.WW[i * C + j];
        uv[pair_index(i, C, U)] = a.gW[(long)b * a.ld_gW + i];
        uv[pair_index(i, C + 1, U)] = a.Wy[i];
    }
    uv[pair_index(C, C, U)] = a.gg[b];
    uv[pair_index(C, C + 1, U)] = a.gy[b];
    uv[pair_index(C + 1, C + 1, U)] = a.yy;

    // Is g (numerically) inside span(W)?  Then X = [W, g] has rank C and the reference's
    // SVD-reduced covariates drop that direction (glimix-core LMM; lstsq in PMat).
    bool use_g = true;
    double logdetXX = 0.0;
    {
        double A[P][P];
#pragma unroll
        for (int i = 0; i < P; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) A[i][j] = uv[pair_index(j, i, U)];
        // Cholesky with the last pivot inspected by hand
        double ld = 0.0;
        bool ok = true;
#pragma unroll
        for (int j = 0; j < P; j++) {
            double d = A[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= A[j][k] * A[j][k];
            if (j == P - 1) {
                // the reference's rule on the singular values of [W, g] where the block was orthogonalised against W
                // (g_drop, blockops.hip); else the relative size of the last pivot
                if (a.g_drop ? a.g_drop[b] != 0 : !(d > 1e-12 * A[j][j])) {
                    use_g = false;
                    break;
                }
                if (!(d > 0.0)) {
                    ok = false;
                    break;
                }
            } else if (!(d > 0.0)) {
                ok = false;
                break;
            }
            const double l = sqrt(d);
            A[j][j] = l;
            ld += 2.0 * log(l);
#pragma unroll
            for (int i = j + 1; i < P; i++) {
                double s = A[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k];
                A[i][j] = s / l;
            }
        }
        logdetXX = ok ? ld : NAN;
    }
    const double p_eff = use_g ? (double)P : (double)C;
    const double df = a.restricted ? n - p_eff : n;

    // one pass over the spectrum: weighted pair sums (+ log-determinant part); with GRAD also the
    // sums with weight (1 - S_j) / D_j^2 that make up d/d(delta) of every bilinear form
    auto spectrum_pass = [&](double delta, bool weighted, double (&S)[NP], double& lsum,
                             bool grad, double (&S2)[NP], double& lsum2) {
#pragma unroll
        for (int i = 0; i < NP; i++) { S[i] = 0.0; S2[i] = 0.0; }
        lsum = 0.0;
        lsum2 = 0.0;
        LogProduct lp;
        int trips = 0;
        const double omd = 1.0 - delta;
        // UNR spectrum entries per lane and trip, all loads issued before the arithmetic: the loop is
        // bound by L2 latency otherwise (two wavefronts per SIMD at this register count)
        constexpr int UNR = C <= 2 ? 4 : 2;
        // a trip whose entries all exist for every lane (j0 - lane + 64 UNR - 1 < r: wave-uniform) runs without the
        // selects that mask the entries beyond r -- a fifth of the pass's instructions; the values are the same
        auto trip = [&](const int j0, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            double t[UNR][U], s0[UNR];
            bool ok[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) {
                const int j = j0 + 64 * q;
                ok[q] = FULL || j < r;
                const int jj = ok[q] ? j : r - 1;
#pragma unroll
                for (int i = 0; i < C; i++) t[q][i] = SH ? sW[i * sld + jj] : R.tW[(long)i * R.ldW + jj];
                t[q][C] = tg[jj];
                t[q][C + 1] = SH ? sy[jj] : R.ty[jj];
                s0[q] = weighted ? (SH ? sS[jj] : R.S0[jj]) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < UNR; q++) {
                double wgt = ok[q] ? 1.0 : 0.0, wgt2 = 0.0;
                if (weighted) {
                    const double D = ok[q] ? omd * s0[q] + delta : 1.0;
                    const double inv = EX ? 1.0 / D : fast_rcp(D);
                    wgt = ok[q] ? inv : 0.0;
                    if (EX) lsum += log(D);   // (D = 1 for the padding entries)
                    else lp.mul(D);
                    if (grad) {
                        const double oms = 1.0 - s0[q];
                        lsum2 += ok[q] ? oms * inv : 0.0;
                        wgt2 = ok[q] ? oms * inv * inv : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const double tw = t[q][u] * wgt;
                    const double tw2 = t[q][u] * wgt2;
#pragma unroll
                    for (int v = u; v < U; v++) {
                        S[pair_index(u, v, U)] += tw * t[q][v];
                        if (grad) S2[pair_index(u, v, U)] += tw2 * t[q][v];
                    }
                }
            }
        };
        for (int j0 = lane; j0 < r; j0 += 64 * UNR) {
            if (j0 - lane + 64 * UNR - 1 < r) trip(j0, std::true_type{});
            else trip(j0, std::false_type{});
            if (!EX && weighted && (++trips & 127) == 0) lp.renorm();
        }
        {
            double red[NP + 1];
#pragma unroll
            for (int i = 0; i < NP; i++) red[i] = S[i];
            red[NP] = weighted ? (EX ? lsum : lp.log_value()) : 0.0;
            butterfly_sums<NP + 1, 32>(red);
#pragma unroll
            for (int i = 0; i < NP; i++) S[i] = red[i];
            if (weighted) lsum = red[NP];
        }
        if (grad) {
            double red[NP + 1];
#pragma unroll
            for (int i = 0; i < NP; i++) red[i] = S2[i];
            red[NP] = lsum2;
            butterfly_sums<NP + 1, 32>(red);
#pragma unroll
            for (int i = 0; i < NP; i++) S2[i] = red[i];
            lsum2 = red[NP];
        }
    };

    // The same pass for four fits per wavefront (G = 4): this lane is the one-fit form's lanes v = sub + 16 q, q = 0 .. 3.
    // Lane v of that form takes the entries j = v + 64 k, k = 0 .. 4 T_v - 1 with T_v = ceil((r - v) / 256) trips of four
    // (entries beyond r: loaded from r - 1 with weight 0 and D = 1, as there); here k runs in trips of four as well, all four
    // q side by side, a trip being unmasked for every lane of the wavefront while 256 t + 255 < r.
    auto spectrum_pass4 = [&](double delta, bool weighted, double (&S)[NP], double& lsum) {
        double Sq[4][NP];
        LogProduct lpq[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int i = 0; i < NP; i++) Sq[q][i] = 0.0;
        const double omd = 1.0 - delta;
        int trips[4];
#pragma unroll
        for (int q = 0; q < 4; q++) trips[q] = sub + 16 * q < r ? (r - (sub + 16 * q) + 255) / 256 : 0;
        const int t_all = (r + 255) / 256;   // trips of the one-fit form's lane 0: no lane has more
        for (int t = 0; t < t_all; t++) {
            const bool full = 256 * t + 255 < r;   // (wave-uniform: every entry of the trip exists for every lane)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!full && t >= trips[q]) continue;
                double tv[4][U], s0[4];
                bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = sub + 16 * q + 64 * (4 * t + k);
                    ok[k] = full || j < r;
                    const int jj = ok[k] ? j : r - 1;
#pragma unroll
                    for (int i = 0; i < C; i++) tv[k][i] = SH ? sW[i * sld + jj] : R.tW[(long)i * R.ldW + jj];
                    tv[k][C] = tg[jj];
                    tv[k][C + 1] = SH ? sy[jj] : R.ty[jj];
                    s0[k] = weighted ? (SH ? sS[jj] : R.S0[jj]) : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    double wgt = ok[k] ? 1.0 : 0.0;
                    if (weighted) {
                        const double D = ok[k] ? omd * s0[k] + delta : 1.0;
                        const double inv = fast_rcp(D);
                        wgt = ok[k] ? inv : 0.0;
                        lpq[q].mul(D);
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const double tw = tv[k][u] * wgt;
#pragma unroll
                        for (int v = u; v < U; v++) Sq[q][pair_index(u, v, U)] += tw * tv[k][v];
                    }
                }
                if (weighted && ((t + 1) & 127) == 0) lpq[q].renorm();
            }
        }
        // the butterfly's levels 32 and 16 pair q with q ^ 2 and then q ^ 1 -- inside this lane; levels 8 .. 1 cross the row
        double red[NP + 1];
#pragma unroll
        for (int i = 0; i < NP; i++) red[i] = (Sq[0][i] + Sq[2][i]) + (Sq[1][i] + Sq[3][i]);
        red[NP] = weighted ? (lpq[0].log_value() + lpq[2].log_value()) + (lpq[1].log_value() + lpq[3].log_value()) : 0.0;
        butterfly_sums<NP + 1, 8>(red);
#pragma unroll
        for (int i = 0; i < NP; i++) S[i] = red[i];
        lsum = weighted ? red[NP] : 0.0;
    };

    double tt[NP];  // t_u' t_v (complement correction)
    {
        double dummy, dummy2, unused[NP];
        if constexpr (G == 1) spectrum_pass(1.0, false, tt, dummy, false, unused, dummy2);
        else spectrum_pass4(1.0, false, tt, dummy);
    }

    double cur_delta = 0.5, cur_scale = 1.0, cur_lml = -INFINITY;
    int nfev = 0;
    // The logistic is clamped to [eps, 1 - eps]: every x beyond +-36.7 is the SAME delta, and the objective there the same
    // number to the last bit -- a phenotype without a random effect (delta -> 1: half of the genes of an eQTL run) sends the
    // reference's bracketing phase through 63, 127, 255, 511, 709 and Brent's iteration after it, dozens of evaluations of
    // one value.  The two clamped points are evaluated once and remembered (bit-identical results, fewer spectrum passes).
    double memo_f[2] = {0.0, 0.0}, memo_scale[2] = {0.0, 0.0}, memo_lml[2] = {0.0, 0.0}, memo_noise[2] = {NAN, NAN};
    bool memo_set[2] = {false, false};
    bool last_clamped = false;   // the last evaluation was one of the two clamped points
    bool want_noise = false;     // the next evaluation also bounds the rounding noise of its value (cur_noise)
    double cur_noise = NAN;
    // f(x) = -lml at d = logistic(x), with beta and scale profiled out
    auto f = [&](double x) -> double {
        nfev++;
        const double delta = logistic_clamped(x);
        const int clamp = delta == 1.0 - EPS_TINY ? 1 : (delta == EPS_TINY ? 0 : -1);
        last_clamped = clamp >= 0;
        if (clamp >= 0 && memo_set[clamp]) {
            cur_delta = delta;
            cur_scale = memo_scale[clamp];
            cur_lml = memo_lml[clamp];
            cur_noise = memo_noise[clamp];
            return memo_f[clamp];
        }
        auto remember = [&](double value) -> double {
            if (clamp >= 0) {
                memo_set[clamp] = true;
                memo_f[clamp] = value;
                memo_scale[clamp] = cur_scale;
                memo_lml[clamp] = cur_lml;
                memo_noise[clamp] = cur_noise;
            }
            return value;
        };
        double S[NP], lsum;
        if constexpr (G == 1) {
            double unused[NP], unused2;
            spectrum_pass(delta, true, S, lsum, false, unused, unused2);
        } else {
            spectrum_pass4(delta, true, S, lsum);
        }
        const double inv_d = 1.0 / delta;
        double K[NP];  // u' Kt^-1 v
#pragma unroll
        for (int i = 0; i < NP; i++) K[i] = S[i] + (uv[i] - tt[i]) * inv_d;
        double A[P][P], rhs[P], xky[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) A[i][j] = K[pair_index(j, i, U)];
            rhs[i] = K[pair_index(i, C + 1, U)];
        }
        if (!use_g) {
#pragma unroll
            for (int j = 0; j < C; j++) A[C][j] = 0.0;
            A[C][C] = 1.0;
            rhs[C] = 0.0;
        }
#pragma unroll
        for (int i = 0; i < P; i++) xky[i] = rhs[i];
        double piv[P];
        double val;
        if (!cholesky_pivots<P>(A, piv)) {
            cur_delta = delta;
            cur_scale = NAN;
            cur_lml = NAN;
            return remember(INFINITY);
        }
        cholesky_solve<P>(A, rhs);  // rhs <- beta
        double rss = K[pair_index(C + 1, C + 1, U)];
#pragma unroll
        for (int i = 0; i < P; i++) rss -= xky[i] * rhs[i];
        const double s = fmax(rss / df, EPS_SMALL);
        // log(delta), log(L_jj), log(s): lane q takes argument q (the others 1), one pass of log, the results read back
        double larg = 1.0;
        larg = sub == 0 ? delta : larg;
#pragma unroll
        for (int j = 0; j < P; j++) larg = sub == 1 + j ? piv[j] : larg;
        larg = sub == P + 1 ? s : larg;
#ifdef CRM_NF_SERIAL_LOGS      // (diagnostic builds: the round-5 form, one logarithm after the other)
        const double log_delta = log(delta), log_s = log(s);
        double logdetH = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) logdetH += 2.0 * log(piv[j]);
        (void)larg;
#else
        const double lres = log(larg);
        // (one fit per wavefront: scalar reads; four: a shuffle inside the row of the fit)
        auto from_sub = [&](int k) -> double {
            if constexpr (G == 1) return read_lane(lres, k);
            else return __shfl(lres, (lane & 48) + k, 64);
        };
        const double log_delta = from_sub(0), log_s = from_sub(P + 1);
        double logdetH = 0.0;
#pragma unroll
        for (int j = 0; j < P; j++) logdetH += 2.0 * from_sub(1 + j);
#endif
        const double logdetK = lsum + (n - (double)r) * log_delta;
        val = -0.5 * (df * LOG2PI + df + n * log_s + logdetK);
        if (a.restricted) val += 0.5 * (logdetXX - (logdetH - p_eff * log_s));
        cur_delta = delta;
        cur_scale = s;
        cur_lml = val;
        if constexpr (TR) {
            if (want_noise || clamp >= 0) {
                double bb[U], sd[U], pl[NP];
#pragma unroll
                for (int i = 0; i < P; i++) bb[i] = fabs(rhs[i]);
                bb[P] = 1.0;
#pragma unroll
                for (int u = 0; u < U; u++) sd[u] = S[pair_index(u, u, U)];
#pragma unroll
                for (int i = 0; i < NP; i++) pl[i] = fabs(uv[i]) + fabs(tt[i]);
                cur_noise = objective_noise_bound<U>(bb, sd, pl, inv_d, rss, df, lsum, (n - (double)r), log_delta, n, log_s, logdetXX,
                                                     logdetH, p_eff);
            }
        }
        return remember(-val);
    };

    // g(x) = d(-lml)/dx with beta and scale profiled out (oracle/lmm.py: _neg_lml_grad_at)
    auto g = [&](double x) -> double {
        const double delta = logistic_clamped(x);
        if (delta <= EPS_TINY || delta >= 1.0 - EPS_TINY) return 0.0;
        double S[NP], S2[NP], lsum, lsum2;
        spectrum_pass(delta, true, S, lsum, true, S2, lsum2);
        const double inv_d = 1.0 / delta, inv_d2 = inv_d * inv_d;
        double K[NP], dK[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const double cpl = uv[i] - tt[i];
            K[i] = S[i] + cpl * inv_d;
            dK[i] = -S2[i] - cpl * inv_d2;
        }
        const double dlogdet = lsum2 + (n - (double)r) * inv_d;
        double A[P][P], dA[P][P], b[P], db[P], beta[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
#pragma unroll
            for (int j = 0; j < P; j++) {
                const int pi = i <= j ? pair_index(i, j, U) : pair_index(j, i, U);
                A[i][j] = K[pi];
                dA[i][j] = dK[pi];
            }
            b[i] = K[pair_index(i, C + 1, U)];
            db[i] = dK[pair_index(i, C + 1, U)];
        }
        if (!use_g) {
#pragma unroll
            for (int j = 0; j < P; j++) {
                A[C][j] = 0.0; A[j][C] = 0.0;
                dA[C][j] = 0.0; dA[j][C] = 0.0;
            }
            A[C][C] = 1.0;
            b[C] = 0.0;
            db[C] = 0.0;
        }
        double ld;
        if (!cholesky<P>(A, ld)) return NAN;
#pragma unroll
        for (int i = 0; i < P; i++) beta[i] = b[i];
        cholesky_solve<P>(A, beta);
        double Rv = K[pair_index(C + 1, C + 1, U)];
        double dR = dK[pair_index(C + 1, C + 1, U)];
#pragma unroll
        for (int i = 0; i < P; i++) {
            Rv -= b[i] * beta[i];
            dR -= 2.0 * db[i] * beta[i];
#pragma unroll
            for (int j = 0; j < P; j++) dR += beta[i] * dA[i][j] * beta[j];
        }
        double d = df * dR / Rv + dlogdet;
        if (a.restricted) {
            double tr = 0.0;
#pragma unroll
            for (int j = 0; j < P; j++) {
                double col[P];
#pragma unroll
                for (int i = 0; i < P; i++) col[i] = dA[i][j];
                cholesky_solve<P>(A, col);
                tr += col[j];
            }
            d += tr;
        }
        return 0.5 * d * delta * (1.0 - delta);
    };

    if (a.probe) {
        // test hook (crm_test_null_fit_probe): the objective at one given x instead of the search, so that the
        // likelihood itself can be compared with the oracle's at the same point
        (void)f(a.probe_x);
        if (sub == 0) {
            NullFitTrial t;
            t.lml = cur_lml; t.delta = cur_delta; t.scale = cur_scale; t.use_g = use_g ? 1 : 0; t.nfev = nfev;
            t.margin = NAN; t.noise = NAN; t.curv = NAN;
            a.trial[(long)b * a.nrho + w] = t;
        }
        return;
    }
    {
        // ---- bracket + Brent localmin (brent_search.h = oracle/brent.py, statement for statement) ----------------
        struct Objective {
            decltype(f)& fn;
            const bool& at_clamp;
            __device__ inline double operator()(double x) { return fn(x); }
            __device__ inline bool clamped() const { return at_clamp; }
        } objective{f, last_clamped};
        BrentTrace trace;
        double bf0;
        double bx0 = brent_search<TR>(objective, trace, bf0);
        if constexpr (G == 1) if (a.polish) {   // (the derivative's pass exists in the one-fit form only)
            // secant steps on the analytic derivative (oracle/lmm.py: _polish)
            const double xs = bx0, fs = bf0;
            double xa = xs, ga = g(xa);
            if (isfinite(ga) && ga != 0.0) {
                double xb = ga > 0.0 ? xa - 1e-4 : xa + 1e-4;
                double gb = g(xb);
                bool reject = false;
                for (int it = 0; it < 8; it++) {
                    if (!isfinite(gb) || gb == ga) break;
                    const double xn = xb - gb * (xb - xa) / (gb - ga);
                    if (!isfinite(xn) || fabs(xn - xs) > 1e-2) { reject = true; break; }
                    const double step = fabs(xn - xb);
                    xa = xb; ga = gb;
                    xb = xn;
                    gb = g(xb);
                    if (gb == 0.0 || step <= 1e-12 * (1.0 + fabs(xn))) break;
                }
                if (!reject && isfinite(gb)) {
                    const double fb = f(xb);
                    if (fb <= fs + 1e-9 * fabs(fs)) bx0 = xb;
                }
            }
        }
        // (tracked kernels: the objective one stopping tolerance to either side of where the search stopped -- how flat the
        // likelihood is there says how far rounding can move the last parabolic steps; include/crm_hip.h)
        double f_up = NAN, f_dn = NAN;
        if constexpr (TR) {
            const double tolx = 1e-6 * fabs(bx0) + 1e-6;
            f_up = f(bx0 + tolx);
            f_dn = f(bx0 - tolx);
        }
        want_noise = true;
        const double f_stop = f(bx0);  // LMM.fit(): beta and scale refreshed at the optimum
        if (sub == 0) {
            NullFitTrial t;
            t.lml = cur_lml;
            t.delta = cur_delta;
            t.scale = cur_scale;
            t.use_g = use_g ? 1 : 0;
            t.nfev = nfev;
            if constexpr (TR) {
                t.margin = fmin(trace.cmp, trace.sign);
                t.curv = 0.5 * (f_up + f_dn) - f_stop;
                t.noise = cur_noise;
            } else {
                (void)f_stop;
                t.margin = NAN; t.curv = NAN; t.noise = NAN;
            }
            a.trial[(long)b * a.nrho + w] = t;
        }
    }
}

template <int C, bool EX, bool TR>
#ifdef CRM_NULLFIT_WAVES
__attribute__((amdgpu_waves_per_eu(CRM_NULLFIT_WAVES, CRM_NULLFIT_WAVES)))
#endif
__global__ __launch_bounds__(64) void nullfit_kernel(NullFitArgs a) {
    nullfit_fit<C, false, EX, TR, 1>(a, (int)blockIdx.x, (int)blockIdx.y, (int)threadIdx.x, nullptr, nullptr, nullptr, 0);
}

// The same fits with the shared vectors of a grid point resident in LDS.  The one-wavefront-per-fit kernel above
// re-reads Q0'W, Q0'y and S0 from L2 on every likelihood evaluation: 4096 variants x 11 grid points x ~30 evaluations
// x 160 KB at config 3 = 36 TB/s at the speed it runs -- the L2's aggregate bandwidth, not the vector ALU, is what
// bounds it.  Here one workgroup of twelve wavefronts per CU copies the three vectors of a grid point into LDS once
// (120 KB at r = 5000) and its wavefronts then draw variants of that grid point from a queue (an atomic counter per
// grid point: likelihood-evaluation counts differ from variant to variant); when the queue is empty the workgroup moves
// on to the next grid point with work left.  Only Q0'g still comes from L2.
// (wavefronts per workgroup = per CU: twelve with one fit each -- 168 registers; eight with four fits each -- 256 registers:
// four sets of accumulators)
constexpr int nf_shared_waves(int G) { return G == 1 ? 12 : 8; }
template <int C, bool EX, bool TR, int G>
__global__ __launch_bounds__(64 * nf_shared_waves(G)) void nullfit_shared_kernel(NullFitArgs a, int variants, int sld,
                                                                             unsigned* __restrict__ queue) {
    extern __shared__ double nf_sm[];   // Q0'W [C][sld], Q0'y [sld], S0 [sld]
    __shared__ int any_left;
    double* const sW = nf_sm;
    double* const sy = nf_sm + C * sld;
    double* const sS = sy + sld;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w0 = (int)(blockIdx.x % (unsigned)a.nrho);
    for (int pass = 0; pass < a.nrho; pass++) {
        const int w = (w0 + pass) % a.nrho;
        if (tid == 0)   // (the counter only grows: a stale "work left" costs one LDS fill, never a missed variant)
            any_left = __hip_atomic_load(&queue[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)variants;
        __syncthreads();
        const bool any = any_left != 0;
        __syncthreads();
        if (!any) continue;
        const NullFitRho R = a.rho[w];
        for (int j = tid; j < R.r; j += 64 * nf_shared_waves(G)) {
#pragma unroll
            for (int i = 0; i < C; i++) sW[i * sld + j] = R.tW[(long)i * R.ldW + j];
            sy[j] = R.ty[j];
            sS[j] = R.S0[j];
        }
        __syncthreads();
        for (;;) {
            // One ticket per wavefront, wave-uniform by construction: every lane issues the atomic, lane 0 adds G and
            // the others add zero, so lane 0's return value is this wavefront's own ticket whatever form the compiler
            // gives the 64 lane-atomics (one per lane, or one per wavefront with a prefix sum) -- no branch on the lane
            // around the atomic, no reliance on a particular optimisation.  Tickets count variants: G per wavefront.
            const unsigned ticket = __hip_atomic_fetch_add(&queue[w], lane == 0 ? (unsigned)G : 0u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            const unsigned b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)ticket);
            if (b0 >= (unsigned)variants) break;
            if constexpr (G == 1) {
                nullfit_fit<C, true, EX, TR, 1>(a, (int)b0, w, lane, sW, sy, sS, sld);
            } else {
                const unsigned b = b0 + (unsigned)(lane >> 4);    // a variant per row of sixteen lanes
                if (b < (unsigned)variants) nullfit_fit<C, true, EX, TR, G>(a, (int)b, w, lane, sW, sy, sS, sld);
            }
        }
        __syncthreads();
    }
}

// rho* = first strictly larger lml over the grid (_cellregmap.py:354-357).  launch_nullfit fills the trial records
// with 0xFF bytes (nfev = -1) before the fit kernels run: a (variant, grid point) no kernel fitted is reported as
// rho_index = -1, which every host consumer turns into CRM_ERR_NUMERIC instead of using what the memory held.
__global__ void select_rho_kernel(const NullFitTrial* __restrict__ trial, int nrho, int variants,
                                  NullFitOut* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= variants) return;
    double best = -INFINITY;
    int bi = -1;
    bool fitted = true;
    for (int i = 0; i < nrho; i++) {
        const NullFitTrial t = trial[(long)b * nrho + i];
        if (t.nfev <= 0) fitted = false;
        if (t.lml > best) {
            best = t.lml;
            bi = i;
        }
    }
    NullFitOut o;
    o.use_g = trial[(long)b * nrho].use_g;
    o.decision = NAN; o.rho_decision = NAN; o.margin = NAN; o.noise = NAN; o.gap = NAN; o.curv = NAN;
    if (!fitted) {
        o.rho_index = -1;
        o.lml = NAN; o.delta = NAN; o.scale = NAN; o.v0 = NAN; o.v1 = NAN;
    } else if (bi < 0) {
        o.rho_index = 0;
        o.lml = NAN; o.delta = NAN; o.scale = NAN; o.v0 = NAN; o.v1 = NAN;
    } else {
        const NullFitTrial t = trial[(long)b * nrho + bi];
        o.rho_index = bi;
        o.lml = best;
        o.delta = t.delta;
        o.scale = t.scale;
        o.v0 = t.scale * (1.0 - t.delta);
        o.v1 = t.scale * t.delta;
        // distance from another outcome in units of the noise bound: the search at rho*, and the choice of rho* itself
        constexpr double ROUNDING = 1.1102230246251565e-16;   // 2^-53
        o.margin = t.margin;
        o.noise = t.noise;
        o.curv = t.curv;
        o.decision = t.margin / (ROUNDING * t.noise);
        double gap = INFINITY, rdec = INFINITY;
        for (int i = 0; i < nrho; i++) {
            if (i == bi) continue;
            const NullFitTrial q = trial[(long)b * nrho + i];
            const double g = best - q.lml;                      // >= 0 (> 0 before bi: first strictly larger wins)
            gap = fmin(gap, g);
            rdec = fmin(rdec, g / (ROUNDING * (t.noise + q.noise)));
        }
        o.gap = gap;
        o.rho_decision = rdec;
    }
    out[b] = o;
}

}  // namespace

// (the trace is built for the default arithmetic; the "exact" test form runs without it and reports no decision distance)
template <int C>
static void launch_c(hipStream_t st, const NullFitArgs& a, int variants) {
    if (a.exact) hipLaunchKernelGGL((nullfit_kernel<C, true, false>), dim3(variants, a.nrho), dim3(64), 0, st, a);
    else if (a.track) hipLaunchKernelGGL((nullfit_kernel<C, false, true>), dim3(variants, a.nrho), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((nullfit_kernel<C, false, false>), dim3(variants, a.nrho), dim3(64), 0, st, a);
}

int launch_nullfit(hipStream_t st, const NullFitArgs& a, int variants, bool force_wide, unsigned* queue) {
    if (variants <= 0) return CRM_OK;
    if (a.nrho < 1 || a.nrho > CRM_MAX_RHO) {
        set_error("null fit: %d grid points (supported 1..%d)", a.nrho, CRM_MAX_RHO);
        return CRM_ERR_UNSUPPORTED;
    }
    // poison the trial records: select_rho_kernel recognises a fit that never ran (nfev = -1)
    CRM_HIP(hipMemsetAsync(a.trial, 0xFF, sizeof(NullFitTrial) * (size_t)variants * a.nrho, st));
    int rmax = 1;
    for (int i = 0; i < a.nrho; i++) rmax = std::max(rmax, a.rho[i].r);
    const int sld = (rmax + 63) / 64 * 64;
    const size_t shared_lds = sizeof(double) * 3 * (size_t)sld;
    // (one covariate column -- the reference's default W = ones -- enough variants to keep 256 x 12 wavefronts busy, and
    // the three vectors of the longest spectrum within LDS)
    if (queue && !force_wide && a.c == 1 && variants >= 1024 && shared_lds <= 144 * 1024 && !form("nullfit_per_wave", 0)) {
        int cus = 256, dev = 0;
        CRM_HIP(hipGetDevice(&dev));
        CRM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        CRM_HIP(hipMemsetAsync(queue, 0, sizeof(unsigned) * CRM_MAX_RHO, st));
        // four fits per wavefront (nullfit_fit: G) unless the derivative polish or the "exact" arithmetic is asked for, or
        // the form "nullfit_one_per_wave" says so (the suite holds the two forms against each other: the same bits)
        // ... and only for spectra up to 2 048 entries: the rows of a wavefront search on their own -- different phases, other
        // numbers of evaluations -- so the pass over the spectrum, which the four fits do not share, pays for the divergence
        // what the shared scalar part saves.  Measured per 4096 x 11 fits: r = 1 020 (BASELINE config 2) 1.62 -> 1.15 ms,
        // r = 5 000 (config 3) 4.34 -> 5.50 ms; the two lines cross near r = 2 200.
        const bool four = !a.exact && !a.polish && rmax <= 2048 && !form("nullfit_one_per_wave", 0);
#define CRM_NF_SHARED(EXv, TRv, Gv)                                                                                        \
    do {                                                                                                                   \
        const void* fn = reinterpret_cast<const void*>(&nullfit_shared_kernel<1, EXv, TRv, Gv>);                           \
        if (shared_lds > 60 * 1024)                                                                                        \
            CRM_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shared_lds));                 \
        hipLaunchKernelGGL((nullfit_shared_kernel<1, EXv, TRv, Gv>), dim3(cus), dim3(64 * nf_shared_waves(Gv)), shared_lds, st, a, \
                           variants, sld, queue);                                                                          \
    } while (0)
        if (a.exact) CRM_NF_SHARED(true, false, 1);
        else if (a.track && four) CRM_NF_SHARED(false, true, 4);
        else if (a.track) CRM_NF_SHARED(false, true, 1);
        else if (four) CRM_NF_SHARED(false, false, 4);
        else CRM_NF_SHARED(false, false, 1);
#undef CRM_NF_SHARED
    } else if (a.c > CRM_MAX_COV_WIDE) {
        CRM_TRY(launch_nullfit_xwide(st, a, variants));
    } else if (force_wide || a.c > CRM_MAX_COV) {
        CRM_TRY(launch_nullfit_wide(st, a, variants));
    } else
    switch (a.c) {
        case 1: launch_c<1>(st, a, variants); break;
        case 2: launch_c<2>(st, a, variants); break;
        case 3: launch_c<3>(st, a, variants); break;
        case 4: launch_c<4>(st, a, variants); break;
        case 5: launch_c<5>(st, a, variants); break;
        case 6: launch_c<6>(st, a, variants); break;
        case 7: launch_c<7>(st, a, variants); break;
        case 8: launch_c<8>(st, a, variants); break;
        default:
            set_error("null fit: %d covariate columns (supported 1..%d)", a.c, CRM_MAX_COV);
            return CRM_ERR_UNSUPPORTED;
    }
    CRM_HIP(hipGetLastError());
    hipLaunchKernelGGL(select_rho_kernel, dim3((variants + 127) / 128), dim3(128), 0, st, a.trial, a.nrho,
                       variants, a.out);
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
