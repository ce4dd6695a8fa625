// Null fits with 63 .. 128 fixed-effect columns: the slower, correct path behind nullfit_wide.hip's 62-column limit.
//
// Who needs it: the association wrappers bind the cellular contexts to the fixed-effect slot
// (cellregmap/_cellregmap.py:498, :529: run_association(y, W, E, G) calls CellRegMap(y, W, E, ...) positionally), so a
// cohort with 100 contexts fits LMMs with 100 covariate columns there; the effect-size estimators fit [W, g, E0]
// (:175, :223).  The reference accepts any width.
//
// Same model, same search (bracket + Brent, rtol = atol = 1e-6, statement for statement as nullfit.hip / oracle/brent.py),
// same weighted-Gram evaluation as nullfit_wide.hip -- one 256-thread workgroup per (variant, rho), 16 x 16 threads with
// TS x TS register tiles over sqrt(w)-scaled rows staged in LDS -- but with what no longer fits LDS kept elsewhere:
//   * the (c+1) x (c+1) system / Cholesky factor in PACKED lower-triangular storage (129 * 130 / 2 doubles = 67 KB),
//   * the complement numerators u'v - t_u't_v (KT x KT, constant over the search) in global memory (NullFitArgs::xwide,
//     KT * KT doubles per workgroup; they stay in L2),
//   * the weighted Gram itself nowhere: every thread adds its tile to the packed system (and to the right-hand side /
//     y'K^-1y) straight from its accumulators,
//   * 32 spectrum entries per staging step instead of 64.
#include "nullfit.h"
#include "brent_search.h"

namespace crm {

namespace {

constexpr double LOG2PI = 1.8378770664093453;
constexpr double EPS_TINY = 2.220446049250313e-16;
constexpr double EPS_SMALL = 1.4901161193847656e-08;
constexpr int CHX = 32;         // spectrum entries per staging step
constexpr int XKT_MAX = 130;    // c + 2 <= 130

__device__ inline double logistic_clamped_x(double x) {
    double v;
    if (x > 0.0) {
        v = 1.0 / (1.0 + exp(-x));
    } else {
        v = exp(x);
        v = v / (v + 1.0);
    }
    return fmin(fmax(v, EPS_TINY), 1.0 - EPS_TINY);
}

__device__ inline int tri(int i, int k) { return i * (i + 1) / 2 + k; }   // k <= i

// In-place Cholesky of the leading P x P block in packed lower storage, all threads.  false on a non-positive pivot.
__device__ bool packed_cholesky(double* H, int P, double* scal, double& logdet) {
    const int tid = threadIdx.x;
    logdet = 0.0;
    for (int j = 0; j < P; j++) {
        __syncthreads();
        if (tid == 0) {
            double d = H[tri(j, j)];
            for (int k = 0; k < j; k++) d -= H[tri(j, k)] * H[tri(j, k)];
            scal[0] = d;
        }
        __syncthreads();
        const double d = scal[0];
        if (!(d > 0.0)) return false;
        const double l = sqrt(d);
        logdet += 2.0 * log(l);
        for (int i = j + 1 + tid; i < P; i += blockDim.x) {
            double s = H[tri(i, j)];
            for (int k = 0; k < j; k++) s -= H[tri(i, k)] * H[tri(j, k)];
            H[tri(i, j)] = s / l;
        }
        if (tid == 0) H[tri(j, j)] = l;
        __syncthreads();
    }
    return true;
}

template <int TS>
__global__ __launch_bounds__(256) void nullfit_xwide_kernel(NullFitArgs a) {
    extern __shared__ __align__(16) double xsm[];
    const int b = blockIdx.x;
    const int w = blockIdx.y;
    const NullFitRho R = a.rho[w];
    const int c = a.c;
    const int P = c + 1, KT = c + 2;
    const int r = R.r;
    const double n = (double)a.n;
    const int tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const double* __restrict__ tg = R.T + (long)b * R.ldT;
    // LDS: S [16 TS][CHX + 1], sd [CHX], Hp [P (P + 1) / 2], rhs [KT], red [256], scal [8]
    double* const S = xsm;
    double* const sd = S + 16 * TS * (CHX + 1);
    double* const Hp = sd + CHX;
    double* const rhs = Hp + (size_t)P * (P + 1) / 2;
    double* const red = rhs + KT;
    double* const scal = red + 256;
    double* const Cp = a.xwide + ((size_t)b * a.nrho + w) * (size_t)KT * KT;   // [KT x KT] complement numerators

    auto row_value = [&](int row, int j) -> double {
        if (row < c) return R.tW[(long)row * R.ldW + j];
        if (row == c) return tg[j];
        return R.ty[j];
    };
    // weighted Gram over the spectrum in register tiles; `sink(row, col, value)` receives every entry of the KT x KT
    // result (row, col < KT) from the thread that holds it; returns sum_j log D_j in lsum
    auto gram_pass = [&](double delta, bool weighted, double& lsum, auto&& sink) {
        double acc[TS][TS];
#pragma unroll
        for (int i = 0; i < TS; i++)
#pragma unroll
            for (int j = 0; j < TS; j++) acc[i][j] = 0.0;
        double lpart = 0.0;
        const double omd = 1.0 - delta;
        for (int c0 = 0; c0 < r; c0 += CHX) {
            if (tid < CHX) {
                const int j = c0 + tid;
                double v = 0.0;
                if (j < r) {
                    if (weighted) {
                        const double D = omd * R.S0[j] + delta;
                        lpart += log(D);
                        v = sqrt(1.0 / D);
                    } else {
                        v = 1.0;
                    }
                }
                sd[tid] = v;
            }
            __syncthreads();
            for (int e = tid; e < 16 * TS * CHX; e += 256) {
                const int row = e / CHX, cc = e - row * CHX;
                const int j = c0 + cc;
                double v = 0.0;
                if (row < KT && j < r) v = row_value(row, j) * sd[cc];
                S[row * (CHX + 1) + cc] = v;
            }
            __syncthreads();
#pragma unroll 2
            for (int cc = 0; cc < CHX; cc++) {
                double x[TS], y[TS];
#pragma unroll
                for (int i = 0; i < TS; i++) {
                    x[i] = S[(ti + 16 * i) * (CHX + 1) + cc];
                    y[i] = S[(tj + 16 * i) * (CHX + 1) + cc];
                }
#pragma unroll
                for (int i = 0; i < TS; i++)
#pragma unroll
                    for (int j = 0; j < TS; j++) acc[i][j] += x[i] * y[j];
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TS; i++) {
            const int row = ti + 16 * i;
#pragma unroll
            for (int j = 0; j < TS; j++) {
                const int col = tj + 16 * j;
                if (row < KT && col < KT) sink(row, col, acc[i][j]);
            }
        }
        red[tid] = lpart;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < CHX; i++) s += red[i];
            scal[1] = s;
        }
        __syncthreads();
        lsum = scal[1];
    };
    // (written by this workgroup's threads, read by others of it: past the CU's vector L1)
    auto cp_at = [&](size_t e) -> double { return __hip_atomic_load(Cp + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto plain = [&](int i, int j) -> double {   // u'v for u, v in {W.., g, y}
        if (i > j) { const int t = i; i = j; j = t; }
        if (j < c) return a.WW[i * c + j];
        if (j == c) return i < c ? a.gW[(long)b * a.ld_gW + i] : a.gg[b];
        return i < c ? a.Wy[i] : (i == c ? a.gy[b] : a.yy);
    };

    // rank of [W, g] and log|X'X| from the Cholesky of the plain Gram
    for (int e = tid; e < P * (P + 1) / 2; e += 256) {
        int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while (tri(i + 1, 0) <= e) i++;
        while (tri(i, 0) > e) i--;
        Hp[e] = plain(i, e - tri(i, 0));
    }
    __syncthreads();
    bool use_g = true;
    double logdetXX = 0.0;
    {
        const bool flagged = a.g_drop && a.g_drop[b] != 0;
        bool ok = packed_cholesky(Hp, c, scal, logdetXX);   // the covariates' block
        if (ok && !flagged) {
            // the variant's row by hand: relative floor on its pivot when no flag came with the block (legacy rule)
            if (tid == 0) {
                for (int k = 0; k < c; k++) {
                    double s = Hp[tri(c, k)];
                    for (int q = 0; q < k; q++) s -= Hp[tri(c, q)] * Hp[tri(k, q)];
                    Hp[tri(c, k)] = s / Hp[tri(k, k)];
                }
                double d = Hp[tri(c, c)];
                const double d0 = d;
                for (int k = 0; k < c; k++) d -= Hp[tri(c, k)] * Hp[tri(c, k)];
                scal[3] = d;
                scal[4] = d0;
            }
            __syncthreads();
            const double d = scal[3];
            if (a.g_drop ? !(d > 0.0) : !(d > 1e-12 * scal[4])) {
                if (a.g_drop) ok = false;
                else use_g = false;
            } else {
                logdetXX += log(d);
            }
        } else if (flagged) {
            use_g = false;
        }
        if (!ok) logdetXX = NAN;
        __syncthreads();
    }
    // complement numerators: plain inner products minus the unweighted Gram, into global memory
    {
        double dummy;
        gram_pass(1.0, false, dummy, [&](int row, int col, double v) { Cp[(size_t)row * KT + col] = plain(row, col) - v; });
        __threadfence();
        __syncthreads();
    }
    const double p_eff = use_g ? (double)P : (double)c;
    const double df = a.restricted ? n - p_eff : n;

    double cur_delta = 0.5, cur_scale = 1.0, cur_lml = -INFINITY;
    int nfev = 0;
    double memo_f[2] = {0.0, 0.0}, memo_scale[2] = {0.0, 0.0}, memo_lml[2] = {0.0, 0.0}, memo_noise[2] = {NAN, NAN};
    bool memo_set[2] = {false, false};
    bool last_clamped = false, want_noise = false;   // (as in nullfit.hip)
    double cur_noise = NAN;
    double* const diagS = S;          // (the tile buffer is free between two passes: diagonal of the spectrum Gram, KT)
    auto f = [&](double x) -> double {
        nfev++;
        const double delta = logistic_clamped_x(x);
        const int clamp = delta == 1.0 - EPS_TINY ? 1 : (delta == EPS_TINY ? 0 : -1);
        last_clamped = clamp >= 0;
        if (clamp >= 0 && memo_set[clamp]) {
            cur_delta = delta; cur_scale = memo_scale[clamp]; cur_lml = memo_lml[clamp]; cur_noise = memo_noise[clamp];
            return memo_f[clamp];
        }
        auto remember = [&](double value) -> double {
            if (clamp >= 0) {
                memo_set[clamp] = true; memo_f[clamp] = value; memo_scale[clamp] = cur_scale; memo_lml[clamp] = cur_lml;
                memo_noise[clamp] = cur_noise;
            }
            return value;
        };
        const bool noise_now = a.track && (want_noise || clamp >= 0);
        const double inv_d = 1.0 / delta;
        double lsum;
        gram_pass(delta, true, lsum, [&](int row, int col, double v) {
            const double k = v + cp_at((size_t)row * KT + col) * inv_d;   // u' Kt^-1 v
            if (row < P && col <= row) {
                double h = k;
                if (!use_g && (row == c || col == c)) h = (row == col) ? 1.0 : 0.0;
                Hp[tri(row, col)] = h;
            }
            if (col == c + 1 && row < P) rhs[row] = (!use_g && row == c) ? 0.0 : k;
            if (row == c + 1 && col == c + 1) scal[5] = k;                // y' Kt^-1 y
            if (noise_now && row == col) diagS[row] = v;
        });
        __syncthreads();
        const double logdetK = lsum + (n - (double)r) * log(delta);
        double logdetH;
        const bool ok = packed_cholesky(Hp, P, scal, logdetH);
        if (!ok) {
            cur_delta = delta; cur_scale = NAN; cur_lml = NAN;
            __syncthreads();
            return remember(INFINITY);
        }
        if (tid == 0) {
            // rss = y'Ky - z'z with L z = b  (one forward substitution)
            double rss = scal[5];
            for (int i = 0; i < P; i++) {
                double s = rhs[i];
                for (int k = 0; k < i; k++) s -= Hp[tri(i, k)] * red[k];
                s /= Hp[tri(i, i)];
                red[i] = s;
                rss -= s * s;
            }
            scal[2] = rss;
            if (noise_now) {
                // the noise bound of nullfit.hip: beta by the backward substitution, then the magnitudes of the terms of
                // rss = b' K b, b = (-beta, 1), by Cauchy-Schwarz on both parts of every entry: (sum_u |b_u| sqrt(m_u))^2
                double* const beta = diagS + KT;
                for (int i = P - 1; i >= 0; i--) {
                    double t = red[i];
                    for (int k = i + 1; k < P; k++) t -= Hp[tri(k, i)] * beta[k];
                    beta[i] = t / Hp[tri(i, i)];
                }
                double root = 0.0;
                for (int u = 0; u <= P; u++) {
                    const int ui = u < P ? u : c + 1;
                    if (u < P && !use_g && u == c) continue;
                    const double bu = u < P ? fabs(beta[u]) : 1.0;
                    root += bu * sqrt(fabs(diagS[ui]) + 2.0 * fabs(plain(ui, ui)) * inv_d);
                }
                scal[3] = root * root;
            }
        }
        __syncthreads();
        const double rss = scal[2];
        const double s = fmax(rss / df, EPS_SMALL);
        double val = -0.5 * (df * LOG2PI + df + n * log(s) + logdetK);
        if (a.restricted) val += 0.5 * (logdetXX - (logdetH - p_eff * log(s)));
        cur_delta = delta; cur_scale = s; cur_lml = val;
        if (noise_now)
            cur_noise = 0.5 * (df * scal[3] / fabs(rss) + fabs(lsum) + fabs((n - (double)r) * log(delta)) + n * fabs(log(s))
                               + df * (LOG2PI + 1.0) + fabs(logdetXX) + fabs(logdetH) + p_eff * fabs(log(s)));
        __syncthreads();
        return remember(-val);
    };

    // ---- bracket + Brent localmin: the search shared with nullfit.hip (brent_search.h) ---------------------------
    struct Objective {
        decltype(f)& fn;
        const bool& at_clamp;
        __device__ inline double operator()(double x) { return fn(x); }
        __device__ inline bool clamped() const { return at_clamp; }
    } objective{f, last_clamped};
    BrentTrace trace;
    double bf0;
    const double bx0 = a.track ? brent_search<true>(objective, trace, bf0) : brent_search<false>(objective, trace, bf0);
    double f_up = NAN, f_dn = NAN;   // (as in nullfit.hip: the objective one stopping tolerance to either side)
    if (a.track) {
        const double tolx = 1e-6 * fabs(bx0) + 1e-6;
        f_up = f(bx0 + tolx);
        f_dn = f(bx0 - tolx);
    }
    want_noise = true;
    const double f_stop = f(bx0);
    if (tid == 0) {
        NullFitTrial t;
        t.lml = cur_lml;
        t.delta = cur_delta;
        t.scale = cur_scale;
        t.use_g = use_g ? 1 : 0;
        t.nfev = nfev;
        t.margin = a.track ? fmin(trace.cmp, trace.sign) : NAN;
        t.curv = a.track ? 0.5 * (f_up + f_dn) - f_stop : NAN;
        t.noise = a.track ? cur_noise : NAN;
        a.trial[(long)b * a.nrho + w] = t;
    }
}

}  // namespace

size_t nullfit_xwide_scratch_doubles(int variants, int nrho, int c) { return (size_t)variants * nrho * (c + 2) * (c + 2); }

int launch_nullfit_xwide(hipStream_t st, const NullFitArgs& a, int variants) {
    const int KT = a.c + 2, P = a.c + 1;
    if (KT > XKT_MAX) {
        set_error("null fit: %d covariate columns (supported up to %d)", a.c, XKT_MAX - 2);
        return CRM_ERR_UNSUPPORTED;
    }
    if (a.polish) {
        set_error("null fit: the derivative polish is only built for up to %d covariate columns", CRM_MAX_COV);
        return CRM_ERR_UNSUPPORTED;
    }
    if (!a.xwide) {
        set_error("null fit: %d covariate columns need the wide scratch buffer (NullFitArgs::xwide)", a.c);
        return CRM_ERR_INTERNAL;
    }
    const int ts = (KT + 15) / 16;
    const size_t lds = sizeof(double) * ((size_t)16 * ts * (CHX + 1) + CHX + (size_t)P * (P + 1) / 2 + KT + 256 + 8);
    dim3 grid(variants, a.nrho);
#define CRM_XWIDE(T)                                                                                 \
    do {                                                                                             \
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nullfit_xwide_kernel<T>),         \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL(nullfit_xwide_kernel<T>, grid, dim3(256), lds, st, a);                    \
    } while (0)
    if (ts <= 5) CRM_XWIDE(5);
    else if (ts == 6) CRM_XWIDE(6);
    else if (ts == 7) CRM_XWIDE(7);
    else if (ts == 8) CRM_XWIDE(8);
    else CRM_XWIDE(9);
#undef CRM_XWIDE
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
