// Null fits with many fixed-effect columns (c up to 62): one 256-thread workgroup per
// (variant, rho).  Needed by the association scans, where the reference's run_association binds the
// cellular contexts to the fixed-effect slot (cellregmap/_cellregmap.py:498, :529), i.e. c = k
// (10-50); also used by the interaction scan when W has more than CRM_MAX_COV columns.
//
// Each likelihood evaluation is a weighted Gram matrix over the spectrum,
//     G(d) = sum_j  w_j(d) t_j t_j',   t_j = (Q0'W_1 .. Q0'W_c, Q0'g, Q0'y)_j,  w_j = 1/((1-d) S0_j + d),
// accumulated by 16 x 16 threads in TS x TS register tiles from sqrt(w)-scaled rows staged in LDS
// (the same structure as assemble.hip's gram_ext), followed by a workgroup-parallel Cholesky of the
// (c+1) x (c+1) block and the closed-form beta / scale / log-likelihood.  The scalar search on top
// (bracket + Brent, rtol = atol = 1e-6) is the same statement sequence as nullfit.hip and
// oracle/brent.py, executed uniformly by all threads.
#include "nullfit.h"
#include "brent_search.h"

namespace crm {

namespace {

constexpr double LOG2PI = 1.8378770664093453;
constexpr double EPS_TINY = 2.220446049250313e-16;
constexpr double EPS_SMALL = 1.4901161193847656e-08;
constexpr int CHW = 64;        // spectrum entries per staging step
constexpr int KT_MAX = 64;     // c + 2 <= 64

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

struct WideShared {
    double S[KT_MAX][CHW + 1];   // sqrt(w)-scaled rows of the current chunk
    double sd[CHW];
    double Gm[KT_MAX * KT_MAX];  // weighted Gram of the current evaluation
    double Cp[KT_MAX * KT_MAX];  // u'v - t_u't_v (complement numerators)
    double H[KT_MAX * KT_MAX];   // (c+1)^2 system / Cholesky factor
    double rhs[KT_MAX];
    double red[256];
    double scal[8];
};

// In-place Cholesky (lower) of the leading P x P block of H (leading dimension KT_MAX), all threads.
// Returns false on a non-positive pivot.  logdet = 2 sum log L_ii.  `last_rel_floor` > 0 makes the
// LAST pivot fail softly when it is below floor * (its diagonal entry): the rank test on g.
__device__ bool block_cholesky(double* H, int P, double* scal, double& logdet, double last_rel_floor,
                               bool& last_dropped) {
    const int tid = threadIdx.x;
    last_dropped = false;
    logdet = 0.0;
    for (int j = 0; j < P; j++) {
        const double diag0 = H[j * KT_MAX + j];
        __syncthreads();
        if (tid == 0) {
            double d = diag0;
            // columns 0..j-1 of row j already hold L; subtract their squares
            for (int k = 0; k < j; k++) d -= H[j * KT_MAX + k] * H[j * KT_MAX + k];
            scal[0] = d;
        }
        __syncthreads();
        const double d = scal[0];
        if (j == P - 1 && last_rel_floor > 0.0 && !(d > last_rel_floor * diag0)) {
            last_dropped = true;
            return true;
        }
        if (!(d > 0.0)) return false;
        const double l = sqrt(d);
        logdet += 2.0 * log(l);
        // column j below the diagonal
        for (int i = j + 1 + tid; i < P; i += blockDim.x) {
            double s = H[i * KT_MAX + j];
            for (int k = 0; k < j; k++) s -= H[i * KT_MAX + k] * H[j * KT_MAX + k];
            H[i * KT_MAX + j] = s / l;
        }
        if (tid == 0) H[j * KT_MAX + j] = l;
        __syncthreads();
    }
    return true;
}

// x <- (L L')^-1 x for the leading P entries (thread 0; P <= 63, called once per evaluation)
__device__ void chol_solve_serial(const double* L, int P, double* x) {
    for (int i = 0; i < P; i++) {
        double s = x[i];
        for (int k = 0; k < i; k++) s -= L[i * KT_MAX + k] * x[k];
        x[i] = s / L[i * KT_MAX + i];
    }
    for (int i = P - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < P; k++) s -= L[k * KT_MAX + i] * x[k];
        x[i] = s / L[i * KT_MAX + i];
    }
}

template <int TS>
__global__ __launch_bounds__(256) void nullfit_wide_kernel(NullFitArgs a, int rho_base) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    WideShared& sh = *reinterpret_cast<WideShared*>(smem_raw);
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

    auto row_value = [&](int row, int j) -> double {
        if (row < c) return R.tW[(long)row * R.ldW + j];
        if (row == c) return tg[j];
        return R.ty[j];
    };
    // weighted Gram into sh.Gm; returns sum_j log D_j (weighted) in lsum
    auto gram_pass = [&](double delta, bool weighted, double& lsum) {
        double acc[TS][TS];
#pragma unroll
        for (int i = 0; i < TS; i++)
#pragma unroll
            for (int j = 0; j < TS; j++) acc[i][j] = 0.0;
        double lpart = 0.0;
        const double omd = 1.0 - delta;
        for (int c0 = 0; c0 < r; c0 += CHW) {
            if (tid < CHW) {
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
                sh.sd[tid] = v;
            }
            __syncthreads();
            for (int e = tid; e < 16 * TS * CHW; e += 256) {
                const int row = e / CHW, cc = e - row * CHW;
                const int j = c0 + cc;
                double v = 0.0;
                if (row < KT && j < r) v = row_value(row, j) * sh.sd[cc];
                sh.S[row][cc] = v;
            }
            __syncthreads();
#pragma unroll 4
            for (int cc = 0; cc < CHW; cc++) {
                double x[TS], y[TS];
#pragma unroll
                for (int i = 0; i < TS; i++) {
                    x[i] = sh.S[ti + 16 * i][cc];
                    y[i] = sh.S[tj + 16 * i][cc];
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
                if (row < KT && col < KT) sh.Gm[row * KT_MAX + col] = acc[i][j];
            }
        }
        sh.red[tid] = lpart;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < CHW; i++) s += sh.red[i];
            sh.scal[1] = s;
        }
        __syncthreads();
        lsum = sh.scal[1];
    };

    // plain inner products u'v (u, v in {W.., g, y}) -> sh.Cp, then subtract the unweighted Gram
    for (int e = tid; e < KT * KT; e += 256) {
        int i = e / KT, j = e - i * KT;
        if (i > j) { int t = i; i = j; j = t; }
        double v;
        if (j < c) v = a.WW[i * c + j];
        else if (j == c) v = i < c ? a.gW[(long)b * a.ld_gW + i] : a.gg[b];
        else v = i < c ? a.Wy[i] : (i == c ? a.gy[b] : a.yy);
        const int ii = e / KT, jj = e - ii * KT;
        sh.Cp[ii * KT_MAX + jj] = v;
    }
    __syncthreads();
    // rank test on g and log|X'X| from the Cholesky of the plain Gram
    bool use_g = true;
    double logdetXX = 0.0;
    {
        for (int e = tid; e < P * P; e += 256) {
            const int i = e / P, j = e - i * P;
            sh.H[i * KT_MAX + j] = sh.Cp[i * KT_MAX + j];
        }
        __syncthreads();
        bool dropped = false;
        // (g_drop: the reference's rule on the singular values of [W, g], decided where the block was orthogonalised
        // against W -- blockops.hip; the factorisation then stops before the variant's row)
        const bool flagged = a.g_drop && a.g_drop[b] != 0;
        const bool ok = block_cholesky(sh.H, flagged ? c : P, sh.scal, logdetXX, a.g_drop ? 0.0 : 1e-12, dropped);
        if (dropped || flagged) use_g = false;
        if (!ok) logdetXX = NAN;
        __syncthreads();
    }
    {
        double dummy;
        gram_pass(1.0, false, dummy);
        for (int e = tid; e < KT * KT; e += 256) {
            const int i = e / KT, j = e - i * KT;
            sh.Cp[i * KT_MAX + j] -= sh.Gm[i * KT_MAX + j];
        }
        __syncthreads();
    }
    const double p_eff = use_g ? (double)P : (double)c;
    const double df = a.restricted ? n - p_eff : n;

    double cur_delta = 0.5, cur_scale = 1.0, cur_lml = -INFINITY;
    int nfev = 0;
    // (the two clamped points delta = eps, 1 - eps are evaluated once and remembered: see nullfit.hip)
    double memo_f[2] = {0.0, 0.0}, memo_scale[2] = {0.0, 0.0}, memo_lml[2] = {0.0, 0.0}, memo_noise[2] = {NAN, NAN};
    bool memo_set[2] = {false, false};
    bool last_clamped = false, want_noise = false;   // (as in nullfit.hip)
    double cur_noise = NAN;
    auto f = [&](double x) -> double {
        nfev++;
        const double delta = logistic_clamped(x);
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
        double lsum;
        gram_pass(delta, true, lsum);
        const double inv_d = 1.0 / delta;
        const double logdetK = lsum + (n - (double)r) * log(delta);
        for (int e = tid; e < P * P; e += 256) {
            const int i = e / P, j = e - i * P;
            double v = sh.Gm[i * KT_MAX + j] + sh.Cp[i * KT_MAX + j] * inv_d;
            if (!use_g && (i == c || j == c)) v = (i == j) ? 1.0 : 0.0;
            sh.H[i * KT_MAX + j] = v;
        }
        for (int i = tid; i < P; i += 256) {
            double v = sh.Gm[i * KT_MAX + (c + 1)] + sh.Cp[i * KT_MAX + (c + 1)] * inv_d;
            if (!use_g && i == c) v = 0.0;
            sh.rhs[i] = v;
        }
        __syncthreads();
        double logdetH;
        bool dropped;
        const bool ok = block_cholesky(sh.H, P, sh.scal, logdetH, 0.0, dropped);
        if (!ok) {
            cur_delta = delta; cur_scale = NAN; cur_lml = NAN;
            __syncthreads();
            return remember(INFINITY);
        }
        if (tid == 0) {
            // rss = y'Ky - b' H^-1 b
            double xk[KT_MAX];
            for (int i = 0; i < P; i++) xk[i] = sh.rhs[i];
            chol_solve_serial(sh.H, P, xk);
            double rss = sh.Gm[(c + 1) * KT_MAX + (c + 1)] + sh.Cp[(c + 1) * KT_MAX + (c + 1)] * inv_d;
            for (int i = 0; i < P; i++) rss -= sh.rhs[i] * xk[i];
            sh.scal[2] = rss;
            if (a.track && (want_noise || clamp >= 0)) {
                // the noise bound of nullfit.hip: magnitudes of the terms of rss = b' K b, b = (-beta, 1)
                // (Gm: spectrum sums, Cp: complements u'v - t_u't_v, whose two parts are each at most sqrt(u'u v'v))
                xk[P] = 1.0;
                const int yi = c + 1;
                auto plain = [&](int i) -> double { return i < c ? a.WW[i * c + i] : (i == c ? a.gg[b] : a.yy); };
                double mag = 0.0;
                for (int u = 0; u <= P; u++) {
                    const int ui = u < P ? u : yi;
                    if (u < P && !use_g && u == c) continue;
                    for (int v = u; v <= P; v++) {
                        const int vi = v < P ? v : yi;
                        if (v < P && !use_g && v == c) continue;
                        const double spec = sqrt(fabs(sh.Gm[ui * KT_MAX + ui] * sh.Gm[vi * KT_MAX + vi]));
                        const double cpl = 2.0 * sqrt(fabs(plain(ui) * plain(vi)));
                        mag += (u == v ? 1.0 : 2.0) * fabs(xk[u]) * fabs(xk[v]) * (spec + cpl * inv_d);
                    }
                }
                sh.scal[3] = mag;
            }
        }
        __syncthreads();
        const double rss = sh.scal[2];
        const double s = fmax(rss / df, EPS_SMALL);
        double val = -0.5 * (df * LOG2PI + df + n * log(s) + logdetK);
        if (a.restricted) val += 0.5 * (logdetXX - (logdetH - p_eff * log(s)));
        cur_delta = delta; cur_scale = s; cur_lml = val;
        if (a.track && (want_noise || clamp >= 0))
            cur_noise = 0.5 * (df * sh.scal[3] / fabs(rss) + fabs(lsum) + fabs((n - (double)r) * log(delta)) + n * fabs(log(s))
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

int launch_nullfit_wide(hipStream_t st, const NullFitArgs& a, int variants) {
    const int KT = a.c + 2;
    if (KT > KT_MAX) {
        set_error("null fit: %d covariate columns (supported up to %d)", a.c, KT_MAX - 2);
        return CRM_ERR_UNSUPPORTED;
    }
    if (a.polish) {
        set_error("null fit: the derivative polish is only built for up to %d covariate columns", CRM_MAX_COV);
        return CRM_ERR_UNSUPPORTED;
    }
    const int ts = (KT + 15) / 16;
    const size_t lds = sizeof(WideShared);
    dim3 grid(variants, a.nrho);
#define CRM_WIDE(T)                                                                                  \
    do {                                                                                             \
        CRM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&nullfit_wide_kernel<T>),          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL(nullfit_wide_kernel<T>, grid, dim3(256), lds, st, a, 0);                  \
    } while (0)
    if (ts <= 1) CRM_WIDE(1);
    else if (ts == 2) CRM_WIDE(2);
    else if (ts == 3) CRM_WIDE(3);
    else CRM_WIDE(4);
#undef CRM_WIDE
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
