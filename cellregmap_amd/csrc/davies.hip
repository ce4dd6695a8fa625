// Mixture weights and p-value per variant: one wavefront per variant.
//
// Reference: chiscore.davies_pvalue(Q, F, True) at cellregmap/_cellregmap.py:333,435
// -> eigvalsh(F) (lower triangle), SKAT's eigenvalue filter, Davies' AS 155 algorithm
// (chi2comb, lim = 10000, acc = 1e-6), modified-Liu fall-back.  Same procedure as
// oracle/davies.py + oracle/qfc.c, re-organised for 64 lanes:
//   * eigenvalues: Householder tridiagonalisation in LDS (lanes over the rows of the trailing
//     block) followed by Sturm-sequence bisection, one eigenvalue index per lane;
//   * qfc: the scalar search logic (truncation point, cut-offs, step) runs wave-uniform;
//     every sum over the eigenvalues is lane-parallel + a wave total (wave_ops.h), and the
//     trapezoid rule puts one abscissa per lane.
#include "nullfit.h"
#include "wave_ops.h"

namespace crm {

namespace {

constexpr double PI = 3.14159265358979323846;
constexpr double LN28 = 0.08664339756999316;  // log(2)/8
constexpr int DAVIES_LIM = 10000;
constexpr double DAVIES_ACC = 1e-6;

__device__ inline double wsum(double v) { return wave_total(v); }
__device__ inline int wsum_i(int v) { return wave_total(v); }

__device__ inline double exp_guard(double x) { return x < -50.0 ? 0.0 : exp(x); }

// log(1+x) when first, else log(1+x) - x; series for small |x| (AS 155 "log1")
__device__ inline double log1p_variant(double x, bool first) {
    if (fabs(x) > 0.1) return first ? log(1.0 + x) : (log(1.0 + x) - x);
    double y = x / (2.0 + x);
    double term = 2.0 * y * y * y;
    double k = 3.0;
    double s = (first ? 2.0 : -x) * y;
    y = y * y;
    double s1 = s + term / k;
    // (|x| <= 0.1: the terms fall by 1 / 400 each, so this ends after ~8 rounds; the cap only keeps a NaN -- for which
    // s1 != s holds for ever -- from hanging the GPU)
    for (int it = 0; it < 64 && s1 != s; it++) {
        k += 2.0;
        term *= y;
        s = s1;
        s1 = s + term / k;
    }
    return s;
}

// ---- regularised upper incomplete gamma and non-central chi-square survival --------------
__device__ double igamc(double a, double x) {
    if (x <= 0.0) return 1.0;
    const double lg = lgamma(a);
    if (x < a + 1.0) {
        // P(a, x) by series
        double ap = a, sum = 1.0 / a, del = sum;
        for (int it = 0; it < 2000; it++) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (fabs(del) < fabs(sum) * 1e-17) break;
        }
        return 1.0 - sum * exp(-x + a * log(x) - lg);
    }
    // continued fraction (modified Lentz)
    const double tiny = 1e-300;
    double bq = x + 1.0 - a;
    double cq = 1.0 / tiny;
    double dq = 1.0 / bq;
    double h = dq;
    for (int i = 1; i < 5000; i++) {
        const double an = -i * (i - a);
        bq += 2.0;
        dq = an * dq + bq;
        if (fabs(dq) < tiny) dq = tiny;
        cq = bq + an / cq;
        if (fabs(cq) < tiny) cq = tiny;
        dq = 1.0 / dq;
        const double del = dq * cq;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return exp(-x + a * log(x) - lg) * h;
}

// P[ ncx2(dof, nc) > x ] as a Poisson mixture of central survival functions
__device__ double ncx2_sf(double x, double dof, double nc) {
    if (!(x > 0.0)) return 1.0;
    const double z = 0.5 * x, lam = 0.5 * nc, a0 = 0.5 * dof;
    double q = igamc(a0, z);                                // Q(a0 + i, z)
    double inc = exp(-z + a0 * log(z) - lgamma(a0 + 1.0));  // z^a e^-z / Gamma(a+1)
    double w = exp(-lam);                                   // Poisson weight
    double sum = 0.0;
    for (int i = 0; i < 5000; i++) {
        const double term = w * q;
        sum += term;
        if (i > lam && term <= sum * 1e-17) break;
        q += inc;
        inc *= z / (a0 + i + 1.0);
        w *= lam / (i + 1.0);
    }
    return sum < 1.0 ? sum : 1.0;
}

struct Qf {
    const double* lb;  // LDS, r positive weights in ascending order
    int r, lane;
    double sigsq, lmax, lmin, mean, c;
    double intl, ersm;
    int count, lim;
    bool fail, overflow;
};

__device__ inline void tick(Qf& q) {
    q.count++;
    if (q.count > q.lim) q.overflow = true;
}

// AS 155 errbd: bound on the tail probability via the mgf (dof 1, no non-centrality)
__device__ double tail_bound(Qf& q, double u, double& cx) {
    tick(q);
    double xconst = u * q.sigsq;
    double sum1 = u * xconst;
    u = 2.0 * u;
    double pc = 0.0, ps = 0.0;
    for (int j = q.lane; j < q.r; j += 64) {
        const double lj = q.lb[j];
        const double x = u * lj, y = 1.0 - x;
        pc += lj / y;
        ps += x * x / y + log1p_variant(-x, false);
    }
    xconst += wsum(pc);
    sum1 += wsum(ps);
    cx = xconst;
    return exp_guard(-0.5 * sum1);
}

// AS 155 ctff
__device__ double cutoff(Qf& q, double accx, double& upn) {
    double u2 = upn, u1 = 0.0, c1 = q.mean, c2 = 0.0, xconst;
    const double rb = 2.0 * ((u2 > 0.0) ? q.lmax : q.lmin);
    double u = u2 / (1.0 + u2 * rb);
    while (!q.overflow && tail_bound(q, u, c2) > accx) {
        u1 = u2;
        c1 = c2;
        u2 = 2.0 * u2;
        u = u2 / (1.0 + u2 * rb);
    }
    u = (c1 - q.mean) / (c2 - q.mean);
    while (!q.overflow && u < 0.9) {
        u = (u1 + u2) / 2.0;
        if (tail_bound(q, u / (1.0 + u * rb), xconst) > accx) {
            u1 = u;
            c1 = xconst;
        } else {
            u2 = u;
            c2 = xconst;
        }
        u = (c1 - q.mean) / (c2 - q.mean);
    }
    upn = u2;
    return c2;
}

// AS 155 truncation
__device__ double trunc_bound(Qf& q, double u, double tausq) {
    tick(q);
    const double sum2 = (q.sigsq + tausq) * u * u;
    double prod1 = 2.0 * sum2;
    u = 2.0 * u;
    double p1 = 0.0, p2 = 0.0, p3 = 0.0;
    int s = 0;
    for (int j = q.lane; j < q.r; j += 64) {
        const double lj = q.lb[j];
        const double x = (u * lj) * (u * lj);
        if (x > 1.0) {
            p2 += log(x);
            p3 += log1p_variant(x, true);
            s += 1;
        } else {
            p1 += log1p_variant(x, true);
        }
    }
    prod1 += wsum(p1);
    double prod2 = wsum(p2), prod3 = wsum(p3);
    s = wsum_i(s);
    const double sum1 = 0.0;  // non-centralities are zero on this path
    prod2 += prod1;
    prod3 += prod1;
    double x = exp_guard(-sum1 - 0.25 * prod2) / PI;
    const double y = exp_guard(-sum1 - 0.25 * prod3) / PI;
    double err1 = (s == 0) ? 1.0 : x * 2.0 / s;
    double err2 = (prod3 > 1.0) ? 2.5 * y : 1.0;
    if (err2 < err1) err1 = err2;
    x = 0.5 * sum2;
    err2 = (x <= y) ? 1.0 : y / x;
    return (err1 < err2) ? err1 : err2;
}

// AS 155 findu
__device__ void find_trunc_point(Qf& q, double& utx, double accx) {
    const double divis[4] = {2.0, 1.4, 1.2, 1.1};
    double ut = utx, u = ut / 4.0;
    if (trunc_bound(q, u, 0.0) > accx) {
        for (u = ut; !q.overflow && trunc_bound(q, u, 0.0) > accx; u = ut) ut *= 4.0;
    } else {
        ut = u;
        for (u = u / 4.0; !q.overflow && trunc_bound(q, u, 0.0) <= accx; u = u / 4.0) ut = u;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u = ut / divis[i];
        if (trunc_bound(q, u, 0.0) <= accx) ut = u;
    }
    utx = ut;
}

// AS 155 integrate: one abscissa per lane.  At an abscissa u the integrand needs  sum_j atan(2 lb_j u)  and
// sum_j log(1 + (2 lb_j u)^2) -- r arctangents and r logarithms, the bulk of the kernel's time at 50 weights.  Both are
// one complex number: prod_j (1 + i x_j) has the argument sum_j atan(x_j) and the modulus exp(sum_j log(1 + x_j^2) / 2).
// With positive weights (what SKAT's filter leaves: always, here) every factor turns the product anticlockwise by less than
// a quarter turn, so the argument is unwrapped by counting the crossings of the negative real axis; the product is kept
// within range by taking out powers of two.  r complex multiplications, one atan2 and one log per abscissa instead of r of
// each; the sums agree with the term-by-term ones to ~r roundings (absolute), far inside acc = 1e-6.  Weights of mixed
// sign (never produced by the score test) take the term-by-term loop.
__device__ void integrate(Qf& q, int nterm, double interv, double tausq, bool mainx) {
    const double inpi = interv / PI;
    double a1 = 0.0, a2 = 0.0;
    const bool positive = q.lmin == 0.0;
    for (int k = nterm - q.lane; k >= 0; k -= 64) {
        const double u = (k + 0.5) * interv;
        double sum1 = -2.0 * u * q.c;
        double sum2 = fabs(sum1);
        double sum3 = -0.5 * q.sigsq * u * u;
        if (positive) {
            double re = 1.0, im = 0.0;
            int turns = 0, expo = 0;
            const double u2 = 2.0 * u;
            for (int j = q.r - 1; j >= 0; j--) {
                const double x = q.lb[j] * u2;
                const double nre = fma(-im, x, re), nim = fma(re, x, im);
                // anticlockwise by less than a quarter turn: the negative real axis is crossed where im goes from >= 0 to < 0
                turns += (im >= 0.0 && nim < 0.0) ? 1 : 0;
                re = nre; im = nim;
                if ((j & 3) == 0) {   // keep the modulus near one: (1 + x^2)^(r/2) leaves the range of a double for large u
                    const int e = __builtin_amdgcn_frexp_exp(fmax(fabs(re), fabs(im)));
                    re = ldexp(re, -e); im = ldexp(im, -e);
                    expo += e;
                }
            }
            const double theta = atan2(im, re) + 2.0 * PI * (double)turns;        // sum_j atan(x_j)
            const double logmod2 = log(re * re + im * im) + 2.0 * 0.6931471805599453 * (double)expo;   // sum_j log(1 + x_j^2)
            sum1 += theta;
            sum2 += theta;                // (every term is positive)
            sum3 -= 0.25 * logmod2;
        } else {
            for (int j = q.r - 1; j >= 0; j--) {
                const double x = 2.0 * q.lb[j] * u;
                const double y = x * x;
                sum3 -= 0.25 * log1p_variant(y, true);
                const double z = atan(x);
                sum1 += z;
                sum2 += fabs(z);
            }
        }
        double x = inpi * exp_guard(sum3) / u;
        if (!mainx) x *= (1.0 - exp_guard(-0.5 * tausq * u * u));
        a1 += sin(0.5 * sum1) * x;
        a2 += 0.5 * sum2 * x;
    }
    q.intl += wsum(a1);
    q.ersm += wsum(a2);
}

// AS 155 cfe.  Weights are positive and ascending, so the |lb|-descending order th[]
// of the original is simply index r-1-k.
__device__ double conv_coef(Qf& q, double x) {
    tick(q);
    double axl = fabs(x);
    const double sxl = (x > 0.0) ? 1.0 : -1.0;
    double sum1 = 0.0;
    for (int j = q.r - 1; j >= 0; j--) {
        const int t = q.r - 1 - j;
        if (q.lb[t] * sxl > 0.0) {
            const double lj = fabs(q.lb[t]);
            const double axl1 = axl - lj;
            const double axl2 = lj / LN28;
            if (axl1 > axl2) {
                axl = axl1;
            } else {
                if (axl > axl2) axl = axl2;
                sum1 = (axl - axl1) / lj;
                sum1 += (double)j;  // one unit per remaining weight
                break;
            }
        }
    }
    if (sum1 > 100.0) {
        q.fail = true;
        return 1.0;
    }
    return pow(2.0, sum1 / 4.0) / (PI * axl * axl);
}

// P[ sum lb_j chi2_1 < c ];  returns the cdf (or -1 when the search gave up)
__device__ double qfc_wave(const double* lb, int r, double c, int lane, int& ifault) {
    Qf q;
    q.lb = lb; q.r = r; q.lane = lane; q.c = c;
    q.sigsq = 0.0; q.intl = 0.0; q.ersm = 0.0; q.count = 0; q.lim = DAVIES_LIM;
    q.fail = false; q.overflow = false;
    ifault = 0;
    double acc1 = DAVIES_ACC, xlim = (double)DAVIES_LIM;
    double sd = 0.0, mean = 0.0;
    for (int j = lane; j < r; j += 64) {
        sd += lb[j] * lb[j] * 2.0;
        mean += lb[j];
    }
    sd = wsum(sd);
    q.mean = wsum(mean);
    q.lmax = lb[r - 1] > 0.0 ? lb[r - 1] : 0.0;  // ascending order
    q.lmin = lb[0] < 0.0 ? lb[0] : 0.0;
    if (sd == 0.0) return (c > 0.0) ? 1.0 : 0.0;
    if (q.lmin == 0.0 && q.lmax == 0.0) {
        ifault = 3;
        return -1.0;
    }
    sd = sqrt(sd);
    const double almx = (q.lmax < -q.lmin) ? -q.lmin : q.lmax;
    double utx = 16.0 / sd, up = 4.5 / sd, un = -up, tausq, intv = 0.0, xnt = 0.0, xntm;
    find_trunc_point(q, utx, 0.5 * acc1);
    if (q.overflow) { ifault = 4; return -1.0; }
    if (c != 0.0 && almx > 0.07 * sd) {
        tausq = 0.25 * acc1 / conv_coef(q, c);
        if (q.fail) {
            q.fail = false;
        } else if (trunc_bound(q, utx, tausq) < 0.2 * acc1) {
            q.sigsq += tausq;
            find_trunc_point(q, utx, 0.25 * acc1);
        }
        if (q.overflow) { ifault = 4; return -1.0; }
    }
    acc1 *= 0.5;
    for (;;) {
        const double d1 = cutoff(q, acc1, up) - c;
        if (q.overflow) { ifault = 4; return -1.0; }
        if (d1 < 0.0) return 1.0;
        const double d2 = c - cutoff(q, acc1, un);
        if (q.overflow) { ifault = 4; return -1.0; }
        if (d2 < 0.0) return 0.0;
        intv = 2.0 * PI / ((d1 > d2) ? d1 : d2);
        xnt = utx / intv;
        xntm = 3.0 / sqrt(acc1);
        if (xnt <= xntm * 1.5) break;
        if (xntm > xlim) { ifault = 1; return -1.0; }
        const int ntm = (int)floor(xntm + 0.5);
        const double intv1 = utx / ntm;
        const double x = 2.0 * PI / intv1;
        if (x <= fabs(c)) break;
        tausq = 0.33 * acc1 / (1.1 * (conv_coef(q, c - x) + conv_coef(q, c + x)));
        if (q.overflow) { ifault = 4; return -1.0; }
        if (q.fail) break;
        acc1 *= 0.67;
        integrate(q, ntm, intv1, tausq, false);
        xlim -= xntm;
        q.sigsq += tausq;
        find_trunc_point(q, utx, 0.25 * acc1);
        if (q.overflow) { ifault = 4; return -1.0; }
        acc1 *= 0.75;
    }
    if (xnt > xlim) { ifault = 1; return -1.0; }
    const int nt = (int)floor(xnt + 0.5);
    integrate(q, nt, intv, 0.0, true);
    const double qfval = 0.5 - q.intl;
    const double upv = q.ersm, x = upv + DAVIES_ACC / 10.0;
    const int rats[4] = {1, 2, 4, 8};
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (rats[j] * x == rats[j] * upv) ifault = 2;
    return qfval;
}

// modified Liu (Lee, Wu & Lin 2012) survival at t for weights lb (dof 1, nc 0)
__device__ double liu_mod_sf(const double* lb, int r, double t, int lane) {
    double c1 = 0.0, c2 = 0.0, c3 = 0.0, c4 = 0.0;
    for (int j = lane; j < r; j += 64) {
        const double l = lb[j], l2 = l * l;
        c1 += l;
        c2 += l2;
        c3 += l2 * l;
        c4 += l2 * l2;
    }
    c1 = wsum(c1); c2 = wsum(c2); c3 = wsum(c3); c4 = wsum(c4);
    const double sq = sqrt(c2);
    const double s1 = c3 / (sq * sq * sq);
    const double s2 = c4 / (c2 * c2);
    const double s12 = s1 * s1;
    double a, delta_x, dof_x;
    if (s12 > s2) {
        a = 1.0 / (s1 - sqrt(s12 - s2));
        delta_x = s1 * a * a * a - a * a;
        dof_x = a * a - 2.0 * delta_x;
    } else {
        delta_x = 0.0;
        a = 1.0 / sqrt(s2);
        dof_x = 1.0 / s2;
    }
    const double mu_q = c1, sigma_q = sqrt(2.0 * c2);
    const double mu_x = dof_x + delta_x, sigma_x = sqrt(2.0 * (dof_x + 2.0 * delta_x));
    const double t_star = (t - mu_q) / sigma_q;
    const double tfinal = t_star * sigma_x + mu_x;
    return ncx2_sf(tfinal, dof_x, delta_x > 1e-9 ? delta_x : 1e-9);
}

// ---- eigenvalues -----------------------------------------------------------------------------
__device__ inline double wave_min(double v) {
    v = fmin(v, __shfl_xor(v, 32, 64));
    v = fmin(v, lane_xor_swizzle(v, 16)); v = fmin(v, lane_xor_swizzle(v, 8)); v = fmin(v, lane_xor_swizzle(v, 4));
    v = fmin(v, lane_xor_quad(v, 2)); v = fmin(v, lane_xor_quad(v, 1));
    return v;
}

// Eigenvalues of the symmetric tridiagonal matrix (dd [k], ee [k - 1], both in LDS and overwritten) by Sturm bisection
// from the Gershgorin interval, one eigenvalue index per lane (LAPACK dstebz's procedure and stopping rule), ascending
// into ev.  The count of eigenvalues below a point is taken from the signs of the leading principal minors
//     p_0 = 1,  p_1 = d_1 - x,  p_i = (d_i - x) p_(i-1) - e_(i-1)^2 p_(i-2)
// rather than from their quotients q_i = p_i / p_(i-1) (dstebz's form, q_i = d_i - x - e^2 / q_(i-1)): the quotient form
// is a chain of k dependent divisions per count, ~55 counts per eigenvalue -- it was half of this kernel's time at 50
// contexts -- where the minors cost one dependent multiply-add each.  The two count the same sign changes; what the
// quotient form buys, safety from over- and underflow, is supplied here by (1) scaling the matrix by a power of two to
// |T| in [1/2, 1) (exact), so a pair of consecutive minors grows by at most 3^8 over eight steps, and taking out its
// exponent every eight steps; (2) a floor of 2^-110 on the scaled e^2 (a perturbation of 2^-55 |T| of an off-diagonal
// entry: an eighth of a rounding), so that a step shrinks the pair by no more than e^2 / 4 -- the step is the matrix
// [[d - x, -e^2], [1, 0]], of determinant e^2 and norm below 4 -- eight steps by no more than 2^-896, and a minor that is
// exactly zero is followed by one that is not.
__device__ void sturm_bisection(double* dd, double* ee, double* ev, int k, int lane, bool bad) {
    double gl = INFINITY, gu = -INFINITY;
    for (int i = lane; i < k; i += 64) {
        const double lo = (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i < k - 1 ? fabs(ee[i]) : 0.0);
        gl = fmin(gl, dd[i] - lo);
        gu = fmax(gu, dd[i] + lo);
    }
    gl = wave_min(gl);
    gu = -wave_min(-gu);
    const double eps = 2.220446049250313e-16;
    const double tnorm = fmax(fabs(gl), fabs(gu));
    __syncthreads();
    if (bad) return;
    if (!(tnorm > 0.0) || !(tnorm < INFINITY)) {    // the zero matrix (or an overflow on the way here)
        for (int i = lane; i < k; i += 64) ev[i] = tnorm == 0.0 ? 0.0 : NAN;
        __syncthreads();
        return;
    }
    const int E = __builtin_amdgcn_frexp_exp(tnorm);
    for (int i = lane; i < k; i += 64) {
        dd[i] = ldexp(dd[i], -E);
        if (i < k - 1) {
            const double e = ldexp(ee[i], -E);
            ee[i] = fmax(e * e, 0x1p-110);
        }
    }
    __syncthreads();
    const double tn = ldexp(tnorm, -E);
    const double slop = 2.1 * tn * eps * k;
    const double lo0 = ldexp(gl, -E) - slop, hi0 = ldexp(gu, -E) + slop;
    const double atol = 2.0 * eps * tn;
    for (int idx = lane; idx < k; idx += 64) {
        double lo = lo0, hi = hi0;
        for (int it = 0; it < 200; it++) {
            const double mid = 0.5 * (lo + hi);
            if (hi - lo <= fmax(atol, 2.0 * eps * fmax(fabs(lo), fabs(hi))) || mid == lo || mid == hi) break;
            // (the signs of the minors go through a shift register, one v_alignbit per step; the sign changes of eight
            // steps are counted at once.  An exactly zero minor needs no rule of its own: p_i = 0 makes p_(i+1) =
            // -e^2 p_(i-1), so whichever sign the zero is read with, the three of them show exactly one change.)
            double p2 = 1.0, p1 = dd[0] - mid;
            unsigned signs = __builtin_amdgcn_alignbit(0u, (unsigned)__double2hiint(p1), 31);   // [.. 0, sign(p_1)]
            int cnt = (int)(signs & 1u);
            auto step = [&](double d, double e2) {
                const double t = e2 * p2;
                const double pn = fma(d - mid, p1, -t);
                signs = __builtin_amdgcn_alignbit(signs, (unsigned)__double2hiint(pn), 31);
                p2 = p1;
                p1 = pn;
            };
            int i = 1;
            for (; i + 7 < k; i += 8) {
                const double d0 = dd[i], d1 = dd[i + 1], d2 = dd[i + 2], d3 = dd[i + 3];
                const double d4 = dd[i + 4], d5 = dd[i + 5], d6 = dd[i + 6], d7 = dd[i + 7];
                const double e0 = ee[i - 1], e1 = ee[i], e2 = ee[i + 1], e3 = ee[i + 2];
                const double e4 = ee[i + 3], e5 = ee[i + 4], e6 = ee[i + 5], e7 = ee[i + 6];
                step(d0, e0); step(d1, e1); step(d2, e2); step(d3, e3);
                step(d4, e4); step(d5, e5); step(d6, e6); step(d7, e7);
                cnt += __popc((signs ^ (signs >> 1)) & 0xFFu);
                const int ex = __builtin_amdgcn_frexp_exp(fmax(fabs(p1), fabs(p2)));
                p1 = ldexp(p1, -ex);
                p2 = ldexp(p2, -ex);
            }
            for (; i < k; i++) {
                step(dd[i], ee[i - 1]);
                cnt += (int)((signs ^ (signs >> 1)) & 1u);
            }
            if (cnt > idx) hi = mid; else lo = mid;
        }
        ev[idx] = ldexp(0.5 * (lo + hi), E);
    }
    __syncthreads();
}

// Householder tridiagonalisation of the full symmetric matrix A [k][ks] (LAPACK dsytd2, lower) for k <= 64: the rows of
// the trailing block one per lane, so lane i holds v_i and w_i of the Householder vector and the update vector in
// registers.  What every lane needs of them -- v_c, w_c, the same for all lanes -- is kept in the part of the matrix that
// step j leaves dead: v in row j right of the diagonal (contiguous), w in column j below the sub-diagonal (w_0 by
// v_readlane: that slot holds the sub-diagonal entry), and read back as LDS broadcasts.  No scratch beside the matrix --
// 20.4 KB at 50 contexts, eight wavefronts to a compute unit, a launch of 4096 variants in two rounds (with scratch it was
// 22.4 KB: seven, three rounds).  The diagonal and the sub-diagonal are left in place (A[i][i], A[i+1][i]).
__device__ void tridiagonalise_narrow(double* A, int k, int ks, int lane) {
    for (int j = 0; j < k - 2; j++) {
        const int m = k - j - 1;
        const bool in = lane < m;
        double* row = A + (j + 1 + (in ? lane : 0)) * ks + (j + 1);   // (idle lanes: a valid row, nothing stored)
        double* V = A + j * ks + (j + 1);           // v_c at V[c]
        double* W = A + (j + 1) * ks + j;           // w_c at W[c * ks], c >= 1
        const double x = in ? row[-1] : 0.0;
        const double sig = wsum(lane >= 1 ? x * x : 0.0);
        const double alpha = read_lane(x, 0);
        double tau = 0.0, beta = alpha, v = x;
        if (sig != 0.0) {
            // beta = -sign(alpha) |x|, tau = (beta - alpha) / beta = 1 + |alpha| / |x|, 1 / (alpha - beta) = sign(alpha) /
            // (|alpha| + |x|): one reciprocal square root and one reciprocal, Newton-refined from the hardware estimates
            // (a square root and two divisions by the IEEE sequences are a fifth of a step's dependent chain)
            const double n2 = alpha * alpha + sig, h = 0.5 * n2;
            double rs = __builtin_amdgcn_rsq(n2);
            rs *= fma(-h * rs, rs, 1.5);
            rs *= fma(-h * rs, rs, 1.5);
            const double nrm = n2 * rs;
            beta = -copysign(nrm, alpha);
            tau = fma(fabs(alpha), rs, 1.0);
            const double den = fabs(alpha) + nrm;
            double rc = __builtin_amdgcn_rcp(den);
            rc = fma(fma(-den, rc, 1.0), rc, rc);
            rc = fma(fma(-den, rc, 1.0), rc, rc);
            v = lane == 0 ? 1.0 : x * copysign(rc, alpha);
        }
        if (lane == 0) row[-1] = beta;
        if (tau != 0.0) {
            if (in) V[lane] = v;
            __syncthreads();
            // (the loops below are bound by the LDS round trip, not by their arithmetic: each batch's operands are
            // requested one batch ahead)
            double p = 0.0;
            int c = 0;
            if (m >= 8) {
                double rr[8], vv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { rr[u] = row[u]; vv[u] = V[u]; }
#pragma unroll 2
                for (; c + 15 < m; c += 8) {
                    double rn[8], vn[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { rn[u] = row[c + 8 + u]; vn[u] = V[c + 8 + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) p += rr[u] * vv[u];
#pragma unroll
                    for (int u = 0; u < 8; u++) { rr[u] = rn[u]; vv[u] = vn[u]; }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) p += rr[u] * vv[u];
                c += 8;
            }
            for (; c < m; c++) p += row[c] * V[c];
            p = in ? p * tau : 0.0;
            const double a2 = -0.5 * tau * wsum(p * v);
            const double w = p + a2 * v;
            if (in && lane >= 1) W[lane * ks] = w;
            const double w0 = read_lane(w, 0);
            __syncthreads();
            // A <- A - v w' - w v'
            auto update = [&](double rv, double vc, double wc) { return fma(-w, vc, fma(-v, wc, rv)); };
            {
                const double r = update(row[0], 1.0, w0);
                if (in) row[0] = r;
            }
            c = 1;
            if (m >= 5) {
                double rr[4], vv[4], ww[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { rr[u] = row[1 + u]; vv[u] = V[1 + u]; ww[u] = W[(1 + u) * ks]; }
#pragma unroll 2
                for (; c + 7 < m; c += 4) {
                    double rn[4], vn[4], wn[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { rn[u] = row[c + 4 + u]; vn[u] = V[c + 4 + u]; wn[u] = W[(c + 4 + u) * ks]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double r = update(rr[u], vv[u], ww[u]);
                        if (in) row[c + u] = r;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) { rr[u] = rn[u]; vv[u] = vn[u]; ww[u] = wn[u]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double r = update(rr[u], vv[u], ww[u]);
                    if (in) row[c + u] = r;
                }
                c += 4;
            }
            for (; c < m; c++) {
                const double r = update(row[c], V[c], W[c * ks]);
                if (in) row[c] = r;
            }
        }
        __syncthreads();
    }
}

// The same reduction for more than 64 contexts: lanes stride the rows, v and w in LDS (hv, hw); diagonal and sub-diagonal
// into dd [k], ee [k - 1].
__device__ void tridiagonalise_wide(double* A, int k, int ks, int lane, double* dd, double* ee, double* hv, double* hw) {
    for (int j = 0; j < k - 2; j++) {
        const int m = k - j - 1;
        double sig = 0.0;
        for (int i = lane; i < m; i += 64) {
            const double x = A[(j + 1 + i) * ks + j];
            hv[i] = x;
            if (i >= 1) sig += x * x;
        }
        sig = wsum(sig);
        __syncthreads();
        const double alpha = hv[0];
        double tau = 0.0, beta = alpha;
        if (sig != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + sig), alpha);
            tau = (beta - alpha) / beta;
            const double sc = 1.0 / (alpha - beta);
            __syncthreads();
            for (int i = lane; i < m; i += 64) hv[i] = i == 0 ? 1.0 : hv[i] * sc;
        }
        if (lane == 0) {
            dd[j] = A[j * ks + j];
            ee[j] = beta;
        }
        __syncthreads();
        if (tau != 0.0) {
            double pv = 0.0;
            for (int i = lane; i < m; i += 64) {
                const double* row = A + (j + 1 + i) * ks + (j + 1);
                double p = 0.0;
                for (int cidx = 0; cidx < m; cidx++) p += row[cidx] * hv[cidx];
                p *= tau;
                hw[i] = p;
                pv += p * hv[i];
            }
            pv = wsum(pv);
            const double a2 = -0.5 * tau * pv;
            __syncthreads();
            for (int i = lane; i < m; i += 64) hw[i] += a2 * hv[i];
            __syncthreads();
            for (int i = lane; i < m; i += 64) {
                double* row = A + (j + 1 + i) * ks + (j + 1);
                const double vi = hv[i], wi = hw[i];
                for (int cidx = 0; cidx < m; cidx++) row[cidx] -= vi * hw[cidx] + wi * hv[cidx];
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
        dd[k - 2] = A[(k - 2) * ks + (k - 2)];
        dd[k - 1] = A[(k - 1) * ks + (k - 1)];
        ee[k - 2] = A[(k - 1) * ks + (k - 2)];
    }
    __syncthreads();
}

// ---- the kernel ------------------------------------------------------------------------------
// LDS, NARROW (k <= 64, and every launch without the eigenvalue step): A [k][ks] (ks = k | 1), over which -- the matrix is
// dead once reduced -- ev [k], kept [k], the tridiagonal's dd [k], ee [k] are laid.  Otherwise (k > 64): A [k][ks] (in
// global memory past 128 contexts), ev [k], kept [k] (doubles as dd), k + 2 doubles for ee, 2k for the Householder vectors.
template <bool NARROW>
__global__ __launch_bounds__(64) void eig_davies_kernel(const double* __restrict__ Fall,
                                                         const double* __restrict__ Qall, int k,
                                                         double* __restrict__ lambda_out,
                                                         double* __restrict__ pv_out,
                                                         int* __restrict__ ifault_out,
                                                         double* __restrict__ liu_out, int do_eig,
                                                         double* __restrict__ scratch) {
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int ks = k | 1;
    const bool a_global = !NARROW && k > 128;                  // (the launcher's rule)
    double* A = a_global ? scratch + (size_t)b * k * ks : sm;  // k * ks  (touched only with do_eig)
    double* ev = (NARROW || a_global) ? sm : sm + (long)k * ks;   // k
    double* kept = ev + k;                                     // k

    bool bad = false;
#ifdef CRM_DAVIES_STAMPS   // (tools/diag/davies_phases.py: phase durations in 10 ns ticks, written over lambda_out[0..3])
    const unsigned long long st0 = wall_clock64();
    unsigned long long st1 = st0, st2 = st0, st3 = st0, stl = st0;
#endif
    if (do_eig) {
        const double* __restrict__ F = Fall + (long)b * k * k;
        // (the lower triangle, as eigvalsh reads it, mirrored: the matrix is read in memory order -- its upper triangle read
        // and dropped -- and e / k is a multiply-high: k * k <= 2^16)
        const unsigned kinv = 0xFFFFFFFFu / (unsigned)k + 1u;
        double amax = 0.0;
        for (int e = lane; e < k * k; e += 64) {
            const int i = (int)__umulhi((unsigned)e, kinv), j = e - i * k;
            const double v = F[e];
            if (i >= j) {
                if (!(fabs(v) < INFINITY)) bad = true;
                amax = fmax(amax, fabs(v));
                A[i * ks + j] = v;
                A[j * ks + i] = v;
            }
        }
        bad = __any(bad);
        __syncthreads();
        // the reduction squares the entries: a matrix far from the middle of the double range is brought there by a power
        // of two first (dsyev's scaling), and its eigenvalues taken back
        int rescale = 0;
        amax = -wave_min(-amax);
        if (!bad && amax != 0.0 && (amax < 0x1p-400 || amax > 0x1p400)) {
            rescale = __builtin_amdgcn_frexp_exp(amax);
            for (int e = lane; e < k * k; e += 64) {
                const int i = (int)__umulhi((unsigned)e, kinv), j = e - i * k;
                A[i * ks + j] = ldexp(A[i * ks + j], -rescale);
            }
            __syncthreads();
        }
#ifdef CRM_DAVIES_STAMPS
        stl = wall_clock64();
#endif
        if constexpr (NARROW) {
            if (!bad) tridiagonalise_narrow(A, k, ks, lane);
            double d = 0.0, e = 0.0;
            if (lane < k) {
                d = A[lane * ks + lane];
                if (lane < k - 1) e = A[(lane + 1) * ks + lane];
            }
            __syncthreads();
            double* dd = sm + 2 * k;
            double* ee = sm + 3 * k;
            if (lane < k) {
                dd[lane] = d;
                ee[lane] = e;
            }
            __syncthreads();
#ifdef CRM_DAVIES_STAMPS
            st1 = wall_clock64();
#endif
            sturm_bisection(dd, ee, ev, k, lane, bad);
        } else {
            double* dd = kept;               // (kept[] is free until the filter)
            double* ee = kept + k;           // k + 2 doubles
            double* hv = ee + 2 * (k / 2 + 1);
            if (!bad) tridiagonalise_wide(A, k, ks, lane, dd, ee, hv, hv + k);
#ifdef CRM_DAVIES_STAMPS
            st1 = wall_clock64();
#endif
            sturm_bisection(dd, ee, ev, k, lane, bad);
        }
        if (rescale != 0) {
            for (int i = lane; i < k; i += 64) ev[i] = ldexp(ev[i], rescale);
            __syncthreads();
        }
        for (int i = lane; i < k; i += 64) lambda_out[(long)b * k + i] = bad ? NAN : ev[i];
    } else {
        for (int i = lane; i < k; i += 64) {
            ev[i] = lambda_out[(long)b * k + i];
            if (!(fabs(ev[i]) < INFINITY)) bad = true;
        }
        bad = __any(bad);
        __syncthreads();
    }
#ifdef CRM_DAVIES_STAMPS
    st2 = wall_clock64();
#endif
    const double Q = Qall[b];
    if (bad || !(fabs(Q) < INFINITY)) {
        if (lane == 0) {
            pv_out[b] = NAN;
            if (ifault_out) ifault_out[b] = -1;
            if (liu_out) liu_out[b] = NAN;
        }
        return;
    }
    // SKAT Get_Lambda filter: lam > mean(lam[lam >= 0]) / 1e5   (ev ascending)
    double sp = 0.0;
    int np = 0;
    for (int i = lane; i < k; i += 64)
        if (ev[i] >= 0.0) { sp += ev[i]; np += 1; }
    sp = wsum(sp);
    np = wsum_i(np);
    const double thr = np > 0 ? (sp / np) / 100000.0 : INFINITY;
    int r = 0;
    if (lane == 0) {
        for (int i = 0; i < k; i++)
            if (ev[i] > thr) kept[r++] = ev[i];
    }
    r = __shfl(r, 0, 64);
    __syncthreads();
    if (r == 0) {  // "No eigenvalue is bigger than 0": the reference raises here
        if (lane == 0) {
            pv_out[b] = NAN;
            if (ifault_out) ifault_out[b] = -2;
            if (liu_out) liu_out[b] = NAN;
        }
        return;
    }
    const double p_liu = liu_mod_sf(kept, r, Q, lane);
    int ifault = 0;
    const double cdf = qfc_wave(kept, r, Q, lane, ifault);
    double p = 1.0 - cdf;
    if (r == 1) p = p_liu;
    if (p > 1.0 || p <= 0.0) p = p_liu;
    if (lane == 0) {
        pv_out[b] = p;
        if (ifault_out) ifault_out[b] = ifault;
        if (liu_out) liu_out[b] = p_liu;
#ifdef CRM_DAVIES_STAMPS
        st3 = wall_clock64();
        if (k >= 4) {
            lambda_out[(long)b * k + 0] = (double)(st1 - st0);
            lambda_out[(long)b * k + 1] = (double)(st2 - st1);
            lambda_out[(long)b * k + 2] = (double)(st3 - st2);
            lambda_out[(long)b * k + 3] = (double)st0;
            if (k >= 5) lambda_out[(long)b * k + 4] = (double)(stl - st0);
        }
#endif
    }
}

}  // namespace

size_t eig_scratch_doubles(int count, int k) { return k > 128 ? (size_t)count * k * (k | 1) : 0; }

int launch_eig_davies(hipStream_t st, const double* F, const double* Q, int count, int k,
                      double* lambda, double* pvalue, int* ifault, double* liu, bool do_eig, double* scratch) {
    if (count <= 0) return CRM_OK;
    if (k < 1 || k > CRM_MAX_K0) {
        set_error("eigen/Davies: k0=%d (supported 1..%d)", k, CRM_MAX_K0);
        return CRM_ERR_UNSUPPORTED;
    }
    const int ks = k | 1;
    const bool narrow = !do_eig || k <= 64;
    const bool global_copy = !narrow && k > 128;
    if (global_copy && !scratch) {
        set_error("eigen/Davies: k0=%d needs the global-memory work space", k);
        return CRM_ERR_ARG;
    }
    size_t lds;
    if (narrow) {
        lds = sizeof(double) * (do_eig ? (size_t)std::max(k * ks, 4 * k) : (size_t)2 * k);
        lds = (lds + 15) / 16 * 16;
        hipLaunchKernelGGL(eig_davies_kernel<true>, dim3(count), dim3(64), lds, st, F, Q, k, lambda, pvalue,
                           ifault, liu, do_eig ? 1 : 0, nullptr);
    } else {
        // A [k x ks], ev [k], kept [k] (doubles as the tridiagonal's diagonal), k + 2 doubles (its sub-diagonal), then 2k
        // doubles for the Householder vectors
        lds = sizeof(double) * ((global_copy ? 0 : (size_t)k * ks) + 2 * k + 2 * (k / 2 + 1) + 2 * k);
        lds = (lds + 15) / 16 * 16;
        hipLaunchKernelGGL(eig_davies_kernel<false>, dim3(count), dim3(64), lds, st, F, Q, k, lambda, pvalue,
                           ifault, liu, 1, global_copy ? scratch : nullptr);
    }
    CRM_HIP(hipGetLastError());
    return CRM_OK;
}

}  // namespace crm
