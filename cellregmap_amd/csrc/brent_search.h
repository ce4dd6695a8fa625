// The scalar search of the reference's null fits, shared by the three null-fit kernels (nullfit.hip, nullfit_wide.hip,
// nullfit_xwide.hip): glimix-core's LMM.fit -> brent-search minimize(rtol = atol = 1e-6) over x = logit(delta)
// (cellregmap/_cellregmap.py:351-352), statement for statement as oracle/brent.py restates it -- a downhill bracketing
// phase with growth factor 2 from (0, 1), then Brent's localmin.
//
// Besides the minimiser the search leaves behind how close it came to taking ANOTHER path (BrentTrace): the search is a
// sequence of decisions on objective VALUES -- f(x2) > f(x1) in the bracketing phase; f(u) <= f(x0), f(u) <= f(x1),
// f(u) <= f(x2) in localmin; the sign of a parabolic step shorter than the tolerance, of which only the sign survives
// (u = x0 +- tol) -- and two faithful implementations of the objective (this one and the reference's numpy) agree on a
// decision unless its margin is within the rounding noise of the objective.  Where they disagree the stopping points
// part by up to a whole tolerance, and the score statistic with them.  The smallest margin of a
// fit, against the noise bound of its objective, is what include/crm_hip.h: CRM_MODEL_FLAT_OPTIMUM is raised on.
// Tracking reads values the search has computed anyway: the search itself is bit for bit the one without it.
#pragma once
#include <hip/hip_runtime.h>

namespace crm {

struct BrentTrace {
    double cmp;    // smallest |f(a) - f(b)| over the value comparisons that steered the search
    double sign;   // smallest distance (in units of f) of a sub-tolerance parabolic step from changing its sign
    __device__ inline void reset() { cmp = INFINITY; sign = INFINITY; }
    __device__ inline void compare(double fa, double fb, bool tie_by_construction) {
        // (two points beyond the clamp of the logistic are the SAME delta: their values are one memoised number on
        // either side of any comparison, here and in the reference -- an exact tie that no rounding can break)
        if (tie_by_construction && fa == fb) return;
        const double m = fabs(fa - fb);
        if (m < cmp || m != m) cmp = m;
    }
};

// f: double -> double, called by every participating thread in lockstep (it may contain barriers); f.clamped(): whether
// the last evaluation was one of the two memoised points delta = eps / 1 - eps.  Returns the minimiser; fx its value.
// TRACK = false: the search alone (tr untouched) -- the scans that do not ask for model flags.
template <bool TRACK, class F>
__device__ __forceinline__ double brent_search(F& f, BrentTrace& tr, double& fx) {
    constexpr double LOGMAX_ = 709.782712893384;   // log(finfo.max)
    constexpr double GOLDEN_ = 0.381966011250105097;
    constexpr int MAXITER_ = 500;
    if constexpr (TRACK) tr.reset();
    // ---- bracket (oracle/brent.py: bracket) ------------------------------------------
    const double lo = -LOGMAX_, hi = LOGMAX_;
    double x0 = 0.0, x1 = 1.0;
    double f0 = f(x0);
    bool c0 = f.clamped();
    double f1 = f(x1);
    bool c1 = f.clamped();
    if constexpr (TRACK) tr.compare(f1, f0, c0 && c1);
    if (f1 > f0) {
        double t = x0; x0 = x1; x1 = t;
        t = f0; f0 = f1; f1 = t;
        const bool tc = c0; c0 = c1; c1 = tc;
    }
    double bl, bm, bh, fm;
    bool bracketed = false;
    for (int it = 0; it < MAXITER_; it++) {
        double x2 = x1 + 2.0 * (x1 - x0);
        x2 = fmin(fmax(x2, lo), hi);
        if (x2 == x1) break;
        const double f2 = f(x2);
        const bool c2 = f.clamped();
        if constexpr (TRACK) tr.compare(f2, f1, c1 && c2);
        if (f2 > f1) {
            bl = x0 < x2 ? x0 : x2;
            bh = x0 < x2 ? x2 : x0;
            bm = x1;
            fm = f1;
            bracketed = true;
            break;
        }
        x0 = x1; f0 = f1; c0 = c1;
        x1 = x2; f1 = f2; c1 = c2;
    }
    if (!bracketed) {
        bl = x0 < x1 ? x0 : x1;
        bh = x0 < x1 ? x1 : x0;
        bm = x1;
        fm = f1;
    }
    // ---- Brent localmin (oracle/brent.py: localmin), rtol = atol = 1e-6 ---------------
    const double rtol = 1e-6, atol = 1e-6;
    double A_ = bl, B_ = bh;
    double bx0 = bm, bf0 = fm;
    double bx1 = bx0, bx2 = bx0, bf1 = bf0, bf2 = bf0;
    bool k0 = c1, k1 = c1, k2 = c1;   // (clamp state of bx0, bx1, bx2)
    double d = 0.0, e = 0.0;
    for (int it = 0; it < MAXITER_; it++) {
        const double m = 0.5 * (A_ + B_);
        const double tol = rtol * fabs(bx0) + atol;
        const double tol2 = 2.0 * tol;
        if (fabs(bx0 - m) <= tol2 - 0.5 * (B_ - A_)) break;
        double p = 0.0, q = 0.0, rr = 0.0;
        if (tol < fabs(e)) {
            rr = (bx0 - bx1) * (bf0 - bf2);
            q = (bx0 - bx2) * (bf0 - bf1);
            p = (bx0 - bx2) * q - (bx0 - bx1) * rr;
            q = 2.0 * (q - rr);
            if (0.0 < q) p = -p;
            q = fabs(q);
            rr = e;
            e = d;
        }
        double u;
        if (fabs(p) < fabs(0.5 * q * rr) && q * (A_ - bx0) < p && p < q * (B_ - bx0)) {
            d = p / q;
            u = bx0 + d;
            if constexpr (TRACK) {
                if (!((u - A_) < tol2 || (B_ - u) < tol2) && fabs(d) < tol) {
                    // only the sign of the step survives below: p = (x0-x2)^2 (f0-f1) - (x0-x1)^2 (f0-f2); a change of the
                    // three values by eta each moves p by at most 2 eta ((x0-x2)^2 + (x0-x1)^2)
                    const double w2 = bx0 - bx2, w1 = bx0 - bx1;
                    const double cp = 2.0 * (w2 * w2 + w1 * w1);
                    const double s = cp > 0.0 ? fabs(p) / cp : 0.0;
                    if (s < tr.sign) tr.sign = s;
                }
            }
            if ((u - A_) < tol2 || (B_ - u) < tol2) d = bx0 < m ? tol : -tol;
        } else {
            e = bx0 < m ? B_ - bx0 : A_ - bx0;
            d = GOLDEN_ * e;
        }
        if (tol <= fabs(d)) u = bx0 + d;
        else if (0.0 < d) u = bx0 + tol;
        else u = bx0 - tol;
        const double fu = f(u);
        const bool ku = f.clamped();
        if constexpr (TRACK) tr.compare(fu, bf0, ku && k0);
        if (fu <= bf0) {
            if (u < bx0) B_ = bx0; else A_ = bx0;
            bx2 = bx1; bf2 = bf1; k2 = k1;
            bx1 = bx0; bf1 = bf0; k1 = k0;
            bx0 = u; bf0 = fu; k0 = ku;
        } else {
            if (u < bx0) A_ = u; else B_ = u;
            if constexpr (TRACK) { if (bx1 != bx0) tr.compare(fu, bf1, ku && k1); }
            if (fu <= bf1 || bx1 == bx0) {
                bx2 = bx1; bf2 = bf1; k2 = k1;
                bx1 = u; bf1 = fu; k1 = ku;
            } else {
                if constexpr (TRACK) { if (bx2 != bx0 && bx2 != bx1) tr.compare(fu, bf2, ku && k2); }
                if (fu <= bf2 || bx2 == bx0 || bx2 == bx1) {
                    bx2 = u; bf2 = fu; k2 = ku;
                }
            }
        }
    }
    fx = bf0;
    return bx0;
}

}  // namespace crm
