"""Scalar minimiser used by glimix-core's ``LMM.fit`` (oracle; test infrastructure).

glimix-core calls ``optimix.Function._maximize_scalar(rtol=1e-6, atol=1e-6)``
which negates the objective and calls ``brent_search.minimize(f, a=lo, b=hi,
rtol, atol)`` with no starting point (reference call sites:
cellregmap/_cellregmap.py:352 ``lmm.fit(verbose=False)``).  brent-search is
absent from this image, so this is a restatement of its published algorithm:
a downhill bracketing phase with geometric growth (factor 2) followed by
Brent's (1973) ``localmin`` -- golden section + successive parabolic
interpolation with ``tol = rtol*|x| + atol``.  **Parity unpinned**: the exact
first bracketing step of brent-search is not recoverable here; any bracketing
of the same basin converges to the same minimiser within ``tol``.

The HIP null-fit kernel (cellregmap_amd/csrc/nullfit.hip) runs this same
procedure, statement for statement, so that both sides take the same steps.
"""
import math

GOLDEN = 0.381966011250105097
GROWTH = 2.0
FIRST_STEP = 1.0
START = 0.0   # logit(delta = 0.5), glimix-core's initial value
MAXITER = 500


def bracket(f, a, b, x0=0.0, step=FIRST_STEP, growth=GROWTH, maxiter=MAXITER):
    """Return (xl, xm, xr, fm) with xl < xm < xr inside [a, b] and f(xm) <= ends,
    or a degenerate triple pinned at a bound when f is monotone up to it."""
    x0 = min(max(x0, a), b)
    x1 = min(max(x0 + step, a), b)
    f0 = f(x0)
    f1 = f(x1)
    if f1 > f0:  # walk downhill: from x0 towards x1
        x0, x1 = x1, x0
        f0, f1 = f1, f0
    # invariant: f1 <= f0, direction = x1 - x0
    for _ in range(maxiter):
        x2 = x1 + growth * (x1 - x0)
        x2 = min(max(x2, a), b)
        if x2 == x1:  # ran into a bound while still going downhill
            break
        f2 = f(x2)
        if f2 > f1:
            lo, hi = (x0, x2) if x0 < x2 else (x2, x0)
            return lo, x1, hi, f1
        x0, f0 = x1, f1
        x1, f1 = x2, f2
    lo, hi = (x0, x1) if x0 < x1 else (x1, x0)
    return lo, x1, hi, f1


def localmin(f, a, b, x0, f0, rtol, atol, maxiter=MAXITER):
    """Brent's localmin on [a, b] started from (x0, f0).  Returns (x, fx, nit)."""
    x1 = x2 = x0
    f1 = f2 = f0
    d = 0.0
    e = 0.0
    nit = 0
    for nit in range(1, maxiter + 1):
        m = 0.5 * (a + b)
        tol = rtol * abs(x0) + atol
        tol2 = 2.0 * tol
        if abs(x0 - m) <= tol2 - 0.5 * (b - a):
            break
        p = q = r = 0.0
        if tol < abs(e):
            r = (x0 - x1) * (f0 - f2)
            q = (x0 - x2) * (f0 - f1)
            p = (x0 - x2) * q - (x0 - x1) * r
            q = 2.0 * (q - r)
            if 0.0 < q:
                p = -p
            q = abs(q)
            r = e
            e = d
        if abs(p) < abs(0.5 * q * r) and q * (a - x0) < p and p < q * (b - x0):
            d = p / q
            u = x0 + d
            if (u - a) < tol2 or (b - u) < tol2:
                d = tol if x0 < m else -tol
        else:
            e = (b - x0) if x0 < m else (a - x0)
            d = GOLDEN * e
        if tol <= abs(d):
            u = x0 + d
        elif 0.0 < d:
            u = x0 + tol
        else:
            u = x0 - tol
        fu = f(u)
        if fu <= f0:
            if u < x0:
                b = x0
            else:
                a = x0
            x2, f2 = x1, f1
            x1, f1 = x0, f0
            x0, f0 = u, fu
        else:
            if u < x0:
                a = u
            else:
                b = u
            if fu <= f1 or x1 == x0:
                x2, f2 = x1, f1
                x1, f1 = u, fu
            elif fu <= f2 or x2 == x0 or x2 == x1:
                x2, f2 = u, fu
    return x0, f0, nit


def minimize(f, a=-math.inf, b=math.inf, rtol=1e-6, atol=1e-6):
    """bracket + localmin.  Returns (x, fx, nfev)."""
    count = [0]

    def g(x):
        count[0] += 1
        return f(x)

    # (module attributes read at call time: tests/test_oracle_brackets.py reruns the scans with other plausible
    # bracketing phases -- the one piece of brent-search this restatement had to guess)
    lo, xm, hi, fm = bracket(g, a, b, x0=START, step=FIRST_STEP, growth=GROWTH)
    x, fx, _ = localmin(g, lo, hi, xm, fm, rtol, atol)
    return x, fx, count[0]
