"""Generate golden vectors from the reference's own ``cellregmap/_math.py``.

Runs ONLY in the build container (``/root/reference`` does not exist on the GPU
box); the produced ``math_golden.npz`` is committed and is what the tests read.

The reference module imports ``numpy_sugar.ddot``, which is not installed here.
``ddot(L, R, left)`` is the diagonal product ``diag(L) @ R`` (``left=True``);
a three-line stand-in with exactly that meaning is registered as
``numpy_sugar`` before loading the module, so the vectors are labelled
"reference _math.py + local ddot stand-in".  Nothing else is substituted: the
classes/functions executed are the reference's, loaded from where they lie.

    python3 -B tests/golden/make_math_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/cellregmap/_math.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "math_golden.npz")


def _load_reference_math():
    shim = types.ModuleType("numpy_sugar")

    def ddot(L, R, left=None, out=None):
        L = np.asarray(L, float)
        R = np.asarray(R, float)
        if left is None:
            left = L.ndim == 1
        return (L[:, None] * R if R.ndim == 2 else L * R) if left else L * R[None, :]

    shim.ddot = ddot
    sys.modules["numpy_sugar"] = shim
    spec = importlib.util.spec_from_file_location("_ref_math", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    m = _load_reference_math()
    out = {}

    # -- the reference's own 3x3 fixture (test_math.py:17-35), frozen: today's
    #    RandomState.multivariate_normal draws a different y than the one the
    #    pinned value 0.49961017073389324 was produced with.
    W3 = np.array([[1.764052345967664, 0.4001572083672233],
                   [0.9787379841057392, 2.240893199201458],
                   [1.8675579901499675, -0.977277879876411]])
    K03 = np.array([[0.9362311369855318, 0.21819440655835556, 0.658821683908891],
                    [0.21819440655835556, 2.3042511132412047, 0.9755059938572415],
                    [0.658821683908891, 0.9755059938572415, 0.7909977981186607]])
    y3 = np.array([-0.5570138122474267, 1.0636510108353878, 0.9015784202772069])
    K3 = 0.2 * K03 + np.eye(3)
    out["kat_W"], out["kat_K"], out["kat_dK"], out["kat_y"] = W3, K3, K03, y3
    out["kat_P"] = m.P_matrix(W3, K3)
    out["kat_Q"] = m.score_statistic(y3, W3, K3, K03)
    out["kat_weights"] = m.score_statistic_distr_weights(W3, K3, K03)

    # -- seeded medium cases through the implicit classes ---------------------------
    rng = np.random.default_rng(1234)
    for tag, (n, r, c, k) in {"a": (40, 7, 2, 5), "b": (120, 30, 3, 9), "c": (48, 60, 1, 4)}.items():
        H = rng.normal(size=(n, r))
        Q0, S0 = m.economic_qs_linear(H)  # r >= n takes the eigh branch (_math.py:255)
        a, b = 0.37 + 0.1 * rng.random(), 0.81 + 0.1 * rng.random()
        X = np.concatenate([np.ones((n, 1)), rng.normal(size=(n, c))], axis=1)
        y = rng.normal(size=n)
        g = rng.normal(size=n)
        E = rng.normal(size=(n, k))
        V = rng.normal(size=(n, 3))
        cov = m.QSCov(Q0, S0, a, b)
        P = m.PMat(cov, X)
        half = g[:, None] * E
        ss = m.ScoreStatistic(P, cov, half)
        out[f"{tag}_H"] = H
        out[f"{tag}_Q0"], out[f"{tag}_S0"] = Q0, S0
        out[f"{tag}_ab"] = np.array([a, b])
        out[f"{tag}_X"], out[f"{tag}_y"], out[f"{tag}_g"], out[f"{tag}_E"], out[f"{tag}_V"] = X, y, g, E, V
        out[f"{tag}_dot"] = cov.dot(V)
        out[f"{tag}_solve"] = cov.solve(V)
        out[f"{tag}_Pdot"] = P.dot(V)
        out[f"{tag}_stat"] = ss.statistic(y)
        out[f"{tag}_F"] = ss.matrix_for_dist_weights()
        Kd = a * (Q0 * S0) @ Q0.T + b * np.eye(n)
        out[f"{tag}_denseP"] = m.P_matrix(X, Kd)
        out[f"{tag}_denseQ"] = m.score_statistic(y, X, Kd, half @ half.T)

    # -- economic_qs on a rank-deficient symmetric matrix (eigh branch) -----------------
    B = rng.normal(size=(30, 11))
    Ksym = B @ B.T
    (q0, q1), s0 = m.economic_qs(Ksym)
    out["eq_K"], out["eq_S0"] = Ksym, s0
    out["eq_proj0"] = q0 @ q0.T  # sign/basis independent
    out["eq_rank"] = np.array([q0.shape[1], q1.shape[1]])

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
