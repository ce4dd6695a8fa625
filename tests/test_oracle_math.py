"""Pins oracle.scoretest / oracle.sugar against the reference.

* known-answer values held by the reference's own tests
  (cellregmap/test/test_math.py:55-83);
* golden vectors generated from the reference's _math.py in the build container
  (tests/golden/make_math_golden.py).
"""
import numpy as np
from numpy.testing import assert_allclose

from oracle import scoretest as st
from oracle import sugar


def test_kat_P_matrix(math_golden):
    g = math_golden
    # test_math.py:55-63 (8 printed digits)
    P = np.array([[0.50355613, -0.24203676, -0.34880245],
                  [-0.24203676, 0.11633617, 0.16765363],
                  [-0.34880245, 0.16765363, 0.24160792]])
    ours = st.dense_P(g["kat_W"], g["kat_K"])
    assert_allclose(ours, P, rtol=1e-6)
    assert_allclose(ours, g["kat_P"], rtol=1e-12)


def test_kat_score_statistic(math_golden):
    g = math_golden
    q = st.dense_Q(g["kat_y"], g["kat_W"], g["kat_K"], g["kat_dK"])
    assert_allclose(q, 0.49961017073389324, rtol=1e-7)  # test_math.py:66-68
    assert_allclose(q, g["kat_Q"], rtol=1e-12)


def test_kat_weights(math_golden):
    g = math_golden
    w = st.dense_weights(g["kat_W"], g["kat_K"], g["kat_dK"])
    # test_math.py:71-73: [4.55266277e-09, 3.46249449e-01] atol 1e-7; the first
    # entry is rounding noise of a singular matrix square root and may be absent
    assert abs(w[-1] - 3.46249449e-01) < 1e-7
    assert np.all(np.abs(w[:-1]) < 1e-7)


def test_kat_liu_params(math_golden):
    g = math_golden
    q = st.dense_Q(g["kat_y"], g["kat_W"], g["kat_K"], g["kat_dK"])
    w = np.array([4.55266277e-09, st.dense_weights(g["kat_W"], g["kat_K"], g["kat_dK"])[-1]])
    par = st.liu_params(q, w)
    # test_math.py:76-83 (default assert_allclose rtol 1e-7)
    assert_allclose(par["pv"], 0.22966744652848403, rtol=1e-7)
    assert_allclose(par["mu_q"], 0.34624945394475326, rtol=1e-7)
    assert_allclose(par["sigma_q"], 0.48967066729451103, rtol=1e-7)
    assert_allclose(par["dof_x"], 1.0, rtol=1e-7)


def test_implicit_algebra_matches_reference_golden(math_golden):
    g = math_golden
    for tag in "abc":
        cov = st.LowRankCov(g[f"{tag}_Q0"], g[f"{tag}_S0"], *g[f"{tag}_ab"])
        V, X, y = g[f"{tag}_V"], g[f"{tag}_X"], g[f"{tag}_y"]
        assert_allclose(st.cov_apply(cov, V), g[f"{tag}_dot"], rtol=1e-12, atol=1e-12)
        assert_allclose(st.cov_solve(cov, V), g[f"{tag}_solve"], rtol=1e-11, atol=1e-12)
        P = st.Projection(cov, X)
        assert_allclose(P.apply(V), g[f"{tag}_Pdot"], rtol=1e-10, atol=1e-12)
        half = g[f"{tag}_g"][:, None] * g[f"{tag}_E"]
        assert_allclose(st.score_Q(P, half, y), g[f"{tag}_stat"], rtol=1e-11)
        assert_allclose(st.score_F(P, half), g[f"{tag}_F"], rtol=1e-10, atol=1e-12)
        # dense twins agree with the implicit forms (the reference's own cross-check)
        assert_allclose(st.score_Q(P, half, y), g[f"{tag}_denseQ"], rtol=1e-8)


def test_economic_qs_linear_matches_reference_golden(math_golden):
    g = math_golden
    for tag in "abc":
        H = g[f"{tag}_H"]
        (Q0,), S0 = sugar.economic_qs_linear(H, return_q1=False)
        assert Q0.shape == g[f"{tag}_Q0"].shape
        assert_allclose(S0, g[f"{tag}_S0"], rtol=1e-10, atol=1e-12)
        # bases may differ by signs: compare the reconstructed covariance
        assert_allclose((Q0 * S0) @ Q0.T, (g[f"{tag}_Q0"] * g[f"{tag}_S0"]) @ g[f"{tag}_Q0"].T,
                        rtol=1e-9, atol=1e-9)
    (q0, q1), s0 = sugar.economic_qs(g["eq_K"])
    assert [q0.shape[1], q1.shape[1]] == list(g["eq_rank"])
    assert_allclose(s0, g["eq_S0"], rtol=1e-10)
    assert_allclose(q0 @ q0.T, g["eq_proj0"], atol=1e-10)


def test_qscov_against_dense_like_reference_test():
    # mirrors cellregmap/test/test_math.py:38-52
    rs = np.random.RandomState(0)
    K = rs.randn(3, 3)
    K = K @ K.T
    K = K[:, :2] @ K[:, :2].T
    (Q0, _), S0 = sugar.economic_qs(K)
    a, b = 0.2, 0.3
    full = a * K + b * np.eye(3)
    cov = st.LowRankCov(Q0, S0, a, b)
    v = np.array([0.3, -0.2, 0.19])
    assert_allclose(full @ v, st.cov_apply(cov, v))
    assert_allclose(st.lstsq_solve(full, v), st.cov_solve(cov, v))
