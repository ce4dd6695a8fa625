"""The background constructor's hand-written symmetric eigen-solver (cellregmap_amd/csrc/eigh*.hip; it
replaces the LAPACK calls behind numpy_sugar.economic_qs_linear, cellregmap/_math.py:204-256) against
numpy on matrices of the kinds the constructor meets: Gram matrices with exact zero blocks (rho = 0 / 1),
rank-deficient, clustered and graded spectra, already tridiagonal input, every size class of the
divide-and-conquer tree."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cellregmap_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.crm_ctx_create(0, ctypes.byref(h)))
    yield lib, h
    lib.crm_ctx_destroy(h)


def _eigh(ctx, A, stage=0):
    from cellregmap_amd import _lib

    lib, h = ctx
    A = np.ascontiguousarray(A, dtype=float)
    if A.ndim == 2:
        A = A[None]
    batch, dim = A.shape[0], A.shape[1]
    lam = np.empty((batch, dim))
    Z = np.empty((batch, dim, dim))
    d = np.empty((batch, dim))
    e = np.empty((batch, dim))
    _lib.check(lib.crm_test_eigh(h, batch, dim, _lib.ptr(A), _lib.ptr(lam), _lib.ptr(Z), stage, _lib.ptr(d), _lib.ptr(e)))
    return lam, Z, d, e


def _check(A, lam, Z, tol=1e-13):
    n = A.shape[0]
    ref = np.linalg.eigvalsh(A)
    scale = max(np.abs(ref).max(), 1e-300)
    assert np.all(np.diff(lam) >= 0)
    assert np.abs(lam - ref).max() <= tol * n * scale
    assert np.abs(A @ Z - Z * lam).max() <= tol * n * scale
    assert np.abs(Z.T @ Z - np.eye(n)).max() <= tol * n


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 31, 32, 33, 48, 64, 65, 100, 129, 257, 500])
def test_random_symmetric(ctx, n):
    rng = np.random.default_rng(n)
    X = rng.normal(size=(n, n))
    A = X + X.T
    lam, Z, d, e = _eigh(ctx, A)
    _check(A, lam[0], Z[0])


def test_tridiagonalisation_is_a_similarity(ctx):
    """Stage 1 alone: the tridiagonal (d, e) has the spectrum of A."""
    rng = np.random.default_rng(5)
    for n in (40, 97, 260):
        X = rng.normal(size=(n, n))
        A = X + X.T
        _, _, d, e = _eigh(ctx, A, stage=1)
        T = np.diag(d[0]) + np.diag(e[0, : n - 1], 1) + np.diag(e[0, : n - 1], -1)
        ref = np.linalg.eigvalsh(A)
        assert np.abs(np.linalg.eigvalsh(T) - ref).max() <= 1e-13 * n * np.abs(ref).max()


def test_tridiagonal_divide_and_conquer(ctx):
    """Stage 2 on input that is tridiagonal already (the reflectors are identities): Wilkinson's W21+ with its
    pairs of eigenvalues that agree to 1e-14, the 1-2-1 Toeplitz matrix, and a glued one (heavy deflation)."""
    n = 21
    W = np.diag(np.abs(np.arange(-10, 11)).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    T = np.diag(np.full(300, 2.0)) + np.diag(np.full(299, -1.0), 1) + np.diag(np.full(299, -1.0), -1)
    Gl = np.kron(np.eye(6), W)
    for k in range(1, 6):
        Gl[k * n - 1, k * n] = Gl[k * n, k * n - 1] = 1e-9
    for A in (W, T, Gl):
        lam, Z, _, _ = _eigh(ctx, A)
        _check(A, lam[0], Z[0])


def test_spectra_of_the_kinds_a_background_has(ctx):
    rng = np.random.default_rng(0)
    n = 300
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    cases = {
        "rank deficient Gram": (lambda X: X @ X.T)(rng.normal(size=(n, 40))),
        "three clusters": (Q * np.repeat([1.0, 2.0, 3.0], n // 3)) @ Q.T,
        "graded over 12 decades": (Q * np.logspace(0, -12, n)) @ Q.T,
        "identity": np.eye(n),
        "diagonal": np.diag(rng.normal(size=n)),
        "zero": np.zeros((n, n)),
    }
    H = rng.normal(size=(700, 60))
    C = H.T @ H
    for rho in (0.0, 0.3, 1.0):  # D C D with the exact zero rows / columns of rho = 0 and rho = 1
        D = np.r_[np.full(10, np.sqrt(rho)), np.full(50, np.sqrt(1 - rho))]
        cases[f"scaled Gram rho={rho}"] = D[:, None] * C * D[None, :]
    for name, A in cases.items():
        A = 0.5 * (A + A.T)
        lam, Z, _, _ = _eigh(ctx, A)
        _check(A, lam[0], Z[0])


def test_batch_of_grid_points(ctx):
    """Eleven matrices at once (the constructor's shape: one Gram matrix rescaled per grid point)."""
    rng = np.random.default_rng(3)
    H = rng.normal(size=(900, 210))
    C = H.T @ H
    A = []
    for rho in np.linspace(0, 1, 11):
        D = np.r_[np.full(10, np.sqrt(rho)), np.full(200, np.sqrt(1 - rho))]
        A.append(D[:, None] * C * D[None, :])
    A = np.stack(A)
    lam, Z, _, _ = _eigh(ctx, A)
    for b in range(11):
        _check(A[b], lam[b], Z[b])


def test_larger_matrix(ctx):
    rng = np.random.default_rng(9)
    n = 1500
    H = rng.normal(size=(4000, n)) * np.logspace(0, -3, n)
    A = H.T @ H
    lam, Z, _, _ = _eigh(ctx, A)
    _check(A, lam[0], Z[0], tol=2e-13)
