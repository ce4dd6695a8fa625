"""Single-kernel parity on a real MI355X, through the C-ABI test hooks."""
import ctypes

import numpy as np
import pytest
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cellregmap_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.crm_ctx_create(0, ctypes.byref(h)))
    yield lib, h
    lib.crm_ctx_destroy(h)


# (tile width, LDS-DMA): auto; register-staged 64- and 128-wide; LDS-DMA 128-wide; LDS-DMA with 64-wide
# tiles for the Khatri-Rao form (plain products fall back to the register-staged 64-wide kernel)
VARIANTS = [(0, 1), (64, 0), (128, 0), (128, 1), (64, 1), (160, 1)]   # 160: Khatri-Rao form only (plain: 128)


@pytest.fixture(params=VARIANTS, ids=lambda v: f"tile{v[0]}-dma{v[1]}")
def variant(request, ctx):
    from cellregmap_amd import _lib

    lib, h = ctx
    _lib.check(lib.crm_test_set_contraction(h, *request.param))
    yield request.param
    _lib.check(lib.crm_test_set_contraction(h, 0, 1))


def _contract(ctx, X, Y, ksplit=1):
    from cellregmap_amd import _lib

    lib, h = ctx
    X, Y = _lib.f64(X), _lib.f64(Y)
    C = np.empty((X.shape[1], Y.shape[1]))
    _lib.check(lib.crm_test_contract(h, X.shape[0], X.shape[1], Y.shape[1], _lib.ptr(X), _lib.ptr(Y),
                                     _lib.ptr(C), ksplit))
    return C


def test_contract_layout_with_integer_data(ctx, variant):
    # exact small integers + asymmetric operands: catches any row/column swap of the
    # v_mfma_f64_16x16x4_f64 fragment maps
    rng = np.random.default_rng(0)
    X = rng.integers(-3, 4, size=(24, 37)).astype(float)
    Y = rng.integers(-3, 4, size=(24, 150)).astype(float)
    X[:, 5] = np.arange(24)
    Y[:, 7] = np.arange(24) ** 2
    assert np.array_equal(_contract(ctx, X, Y), X.T @ Y)


@pytest.mark.parametrize("cells,M,N,ksplit", [(1000, 130, 257, 1), (4096, 256, 128, 4), (777, 16, 16, 1),
                                               (20000, 200, 1275, 5), (7, 140, 130, 1), (20, 129, 300, 1),
                                               (48, 300, 129, 1)])  # 1, 2 and 3 stages of 16 cells
def test_contract_random(ctx, variant, cells, M, N, ksplit):
    rng = np.random.default_rng(cells + M)
    X = rng.normal(size=(cells, M))
    Y = rng.normal(size=(cells, N))
    ref = X.T @ Y
    assert_allclose(_contract(ctx, X, Y, ksplit), ref, rtol=0, atol=1e-11 * np.sqrt(cells))


@pytest.mark.parametrize("cells,B,k0,N", [(500, 7, 10, 140), (2048, 20, 50, 300), (333, 40, 3, 64),
                                           (1024, 5, 128, 130), (640, 300, 1, 128), (5, 9, 50, 200),
                                           (30, 40, 7, 130), (40, 6, 33, 129), (100, 3, 97, 140),
                                           # more than 128 contexts: the slower form with context tiles twice as wide
                                           (200, 5, 129, 140), (333, 7, 160, 300), (64, 3, 256, 50), (1000, 4, 200, 129)])
def test_contract_khatri_rao(ctx, variant, cells, B, k0, N):
    from cellregmap_amd import _lib

    lib, h = ctx
    rng = np.random.default_rng(B * k0)
    G = _lib.f64(rng.normal(size=(cells, B)))
    E = _lib.f64(rng.normal(size=(cells, k0)))
    Y = _lib.f64(rng.normal(size=(cells, N)))
    C = np.empty((B * k0, N))
    _lib.check(lib.crm_test_contract_kr(h, cells, B, k0, N, _lib.ptr(G), _lib.ptr(E), _lib.ptr(Y),
                                        _lib.ptr(C)))
    KR = (G[:, :, None] * E[:, None, :]).reshape(cells, B * k0)
    assert_allclose(C, KR.T @ Y, rtol=0, atol=1e-11 * np.sqrt(cells))


@pytest.mark.parametrize("every", [1, 3])
def test_khatri_rao_contraction_in_persistent_generations(ctx, every):
    """The optional persistent form (8 x 64 workgroups, soft re-alignment per XCD group) on a launch of
    more than 1024 tiles, ragged at both ends."""
    from cellregmap_amd import _lib

    lib, h = ctx
    cells, B, k0, N = 1040, 1300, 13, 1100       # 133 row tiles x 9 column tiles = 1197 tiles; 65 stages (the form takes >= 64)
    rng = np.random.default_rng(every)
    G = rng.normal(size=(cells, B))
    E = rng.normal(size=(cells, k0))
    Y = rng.normal(size=(cells, N))
    C = np.empty((B * k0, N))
    _lib.check(lib.crm_test_set_contraction(h, 0, 1))
    _lib.check(lib.crm_test_set_contraction_sync(h, every))
    try:
        _lib.check(lib.crm_test_contract_kr(h, cells, B, k0, N, _lib.ptr(G), _lib.ptr(E), _lib.ptr(Y), _lib.ptr(C)))
    finally:
        _lib.check(lib.crm_test_set_contraction_sync(h, 1))   # the default
    KR = (G[:, :, None] * E[:, None, :]).reshape(cells, B * k0)
    assert_allclose(C, KR.T @ Y, rtol=0, atol=1e-11 * np.sqrt(cells))


@pytest.mark.parametrize("cells,B,k0,N", [(64, 3, 5, 17), (1000, 37, 50, 300), (320, 130, 4, 129), (4096, 8, 128, 64),
                                          (208, 9, 20, 40), (224, 700, 3, 51), (1600, 64, 50, 51), (96, 40, 33, 64),
                                          (208, 5, 129, 40), (500, 7, 160, 300), (64, 3, 256, 50), (96, 4, 200, 129)])
def test_khatri_rao_contraction_with_transposed_store(ctx, cells, B, k0, N):
    """The shared-H route of the multi-gene scan stores (KR(G,E)' H)' directly (operands of the MFMA
    swapped, stores along M)."""
    from cellregmap_amd import _lib

    lib, h = ctx
    rng = np.random.default_rng(cells + B)
    G = rng.normal(size=(cells, B))
    E = rng.normal(size=(cells, k0))
    Y = rng.normal(size=(cells, N))
    CT = np.empty((N, B * k0))
    _lib.check(lib.crm_test_contract_kr_t(h, cells, B, k0, N, _lib.ptr(G), _lib.ptr(E), _lib.ptr(Y), _lib.ptr(CT)))
    KR = (G[:, :, None] * E[:, None, :]).reshape(cells, B * k0)
    assert_allclose(CT, Y.T @ KR, rtol=0, atol=1e-11 * np.sqrt(cells))


@pytest.mark.parametrize("k", [1, 2, 7, 50, 64, 65, 128, 129, 160, 256])   # past 128: working copy in global memory
def test_eigvalsh_batched(ctx, k):
    from cellregmap_amd import _lib

    lib, h = ctx
    rng = np.random.default_rng(k)
    count = 9
    F = np.empty((count, k, k))
    for i in range(count):
        A = rng.normal(size=(k, max(1, k // (1 + i % 3))))
        S = A @ A.T
        F[i] = S + 1e-14 * rng.normal(size=(k, k))  # asymmetric noise: only the lower triangle counts
    lam = np.empty((count, k))
    _lib.check(lib.crm_test_eigvalsh(h, count, k, _lib.ptr(F), _lib.ptr(lam)))
    for i in range(count):
        ref = np.linalg.eigvalsh(F[i])  # UPLO='L'
        assert_allclose(lam[i], ref, rtol=0, atol=1e-13 * max(1.0, abs(ref).max()))


@pytest.mark.parametrize("k", [3, 8, 21, 50, 64, 70])
def test_eigvalsh_on_structured_matrices(ctx, k):
    """The cases the bisection's count -- signs of the leading minors, rescaled every eight steps (csrc/davies.hip:
    sturm_bisection) -- has to survive without the quotient form's pivmin: the zero matrix, multiples of the identity,
    diagonal matrices with zeros and repeats, decoupled blocks (exact zeros off the diagonal, a zero block), minors that
    vanish exactly at a bisection point (antidiagonal pairs: eigenvalues +-1 around the midpoint 0), Wilkinson's close
    pairs, rank one, negative definite, and entries near both ends of the double range."""
    from cellregmap_amd import _lib

    lib, h = ctx
    rng = np.random.default_rng(100 + k)
    mats = [np.zeros((k, k)), 3.5 * np.eye(k), -2.0 * np.eye(k)]
    d = rng.normal(size=k); d[::3] = 0.0; d[1::4] = d[1]
    mats.append(np.diag(d))
    B = np.zeros((k, k)); h2 = k // 2
    A = rng.normal(size=(h2, h2)); B[:h2, :h2] = A @ A.T            # a block and a zero block
    mats.append(B)
    C = np.zeros((k, k))
    for i in range(0, k - 1, 2):
        C[i, i + 1] = C[i + 1, i] = 1.0                             # 2 x 2 antidiagonal blocks: p_1(0) = 0 exactly
    mats.append(C)
    W = np.diag(np.abs(np.arange(k) - (k - 1) / 2.0)) + np.diag(np.ones(k - 1), 1) + np.diag(np.ones(k - 1), -1)
    mats.append(W)                                                  # Wilkinson: pairs agreeing to ~1e-14
    T = np.diag(rng.normal(size=k)) + np.diag(np.full(k - 1, 1e-200), 1) + np.diag(np.full(k - 1, 1e-200), -1)
    mats.append(T)                                                  # couplings that underflow when squared
    u = rng.normal(size=k)
    mats.append(np.outer(u, u))                                     # rank one
    A = rng.normal(size=(k, k))
    mats.append(-(A @ A.T) - np.eye(k))                             # negative definite
    S = A @ A.T
    mats += [S * 1e-150, S * 1e150, S * 2.0 ** -1000, S + 1e8 * np.eye(k)]
    P = np.eye(k)[rng.permutation(k)]
    mats.append(P @ np.diag(np.r_[np.zeros(k - 2), 1.0, 1.0]) @ P.T)  # two ones and zeros, scrambled
    F = np.ascontiguousarray(np.stack(mats))
    lam = np.empty((len(mats), k))
    _lib.check(lib.crm_test_eigvalsh(h, len(mats), k, _lib.ptr(F), _lib.ptr(lam)))
    for i, M in enumerate(mats):
        ref = np.linalg.eigvalsh(M)
        scale = np.abs(ref).max()
        assert np.all(np.isfinite(lam[i])), i
        assert np.all(np.diff(lam[i]) >= 0), i
        assert np.abs(lam[i] - ref).max() <= 4e-14 * scale + (0.0 if scale > 0 else 1e-300), (i, np.abs(lam[i] - ref).max(), scale)


def test_davies_on_the_published_as155_case(ctx):
    """Q = 6 chi2_1 + 3 chi2_1 + chi2_1 (Imhof 1961 / Davies 1980, Table 1): P[Q < 7] = .4936,
    P[Q < 20] = .8760; at x = 1 the reference's settings (lim 10000, acc 1e-6) run out of terms and the
    modified-Liu value is returned instead."""
    from cellregmap_amd import _lib
    from oracle.davies import pvalue_from_weights

    lib, h = ctx
    lam = np.tile(np.array([1.0, 3.0, 6.0]), (3, 1))
    Q = np.array([7.0, 20.0, 1.0])
    pv = np.empty(3); ifault = np.empty(3, np.int32); liu = np.empty(3)
    _lib.check(lib.crm_test_davies(h, 3, 3, _lib.ptr(Q), _lib.ptr(lam), _lib.ptr(pv), _lib.ptr(ifault), _lib.ptr(liu)))
    assert abs(pv[0] - (1 - 0.4936)) < 1e-4 and abs(pv[1] - (1 - 0.8760)) < 1e-4
    assert list(ifault) == [0, 0, 1]
    p_ref, info = pvalue_from_weights(1.0, lam[2])
    assert info["ifault"] == 1
    assert_allclose(pv[2], p_ref, rtol=1e-8)
    assert_allclose(pv[2], liu[2], rtol=0, atol=0)


def test_davies_matches_oracle(ctx):
    from cellregmap_amd import _lib
    from oracle.davies import filter_weights, pvalue_from_weights

    lib, h = ctx
    rng = np.random.default_rng(5)
    k, count = 12, 64
    lam = np.sort(rng.gamma(0.7, 1.0, size=(count, k)), axis=1)
    lam[::7, : k - 3] *= 1e-9          # mostly filtered away
    lam[3::11, : k - 1] = -1e-12        # a single survivor -> Liu
    frac = rng.choice([0.2, 1.0, 2.0, 4.0, 8.0, 20.0, 60.0], size=count)
    Q = frac * lam.sum(1)
    pv = np.empty(count); ifault = np.empty(count, np.int32); liu = np.empty(count)
    _lib.check(lib.crm_test_davies(h, count, k, _lib.ptr(Q), _lib.ptr(lam), _lib.ptr(pv), _lib.ptr(ifault),
                                   _lib.ptr(liu)))
    for i in range(count):
        nonneg = lam[i][lam[i] >= 0]
        kept = lam[i][lam[i] > nonneg.mean() / 1e5]
        p_ref, info = pvalue_from_weights(Q[i], kept)
        # Davies' integral is a 0.5 - sum cancellation: absolute floor from the summation order
        assert abs(pv[i] - p_ref) <= 1e-5 * p_ref + 1e-13, (i, pv[i], p_ref, info)
        assert_allclose(liu[i], info["liu_pval"], rtol=1e-8, atol=1e-300)
        assert ifault[i] == info["ifault"]
