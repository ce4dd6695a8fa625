"""The C-ABI under wrong arguments: every case must come back with a status code and a text (include/crm_hip.h: 0 OK,
-1 HIP, -2 argument, -3 unsupported, -4 numeric, -5 internal) -- never a crash, never a silent success -- and the
context must remain usable afterwards.  Through ctypes, as a foreign caller would."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OK, ERR_ARG, ERR_UNSUPPORTED, ERR_NUMERIC = 0, -2, -3, -4


@pytest.fixture(scope="module")
def world():
    from cellregmap_amd import _lib
    from cellregmap_amd.synth import make_cohort

    lib = _lib.load()
    ctx, other = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.crm_ctx_create(0, ctypes.byref(ctx)) == OK
    assert lib.crm_ctx_create(0, ctypes.byref(other)) == OK
    c = make_cohort(6, 20, 3, 24, seed=9)
    n = c.y.size
    E, hK, rho = _lib.f64(c.E), _lib.f64(c.hK), _lib.f64(np.linspace(0, 1, 11))
    bg = ctypes.c_void_p()
    assert lib.crm_background_create(ctx, n, _lib.ptr(E), E.shape[1], _lib.ptr(hK), hK.shape[1], 11, _lib.ptr(rho), 0.0,
                                     ctypes.byref(bg)) == OK
    y, W = _lib.f64(c.y), _lib.f64(c.W)
    gene = ctypes.c_void_p()
    assert lib.crm_gene_create(bg, _lib.ptr(y), _lib.ptr(W), W.shape[1], _lib.ptr(E), E.shape[1], ctypes.byref(gene)) == OK
    G = _lib.f64(c.G)
    panel = ctypes.c_void_p()
    assert lib.crm_panel_create(ctx, n, _lib.ptr(G), G.shape[1], G.shape[1], ctypes.byref(panel)) == OK
    yield dict(lib=lib, _lib=_lib, ctx=ctx, other=other, c=c, n=n, E=E, hK=hK, rho=rho, bg=bg, y=y, W=W, gene=gene, G=G, panel=panel)
    lib.crm_gene_destroy(gene)
    lib.crm_panel_destroy(panel)
    lib.crm_background_destroy(bg)
    lib.crm_ctx_destroy(other)
    lib.crm_ctx_destroy(ctx)


def _scan(w, first, count, idx_E=None, idx_G=None, gene=None, panel=None):
    _lib = w["_lib"]
    pv = np.full(max(count, 1), -1.0)
    rc = w["lib"].crm_scan_interaction(gene or w["gene"], panel or w["panel"], first, count, _lib.ptr(idx_E), _lib.ptr(idx_G),
                                       _lib.ptr(pv), *([None] * 10))
    return rc, pv


def test_null_handles_and_pointers(world):
    lib, _lib = world["lib"], world["_lib"]
    out = ctypes.c_void_p()
    assert lib.crm_ctx_create(0, None) == ERR_ARG
    assert lib.crm_ctx_create(4096, ctypes.byref(out)) == ERR_ARG and b"not present" in lib.crm_last_error()
    assert lib.crm_ctx_synchronize(None) == ERR_ARG and lib.crm_ctx_trim(None) == ERR_ARG
    assert lib.crm_background_create(None, 10, _lib.ptr(world["E"]), 3, None, 0, 11, _lib.ptr(world["rho"]), 0.0,
                                     ctypes.byref(out)) == ERR_ARG
    assert lib.crm_background_create(world["ctx"], world["n"], None, 3, None, 0, 11, _lib.ptr(world["rho"]), 0.0,
                                     ctypes.byref(out)) == ERR_ARG
    assert lib.crm_gene_create(None, _lib.ptr(world["y"]), _lib.ptr(world["W"]), 1, _lib.ptr(world["E"]), 3,
                               ctypes.byref(out)) == ERR_ARG
    assert lib.crm_gene_create(world["bg"], None, _lib.ptr(world["W"]), 1, _lib.ptr(world["E"]), 3, ctypes.byref(out)) == ERR_ARG
    assert lib.crm_panel_create(world["ctx"], world["n"], None, 24, 24, ctypes.byref(out)) == ERR_ARG
    assert lib.crm_scan_interaction(None, world["panel"], 0, 1, *([None] * 13)) == ERR_ARG
    assert lib.crm_scan_interaction(world["gene"], None, 0, 1, *([None] * 13)) == ERR_ARG
    assert lib.crm_scan_association(None, world["panel"], 0, 1, 0, None, None, None) == ERR_ARG
    assert lib.crm_lmm_fit(None, 1, None, None) == ERR_ARG
    assert lib.crm_background_rank(None, 0) == -1 and lib.crm_background_rank(world["bg"], 99) == -1
    lib.crm_gene_destroy(None), lib.crm_panel_destroy(None), lib.crm_background_destroy(None), lib.crm_ctx_destroy(None)  # no-ops


def test_sizes_out_of_range(world):
    lib, _lib = world["lib"], world["_lib"]
    n, out = world["n"], ctypes.c_void_p()
    many = _lib.f64(np.linspace(0, 1, 17))
    assert lib.crm_background_create(world["ctx"], n, _lib.ptr(world["E"]), 3, None, 0, 17, _lib.ptr(many), 0.0,
                                     ctypes.byref(out)) == ERR_UNSUPPORTED and b"grid points" in lib.crm_last_error()
    # up to 128 covariate columns bind (the association scans and LMM fits take them: nullfit_xwide.hip); 129 do not
    Wwide = _lib.f64(np.random.default_rng(0).normal(size=(n, 129)))
    assert lib.crm_gene_create(world["bg"], _lib.ptr(world["y"]), _lib.ptr(Wwide), 129, _lib.ptr(world["E"]), 3,
                               ctypes.byref(out)) == ERR_UNSUPPORTED
    # up to 256 contexts bind (past 128 the interaction scan takes the slower kernel forms); 257 do not
    Ewide = _lib.f64(np.random.default_rng(1).normal(size=(n, 257)))
    assert lib.crm_gene_create(world["bg"], _lib.ptr(world["y"]), _lib.ptr(world["W"]), 1, _lib.ptr(Ewide), 257,
                               ctypes.byref(out)) == ERR_UNSUPPORTED
    assert lib.crm_gene_create(world["bg"], _lib.ptr(world["y"]), _lib.ptr(world["W"]), 0, _lib.ptr(world["E"]), 3,
                               ctypes.byref(out)) == ERR_UNSUPPORTED
    assert lib.crm_panel_create(world["ctx"], n, _lib.ptr(world["G"]), 10, 24, ctypes.byref(out)) == ERR_ARG     # ld < p
    assert lib.crm_panel_create(world["ctx"], 0, _lib.ptr(world["G"]), 24, 24, ctypes.byref(out)) == ERR_ARG
    assert lib.crm_set_block_variants(world["ctx"], -5) == ERR_ARG
    assert lib.crm_test_set_contraction(world["ctx"], 96, 1) == ERR_ARG


def test_variant_ranges_and_permutation_indices(world):
    lib, n = world["lib"], world["n"]
    for first, count in ((-1, 3), (0, 25), (20, 5), (24, 1), (0, -2)):
        rc, pv = _scan(world, first, count)
        assert rc == ERR_ARG and b"outside the panel" in lib.crm_last_error(), (first, count)
        assert np.all(pv == -1.0)                                   # outputs untouched
    rc, _ = _scan(world, 24, 0)                                     # an empty range at the end is fine
    assert rc == OK
    bad = np.arange(n, dtype=np.int32)
    bad[7] = n
    rc, _ = _scan(world, 0, 4, idx_E=bad)
    assert rc == ERR_ARG and b"permutation index" in lib.crm_last_error()
    bad[7] = -1
    rc, _ = _scan(world, 0, 4, idx_G=bad)
    assert rc == ERR_ARG


def test_objects_of_different_contexts_and_shapes(world):
    lib, _lib = world["lib"], world["_lib"]
    n = world["n"]
    foreign = ctypes.c_void_p()
    assert lib.crm_panel_create(world["other"], n, _lib.ptr(world["G"]), 24, 24, ctypes.byref(foreign)) == OK
    try:
        rc, _ = _scan(world, 0, 4, panel=foreign)
        assert rc == ERR_ARG and b"different contexts" in lib.crm_last_error()
        assert lib.crm_scan_association(world["gene"], foreign, 0, 4, 1, None, None, None) == ERR_ARG
    finally:
        lib.crm_panel_destroy(foreign)
    short = ctypes.c_void_p()
    Gs = _lib.f64(world["G"][: n - 8])
    assert lib.crm_panel_create(world["ctx"], n - 8, _lib.ptr(Gs), 24, 24, ctypes.byref(short)) == OK
    try:
        rc, _ = _scan(world, 0, 4, panel=short)
        assert rc == ERR_ARG and b"cells" in lib.crm_last_error()
    finally:
        lib.crm_panel_destroy(short)
    group = np.zeros(n, np.int32)
    group[3] = 9
    out = ctypes.c_void_p()
    Gd = _lib.f64(np.ones((6, 24)))
    assert lib.crm_panel_create_grouped(world["ctx"], n, _lib.ptr(group), 6, _lib.ptr(Gd), 24, 24, ctypes.byref(out)) == ERR_ARG
    assert b"group index" in lib.crm_last_error()


def test_non_finite_inputs(world):
    lib, _lib = world["lib"], world["_lib"]
    out = ctypes.c_void_p()
    ybad = world["y"].copy()
    ybad[5] = np.nan
    assert lib.crm_gene_create(world["bg"], _lib.ptr(ybad), _lib.ptr(world["W"]), 1, _lib.ptr(world["E"]), 3,
                               ctypes.byref(out)) == ERR_NUMERIC and b"non-finite" in lib.crm_last_error()
    Gbad = world["G"].copy()
    Gbad[2, 3] = np.inf
    grouped = ctypes.c_int(7)
    assert lib.crm_panel_create_auto(world["ctx"], world["n"], _lib.ptr(Gbad), 24, 24, None, 0, None, ctypes.byref(out),
                                     ctypes.byref(grouped)) == ERR_NUMERIC
    assert grouped.value == 0


def test_the_context_is_still_sound(world):
    """... after all of the above: a scan gives the results of the Python host."""
    from cellregmap_amd import CellRegMap, GenotypePanel

    rc, pv = _scan(world, 0, 24)
    assert rc == OK
    c = world["c"]
    from cellregmap_amd import _engine

    lib, host_ctx = world["_lib"].load(), _engine._context(0)     # (the Python host's own context, not the world's)
    crm = CellRegMap(c.y, c.E, W=c.W, hK=c.hK)
    # the world's background was built through the C-ABI without the donor structure of hK; the Python host announces it,
    # so compare bit for bit on the same (direct) route and to the north-star tolerance on the host's own
    world["_lib"].check(lib.crm_test_set_kinship_route(host_ctx, 0))
    try:
        ref, _ = crm.scan_interaction(GenotypePanel(c.G, groups=None), progress=False)
    finally:
        world["_lib"].check(lib.crm_test_set_kinship_route(host_ctx, 1))
    assert np.array_equal(pv, ref)
    host, _ = crm.scan_interaction(GenotypePanel(c.G, groups=None), progress=False)
    assert np.all(np.abs(host - pv) <= 1e-5 * pv + 1e-13)


def test_a_kinship_structure_the_half_factor_does_not_have_is_refused(world):
    """crm_background_set_kinship_groups checks the announcement against the background's own half factor entry by entry:
    other donor-level rows, or cells assigned to the wrong donor, are an argument error and the background keeps the direct
    route; the right announcement is accepted before and after."""
    from cellregmap_amd import _engine

    lib, _lib, n = world["lib"], world["_lib"], world["n"]
    bg = ctypes.c_void_p()
    assert lib.crm_background_create(world["ctx"], n, _lib.ptr(world["E"]), world["E"].shape[1], _lib.ptr(world["hK"]),
                                     world["hK"].shape[1], 11, _lib.ptr(world["rho"]), 0.0, ctypes.byref(bg)) == OK
    try:
        group, hKd = _engine._kinship_groups(world["c"].hK)
        group, hKd, ones = np.ascontiguousarray(group, np.int32), _lib.f64(hKd), _lib.f64(np.ones((n, 1)))
        announce = lambda g, h: lib.crm_background_set_kinship_groups(bg, _lib.ptr(g), h.shape[0], _lib.ptr(h), h.shape[1],
                                                                      _lib.ptr(ones), 1)
        assert lib.crm_background_kinship_groups(bg) == 0
        assert announce(group, hKd) == OK and lib.crm_background_kinship_groups(bg) == hKd.shape[0]
        wrong = hKd.copy()
        wrong[1, 0] += 1e-6
        assert announce(group, wrong) == ERR_ARG and b"half factor" in lib.crm_last_error()
        assert lib.crm_background_kinship_groups(bg) == 0          # ... and the first announcement is gone with it
        swapped = group.copy()
        swapped[0] = (swapped[0] + 1) % hKd.shape[0]
        assert announce(swapped, hKd) == ERR_ARG
        out_of_range = group.copy()
        out_of_range[3] = hKd.shape[0]
        assert announce(out_of_range, hKd) == ERR_ARG and b"outside" in lib.crm_last_error()
        assert announce(group, hKd) == OK and lib.crm_background_kinship_groups(bg) == hKd.shape[0]
    finally:
        lib.crm_background_destroy(bg)


def test_out_of_device_memory_is_a_status_code(world):
    """A background that cannot fit (one grid point of 4 000 000 cells x 10 112 padded columns = 324 GB on a 288 GB device):
    the allocation fails inside the library, which releases its idle caches, tries once more and then reports
    CRM_ERR_HIP with the size in the text; nothing is read from the host pointers, nothing leaks into the next
    call."""
    lib, _lib = world["lib"], world["_lib"]
    rho = _lib.f64(np.array([1.0]))
    r = np.array([10000], np.int32)
    dummy = np.zeros(8)
    PP = ctypes.c_void_p * 1
    out = ctypes.c_void_p()
    rc = lib.crm_background_create_qs(world["ctx"], 4_000_000, 1, _lib.ptr(rho), _lib.ptr(r), PP(dummy.ctypes.data),
                                      PP(dummy.ctypes.data), ctypes.byref(out))
    assert rc == -1 and not out.value, rc
    assert b"device allocation of 323584000000 bytes failed" in lib.crm_last_error(), lib.crm_last_error()
    rc, pv = _scan(world, 0, 24)
    assert rc == OK and np.all((pv > 0) & (pv <= 1))


def test_the_overrun_detector_detects(world):
    """CRM_POISON=1 puts 4 KiB of 0xFF behind every device buffer and inspects it on release; the self-test writes eight
    bytes past a scratch buffer on purpose.  In a poison run it must be counted (and is then subtracted again by the
    session's bookkeeping: tests/conftest.py), in a normal run nothing is instrumented."""
    import os

    lib = world["lib"]
    before = lib.crm_test_overruns()
    grew = lib.crm_test_overrun_selftest(world["ctx"])
    poison = os.environ.get("CRM_POISON", "0") not in ("", "0")
    assert grew == (1 if poison else 0)
    assert lib.crm_test_overruns() == before + grew
    if poison:
        os.environ["CRM_TEST_EXPECTED_OVERRUNS"] = str(int(os.environ.get("CRM_TEST_EXPECTED_OVERRUNS", "0")) + 1)
