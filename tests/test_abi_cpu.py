"""The C-ABI library loads without a GPU and exports every symbol include/crm_hip.h declares;
the host-side mirror keeps the reference's public surface."""
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    found = set()
    for header in ("crm_hip.h", "crm_hip_test.h"):  # the boundary, and the unit-test hooks beside it
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        found |= set(re.findall(r"\b(crm_[a-z_0-9]+)\s*\(", text))
    return sorted(found)


def test_library_exports_every_declared_symbol():
    from cellregmap_amd import _lib

    lib = _lib.load()
    declared = _declared_symbols()
    assert declared, "no prototypes found in include/*.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.crm_version().decode().count(".") == 2


def test_public_surface_matches_reference():
    import cellregmap_amd as pkg

    # cellregmap/__init__.py:1-20
    for name in ("CellRegMap", "run_association", "run_association_fast", "run_interaction", "estimate_betas",
                 "Term", "__version__"):
        assert hasattr(pkg, name)
    sig = inspect.signature(pkg.CellRegMap.__init__)
    assert list(sig.parameters)[:7] == ["self", "y", "E", "W", "Ls", "E1", "hK"]  # _cellregmap.py:63
    sig = inspect.signature(pkg.CellRegMap.scan_interaction)
    assert list(sig.parameters)[:4] == ["self", "G", "idx_E", "idx_G"]            # :317-319
    sig = inspect.signature(pkg.run_interaction)
    assert list(sig.parameters)[:8] == ["y", "E", "G", "W", "E1", "E2", "hK", "idx_G"]  # :547
    sig = inspect.signature(pkg.run_association)
    assert list(sig.parameters)[:5] == ["y", "W", "E", "G", "hK"]                 # :471
    assert pkg.Term.FIXED.value == 1 and pkg.Term.RANDOM.value == 2


def test_get_L_values_is_the_hadamard_factorisation():
    # proof.md: sum_i L_i L_i' == K o EE'
    from cellregmap_amd import get_L_values

    rng = np.random.default_rng(0)
    hK = rng.normal(size=(30, 4))
    E = rng.normal(size=(30, 3))
    Ls = get_L_values(hK, E)
    lhs = sum(L @ L.T for L in Ls)
    assert np.allclose(lhs, (hK @ hK.T) * (E @ E.T), atol=1e-10)
    # what the device is handed in the place of U S: the contexts themselves -- the same covariance in another basis of
    # the same column space (cellregmap_amd/_engine.py: HadamardHalves.device_us); the list the caller sees is unchanged
    assert Ls.device_us is not Ls.us and np.array_equal(Ls.device_us, E)
    U, S, _ = np.linalg.svd(E, full_matrices=False)
    assert np.allclose(np.abs(Ls.us), np.abs(U * S), atol=1e-12)
    dev = sum((Ls.device_us[:, [i]] * hK) @ (Ls.device_us[:, [i]] * hK).T for i in range(E.shape[1]))
    assert np.allclose(dev, lhs, atol=1e-10)
    # contexts of deficient rank: U S has fewer columns than E and stays what the device gets
    E_def = np.concatenate([E, E[:, :1] + E[:, 1:2]], axis=1)
    Ld = get_L_values(hK, E_def)
    assert Ld.us.shape[1] == 3 and Ld.device_us is Ld.us


def test_no_gpu_means_loud_failure():
    """Without a GPU the product path must raise, never fall back to a CPU computation."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cellregmap_amd import CellRegMap, _lib

    rng = np.random.default_rng(1)
    with pytest.raises(_lib.CrmError):
        CellRegMap(rng.normal(size=20), rng.normal(size=(20, 2)))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "cellregmap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libcrm_oracle" not in src, f


def test_detect_groups_finds_exactly_the_donor_structure():
    from cellregmap_amd import detect_groups

    rng = np.random.default_rng(0)
    Gd = rng.integers(0, 3, size=(7, 300)).astype(float)
    donor = rng.integers(0, 7, size=90)
    donor[:7] = np.arange(7)
    G = Gd[donor]
    group, reps = detect_groups(G)
    assert group.dtype == np.int32 and len(reps) == 7
    assert np.array_equal(G[reps][group], G)
    # one cell deviating in one (unsampled) column breaks the structure -> no collapse
    G2 = G.copy()
    G2[50, 123] += 1.0
    found = detect_groups(G2)
    assert found is None or np.array_equal(G2[found[1]][found[0]], G2)
    assert detect_groups(rng.normal(size=(40, 20))) is None


def test_a_cpp_exception_stops_at_the_boundary():
    """include/crm_hip.h: "no exceptions across the boundary".  An absurd donor count makes std::vector throw
    std::length_error inside crm_panel_create_grouped_i8 before any GPU call (so this runs without a GPU); the
    caller must see a status code and a text, not std::terminate."""
    import ctypes

    from cellregmap_amd import _lib

    lib = _lib.load()
    fake_ctx = ctypes.create_string_buffer(4096)          # never dereferenced before the allocation that throws
    group = np.zeros(1, np.int32)
    dosage = np.zeros(1, np.int8)
    out = ctypes.c_void_p()
    rc = lib.crm_panel_create_grouped_i8(ctypes.cast(fake_ctx, ctypes.c_void_p), 1, _lib.ptr(group), 1 << 61,
                                         _lib.ptr(dosage), 1, 1, 0, ctypes.byref(out))
    assert rc == -5 and not out.value                     # CRM_ERR_INTERNAL
    msg = lib.crm_last_error().decode()
    assert "crm_panel_create_grouped_i8" in msg and "exception" in msg or "memory" in msg
    with pytest.raises(_lib.CrmError, match="crm_panel_create_grouped_i8"):
        _lib.check(rc)


def test_every_entry_point_runs_behind_the_exception_guard():
    """Structural: each `int crm_*` the headers declare is defined with its body inside crm::guarded(...) or
    crm::guarded_on(..., context, ...) -- the latter also holds the context's lock -- (one-line getters that cannot throw
    excepted), each `void crm_*_destroy` inside try / catch (...)."""
    csrc = os.path.join(ROOT, "cellregmap_amd", "csrc")
    text = "\n".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith(".hip"))
    exempt = {"crm_last_error", "crm_version", "crm_test_tail_launches", "crm_test_spectrum_tail_launches", "crm_test_dense_repeats", "crm_test_sync_fallbacks",
              "crm_test_donor_pair_blocks", "crm_test_tests_without_pair",
              "crm_test_overruns",
              "crm_background_kinship_groups", "crm_background_kinship_folded"}
    for name in _declared_symbols():
        if name in exempt:
            continue
        m = re.search(r'^(?:extern "C" )?(int|void) %s\((?:[^{;]|\n)*\{\n(.*)$' % name, text, flags=re.M)
        assert m, f"definition of {name} not found"
        first = m.group(2).strip()
        if m.group(1) == "int":
            assert (first.startswith('return crm::guarded("%s"' % name) or
                    first.startswith('return crm::guarded_on("%s"' % name)), f"{name}: {first}"
        else:
            assert first.startswith("try {"), f"{name}: {first}"


def test_the_committed_traffic_profile_is_of_this_library():
    """bench.py quotes `roofline.traffic` from profiles/r06_pmc_summary.json only when the profile's kernel form (library
    version, route, tile walk) is the running one; a version bump without new counter passes would silently turn the figure
    into null in the driver's bench line -- fail here instead."""
    import json
    import os

    import cellregmap_amd
    from cellregmap_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    form = json.load(open(os.path.join(root, "profiles", "r06_pmc_summary.json")))["kernel_form"]
    version = _lib.load().crm_version().decode()
    assert version == cellregmap_amd.__version__
    assert form == {"contraction_sync": True, "tail_launch": True, "library": version, "kinship_route": True, "tile_band": 8}


def test_kinship_groups_of_expanded_factors():
    """``_engine._kinship_groups`` (what the Python host announces through crm_background_set_kinship_groups): exact donor
    structure or nothing.  Indicator factors (one 1 per row: every sampled column looks alike) and cells in shuffled order
    are found; rows that agree only to rounding, and factors without repeated rows, are not."""
    from cellregmap_amd import _engine

    rng = np.random.default_rng(4)
    donors, cells = 23, 400
    donor = rng.integers(0, donors, size=cells)
    donor[:donors] = rng.permutation(donors)                    # every donor present, in no particular order
    for hKd in (rng.normal(size=(donors, donors)),              # dense donor-level factor
                np.eye(donors),                                 # indicator factor (unrelated donors), 23 columns
                np.eye(120)[:donors]):                          # ... with more than 64 columns: the sampled pass sees mostly zeros
        hK = np.ascontiguousarray(hKd[donor])
        group, rows = _engine._kinship_groups(hK)
        assert group.dtype == np.int32 and group.shape == (cells,) and rows.shape == hKd.shape
        assert np.array_equal(rows[group], hK)
        assert len(np.unique(group)) == donors
    # twins: two donors with the same row fall into one group -- the structure H = us o hKd[group] holds all the same
    twins = rng.normal(size=(donors, 9))
    twins[5] = twins[11]
    group, rows = _engine._kinship_groups(np.ascontiguousarray(twins[donor]))
    assert rows.shape[0] == donors - 1 and np.array_equal(rows[group], twins[donor])
    # a row that differs from its donor's in the last bit is a row of its own
    hK = np.ascontiguousarray(rng.normal(size=(donors, 12))[donor])
    hK[37, 7] = np.nextafter(hK[37, 7], np.inf)
    group, rows = _engine._kinship_groups(hK)
    assert rows.shape[0] == donors + 1 and np.array_equal(rows[group], hK)
    # no repeated rows at all
    assert _engine._kinship_groups(rng.normal(size=(300, 10))) is None


def test_stream_chunk_setting(monkeypatch):
    """CELLREGMAP_AMD_STREAM_CHUNK: variants per chunk of the streamed scan of a host matrix; 0 disables, junk falls back."""
    from cellregmap_amd import _engine

    monkeypatch.delenv("CELLREGMAP_AMD_STREAM_CHUNK", raising=False)
    assert _engine._stream_chunk() == 8192
    for text, want in (("0", 0), ("4096", 4096), ("-7", 0), ("many", 8192)):
        monkeypatch.setenv("CELLREGMAP_AMD_STREAM_CHUNK", text)
        assert _engine._stream_chunk() == want
