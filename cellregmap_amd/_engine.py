"""Host-side mirror of the reference's scan API on top of libcrm_hip.so.

Same names, argument meaning, defaults, positional quirks and error behaviour as
``cellregmap/_cellregmap.py`` (reference file:line cited per method); everything
numerical happens in the HIP library through the C-ABI of ``include/crm_hip.h``.
There is no CPU path here: without the library or without a GPU, calls raise.
"""
import ctypes
import threading
import hashlib
import weakref
from collections import OrderedDict
from collections.abc import Sequence
from typing import Optional

import numpy as np

from . import _lib

_RHO_GRID = np.linspace(0, 1, 11)
_SQRT_EPS = float(np.sqrt(np.finfo(float).eps))

_contexts = {}


def _context(device=0):
    """One library context (device + stream + workspace) per device, created lazily."""
    if device not in _contexts:
        lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(lib.crm_ctx_create(int(device), ctypes.byref(h)))
        _contexts[device] = h
    return _contexts[device]


def release_workspaces(device=0):
    """Hand the cached device work buffers of ``device``'s context back (the constructor keeps its eigen-solver
    workspace between calls; it is released on its own when memory runs short)."""
    _lib.check(_lib.load().crm_ctx_trim(_context(device)))


def _economic_svd(X):
    """Thin SVD with singular values < sqrt(eps) dropped (numpy_sugar.economic_svd,
    used at _cellregmap.py:540)."""
    U, s, Vt = np.linalg.svd(np.asarray(X, float), full_matrices=False)
    ok = s >= _SQRT_EPS
    return U[:, ok], s[ok], Vt[ok, :]


class HadamardHalves(Sequence):
    """The list ``[diag(us[:, i]) @ hK for i]`` that ``get_L_values`` returns, kept in factored form:
    indexing / iterating yields the same arrays as the reference's list, but ``CellRegMap`` hands the
    two factors to the device and never materialises the n x (k*m) concatenation on the host.

    ``us`` may be given as a callable: it is then computed on first use (``get_L_values`` with contexts of full column
    rank: the device takes the contexts themselves, below, and the thin SVD -- a tenth of a second at 20 000 cells -- is
    only ever needed by a caller that indexes the list)."""

    def __init__(self, us, hK, contexts=None, columns=None):
        self._us = None if callable(us) else np.ascontiguousarray(us, dtype=float)
        self._make_us = us if callable(us) else None
        self.hK = np.ascontiguousarray(hK, dtype=float)
        self._k2 = int(columns) if self._us is None else self._us.shape[1]
        # What the device is handed in the place of ``us``.  us = U S = E V with V orthogonal when E has full column rank,
        # so [diag(E[:, i]) hK for i] is the same covariance sum_i L_i L_i' = K o EE' in another basis of the same column
        # space (H -> H blockdiag(I, V (x) I): same Gram spectrum, same Q0 S0 Q0').  In that basis the per-donor sums of the
        # kinship-structure route against the kinship term's contexts are sums against the scan's own contexts -- symmetric
        # when E2 = E, the reference's default -- which halves their product (csrc/scan.hip: donor pairs).
        self._device_us = None
        if contexts is not None:
            E = np.ascontiguousarray(contexts, dtype=float)
            if E.shape == (self.hK.shape[0], self._k2):
                self._device_us = E

    @property
    def us(self):
        if self._us is None:
            self._us = np.ascontiguousarray(self._make_us(), dtype=float)
            self._make_us = None
        return self._us

    @property
    def device_us(self):
        return self._device_us if self._device_us is not None else self.us

    @property
    def shape_us(self):
        """(n, columns of us) without forming it."""
        return self.hK.shape[0], self._k2

    def __len__(self):
        return self._k2

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        return self.us[:, [i]] * self.hK


_us_cache = OrderedDict()  # digest of E -> U * S (an eQTL run calls get_L_values with one E per gene)


def _cached_us(E, key):
    us = _us_cache.get(key)
    if us is None:
        U, S, _ = _economic_svd(E)
        us = U * S
        _us_cache[key] = us
        while len(_us_cache) > 4:
            _us_cache.popitem(last=False)
    else:
        _us_cache.move_to_end(key)
    return us


def get_L_values(hK, E):
    """L_i = diag((U S)[:, i]) hK with U, S from the economic SVD of E
    (_cellregmap.py:533-545); sum_i L_i L_i' = K o EE' (proof.md).

    The SVD is taken when somebody looks at the list (``HadamardHalves``) -- or right away unless the contexts are
    certainly of full column rank by the reference's rule (numpy_sugar.economic_svd keeps singular values >= sqrt(eps)):
    certified here from the Gram matrix E'E, whose eigenvalues are the squared singular values to ~1e-12 of the largest,
    by asking for a smallest one above 1e-8 of the largest AND above 1e-6 -- both far on the safe side of the rule."""
    E = np.asarray(E, float)
    key = _digest(E)
    if E.ndim == 2 and E.shape[0] >= E.shape[1] >= 1 and key not in _us_cache:
        lam = np.linalg.eigvalsh(E.T @ E)
        if np.all(np.isfinite(lam)) and lam[0] > 1e-8 * lam[-1] and lam[0] > 1e-6:
            return HadamardHalves(lambda: _cached_us(E, key), np.asarray(hK, float), contexts=E, columns=E.shape[1])
    return HadamardHalves(_cached_us(E, key), np.asarray(hK, float), contexts=E)


class _Background:
    """Owner of a ``crm_background`` handle (device-resident Q0/S0 per rho)."""

    def __init__(self, handle, rho, device):
        self.handle = handle
        self.rho = np.asarray(rho, float)
        self.device = device
        self._fin = weakref.finalize(self, _lib.load().crm_background_destroy, handle)

    def rank(self, i):
        return _lib.load().crm_background_rank(self.handle, i)

    def read(self, i, n):
        r = self.rank(i)
        Q0 = np.empty((n, r))
        S0 = np.empty(r)
        _lib.check(_lib.load().crm_background_read(self.handle, i, _lib.ptr(Q0), _lib.ptr(S0)))
        return Q0, S0


_bg_cache = OrderedDict()
BACKGROUND_CACHE_SIZE = 2


try:  # ~10 GB/s; the cache key of a config-3 background covers 32 MB per CellRegMap(...)
    import xxhash as _xxhash
except ImportError:  # pragma: no cover
    _xxhash = None


def _digest(*arrays):
    h = _xxhash.xxh3_128() if _xxhash is not None else hashlib.blake2b(digest_size=16)
    for a in arrays:
        if a is None:
            h.update(b"-")
        else:
            h.update(str(a.shape).encode())
            h.update(np.ascontiguousarray(a).view(np.uint8).data)
    return h.hexdigest()


def _make_background(E1, B, rho, device, rel_tol=0.0, cache=True):
    if isinstance(B, HadamardHalves):
        return _make_background_hadamard(E1, B, rho, device, rel_tol, cache)
    return _make_background_dense(E1, B, rho, device, rel_tol, cache)


_kin_cache = OrderedDict()   # digest of hK -> (group of cell, donor-level rows) or None


def _kinship_groups(hK):
    """Donor structure of an "expanded" kinship factor (_cellregmap.py:559: hK is n x m with the rows of a donor-level
    factor repeated for the cells of each donor): ``(group int32 (n,), hKd (groups, m))`` with ``hK == hKd[group]`` exactly,
    or ``None`` when the rows do not repeat (more than n / 2 or 2048 distinct rows)."""
    key = _digest(hK)
    if key in _kin_cache:
        _kin_cache.move_to_end(key)
        return _kin_cache[key]
    found = None
    # (a sample of the columns first: rows that already differ there -- the concatenated halves of an explicit mode C list --
    # are dismissed for the price of n x 64 entries; then every column: an indicator factor has one 1 per row)
    if hK.shape[1] <= 64 or candidate_groups(hK, max_groups=2048, sample_columns=64) is not None:
        found = detect_groups(hK, max_groups=2048, sample_columns=hK.shape[1])
    out = None
    if found is not None:
        group, reps = found
        out = (np.ascontiguousarray(group, dtype=np.int32), np.ascontiguousarray(hK[reps], dtype=float))
    _kin_cache[key] = out
    while len(_kin_cache) > 4:
        _kin_cache.popitem(last=False)
    return out


def _announce_kinship_groups(bg, halves):
    """Tell the library about the donor structure of the kinship factor (``crm_background_set_kinship_groups``): the
    dense scan then forms H'(g o E0) donor by donor instead of contracting every variant against Q0(rho*) over all cells.
    ``halves``: the factored halves of mode C, or -- mode B, ``hS = [sqrt(rho) E1, sqrt(1 - rho) hK]`` -- the kinship factor
    itself, which is the same structure with a single column of ones in the place of ``us``."""
    if isinstance(halves, HadamardHalves):
        hK, us = halves.hK, halves.device_us
    else:
        hK = np.ascontiguousarray(halves, dtype=float)
        us = np.ones((hK.shape[0], 1))
    found = _kinship_groups(hK)
    if found is None:
        return
    group, hKd = found
    us = _lib.f64(us)
    rc = _lib.load().crm_background_set_kinship_groups(bg.handle, _lib.ptr(group), hKd.shape[0], _lib.ptr(hKd),
                                                       hKd.shape[1], _lib.ptr(us), us.shape[1])
    if rc != 0:
        # The announcement is an optimisation: a structure the library declines (CRM_ERR_UNSUPPORTED) or scratch memory it
        # cannot get leaves the background as it is, scanning by the direct route (include/crm_hip.h) -- say so, do not
        # fail the constructor.  Anything else (the library's own check of the announcement against its half factor,
        # CRM_ERR_ARG; a HIP error) is a defect of the detection above or of the device state and is raised.
        import warnings

        msg = _lib.load().crm_last_error()
        text = msg.decode() if msg else ""
        if rc != -3 and "out of memory" not in text.lower():
            raise _lib.CrmError(f"libcrm_hip error {rc}: {text}")
        warnings.warn("kinship structure not used (libcrm_hip status %d: %s); scans take the direct route" % (rc, text),
                      RuntimeWarning, stacklevel=3)


def _make_background_hadamard(E1, halves, rho, device, rel_tol, cache):
    lib = _lib.load()
    key = (device, "hadamard", _digest(E1, halves.device_us, halves.hK), tuple(np.asarray(rho, float)), rel_tol)
    if cache and key in _bg_cache:
        _bg_cache.move_to_end(key)
        return _bg_cache[key]
    ctx = _context(device)
    E1c, us, hK, rho = _lib.f64(E1), halves.device_us, halves.hK, _lib.f64(rho)
    h = ctypes.c_void_p()
    finder = None
    if cache:
        # the donor structure of the kinship factor is looked for on the host (row hashes, then entry by entry) while the
        # device decomposes: the call below releases the GIL for its 0.8 s
        finder = threading.Thread(target=_kinship_groups, args=(hK,), daemon=True)
        finder.start()
    try:
        _lib.check(lib.crm_background_create_hadamard(ctx, E1c.shape[0], _lib.ptr(E1c), E1c.shape[1], _lib.ptr(us),
                                                      us.shape[1], _lib.ptr(hK), hK.shape[1], rho.shape[0],
                                                      _lib.ptr(rho), float(rel_tol), ctypes.byref(h)))
    finally:
        if finder is not None:
            finder.join()
    bg = _Background(h, rho, device)
    if cache:   # (the per-SNP backgrounds of the effect-size path are never scanned: nothing to announce)
        _announce_kinship_groups(bg, halves)        # (finds the structure in the cache the thread filled)
    if cache:
        _bg_cache[key] = bg
        while len(_bg_cache) > BACKGROUND_CACHE_SIZE:
            _bg_cache.popitem(last=False)
    return bg


def _make_background_dense(E1, B, rho, device, rel_tol=0.0, cache=True):
    """hS(rho) = [sqrt(rho) E1, sqrt(1-rho) B] -> economic eigendecompositions on the device.
    Re-used across objects with identical (E1, B, rho): the reference redoes the 11
    decompositions per ``CellRegMap(...)``, i.e. per gene."""
    lib = _lib.load()
    key = (device, _digest(E1, B), tuple(np.asarray(rho, float)), rel_tol) if cache else None
    if cache and key in _bg_cache:
        _bg_cache.move_to_end(key)
        return _bg_cache[key]
    ctx = _context(device)
    E1c = _lib.f64(E1)
    Bc = None if B is None else _lib.f64(B)
    rho = _lib.f64(rho)
    h = ctypes.c_void_p()
    _lib.check(lib.crm_background_create(ctx, E1c.shape[0], _lib.ptr(E1c), E1c.shape[1], _lib.ptr(Bc),
                                         0 if Bc is None else Bc.shape[1], rho.shape[0], _lib.ptr(rho),
                                         float(rel_tol), ctypes.byref(h)))
    bg = _Background(h, rho, device)
    if cache and Bc is not None:   # mode B: B is the kinship factor itself (anything else is dismissed on a column sample)
        _announce_kinship_groups(bg, Bc)
    if cache:
        _bg_cache[key] = bg
        while len(_bg_cache) > BACKGROUND_CACHE_SIZE:
            _bg_cache.popitem(last=False)
    return bg


class BackgroundBuilder:
    """The constructor's three phases (``crm_background_begin / _complete / _seal``) for a process that owns
    only the grid points flagged in ``mine``; the others arrive through ``import_slot``
    (``cellregmap_amd.distributed.sharded_background`` drives several of these, one per GPU)."""

    SLOTS = {"Q0": 0, "S0": 1, "Mix": 2}

    def __init__(self, E1, B, rho, device=0, mine=None, rel_tol=0.0):
        lib = _lib.load()
        self.device = device
        self.rho = _lib.f64(rho)
        nrho = self.rho.shape[0]
        flags = np.ones(nrho, np.int32) if mine is None else np.ascontiguousarray(mine, dtype=np.int32)
        E1c = _lib.f64(E1)
        h = ctypes.c_void_p()
        self._halves = B if (isinstance(B, HadamardHalves) or B is not None) else None
        # (the donor structure of the kinship factor is looked for on the host while the device decomposes, as in
        # _make_background_hadamard; seal() then finds it in the cache)
        hK_host = B.hK if isinstance(B, HadamardHalves) else (np.ascontiguousarray(B, dtype=float) if B is not None else None)
        finder = None
        if hK_host is not None:
            finder = threading.Thread(target=_kinship_groups, args=(hK_host,), daemon=True)
            finder.start()
        try:
            self._begin(lib, E1c, B, nrho, flags, rel_tol, h, device)
        finally:
            if finder is not None:
                finder.join()
        self._bg = _Background(h, self.rho, device)   # (owns the handle from here on)
        self.nrho = nrho

    def _begin(self, lib, E1c, B, nrho, flags, rel_tol, h, device):
        if isinstance(B, HadamardHalves):
            _lib.check(lib.crm_background_begin(_context(device), E1c.shape[0], _lib.ptr(E1c), E1c.shape[1], None,
                                                B.shape_us[1] * B.hK.shape[1], _lib.ptr(B.device_us), B.shape_us[1],
                                                _lib.ptr(B.hK), B.hK.shape[1], nrho, _lib.ptr(self.rho), _lib.ptr(flags),
                                                float(rel_tol), ctypes.byref(h)))
        else:
            Bc = None if B is None else _lib.f64(B)
            _lib.check(lib.crm_background_begin(_context(device), E1c.shape[0], _lib.ptr(E1c), E1c.shape[1], _lib.ptr(Bc),
                                                0 if Bc is None else Bc.shape[1], None, 0, None, 0, nrho,
                                                _lib.ptr(self.rho), _lib.ptr(flags), float(rel_tol), ctypes.byref(h)))

    def rank(self, i):
        """Rank of an owned grid point after ``begin``; -1 for the others."""
        return _lib.load().crm_background_rank(self._bg.handle, i)

    def complete(self, ranks):
        ranks = np.ascontiguousarray(ranks, dtype=np.int32)
        _lib.check(_lib.load().crm_background_complete(self._bg.handle, _lib.ptr(ranks)))

    def layout(self):
        """Slots to exchange and their sizes in doubles: {"S0": ldq, "Mix": ldh * ldq} when the background keeps its
        half factor (thin branch), else {"Q0": n_pad * ldq, "S0": ldq}."""
        n_pad, ldq, ldh, mix = ctypes.c_long(), ctypes.c_long(), ctypes.c_long(), ctypes.c_int()
        _lib.check(_lib.load().crm_background_layout(self._bg.handle, ctypes.byref(n_pad), ctypes.byref(ldq),
                                                     ctypes.byref(ldh), ctypes.byref(mix)))
        if mix.value:   # thin branch: every rank holds H and forms Q0 = H Mix itself, on first use
            return {"S0": ldq.value, "Mix": ldh.value * ldq.value}
        return {"Q0": n_pad.value * ldq.value, "S0": ldq.value}

    def export_slot(self, i, what, tensor):
        """Copy slot ``what`` of grid point i into ``tensor`` (float64; on this GPU, or a CPU tensor under gloo)."""
        _lib.check(_lib.load().crm_background_export(self._bg.handle, i, self.SLOTS[what], ctypes.c_void_p(tensor.data_ptr())))

    def import_slot(self, i, what, tensor):
        _lib.check(_lib.load().crm_background_import(self._bg.handle, i, self.SLOTS[what], ctypes.c_void_p(tensor.data_ptr())))

    def seal(self):
        _lib.check(_lib.load().crm_background_seal(self._bg.handle))
        if self._halves is not None:
            _announce_kinship_groups(self._bg, self._halves)
        return self._bg


def background_from_qs(qs_list, rho, device=0):
    """Background from precomputed ``((Q0,), S0)`` pairs (one per rho), e.g. LAPACK's."""
    lib = _lib.load()
    ctx = _context(device)
    Q0s = [_lib.f64(q[0][0]) for q in qs_list]
    S0s = [_lib.f64(q[1]) for q in qs_list]
    n = Q0s[0].shape[0]
    r = np.asarray([q.shape[1] for q in Q0s], np.int32)
    rho = _lib.f64(rho)
    PP = ctypes.c_void_p * len(Q0s)
    qp = PP(*[q.ctypes.data for q in Q0s])
    sp = PP(*[s.ctypes.data for s in S0s])
    h = ctypes.c_void_p()
    _lib.check(lib.crm_background_create_qs(ctx, n, len(Q0s), _lib.ptr(rho), _lib.ptr(r), qp, sp,
                                            ctypes.byref(h)))
    return _Background(h, rho, device)


_projections = {}


def _projection(k):
    if k not in _projections:
        _projections[k] = np.random.default_rng(0x5EED).integers(1, 2 ** 63, size=k, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    return _projections[k]


def candidate_groups(G, max_groups=2048, sample_columns=64):
    """Candidate donor structure of an expanded genotype matrix from a column sample: cells with
    identical sampled entries share a group.  Returns ``(group_of_cell int32 (n,), representative
    row per group int64 (m,))`` -- groups labelled in order of first appearance -- or ``None`` when the
    sample already shows more than ``max_groups`` (or n/2) distinct rows.  The candidates still have
    to be verified on every column (``detect_groups`` on the host, ``crm_panel_create_auto`` on the
    device)."""
    G = np.asarray(G)
    n, p = G.shape
    if n < 2 or p < 1:
        return None
    cols = np.unique(np.linspace(0, p - 1, min(p, sample_columns)).astype(int))
    key = np.ascontiguousarray(G[:, cols], dtype=float)
    # rows are grouped through a 64-bit hash of their bit patterns (wrap-around integer arithmetic: exact
    # and order-independent, unlike a floating-point projection) and the grouping is then checked on the
    # sampled entries; a row-wise np.unique, ~10x slower, only serves as the fallback for a collision
    key += 0.0  # -0.0 -> +0.0
    bits = key.view(np.uint64)
    # (the bit patterns of small integers as doubles end in ~50 zero bits -- 1.0 is 0x3FF0000000000000 -- and a product
    # with an odd multiplier keeps that many zero bits: mix the pattern first, splitmix64's finaliser, so that an
    # indicator or dosage matrix does not collide into a few thousand hash values and fall back to the row-wise unique)
    with np.errstate(over="ignore"):
        mixed = bits ^ (bits >> np.uint64(30))
        mixed = mixed * np.uint64(0xBF58476D1CE4E5B9)
        mixed ^= mixed >> np.uint64(27)
        mixed = mixed * np.uint64(0x94D049BB133111EB)
        mixed ^= mixed >> np.uint64(31)
        proj = (mixed * _projection(key.shape[1])).sum(axis=1, dtype=np.uint64)
    _, first, inv = np.unique(proj, return_index=True, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    if not (bits == bits[first[inv]]).all():
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
    m = first.size
    if m > max_groups or m > n // 2:
        return None
    order = np.argsort(first, kind="stable")
    relabel = np.empty(m, np.int32)
    relabel[order] = np.arange(m, dtype=np.int32)
    return relabel[inv].astype(np.int32), first[order].astype(np.int64)


def detect_groups(G, max_groups=2048, sample_columns=64, chunk=2048):
    """``candidate_groups`` followed by an exact host-side check of every column.  Returns
    ``(group_of_cell, representative rows)`` or ``None`` when the rows do not collapse."""
    G = np.asarray(G)
    found = candidate_groups(G, max_groups, sample_columns)
    if found is None:
        return None
    group, reps = found
    rep = reps[group]
    for c0 in range(0, G.shape[1], chunk):
        blk = G[:, c0:c0 + chunk]
        if not np.array_equal(blk, blk[rep]):
            return None
    return group, reps


_last_grouping = {}   # cells -> (group of cell int32, representative rows int64) of the last donor-constant panel


class GenotypePanel:
    """A genotype matrix (n x p) resident in HBM; build once, scan many genes against it.

    ``groups``: ``"auto"`` (default) looks for the donor structure of expanded genotypes
    (``detect_groups``) and, when found, stores one row per donor -- scans then run the exact
    donor-collapsed path; ``None`` keeps the matrix dense (general G)."""

    def __init__(self, G, device=0, groups="auto"):
        lib = _lib.load()
        G = np.asarray(G, float)
        assert G.ndim == 2
        # a block of columns of a row-major matrix (rows a fixed number of doubles apart) goes to the library as it lies --
        # the C-ABI takes the leading dimension -- so that a large host matrix can be uploaded in column chunks without
        # a host copy of each (``CellRegMap._scan_streamed``); anything else is made contiguous first
        ldg = G.shape[1]
        if not G.flags.c_contiguous:
            if (G.shape[0] > 1 and G.shape[1] > 0 and G.strides[1] == G.itemsize and G.strides[0] % G.itemsize == 0
                    and G.strides[0] >= G.shape[1] * G.itemsize):
                ldg = G.strides[0] // G.itemsize
            else:
                G = np.ascontiguousarray(G)
        self.shape = G.shape
        self.device = device
        self.n_groups = None
        h = ctypes.c_void_p()
        grouped = ctypes.c_int(0)

        def create(hint):
            if hint is None:
                return lib.crm_panel_create_auto(_context(device), G.shape[0], _lib.ptr(G), ldg, G.shape[1], None, 0,
                                                 None, ctypes.byref(h), ctypes.byref(grouped))
            group, reps = hint
            return lib.crm_panel_create_auto(_context(device), G.shape[0], _lib.ptr(G), ldg, G.shape[1],
                                             _lib.ptr(group), reps.shape[0], _lib.ptr(reps), ctypes.byref(h),
                                             ctypes.byref(grouped))

        hint = None
        if isinstance(groups, str) and groups == "auto":
            # An eQTL run scans gene after gene on ONE cohort: the donor structure found for the previous panel with
            # this many cells is tried first -- the device verifies it exactly on every entry -- and only if it does
            # not fit is the structure searched again on the host (a strided gather over the matrix, ~40 ms at config 3)
            hint = _last_grouping.get(G.shape[0])
            rc = create(hint) if hint is not None else None
            if hint is None or (rc == 0 and not grouped.value):
                if hint is not None:
                    lib.crm_panel_destroy(h)
                    h = ctypes.c_void_p()
                hint = candidate_groups(G)
                if hint is not None:
                    hint = (np.ascontiguousarray(hint[0], dtype=np.int32), np.ascontiguousarray(hint[1], dtype=np.int64))
                rc = create(hint)
        else:
            rc = create(None)
        if rc == -4:  # CRM_ERR_NUMERIC: the reference's LMM raises ValueError on non-finite covariates
            raise ValueError("There are non-finite values in the covariates matrix.")
        _lib.check(rc)
        if grouped.value:
            self.n_groups = int(hint[1].shape[0])
            _last_grouping[G.shape[0]] = hint
            while len(_last_grouping) > 4:
                _last_grouping.pop(next(iter(_last_grouping)))
        self.handle = h
        self._fin = weakref.finalize(self, lib.crm_panel_destroy, h)

    def _create_grouped(self, lib, group, Gd, device, h):
        group = np.ascontiguousarray(group, dtype=np.int32)
        Gd = _lib.f64(Gd)
        self.n_groups = int(Gd.shape[0])
        _lib.check(lib.crm_panel_create_grouped(_context(device), group.shape[0], _lib.ptr(group), Gd.shape[0],
                                                _lib.ptr(Gd), Gd.shape[1], Gd.shape[1], ctypes.byref(h)))

    @classmethod
    def from_donors(cls, Gd, donor_of_cell, device=0):
        """Panel from donor-level genotypes (m x p) and the donor index of every cell (n,),
        without materialising the expanded n x p matrix."""
        lib = _lib.load()
        self = cls.__new__(cls)
        Gd = np.asarray(Gd, float)
        donor_of_cell = np.asarray(donor_of_cell)
        assert Gd.ndim == 2 and donor_of_cell.ndim == 1
        self.shape = (donor_of_cell.shape[0], Gd.shape[1])
        self.device = device
        h = ctypes.c_void_p()
        self._create_grouped(lib, donor_of_cell, Gd, device, h)
        self.handle = h
        self._fin = weakref.finalize(self, lib.crm_panel_destroy, h)
        return self


def _panel_from_dosages(cls, dosage, donor_of_cell, standardize=True, device=0):
    """Panel from donor-level allele counts (m x p, integers in [-128, 127], one byte each on the wire) and the
    donor index of every cell; ``standardize=True`` centres and scales every variant on the device by the mean
    and standard deviation of its expanded column (donors weighted by their cell counts) -- the float64
    matrix a caller of the reference would build on the host and expand is never formed."""
    lib = _lib.load()
    self = cls.__new__(cls)
    D = np.ascontiguousarray(dosage)
    if D.dtype != np.int8:
        if not np.array_equal(D, np.rint(D)) or np.abs(D).max(initial=0) > 127:
            raise ValueError("dosages must be integers in [-128, 127]")
        D = D.astype(np.int8)
    group = np.ascontiguousarray(donor_of_cell, dtype=np.int32)
    assert D.ndim == 2 and group.ndim == 1
    self.shape = (group.shape[0], D.shape[1])
    self.device = device
    self.n_groups = int(D.shape[0])
    h = ctypes.c_void_p()
    rc = lib.crm_panel_create_grouped_i8(_context(device), group.shape[0], _lib.ptr(group), D.shape[0], _lib.ptr(D),
                                         D.shape[1], D.shape[1], int(bool(standardize)), ctypes.byref(h))
    if rc == -4:
        raise ValueError("a monomorphic variant cannot be standardised")
    _lib.check(rc)
    self.handle = h
    self._fin = weakref.finalize(self, lib.crm_panel_destroy, h)
    return self


GenotypePanel.from_dosages = classmethod(_panel_from_dosages)


def _release_gene(lib, handle, _background_kept_alive):
    lib.crm_gene_destroy(handle)


_PROGRESS_CB = ctypes.CFUNCTYPE(None, ctypes.c_long, ctypes.c_long, ctypes.c_void_p)


def _stream_chunk():
    """Variants per chunk of the streamed scan of a host matrix (``CellRegMap._scan_streamed``): two blocks of the
    scan's largest block size; CELLREGMAP_AMD_STREAM_CHUNK overrides (0: never stream)."""
    import os

    try:
        return max(0, int(os.environ.get("CELLREGMAP_AMD_STREAM_CHUNK", "8192")))
    except ValueError:
        return 8192


_progress_stack = {}   # (device, thread) -> callbacks installed by the scans in flight on that thread (innermost last)


def _progress_default():
    """The reference shows a tqdm bar over the variants of every scan (_cellregmap.py:270,340); so does this engine
    unless ``progress=False`` is passed or CELLREGMAP_AMD_PROGRESS=0 is set in the environment."""
    import os

    return os.environ.get("CELLREGMAP_AMD_PROGRESS", "1").lower() not in ("0", "false", "no", "off")


class _progress:
    """Context manager: installs a per-block progress callback on the device's context for one scan.
    ``progress``: None = the reference's behaviour (a tqdm bar, see ``_progress_default``), True = a tqdm bar,
    False = silent, or a callable ``(done, total)``.  A scan started from inside a callback (or a nested call)
    gets its own callback and the outer one is put back afterwards; an exception raised by a user callback is
    re-raised once the scan has returned (ctypes cannot carry it through the C frames)."""

    def __init__(self, device, progress, total, offset=0, grand_total=None):
        if progress is None:
            progress = _progress_default()
        self.device, self.progress, self.total = device, progress, total
        self.offset, self.grand_total = offset, grand_total
        self.bar = None
        self.error = None

    def __enter__(self):
        if not self.progress:
            return self
        if self.progress is True:
            try:
                from tqdm import tqdm
            except ImportError:  # pragma: no cover  (the reference depends on tqdm; without it: no bar)
                self.progress = False
                return self

            self.bar = tqdm(total=self.total)
            state = {"done": 0}

            def cb(done, total, _user):
                self.bar.update(done - state["done"])
                state["done"] = done
        else:
            fn = self.progress

            def cb(done, total, _user):
                if self.error is None:
                    try:
                        fn(done + self.offset, total if self.grand_total is None else self.grand_total)
                    except BaseException as exc:  # noqa: BLE001 -- re-raised in __exit__
                        self.error = exc
        self._cb = _PROGRESS_CB(cb)   # (kept alive until __exit__)
        self._key = (self.device, threading.get_ident())   # (the library keeps one callback per calling thread)
        _progress_stack.setdefault(self._key, []).append(self._cb)
        _lib.check(_lib.load().crm_set_progress_callback(_context(self.device), ctypes.cast(self._cb, ctypes.c_void_p), None))
        return self

    def __exit__(self, *exc):
        if self.progress:
            stack = _progress_stack[self._key]
            stack.remove(self._cb)
            outer = ctypes.cast(stack[-1], ctypes.c_void_p) if stack else None
            _lib.check(_lib.load().crm_set_progress_callback(_context(self.device), outer, None))
            if self.bar is not None:
                self.bar.close()
            if self.error is not None and exc[0] is None:
                raise self.error
        return False


def _permutation(idx, n):
    """Index argument of the permutation hooks (_cellregmap.py:398-413) as int32 row indices: a
    boolean mask selects rows like numpy's ``E0[idx, :]`` does, negative entries count from the
    end; the hooks need one entry per sample."""
    if idx is None:
        return None
    idx = np.asarray(idx)
    if idx.dtype == bool:
        idx = np.flatnonzero(idx)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    if idx.shape != (n,):
        raise ValueError("permutation index must have one entry per sample")
    idx = np.where(idx < 0, idx + n, idx)
    return np.ascontiguousarray(idx, dtype=np.int32)


_DEFERRED = object()  # constructor argument: build the background decompositions on first use


class CellRegMap:
    """Mixed model with genetic effect heterogeneity -- MI355X engine.

    Drop-in for ``cellregmap.CellRegMap`` (_cellregmap.py:23-440) on the score-test path:
    the constructor (:63-131) builds the rho grid of background covariances as economic
    eigendecompositions -- here on the device -- and ``scan_interaction`` (:317-440) runs the
    per-variant null fits, score statistic and Davies p-value in batched HIP kernels.

    Extra keyword-only arguments (not in the reference): ``device`` (GPU ordinal),
    ``background`` (a prebuilt background to share across genes).
    """

    def __init__(self, y, E, W=None, Ls=None, E1=None, hK=None, *, device=0, background=None):
        # coercions and checks exactly as _cellregmap.py:64-91
        self._y = np.asarray(y, float).flatten()
        self._E0 = np.asarray(E, float)
        Ls = [] if Ls is None else Ls
        if W is not None:
            self._W = np.asarray(W, float)
        else:
            self._W = np.ones((self._y.shape[0], 1))
        if E1 is not None:
            self._E1 = np.asarray(E1, float)
        else:
            self._E1 = np.asarray(E, float)
        if isinstance(Ls, HadamardHalves):
            self._Ls = Ls  # factored; elements are formed on demand
        else:
            self._Ls = list(np.asarray(L, float) for L in Ls)

        assert self._W.ndim == 2
        assert self._E0.ndim == 2
        assert self._E1.ndim == 2
        assert self._y.shape[0] == self._W.shape[0]
        assert self._y.shape[0] == self._E0.shape[0]
        assert self._y.shape[0] == self._E1.shape[0]
        if isinstance(Ls, HadamardHalves):
            assert self._y.shape[0] == Ls.shape_us[0] == Ls.hK.shape[0]
        else:
            for L in Ls:
                assert self._y.shape[0] == L.shape[0]
                assert L.ndim == 2

        self._device = device
        # background modes of _cellregmap.py:101-131
        if len(Ls) == 0:
            if hK is None:
                self._rho1 = [1.0]
                B = None
            else:
                self._rho1 = _RHO_GRID
                B = np.asarray(hK, float)
        else:
            self._rho1 = _RHO_GRID
            B = self._Ls if isinstance(self._Ls, HadamardHalves) else np.concatenate(self._Ls, axis=1)
        self._bg_halves = B
        if background is _DEFERRED:  # estimate_betas: predict_interaction never reads them
            self._bg_lazy = None
        elif background is not None:
            self._bg_lazy = background
        else:
            self._bg_lazy = _make_background(self._E1, B, self._rho1, device)
        self._gene = None
        self._gene_fin = None

    @property
    def _bg(self):
        if self._bg_lazy is None:
            self._bg_lazy = _make_background(self._E1, self._bg_halves, self._rho1, self._device)
        return self._bg_lazy

    @property
    def n_samples(self):
        return self._y.shape[0]

    # -- device objects --------------------------------------------------------------------
    def _fixed_effect_basis(self):
        """The basis of span(W) the reference's LMM works in: ``U * s`` of numpy_sugar.economic_svd(W) (singular values
        below sqrt(eps) dropped; glimix-core ``LMM.__init__``: ``tX = ddot(U, S)``).  The scans depend on W through
        its column space only (LMM profiles the fixed effects out, PMat solves by lstsq), the columns come out
        mutually orthogonal, and the library then orthogonalises every variant against them in the cell axis
        (csrc/blockops.hip: launch_ortho_block) -- together the reference's economic_svd([W, g]) basis."""
        W = self._W
        if W.shape[1] == 0:
            raise ValueError("W has no columns")
        if W.shape[1] == 1 and np.abs(W).max(initial=0.0) >= _SQRT_EPS:
            return W  # one non-zero column (norm >= its largest entry): full rank without asking the SVD
        U, s, _ = _economic_svd(W)
        return U * s

    def _bind_gene(self, like=None):
        """``like``: a ``CellRegMap`` of the same cohort (same background, W and E: checked by the caller) that is bound
        already -- its covariates and contexts are copied on the device (``crm_gene_create_like``) instead of being
        decomposed, hashed and uploaded again."""
        if self._gene is not None:
            return self._gene
        lib = _lib.load()
        if not np.all(np.isfinite(self._y)):
            raise ValueError("There are non-finite values in the outcome.")
        y = _lib.f64(self._y)
        h = ctypes.c_void_p()
        if like is not None and like._gene is not None and like._bg is self._bg:
            _lib.check(lib.crm_gene_create_like(like._gene, _lib.ptr(y), ctypes.byref(h)))
            ncov = like._ncov
        else:
            if not np.all(np.isfinite(self._W)):
                raise ValueError("There are non-finite values in the covariates matrix.")
            Wb = _lib.f64(self._fixed_effect_basis())
            E0 = _lib.f64(self._E0)
            _lib.check(lib.crm_gene_create(self._bg.handle, _lib.ptr(y), _lib.ptr(Wb), Wb.shape[1], _lib.ptr(E0),
                                           E0.shape[1], ctypes.byref(h)))
            ncov = Wb.shape[1]
        self._gene = h
        self._ncov = ncov
        n, rmax = self.n_samples, max(self._bg.rank(i) for i in range(len(self._rho1)))
        if rmax + ncov + 1 >= n:
            import warnings

            warnings.warn(f"saturated model: the background covariance has rank {rmax} and with the {ncov} covariate "
                          f"column(s) and the variant it spans all {n} cells; the reference's likelihood then divides "
                          "rounding noise by delta and its results (and these) are not reproducible to the usual "
                          "tolerances (scan_interaction_info flags such variants)", RuntimeWarning, stacklevel=3)
        # (the background object rides along so that it outlives the gene whatever the collection order)
        self._gene_fin = weakref.finalize(self, _release_gene, lib, h, self._bg)
        return h

    def _panel(self, G, groups="auto"):
        if isinstance(G, GenotypePanel):
            panel = G
        else:
            G = np.asarray(G, float)
            if G.ndim != 2 or G.shape[0] != self.n_samples:
                raise ValueError(f"G must be {self.n_samples} x p, got {G.shape}")
            panel = GenotypePanel(G, self._device, groups)  # raises ValueError on non-finite entries
        if panel.shape[0] != self.n_samples:
            raise ValueError(f"G has {panel.shape[0]} rows, expected {self.n_samples}")
        return panel

    # -- interaction scan (_cellregmap.py:317-440) ----------------------------------------------
    def scan_interaction(self, G, idx_E: Optional[any] = None, idx_G: Optional[any] = None,
                         return_stats: bool = False, progress=None, groups="auto"):
        """Per-variant GxC score test.  ``G`` is n x p (array-like) or a ``GenotypePanel``; an array goes to the device as
        ``GenotypePanel(G, groups=groups)`` would take it (``"auto"``: donor-constant genotypes are found and scanned on
        the donor-collapsed path; ``None``: kept dense), in column chunks beside the scan when it has many variants
        (``_scan_streamed``).
        ``progress``: the reference always shows a tqdm bar over the variants (:340) and so does this method by
        default (it advances block by block); ``False`` silences it (or CELLREGMAP_AMD_PROGRESS=0 in the
        environment), a callable ``(done, total)`` replaces it.

        Returns ``(pvalues, info)`` with ``info = {rho1, e2, g2, eps2}`` as the reference
        (:439-440); with ``return_stats=True`` additionally a dict holding Q, the eigenvalues
        of F, F itself and the null-fit scalars (for parity tests)."""
        lib = _lib.load()
        k0 = self._E0.shape[1]
        if not isinstance(G, GenotypePanel) and np.asarray(G).ndim == 2 and np.asarray(G).shape[1] == 0:
            # no variants: the reference's loop body never runs (_cellregmap.py:340)
            if np.asarray(G).shape[0] != self.n_samples:
                raise ValueError(f"G must be {self.n_samples} x p, got {np.asarray(G).shape}")
            empty = {key: np.empty(0) for key in ("rho1", "e2", "g2", "eps2")}
            if return_stats:
                return np.empty(0), empty, {"Q": np.empty(0), "lml": np.empty(0), "delta": np.empty(0),
                                            "scale": np.empty(0), "lambda": np.empty((0, k0)),
                                            "F": np.empty((0, k0, k0))}
            return np.empty(0), empty
        if not isinstance(G, GenotypePanel):
            G = np.asarray(G, float)
            if G.ndim == 2 and G.shape[0] == self.n_samples and G.shape[1] >= 2 * _stream_chunk() > 0:
                return self._scan_streamed(lib, G, k0, idx_E, idx_G, return_stats, progress, groups)
        panel = self._panel(G, groups)
        n, p = panel.shape
        gene = self._bind_gene()

        iE, iG = _permutation(idx_E, n), _permutation(idx_G, n)
        with _progress(self._device, progress, p):
            return self._scan_interaction(lib, gene, panel, p, k0, iE, iG, return_stats)

    def _streamed_panels(self, G, groups="auto"):
        """Generator over ``(first, last, panel)``: the column chunks of a host matrix, uploaded one after the other by a
        second thread on the library's upload stream, outside the context's lock -- at most three chunks on the device
        at a time: the one being scanned, one waiting, one being built.  The
        first chunk -- the only one nothing hides -- is one block of the scan when the chunk is a multiple of it; every
        chunk looks for the donor structure by itself, as one panel would.  An error of the uploading thread (the
        reference's ValueError on non-finite entries) is raised here; closing the generator stops the thread."""
        import queue

        p = G.shape[1]
        chunk = _stream_chunk()
        first = 4096 if chunk % 4096 == 0 else (chunk // 2 if chunk % 256 == 0 else chunk)
        bounds = [(0, min(p, first))] + [(j0, min(p, j0 + chunk)) for j0 in range(first, p, chunk)]
        ready = queue.Queue(maxsize=1)
        stop = threading.Event()

        def upload():
            try:
                for j0, j1 in bounds:
                    if stop.is_set():
                        return
                    ready.put(GenotypePanel(G[:, j0:j1], self._device, groups))
            except BaseException as exc:  # noqa: BLE001 -- handed to the consuming thread, which raises it
                ready.put(exc)

        worker = threading.Thread(target=upload, name="cellregmap-amd-upload", daemon=True)
        worker.start()
        try:
            for j0, j1 in bounds:
                item = ready.get()
                if isinstance(item, BaseException):
                    raise item
                yield j0, j1, item
                del item
        finally:
            stop.set()
            while worker.is_alive():      # (let a blocked put() through, drop what it still uploads)
                try:
                    ready.get(timeout=0.05)
                except queue.Empty:
                    pass
            worker.join()

    @staticmethod
    def _one_bar(progress, total):
        """``progress`` of a scan that runs chunk by chunk: a tqdm bar becomes ONE bar over all chunks.  Returns the
        callable (or False) to hand to every chunk's ``_progress`` and the bar to close."""
        if progress is None:
            progress = _progress_default()
        if progress is not True:
            return progress, None
        try:
            from tqdm import tqdm
        except ImportError:  # pragma: no cover
            return False, None
        bar, seen = tqdm(total=total), {"done": 0}

        def advance(done, _total):
            bar.update(done - seen["done"])
            seen["done"] = done

        return advance, bar

    def _scan_streamed(self, lib, G, k0, idx_E, idx_G, return_stats, progress, groups="auto"):
        """A host matrix of many variants goes to the device in column chunks from a second thread while this one scans
        the chunks that have arrived: PCIe beside the scan, and device memory for three chunks instead of the whole
        matrix.  Every chunk is scanned as a panel of its own: where the chunk bounds fall on the block bounds of the
        one-panel scan (the default 8192-variant chunks at the BASELINE configurations: blocks of 4096) the results are
        identical bit for bit, otherwise -- other block sizes, donor structure found in some chunks only -- they agree
        to the rounding of a different summation order."""
        n, p = G.shape
        iE, iG = _permutation(idx_E, n), _permutation(idx_G, n)
        progress, bar = self._one_bar(progress, p)
        panels = self._streamed_panels(G, groups)
        parts = []
        try:
            gene = self._bind_gene()      # (beside the first chunk's upload)
            for j0, j1, panel in panels:
                with _progress(self._device, progress, j1 - j0, offset=j0, grand_total=p):
                    parts.append(self._scan_interaction(lib, gene, panel, j1 - j0, k0, iE, iG, return_stats))
                panel = None      # (released before the next chunk is taken from the queue)
        finally:
            panels.close()
            if bar is not None:
                bar.close()
        pv = np.concatenate([part[0] for part in parts])
        info = {key: np.concatenate([part[1][key] for part in parts]) for key in parts[0][1]}
        if return_stats:
            return pv, info, {key: np.concatenate([part[2][key] for part in parts]) for key in parts[0][2]}
        return pv, info

    def _scan_interaction(self, lib, gene, panel, p, k0, iE, iG, return_stats):
        out = {k: np.empty(p) for k in ("pv", "rho1", "e2", "g2", "eps2")}
        extra = {}
        if return_stats:
            extra = {"Q": np.empty(p), "lml": np.empty(p), "delta": np.empty(p), "scale": np.empty(p),
                     "lambda": np.empty((p, k0)), "F": np.empty((p, k0, k0))}
        _lib.check(lib.crm_scan_interaction(
            gene, panel.handle, 0, p, _lib.ptr(iE), _lib.ptr(iG),
            _lib.ptr(out["pv"]), _lib.ptr(out["rho1"]), _lib.ptr(out["e2"]), _lib.ptr(out["g2"]),
            _lib.ptr(out["eps2"]), _lib.ptr(extra.get("Q")), _lib.ptr(extra.get("lml")),
            _lib.ptr(extra.get("delta")), _lib.ptr(extra.get("scale")), _lib.ptr(extra.get("lambda")),
            _lib.ptr(extra.get("F"))))
        info = {key: out[key] for key in ("rho1", "e2", "g2", "eps2")}
        if return_stats:
            return out["pv"], info, extra
        return out["pv"], info

    def scan_interaction_permutations(self, G, idx_E_list=None, idx_G_list=None, return_Q=False):
        """``scan_interaction(G, idx_E=perm)`` for a whole list of permutations in one call -- the loop of the reference's
        calibration test (cellregmap/test/test_struct_lmm2.py:208-209) and of any permutation driver around the hooks at
        cellregmap/_cellregmap.py:398-413.  The hooks enter only the test direction ``ddot(g[idx_G], E0[idx_E])``: the
        eleven null fits, rho*, the variance components and the rotations of the variants do not depend on them and are done
        once per block of variants (``crm_scan_interaction_permuted``); every permutation then runs the score test proper.

        ``idx_E_list`` / ``idx_G_list``: sequences of B index arrays (either may be ``None``: no permutation of that kind;
        an entry may be ``None`` too = the identity).  Returns ``(pvalues (B, p), info)`` with the reference's four ``info``
        arrays (they are the same for every permutation); row b of ``pvalues`` is bit for bit what
        ``scan_interaction(G, idx_E_list[b], idx_G_list[b])`` returns.  ``return_Q``: also the score statistics (B, p)."""
        lib = _lib.load()
        panel = self._panel(G)
        n, p = panel.shape
        nb = len(idx_E_list) if idx_E_list is not None else (len(idx_G_list) if idx_G_list is not None else 0)
        if nb < 1 or (idx_E_list is not None and idx_G_list is not None and len(idx_G_list) != nb):
            raise ValueError("idx_E_list / idx_G_list: one entry per permutation, the same number in both")

        def stack(lst):
            if lst is None:
                return None
            rows = [_permutation(v, n) for v in lst]
            ident = np.arange(n, dtype=np.int32)
            return np.ascontiguousarray(np.stack([ident if r is None else r for r in rows]), dtype=np.int32)

        iE, iG = stack(idx_E_list), stack(idx_G_list)
        gene = self._bind_gene()
        pv = np.empty((nb, p))
        Q = np.empty((nb, p)) if return_Q else None
        info = {k: np.empty(p) for k in ("rho1", "e2", "g2", "eps2")}
        if p > 0:
            with _progress(self._device, False, p):
                _lib.check(lib.crm_scan_interaction_permuted(gene, panel.handle, 0, p, nb, _lib.ptr(iE), _lib.ptr(iG), _lib.ptr(pv),
                                                             _lib.ptr(info["rho1"]), _lib.ptr(info["e2"]), _lib.ptr(info["g2"]),
                                                             _lib.ptr(info["eps2"]), _lib.ptr(Q)))
        return (pv, info, Q) if return_Q else (pv, info)

    def scan_interaction_info(self, G, idx_E=None, idx_G=None):
        """The p-values together with chiscore's ``info`` of ``davies_pvalue(Q, F, True)`` (which the reference
        computes at :435 and drops) and with how reproducible every variant's result is:
        ``(pvalues, {"liu_pval", "Is_Converged", "ifault", "model_flags", "degenerate", "flat_optimum",
        "statistic_at_tolerance", "rho_tie", "bound_Q", "bound_p"})``.

        ``bound_Q`` / ``bound_p``: how far the score statistic (relative to max(Q, tr F)) and the p-value (relative) of two
        faithful runs of the reference's procedure may differ -- the reference stops its null fit at a tolerance of 1e-6 on
        logit(delta), and where exactly a search stops within that tolerance is decided by rounding noise
        (include/crm_hip.h: crm_scan_interaction_bounds).  ``flat_optimum`` = ``bound_p > 1e-5`` (about 2 % of scans),
        ``statistic_at_tolerance`` = ``bound_Q > 1e-6`` (more than a third: Q moves by ~1e-6 per tolerance).
        ``model_flags`` (bits ``MODEL_SATURATED`` = 1, ``MODEL_DELTA_AT_ZERO`` = 2, ``MODEL_G_IN_SPAN_W`` = 4,
        ``MODEL_FLAT_OPTIMUM`` = 8, ``MODEL_RHO_TIE`` = 16, ``MODEL_STATISTIC_AT_TOLERANCE`` = 32) / ``degenerate`` mark the
        variants where the reference's own result is decided by rounding noise outright (saturated model, null fit ending at
        delta = 0); ``rho_tie`` those whose rho* is within the likelihood's noise of another grid point's
        (``info["rho1"]`` may differ between two faithful runs)."""
        lib = _lib.load()
        panel = self._panel(G)
        n, p = panel.shape
        gene = self._bind_gene()
        iE, iG = _permutation(idx_E, n), _permutation(idx_G, n)
        pv, liu = np.empty(p), np.empty(p)
        ifault, flags = np.empty(p, np.int32), np.empty(p, np.int32)
        bq, bp = np.empty(p), np.empty(p)
        _lib.check(lib.crm_scan_interaction_bounds(gene, panel.handle, 0, p, _lib.ptr(iE), _lib.ptr(iG), _lib.ptr(pv),
                                                   _lib.ptr(ifault), _lib.ptr(liu), _lib.ptr(flags), _lib.ptr(bq), _lib.ptr(bp)))
        return pv, {"liu_pval": liu, "Is_Converged": (ifault == 0).astype(int), "ifault": ifault,
                    "model_flags": flags, "degenerate": (flags & 3) != 0, "flat_optimum": (flags & 8) != 0,
                    "rho_tie": (flags & 16) != 0, "statistic_at_tolerance": (flags & 32) != 0, "bound_Q": bq, "bound_p": bp}

    # -- association scans (_cellregmap.py:246-314) --------------------------------------------------
    def _scan_association(self, G, fast, return_stats=False, progress=None):
        lib = _lib.load()
        p_user = None
        if not isinstance(G, GenotypePanel) and np.asarray(G).ndim == 2 and np.asarray(G).shape[1] == 0:
            # no SNPs: the null model is still fitted and reported (_cellregmap.py:250-266)
            p_user = 0
            G = np.zeros((np.asarray(G).shape[0], 1))
        null = np.empty(6)
        if p_user is None and not isinstance(G, GenotypePanel):
            G = np.asarray(G, float)
            if G.ndim == 2 and G.shape[0] == self.n_samples and G.shape[1] >= 2 * _stream_chunk() > 0:
                # a host matrix of many SNPs: column chunks uploaded beside the scan (``_streamed_panels``); every call
                # fits the null model again -- the same eleven fits, the same numbers
                p = G.shape[1]
                pv, alt = np.empty(p), np.empty(p)
                progress, bar = self._one_bar(progress, p)
                panels = self._streamed_panels(G)
                try:
                    gene = self._bind_gene()
                    for j0, j1, panel in panels:
                        with _progress(self._device, progress, j1 - j0, offset=j0, grand_total=p):
                            _lib.check(lib.crm_scan_association(gene, panel.handle, 0, j1 - j0, int(bool(fast)),
                                                                _lib.ptr(pv[j0:j1]), _lib.ptr(alt[j0:j1]), _lib.ptr(null)))
                finally:
                    panels.close()
                    if bar is not None:
                        bar.close()
                return self._association_result(pv, alt, null, return_stats)
        panel = self._panel(G)
        n, p = panel.shape
        if p_user is not None:
            p = p_user
        gene = self._bind_gene()
        pv = np.empty(p)
        alt = np.empty(p)
        with _progress(self._device, progress, p):
            _lib.check(lib.crm_scan_association(gene, panel.handle, 0, p, int(bool(fast)), _lib.ptr(pv),
                                                _lib.ptr(alt), _lib.ptr(null)))
        return self._association_result(pv, alt, null, return_stats)

    @staticmethod
    def _association_result(pv, alt, null, return_stats):
        info = {"rho1": np.asarray([null[0]], float), "e2": np.asarray([null[1]], float),
                "g2": np.asarray([null[2]], float), "eps2": np.asarray([null[3]], float)}
        if return_stats:
            return pv, info, {"alt_lml": alt, "null_lml": null[4], "null_delta": null[5]}
        return pv, info

    def scan_association(self, G, return_stats: bool = False, progress=None):
        """Persistent-effect LRT with a full ML refit per SNP (_cellregmap.py:246-281; ``progress`` as in
        ``scan_interaction`` -- the reference shows a tqdm bar over the SNPs, :270)."""
        return self._scan_association(G, False, return_stats, progress)

    def scan_association_fast(self, G, return_stats: bool = False, progress=False):
        """Persistent-effect LRT with the covariance ratio frozen at the null model
        (glimix-core FastScanner; _cellregmap.py:284-314 -- run with ``verbose=False`` there: no bar by default)."""
        return self._scan_association(G, True, return_stats, progress)

    # -- effect sizes (_cellregmap.py:137-244) -----------------------------------------------------
    def _lmm_fit(self, bg, M):
        """Best restricted fit of LMM(y, M, QS(rho)) over the grid of ``bg``:
        (rho, v0, v1, beta, grid index).  Rank-deficient M goes through the SVD basis like
        glimix-core's LMM (beta is the minimum-norm solution)."""
        lib = _lib.load()
        U, s, Vt = _economic_svd(M)
        X = _lib.f64(U * s)   # (mutually orthogonal columns: the basis glimix-core's LMM works in)
        y = _lib.f64(self._y)
        E0 = _lib.f64(self._E0)
        h = ctypes.c_void_p()
        _lib.check(lib.crm_gene_create(bg.handle, _lib.ptr(y), _lib.ptr(X), X.shape[1], _lib.ptr(E0), E0.shape[1],
                                       ctypes.byref(h)))
        try:
            fit = np.empty(6)
            beta = np.empty(X.shape[1])
            _lib.check(lib.crm_lmm_fit(h, 1, _lib.ptr(fit), _lib.ptr(beta)))
        finally:
            lib.crm_gene_destroy(h)
        beta = Vt.T @ beta
        return fit[0], fit[1], fit[2], beta, int(fit[5])

    def _snp_background(self, gE):
        """Decompositions of [sqrt(rho) g o E0, sqrt(1-rho) L..] over the grid (:158-175).  As in the
        reference only ``Ls`` enters here -- an ``hK`` passed to the constructor does not."""
        if isinstance(self._Ls, HadamardHalves):
            B = self._Ls
        elif len(self._Ls) == 0:
            B = None
        else:
            B = np.concatenate(self._Ls, axis=1)
        return _make_background(gE, B, self._rho1, self._device, cache=False)

    @staticmethod
    def _cov_solve(bg, ri, v0, v1, rhs):
        rhs = _lib.f64(np.asarray(rhs, float).reshape(rhs.shape[0], -1))
        out = np.empty_like(rhs)
        _lib.check(_lib.load().crm_cov_solve(bg.handle, ri, float(v0), float(v1), _lib.ptr(rhs), rhs.shape[1],
                                             _lib.ptr(out)))
        return out

    def predict_interaction(self, G, MAF):
        """Persistent effect and cell-level GxC effects per SNP (_cellregmap.py:137-205): returns
        ``(beta_g (p,), beta_gxe (1, n, p))`` -- the reference's shapes."""
        G = np.asarray(G, float)
        E0, W = self._E0, self._W
        maf = np.asarray(np.atleast_1d(MAF), float)
        normalization = 1 / np.sqrt(2 * maf * (1 - maf))
        beta_g_s, beta_gxe_s = [], []
        for i in range(G.shape[1]):
            g = G[:, [i]]
            M = np.concatenate((W, g, E0), axis=1)
            gE = g * E0
            bg = self._snp_background(gE)
            rho1, v0, v1, beta, ri = self._lmm_fit(bg, M)
            yadj = (self._y - M @ beta).reshape(-1, 1)
            v = self._cov_solve(bg, ri, v0, v1, yadj)
            beta_g_s.append(beta[W.shape[1]])
            beta_gxe_s.append((v0 * rho1) * E0 @ (gE.T @ v) * normalization[i])
        return np.asarray(beta_g_s), np.stack(beta_gxe_s).T

    def estimate_aggregate_environment(self, g):
        """_cellregmap.py:207-244: the fit runs on the object's own background; only the final solve
        uses the decomposition of the SNP's covariance halves."""
        g = np.atleast_2d(np.asarray(g, float)).reshape((np.asarray(g).size, 1))
        E0, W = self._E0, self._W
        gE = g * E0
        M = np.concatenate((W, g, E0), axis=1)
        rho1, v0, v1, beta, ri = self._lmm_fit(self._bg, M)
        yadj = self._y - M @ beta
        bg = self._snp_background(gE)
        v = self._cov_solve(bg, ri, v0, v1, yadj.reshape(-1, 1))[:, 0]
        return E0 @ ((rho1 * v0) * gE.T @ v)


def _bind_genes_like(first, others, batch=64):
    """Bind the phenotypes of ``others`` (same cohort as ``first``, which is bound) through ``crm_gene_create_batch``."""
    lib = _lib.load()
    for a in range(0, len(others), batch):
        part = others[a:a + batch]
        for c in part:
            if not np.all(np.isfinite(c._y)):
                raise ValueError("There are non-finite values in the outcome.")
        Y = np.ascontiguousarray(np.stack([c._y for c in part], axis=1), dtype=np.float64)
        handles = (ctypes.c_void_p * len(part))()
        _lib.check(lib.crm_gene_create_batch(first._gene, _lib.ptr(Y), Y.shape[1], len(part), handles))
        n, rmax = first.n_samples, max(first._bg.rank(i) for i in range(len(first._rho1)))
        for c, h in zip(part, handles):
            c._gene = ctypes.c_void_p(h)
            c._ncov = first._ncov
            c._gene_fin = weakref.finalize(c, _release_gene, lib, c._gene, c._bg)
        if rmax + first._ncov + 1 >= n:
            import warnings

            warnings.warn(f"saturated model: the background covariance has rank {rmax} and with the {first._ncov} covariate "
                          f"column(s) and the variant it spans all {n} cells; the reference's likelihood then divides "
                          "rounding noise by delta and its results (and these) are not reproducible to the usual "
                          "tolerances (scan_interaction_info flags such variants)", RuntimeWarning, stacklevel=3)


def _cis_runs(cis_index, ngenes, p, dense_limit=1 << 26):
    """Split the variant axis of a panel into maximal runs over which the set of phenotypes that test the
    variant does not change.  ``cis_index[i]``: the variants of phenotype i -- a ``(start, stop)`` pair, a
    ``slice`` or an array of column indices (any order, repeats allowed).  Returns ``(columns, runs)``:
    ``columns[i]`` the int64 column indices of phenotype i as given, ``runs`` a list of
    ``(first, count, genes)`` with ``genes`` the sorted phenotype numbers active on ``[first, first+count)``;
    variants nobody asks for are in no run."""
    if len(cis_index) != ngenes:
        raise ValueError(f"cis_index has {len(cis_index)} entries for {ngenes} phenotypes")
    columns = []
    for i, sel in enumerate(cis_index):
        if isinstance(sel, slice):
            cols = np.arange(p, dtype=np.int64)[sel]
        elif (isinstance(sel, tuple) and len(sel) == 2 and all(isinstance(v, (int, np.integer)) for v in sel)):
            if not 0 <= sel[0] <= sel[1] <= p:
                raise ValueError(f"cis_index[{i}] = {sel} is not a range inside [0, {p}]")
            cols = np.arange(sel[0], sel[1], dtype=np.int64)
        else:
            cols = np.asarray(sel)
            if cols.dtype == bool:
                if cols.shape != (p,):
                    raise ValueError(f"cis_index[{i}]: a boolean mask must have one entry per variant")
                cols = np.flatnonzero(cols)
            cols = cols.astype(np.int64, copy=False).ravel()
            if cols.size and (cols.min() < -p or cols.max() >= p):
                raise ValueError(f"cis_index[{i}] has a variant index outside the panel (p = {p})")
            cols = np.where(cols < 0, cols + p, cols)
        columns.append(cols)
    # +1 / -1 events on the variant axis per phenotype, from the sorted distinct columns of each
    active = np.zeros((ngenes, p), dtype=bool) if ngenes * p <= dense_limit else None
    runs = []
    if active is not None:
        for i, cols in enumerate(columns):
            active[i, cols] = True
        if p == 0:
            return columns, runs
        change = np.flatnonzero((active[:, 1:] != active[:, :-1]).any(axis=0)) + 1
        bounds = np.concatenate(([0], change, [p]))
        for a, b in zip(bounds[:-1], bounds[1:]):
            genes = np.flatnonzero(active[:, a])
            if genes.size:
                runs.append((int(a), int(b - a), genes))
        return columns, runs
    # large panels x many phenotypes: sweep over the run boundaries of each phenotype instead of a dense mask
    starts, stops = [], []
    for i, cols in enumerate(columns):
        u = np.unique(cols)
        if u.size == 0:
            continue
        brk = np.flatnonzero(np.diff(u) > 1)
        a = np.concatenate(([u[0]], u[brk + 1]))
        b = np.concatenate((u[brk] + 1, [u[-1] + 1]))
        starts += [(int(x), i) for x in a]
        stops += [(int(x), i) for x in b]
    events = sorted([(x, 1, i) for x, i in starts] + [(x, 0, i) for x, i in stops])
    current, at = set(), 0
    k = 0
    while k < len(events):
        x = events[k][0]
        if current and x > at:
            runs.append((at, x - at, np.array(sorted(current))))
        while k < len(events) and events[k][0] == x:
            _, opening, i = events[k]
            (current.add if opening else current.discard)(i)
            k += 1
        at = x
    return columns, runs


def scan_interaction_many(crms, G, idx_E=None, idx_G=None, cis_index=None, progress=False):
    """Interaction scans of several phenotypes against one genotype panel in a single pass.

    ``crms``: ``CellRegMap`` objects that share the background, ``W`` and ``E`` (e.g. built with
    ``background=crms[0]._bg``).  Work that does not depend on the phenotype is done once per block
    of variants (the reference redoes everything per gene).  Returns
    ``(pvalues, info)`` with arrays of shape (len(crms), p); row i equals
    ``crms[i].scan_interaction(G, idx_E, idx_G)``.

    ``cis_index`` (optional): one entry per phenotype naming the variants it is tested against (its cis
    window) -- a ``(start, stop)`` pair, a slice, a boolean mask or an array of column indices.  The panel is
    walked once; each stretch of variants is scanned for exactly the phenotypes whose window covers it.  The
    results are then lists: entry i holds the arrays of ``crms[i].scan_interaction(G[:, cis_index[i]], ...)``.

    ``progress`` (default off -- this entry point has no counterpart in the reference): ``True`` for a tqdm bar, or a
    callable ``(done, total)`` counted in variants of the panel that at least one phenotype tests."""
    lib = _lib.load()
    crms = list(crms)
    if not crms:
        raise ValueError("no phenotypes given")
    first = crms[0]
    for c in crms[1:]:
        if c._bg is not first._bg:
            raise ValueError("all CellRegMap objects must share one background (pass background=...)")
        if c._W.shape != first._W.shape or c._E0.shape != first._E0.shape:
            raise ValueError("all CellRegMap objects must share W and E")
        # (the shared pass takes g'W, the context features and the donor tables from the first object)
        if not (c._W is first._W or np.array_equal(c._W, first._W)):
            raise ValueError("all CellRegMap objects of one pass must hold the same covariates W")
        if not (c._E0 is first._E0 or np.array_equal(c._E0, first._E0)):
            raise ValueError("all CellRegMap objects of one pass must hold the same contexts E")
    panel = first._panel(G)
    n, p = panel.shape
    # (the phenotypes share W and E -- checked above: the first one's copies on the device serve the others, and their
    # rotations are taken in batches)
    first._bind_gene()
    _bind_genes_like(first, [c for c in crms[1:] if c._gene is None and c._bg is first._bg])
    genes = [c._bind_gene(like=first) for c in crms]
    ng = len(genes)

    iE, iG = _permutation(idx_E, n), _permutation(idx_G, n)
    keys = ("pv", "rho1", "e2", "g2", "eps2")
    if cis_index is not None:
        columns, runs = _cis_runs(cis_index, ng, p)
        # Donor-level panels (the collapsed path): what the phenotypes of a run share is small there, and a call per run of
        # constant phenotypes -- 128 short runs for 64 overlapping windows -- costs more than it saves.  Contiguous windows
        # are then scanned window by window on the resident panel, one call per phenotype (measured at BASELINE config 3,
        # 1024-variant windows: 206 000 against 147 000 variant-tests/s; general genotypes keep the runs, where the shared
        # rotations are what a block costs: 36 000 against 28 500).
        contiguous = all(c.size == 0 or (c.size == int(c[-1] - c[0]) + 1 and np.all(np.diff(c) == 1)) for c in columns)
        if contiguous and panel.n_groups is not None:
            res = {k: [] for k in keys}
            tested = int(sum(c.size for c in columns))
            bar = None
            if progress is True:
                from tqdm import tqdm

                bar = tqdm(total=tested)
            seen = 0
            for i, cols in enumerate(columns):
                out = {k: np.empty(cols.size) for k in keys}
                if cols.size:
                    _lib.check(lib.crm_scan_interaction(genes[i], panel.handle, int(cols[0]), int(cols.size), _lib.ptr(iE),
                                                        _lib.ptr(iG), *[_lib.ptr(out[k]) for k in keys],
                                                        None, None, None, None, None, None))
                seen += cols.size
                if bar is not None:
                    bar.update(cols.size)
                elif callable(progress):
                    progress(seen, tested)
                for k in keys:
                    res[k].append(out[k])
            if bar is not None:
                bar.close()
            return res["pv"], {k: res[k] for k in keys[1:]}
        full = {k: [np.full(p, np.nan) if columns[i].size else None for i in range(ng)] for k in keys}
        tested = int(sum(count for _, count, _ in runs))
        bar = None
        if progress is True:   # one bar over all runs
            from tqdm import tqdm

            bar = tqdm(total=tested)
            state = {"done": 0}

            def progress(done, total, _bar=bar, _state=state):
                _bar.update(done - _state["done"])
                _state["done"] = done
        seen = 0
        for a, count, active in runs:
            handles = (ctypes.c_void_p * len(active))(*[genes[i].value for i in active])
            out = {k: np.empty((len(active), count)) for k in keys}
            with _progress(first._device, progress, count, offset=seen, grand_total=tested):
                _lib.check(lib.crm_scan_interaction_multi(handles, len(active), panel.handle, a, count, _lib.ptr(iE),
                                                          _lib.ptr(iG), *[_lib.ptr(out[k]) for k in keys], None))
            seen += count
            for row, i in enumerate(active):
                for k in keys:
                    full[k][i][a:a + count] = out[k][row]
        if bar is not None:
            bar.close()
        res = {k: [full[k][i][columns[i]] if columns[i].size else np.empty(0) for i in range(ng)] for k in keys}
        return res["pv"], {k: res[k] for k in keys[1:]}
    handles = (ctypes.c_void_p * ng)(*[g.value for g in genes])
    out = {k: np.empty((ng, p)) for k in keys}
    with _progress(first._device, progress, p):
        _lib.check(lib.crm_scan_interaction_multi(handles, ng, panel.handle, 0, p, _lib.ptr(iE), _lib.ptr(iG),
                                                  _lib.ptr(out["pv"]), _lib.ptr(out["rho1"]), _lib.ptr(out["e2"]),
                                                  _lib.ptr(out["g2"]), _lib.ptr(out["eps2"]), None))
    return out["pv"], {k: out[k] for k in ("rho1", "e2", "g2", "eps2")}


def scan_interaction_resumable(crm, G, checkpoint, idx_E=None, idx_G=None, chunk=8192, scan=None):
    """``crm.scan_interaction(G, idx_E, idx_G)`` over column chunks of the host matrix ``G`` with the finished chunks kept in
    ``checkpoint`` (an ``.npz`` file, rewritten atomically after every chunk): a job that is killed -- a pre-empted node, a
    wall-clock limit -- and started again with the same arguments scans only what is missing.  The reference has no such
    hook (SURVEY.md section 5 lists checkpoint / resume among the optional ones); its per-variant loop
    (cellregmap/_cellregmap.py:340) is what makes it safe: variants are independent, so any split into chunks gives the
    scan's own results (chunks that are multiples of the scan's block of 4096 variants: bit for bit).

    The file carries a fingerprint of the problem (phenotype, contexts, covariates, permutation hooks, the shape of ``G``
    and a digest of its first, middle and last columns); a checkpoint of another problem is refused, not overwritten.
    ``scan`` overrides the per-chunk call (tests inject the CPU oracle).  Returns ``(pvalues, info)`` like
    ``scan_interaction``."""
    import os
    import tempfile

    G = np.asarray(G)
    n, p = G.shape
    chunk = max(1, int(chunk))
    keys = ("pv", "rho1", "e2", "g2", "eps2")
    cols = sorted({0, p // 2, p - 1}) if p else []
    h = hashlib.blake2b(digest_size=16)      # (not _digest: the file must mean the same on a host without xxhash)
    for a in (crm._y, crm._E0, crm._W, [n, p, chunk], -1.0 if idx_E is None else idx_E, -1.0 if idx_G is None else idx_G,
              G[:, cols]):
        a = np.ascontiguousarray(a, dtype=float)
        h.update(str(a.shape).encode())
        h.update(a.view(np.uint8).data)
    finger = np.frombuffer(h.digest(), dtype=np.uint8)
    nchunks = (p + chunk - 1) // chunk
    out = {k: np.full(p, np.nan) for k in keys}
    done = np.zeros(nchunks, bool)
    if os.path.exists(checkpoint):
        with np.load(checkpoint) as old:
            if "fingerprint" not in old or not np.array_equal(old["fingerprint"], finger):
                raise ValueError(f"{checkpoint}: holds the checkpoint of another problem (other inputs, hooks or chunk size)")
            done = old["done"].copy()
            for k in keys:
                out[k] = old[k].copy()
    fn = scan if scan is not None else crm.scan_interaction

    def save():
        d = os.path.dirname(os.path.abspath(checkpoint))
        fd, tmp = tempfile.mkstemp(dir=d, suffix=".npz")
        os.close(fd)
        try:
            np.savez(tmp, fingerprint=finger, done=done, **out)
            os.replace(tmp, checkpoint)       # (atomic on POSIX: a reader never sees half a file)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    for ci in range(nchunks):
        if done[ci]:
            continue
        j0, j1 = ci * chunk, min(p, (ci + 1) * chunk)
        pv, info = fn(np.ascontiguousarray(G[:, j0:j1], dtype=float), idx_E, idx_G)
        out["pv"][j0:j1] = pv
        for k in keys[1:]:
            out[k][j0:j1] = info[k]
        done[ci] = True
        save()
    if nchunks == 0:
        save()
    return out["pv"], {k: out[k] for k in keys[1:]}


def run_interaction_many(Y, E, G, W=None, E1=None, E2=None, hK=None, *, cis_index=None, device=0):
    """``run_interaction`` for the columns of ``Y`` (n x genes) with one background decomposition,
    one genotype upload and shared per-variant work.  Returns arrays of shape (genes, p); with
    ``cis_index`` (see ``scan_interaction_many``) lists of per-phenotype arrays over each phenotype's own
    variants."""
    Y = np.asarray(Y, float)
    if Y.ndim != 2:
        raise ValueError("Y must be n x genes")
    if E1 is None:
        E1 = E
    if E2 is None:
        E2 = E
    Ls = None if hK is None else get_L_values(hK, E2)
    first = CellRegMap(y=Y[:, 0], E=E, W=W, E1=E1, Ls=Ls, device=device)
    crms = [first] + [CellRegMap(y=Y[:, i], E=E, W=W, E1=E1, Ls=Ls, device=device, background=first._bg)
                      for i in range(1, Y.shape[1])]
    return scan_interaction_many(crms, G, cis_index=cis_index)


def lrt_pvalues(null_lml, alt_lmls, dof=1):
    """Likelihood-ratio p-values with the reference's clips (_cellregmap.py:443-469)."""
    from scipy.stats import chi2

    tiny = float(np.finfo(float).eps)
    super_tiny = float(np.finfo(float).tiny)
    lrs = np.clip(-2 * null_lml + 2 * np.asarray(alt_lmls, float), super_tiny, np.inf)
    pv = chi2(df=dof).sf(lrs)
    return np.clip(pv, super_tiny, 1 - tiny)


def run_interaction(y, E, G, W=None, E1=None, E2=None, hK=None, idx_G=None, *, device=0):
    """Interaction test (_cellregmap.py:547-587).

    As in the reference, ``idx_G`` is forwarded positionally and therefore lands in
    ``scan_interaction``'s ``idx_E`` slot (:586 vs :318): it permutes the rows of the
    contexts inside the test direction, not the genotypes."""
    if E1 is None:
        E1 = E
    if E2 is None:
        E2 = E
    if hK is None:
        Ls = None
    else:
        Ls = get_L_values(hK, E2)
    crm = CellRegMap(y=y, E=E, W=W, E1=E1, Ls=Ls, device=device)
    pv = crm.scan_interaction(G, idx_G)
    return pv


def run_association(y, W, E, G, hK=None, *, device=0):
    """Association test (_cellregmap.py:471-500).  The reference's positional constructor call
    (:498) binds ``W`` to the contexts slot and ``E`` to the covariates slot; kept as is."""
    crm = CellRegMap(y, W, E, hK=hK, device=device)
    pv = crm.scan_association(G)
    return pv


def run_association_fast(y, W, E, G, hK=None, *, device=0):
    """Fast association test (_cellregmap.py:502-531); same positional binding (:529)."""
    crm = CellRegMap(y, W, E, hK=hK, device=device)
    pv = crm.scan_association_fast(G)
    return pv


def compute_maf(X):
    """Minor allele frequencies of a 0 / 1 / 2 (or dosage) matrix, samples along the first axis, NaN = missing
    (_cellregmap.py:589-638).  Like the reference it keeps the container it is given: a pandas DataFrame yields
    a Series named "maf" (indexed by the variant columns), an xarray DataArray a DataArray named "maf" (reduced
    over its "sample" dimension when it has one, else over axis 0), a dask array is reduced lazily and computed,
    anything else goes through numpy."""
    kind = type(X).__module__.split(".")[0]
    if kind == "dask":
        total = np.asarray(X.shape[0] - np.isnan(X).sum(axis=0))   # (dispatches to dask, then materialises)
        freq = np.asarray(np.nansum(X, axis=0)) / (2 * total)
    elif kind == "pandas" and hasattr(X, "isna") and getattr(X, "ndim", 0) == 2:
        freq = X.sum(axis=0, skipna=True) / (2 * X.notna().sum(axis=0))
    elif kind == "xarray":
        over = {"dim": "sample"} if "sample" in X.dims else {"axis": 0}
        freq = X.sum(skipna=True, **over) / (2 * X.notnull().sum(**over))
    else:
        X = np.asarray(X, float)
        freq = np.nansum(X, axis=0) / (2 * np.logical_not(np.isnan(X)).sum(axis=0))
    maf = np.minimum(freq, 1 - freq)
    if hasattr(maf, "name"):
        maf.name = "maf"
    return maf


def estimate_betas(y, W, E, G, maf=None, E1=None, E2=None, hK=None, *, device=0):
    """Effect sizes (_cellregmap.py:640-682): persistent effects and cell-level GxC effects."""
    E1 = E if E1 is None else E1
    E2 = E if E2 is None else E2
    Ls = None if hK is None else get_L_values(hK, E2)
    crm = CellRegMap(y=y, E=E, W=W, E1=E1, Ls=Ls, device=device, background=_DEFERRED)
    if maf is None:
        maf = compute_maf(G)
    return crm.predict_interaction(G, maf)
