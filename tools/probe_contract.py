"""GPU probe: contraction hooks vs numpy (binds only the symbols it needs)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "cellregmap_amd", "libcrm_hip.so"))
vp = ctypes.c_void_p
lib.crm_last_error.restype = ctypes.c_char_p
h = vp()
assert lib.crm_ctx_create(0, ctypes.byref(h)) == 0, lib.crm_last_error()
rng = np.random.default_rng(0)
X = rng.integers(-3, 4, size=(24, 37)).astype(float); Y = rng.integers(-3, 4, size=(24, 150)).astype(float)
X[:, 5] = np.arange(24); Y[:, 7] = np.arange(24) ** 2
C = np.empty((37, 150))
lib.crm_test_contract.argtypes = [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int]
rc = lib.crm_test_contract(h, 24, 37, 150, X.ctypes.data, Y.ctypes.data, C.ctypes.data, 1)
print("rc", rc, lib.crm_last_error(), "exact:", np.array_equal(C, X.T @ Y), "maxdiff", np.abs(C - X.T @ Y).max())
if not np.array_equal(C, X.T @ Y):
    print("transposed?", np.array_equal(C, (Y.T @ X).T))
    D = C - X.T @ Y
    print(np.argwhere(D != 0)[:10])
for cells, M, N, ks in [(1000, 130, 257, 1), (4096, 256, 128, 4), (20000, 200, 1275, 5)]:
    X = rng.normal(size=(cells, M)); Y = rng.normal(size=(cells, N)); C = np.empty((M, N))
    rc = lib.crm_test_contract(h, cells, M, N, X.ctypes.data, Y.ctypes.data, C.ctypes.data, ks)
    print(cells, M, N, ks, "rc", rc, "maxdiff", np.abs(C - X.T @ Y).max())
lib.crm_test_contract_kr.argtypes = [vp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
for cells, B, k0, N in [(500, 7, 10, 140), (2048, 20, 50, 300), (333, 40, 3, 64), (1024, 5, 128, 130), (640, 300, 1, 128)]:
    G = rng.normal(size=(cells, B)); E = rng.normal(size=(cells, k0)); Y = rng.normal(size=(cells, N)); C = np.empty((B * k0, N))
    rc = lib.crm_test_contract_kr(h, cells, B, k0, N, G.ctypes.data, E.ctypes.data, Y.ctypes.data, C.ctypes.data)
    KR = (G[:, :, None] * E[:, None, :]).reshape(cells, B * k0)
    print("KR", cells, B, k0, N, "rc", rc, "maxdiff", np.abs(C - KR.T @ Y).max())
