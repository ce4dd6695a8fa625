import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); hip.hipFree(p)
for gb in (1, 8, 8, 32, 32):
    t0 = time.time(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gb << 30)); t1 = time.time()
    hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30)); hip.hipDeviceSynchronize(); t2 = time.time()
    hip.hipMemset(p, 0, ctypes.c_size_t(gb << 30)); hip.hipDeviceSynchronize(); t3 = time.time()
    hip.hipFree(p); t4 = time.time()
    print(f"{gb} GB: malloc {t1-t0:.3f} s (rc {rc}), first memset {t2-t1:.3f} s, second memset {t3-t2:.3f} s, free {t4-t3:.3f} s")
