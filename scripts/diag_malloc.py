"""hipMalloc / hipFree cost by size (the local predictor's scratch slab is tens of GiB)."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
for rep in range(2):
    for gib in (1, 4, 8, 16, 32):
        p = ctypes.c_void_p()
        t0 = time.perf_counter()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(gib << 30))
        t1 = time.perf_counter()
        hip.hipMemset(p, 0, ctypes.c_size_t(1 << 20)); hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipFree(p)
        t3 = time.perf_counter()
        print(f"rep {rep} {gib:3d} GiB  malloc {1e3*(t1-t0):8.1f} ms  first touch {1e3*(t2-t1):6.1f} ms  free {1e3*(t3-t2):8.1f} ms  rc {rc}", flush=True)
