"""hipMalloc / hipFree cost: the local predictor keeps its scratch slab between calls because growing it
(free a few GiB that were written, then allocate tens of GiB) can stall for more than a second."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")

def malloc(gib):
    p = ctypes.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(int(gib * (1 << 30))))
    return p, 1e3 * (time.perf_counter() - t0), rc

def touch(p, gib):
    t0 = time.perf_counter()
    hip.hipMemset(p, 0, ctypes.c_size_t(int(gib * (1 << 30))))
    hip.hipDeviceSynchronize()
    return 1e3 * (time.perf_counter() - t0)

def free(p):
    t0 = time.perf_counter()
    hip.hipFree(p)
    return 1e3 * (time.perf_counter() - t0)

p, t, _ = malloc(0.1); touch(p, 0.1); free(p)
for rep in range(3):
    for small, big, order in ((5, 32, "free-then-malloc"), (5, 32, "malloc-then-free"), (0.05, 32, "free-then-malloc")):
        p, tm, _ = malloc(small)
        tt = touch(p, small)
        if order == "free-then-malloc":
            tf = free(p)
            q, tb, rc = malloc(big)
        else:
            q, tb, rc = malloc(big)
            tf = free(p)
        tq = touch(q, 1)
        tf2 = free(q)
        print(f"rep {rep} {order:17s} small {small:5.2f} GiB: malloc {tm:7.1f} touch {tt:7.1f} free {tf:7.1f} | "
              f"big {big} GiB: malloc {tb:8.1f} touch(1GiB) {tq:6.1f} free {tf2:7.1f} ms rc {rc}", flush=True)
