"""How long does the driver take to hand out a big buffer?  (the first render of a scene allocates its path-state workspace: 44 GB for C2)
hipMalloc + first touch (hipMemset) + a second touch, for several sizes, twice, through libamdhip64 directly."""
import ctypes as C
import time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
hip.hipFree(None)
for rnd in range(2):
    for gb in (1, 7, 14, 44):
        n = gb << 30
        p = C.c_void_p()
        t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), n); hip.hipDeviceSynchronize(); t1 = time.perf_counter()
        hip.hipMemset(p, 0, n); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipMemset(p, 1, n); hip.hipDeviceSynchronize(); t3 = time.perf_counter()
        hip.hipFree(p); hip.hipDeviceSynchronize(); t4 = time.perf_counter()
        print("round %d: %2d GiB: hipMalloc %7.1f ms (rc %d), first memset %7.1f ms, second %7.1f ms, hipFree %7.1f ms" % (rnd, gb, (t1 - t0) * 1e3, rc, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
