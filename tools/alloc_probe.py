"""Round 5: why does the first C2 render take 0.5 - 1 s longer on some boxes?  The 51 GB path-state buffer is ONE hipMalloc.
Times, each in a fresh process (the runtime caches freed memory): one 51 GiB hipMalloc; the same bytes as 4 / 16 pieces; and a
contiguous virtual range backed by 1 GiB physical chunks (hipMemAddressReserve + hipMemCreate + hipMemMap + hipMemSetAccess)."""
import ctypes as C
import subprocess
import sys
import time

GB = 1 << 30


def run(mode):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipSetDevice(0)
    hip.hipFree(None)
    total = 51 * GB
    t0 = time.perf_counter()
    if mode.startswith("malloc"):
        pieces = int(mode[6:] or 1)
        ps = []
        for i in range(pieces):
            p = C.c_void_p()
            rc = hip.hipMalloc(C.byref(p), total // pieces)
            assert rc == 0, rc
            ps.append(p)
        hip.hipDeviceSynchronize()
        t1 = time.perf_counter()
        hip.hipMemset(ps[0], 0, total // pieces)
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        print("%-10s alloc %8.1f ms, first touch of piece 0 (%5.1f GiB) %7.1f ms" % (mode, (t1 - t0) * 1e3, total / pieces / GB, (t2 - t1) * 1e3))
    else:
        chunk = int(mode[3:]) * GB  # vmmN: N GiB chunks

        class Prop(C.Structure):
            _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("loc_type", C.c_int), ("loc_id", C.c_int), ("win32", C.c_void_p),
                        ("flags", C.c_ubyte * 8)]

        class Access(C.Structure):
            _fields_ = [("loc_type", C.c_int), ("loc_id", C.c_int), ("flags", C.c_int)]
        prop = Prop(type=1, requestedHandleType=0, loc_type=1, loc_id=0)  # hipMemAllocationTypePinned, hipMemLocationTypeDevice
        gran = C.c_size_t(0)
        rc = hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 0)
        base = C.c_void_p()
        hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        rc = hip.hipMemAddressReserve(C.byref(base), total, 0, None, 0)
        assert rc == 0, ("reserve", rc)
        hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(Prop), C.c_ulonglong]
        hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Access), C.c_size_t]
        acc = Access(loc_type=1, loc_id=0, flags=3)
        off = 0
        while off < total:
            h = C.c_void_p()
            n = min(chunk, total - off)
            rc = hip.hipMemCreate(C.byref(h), n, C.byref(prop), 0)
            assert rc == 0, ("create", rc)
            rc = hip.hipMemMap(C.c_void_p(base.value + off), n, 0, h, 0)
            assert rc == 0, ("map", rc)
            off += n
        rc = hip.hipMemSetAccess(base, total, C.byref(acc), 1)
        assert rc == 0, ("access", rc)
        hip.hipDeviceSynchronize()
        t1 = time.perf_counter()
        hip.hipMemset(base, 0, 4 * GB)
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        print("%-10s granularity %d: reserve + create + map + access %8.1f ms, first touch of 4 GiB %7.1f ms" % (mode, gran.value, (t1 - t0) * 1e3, (t2 - t1) * 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for rep in range(2):
            for mode in ("malloc1", "malloc4", "malloc16", "vmm1", "vmm4"):
                r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True, timeout=300)
                sys.stdout.write(r.stdout if r.returncode == 0 else "%s FAILED: %s\n" % (mode, r.stderr[-400:]))
                sys.stdout.flush()
