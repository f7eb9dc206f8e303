"""Times dr_bvh_build (host SAH builder) for several thread counts and checks that every
thread count yields the same bytes."""
import ctypes as C, os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dartray_amd import scenes, core, _abi

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
threads = [int(t) for t in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "8", "32", "64"])]
t = time.time(); prims, _ = scenes.config(cfg); print("scene gen %.2f s" % (time.time() - t))
os.environ["DARTRAY_BUILD_THREADS"] = "1"
acc = core.BVHAccel(prims)
inv = np.empty_like(acc.order); inv[acc.order] = np.arange(len(acc.order))
refined = np.ascontiguousarray(acc.tri_idx[inv]); n = len(refined)
lib = _abi.lib()
ref = hashlib.sha1(acc.nodes.tobytes()).hexdigest()
for th in threads:
    os.environ["DARTRAY_BUILD_THREADS"] = str(th)
    best = 1e9
    for rep in range(2):
        nodes = np.ones(2 * n - 1, dtype=core.NODE_DTYPE); order = np.ones(n, np.uint32); nn = C.c_uint64(); d = C.c_uint32()
        t = time.time()
        lib.dr_bvh_build(acc.verts.ctypes.data, len(acc.verts), refined.ctypes.data, n, 4, nodes.ctypes.data, C.byref(nn), order.ctypes.data, C.byref(d))
        best = min(best, time.time() - t)
    ok = hashlib.sha1(nodes[:nn.value].tobytes()).hexdigest() == ref and np.array_equal(order, acc.order)
    print("%s: %d tris, %d nodes, depth %d, threads %3d: %.3f s  identical=%s" % (cfg, n, nn.value, d.value, th, best, ok))
