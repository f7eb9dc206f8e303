"""dr_scene_create's work on the marshalled tree, on the device (dartray_amd/csrc/dr_scene_prep.hip): validation of every node
and primitive, the tree's height, the sibling-pair records in their memory order, the union check.  The serial host loops of
round 3 (DARTRAY_SCENE_PREP=host through dr_set_option) are the reference: same errors, same height, the same pair records byte
for byte."""
import ctypes as C

import numpy as np
import pytest

from dartray_amd import _abi, core, scenes

pytestmark = pytest.mark.gpu


def _pairs(dev):
    lib = _abi.lib()
    n, top, depth = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
    _abi.check(lib.dr_scene_get_pairs(dev.handle, None, 0, C.byref(n), C.byref(top), C.byref(depth)))
    out = np.zeros((n.value, 16), dtype=np.uint32)
    if n.value:
        _abi.check(lib.dr_scene_get_pairs(dev.handle, out.ctypes.data, out.nbytes, C.byref(n), C.byref(top), C.byref(depth)))
    return out, int(top.value), int(depth.value)


def _create(acc, prep):
    lib = _abi.lib()
    _abi.check(lib.dr_set_option(b"SCENE_PREP", prep))
    try:
        return core._DeviceScene(acc, acc.lights())
    finally:
        _abi.check(lib.dr_set_option(b"SCENE_PREP", None))


@pytest.mark.parametrize("blob", [(2, 2), (8, 4), (60, 30), (333, 77)])   # trees lower and higher than the breadth-first top (12 levels)
def test_device_prep_equals_the_host_loops(gpu, blob):
    prims, _ = scenes.config("C2", xres=8, yres=8, spp=1, blob=blob)
    acc = core.BVHAccel(prims)
    dev, host = _create(acc, None), _create(acc, b"host")
    pd, topd, depthd = _pairs(dev)
    ph, toph, depthh = _pairs(host)
    assert len(pd) == (len(acc.nodes) - 1) // 2 and depthd == depthh == acc.depth and topd == toph
    assert pd.tobytes() == ph.tobytes()
    assert topd > 0
    # the height is measured when the host passes none, and a bound below it is refused
    acc2 = core.BVHAccel(prims)
    acc2.depth = 0
    assert _pairs(_create(acc2, None))[2] == acc.depth
    acc2.depth = acc.depth - 1
    for prep in (None, b"host"):
        with pytest.raises(_abi.DartRayHipError, match="bvh_depth is smaller"):
            _create(acc2, prep)


def test_full_size_c4_pairs_equal(gpu):
    prims, _ = scenes.config("C4")
    acc = core.BVHAccel(prims)
    import time
    t0 = time.time()
    dev = _create(acc, None)
    t_dev = time.time() - t0
    pd, topd, depthd = _pairs(dev)
    dev.destroy()
    t0 = time.time()
    host = _create(acc, b"host")
    t_host = time.time() - t0
    ph, toph, depthh = _pairs(host)
    print("C4 dr_scene_create: device prep %.3f s, host loops %.3f s; %d pairs, top %d, depth %d" % (t_dev, t_host, len(pd), topd, depthd))
    assert topd == toph and depthd == depthh and pd.tobytes() == ph.tobytes()
    assert t_dev < 0.45  # (measured ~0.15 s; the host loops 0.6 s)


def _bad(acc, mutate, match):
    for prep in (None, b"host"):
        a = core.BVHAccel(acc.prims_in)
        mutate(a)
        with pytest.raises(_abi.DartRayHipError, match=match):
            _create(a, prep)


def test_malformed_trees_and_tables_are_refused_like_before(gpu):
    """A foreign host's arrays are input: a malformed node, a primitive range or an index out of bounds must come back as
    an error from dr_scene_create, by the device-side checks as by the host loops."""
    prims, _ = scenes.config("C2", xres=8, yres=8, spp=1, blob=(24, 12))
    acc = core.BVHAccel(prims)
    interior = np.flatnonzero(acc.nodes["nprims"] == 0)
    leaves = np.flatnonzero(acc.nodes["nprims"] != 0)
    i, j, l = int(interior[len(interior) // 2]), int(interior[3]), int(leaves[len(leaves) // 3])

    def set_node(k, field, value):
        def f(a):
            a.nodes[k][field] = value
        return f

    _bad(acc, set_node(i, "offset", i + 1), "malformed BVH node")            # second child == first child
    _bad(acc, set_node(i, "offset", 3), "malformed BVH node")                # points backwards: an endless walk
    _bad(acc, set_node(j, "offset", len(acc.nodes) + 7), "malformed BVH node")
    _bad(acc, set_node(j, "axis", 3), "malformed BVH node")
    _bad(acc, set_node(l, "offset", len(acc.tri_idx)), "leaf primitive range")

    def bad_vertex(a):
        a.tri_idx[5, 1] = len(a.verts) + 1
    _bad(acc, bad_vertex, "vertex index out of range")

    def bad_material(a):
        a.tri_material[7] = 99
    _bad(acc, bad_material, "material index out of range")

    # a node that is the child of two nodes is not a tree: here the root's whole right sub-tree hangs a second time below a
    # small node of the left one, so the device walk reaches more nodes than the array holds (the host loops only measure
    # a height; either way every traversal stays finite -- children always lie behind their parent)
    a = core.BVHAccel(acc.prims_in)
    right = int(a.nodes[0]["offset"])
    k = int(interior[interior < right - 2][-1])
    a.nodes[k]["offset"] = right
    with pytest.raises(_abi.DartRayHipError, match="child of two nodes|deeper than"):
        _create(a, None)

    # a box that is not the union of its children's: still a valid scene, but the pair kernels cannot be used
    a = core.BVHAccel(acc.prims_in)
    a.nodes[i]["bmax"][0] += 1.0
    for prep in (None, b"host"):
        d = _create(a, prep)
        assert len(_pairs(d)[0]) == 0


@pytest.mark.parametrize("prep", [None, b"host"])
def test_pair_records_of_either_prep_trace_the_oracles_hits(gpu, ob, prep):
    """The pair records as the device lays them out, and as the host loops do, give the oracle's hits (dr_intersect through the
    pair kernels).  (Round 3's other memory orders of the records: experiments/r06_runtime_switches.diff.)"""
    prims, _ = scenes.config("C2", xres=8, yres=8, spp=1, blob=(40, 20))
    acc = core.BVHAccel(prims)
    lib = _abi.lib()
    rng = np.random.default_rng(3)
    n = 20000
    o = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ray = core.Ray(o, d, 1e-3, np.inf)
    osc = ob.OracleScene(prims)
    rays = np.zeros(n, dtype=ob.RAY_DTYPE)
    rays["o"], rays["d"], rays["tmin"], rays["tmax"] = o, d, 1e-3, np.inf
    ref = osc.intersect(rays, any_hit=False)
    try:
        _abi.check(lib.dr_set_option(b"TRACE_IMPL", b"3"))
        dev = _create(acc, prep)
        out = dev.intersect(ray, any_hit=False)
    finally:
        _abi.check(lib.dr_set_option(b"TRACE_IMPL", None))
    assert len(_pairs(dev)[0]) > 0
    assert np.array_equal(out["prim"], ref["prim"]) and np.array_equal(out["t"], ref["t"])
