"""SURVEY.md section 8 row f1 on the device: dr_bvh_build_device (HIP: device-wide reductions for the bounds, the
centroid bounds and the 12 SAH buckets, the reference's two-pointer `partition` as rank-paired swaps, small sub-trees
finished by one thread each) must write the bytes of dr_bvh_build -- which the oracle's serial restatement of
bvh_accel.dart:41-91,228-437 / common.dart:256-297 and the Python one equal (tests/test_host_logic.py,
tests/test_restatement.py) -- node for node and primitive for primitive."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from dartray_amd import core, scenes

pytestmark = pytest.mark.gpu


def _inputs(prims):
    acc = core.BVHAccel(prims, builder="host")
    inv = np.empty_like(acc.order)
    inv[acc.order] = np.arange(len(acc.order), dtype=acc.order.dtype)
    refined = np.ascontiguousarray(acc.tri_idx[inv])  # the refined (input) primitive order
    qb = np.zeros((max(len(acc.quadrics), 1), 6), dtype=np.float32)
    for i, q in enumerate(acc.quadrics):
        lo, hi = q.worldBound()
        qb[i, :3], qb[i, 3:] = lo, hi
    return acc, refined, qb


def _same(acc, refined, qb, max_prims=4):
    t0 = time.time()
    nodes, order, nn, depth, ran = core.build_bvh_arrays(acc.verts, refined, qb, len(acc.quadrics), max_prims, "device")
    dt = time.time() - t0
    assert ran == "device"
    hn, ho, hnn, hdepth, _ = core.build_bvh_arrays(acc.verts, refined, qb, len(acc.quadrics), max_prims, "host")
    assert nn == hnn and depth == hdepth
    assert np.array_equal(order[:len(refined)], ho[:len(refined)])
    assert nodes[:nn].tobytes() == hn[:hnn].tobytes()
    return dt, nn, depth


@pytest.mark.parametrize("case", ["c1", "c2small", "cspec", "cenv", "cquad", "blob6k"])
def test_device_builder_equals_the_host_builder_on_the_restatement_scenes(gpu, case):
    import test_restatement as tr
    if case == "blob6k":
        prims, _ = scenes.config("C2", xres=8, yres=8, spp=1, blob=(80, 40))
    else:
        prims = tr._cases()["restatement_%s.npz" % case][1]
    acc, refined, qb = _inputs(prims)
    _same(acc, refined, qb)
    # and through the host class: the aggregate the renderer uses
    dev = core.BVHAccel(prims, builder="device")
    assert dev.builder == "device"
    assert dev.nodes.tobytes() == acc.nodes.tobytes() and np.array_equal(dev.tri_idx, acc.tri_idx)
    # and DIRECTLY against the oracle's serial restatement of bvh_accel.dart:228-437 (not only through the host builder)
    import oracle.binding as ob
    onodes, otri, _, _ = ob.OracleScene(prims).bvh()
    assert dev.nodes.tobytes() == onodes.tobytes()
    if case != "cquad":  # (a quadric's row of tri_idx is a tag + its index: the product's own encoding)
        assert np.array_equal(dev.tri_idx, otri)


@pytest.mark.parametrize("blob,max_prims", [((200, 100), 4), ((200, 100), 1), ((200, 100), 16), ((200, 100), 255), ((333, 77), 4)])
def test_levels_small_sub_trees_and_max_prims(gpu, blob, max_prims):
    """40 000 / 51 282 triangles: several levels by all threads, then ~1000 single-thread sub-trees; maxnodeprims from 1
    to 255 (leaves of several primitives appear when the SAH says so, bvh_accel.dart:377-385)."""
    prims, _ = scenes.config("C2", xres=8, yres=8, spp=1, blob=blob)
    acc, refined, qb = _inputs(prims)
    _same(acc, refined, qb, max_prims)


def test_coincident_centroids_and_signed_zeros(gpu):
    """Leaves of several primitives where every centroid coincides on the widest axis (bvh_accel.dart:265-274), and
    boxes whose extreme coordinate is a zero of both signs: Math.min(0.0, -0.0) is -0.0 and Math.max is +0.0 whatever the
    order (dart:math), on the device's integer keys as in the host builder."""
    rng = np.random.default_rng(7)
    tris = []
    for k in range(300):  # stacks of identical triangles
        base = rng.uniform(-5, 5, 3).astype(np.float32)
        t = np.array([base, base + [1, 0, 0], base + [0, 1, 0]], np.float32)
        for _ in range(int(rng.integers(1, 6))):
            tris.append(t.copy())
    for k in range(400):  # triangles touching the planes x = +-0, y = +-0 with both zero signs
        a = rng.uniform(0, 3, 3).astype(np.float32)
        t = np.array([[0.0, a[1], a[2]], [a[0], -0.0 if k & 1 else 0.0, a[2]], [-0.0 if k & 2 else 0.0, a[1], -a[2]]], np.float32)
        tris.append(t)
    verts = np.ascontiguousarray(np.concatenate(tris), np.float32)
    idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    qb = np.zeros((1, 6), np.float32)
    hn, ho, hnn, hd, _ = core.build_bvh_arrays(verts, idx, qb, 0, 4, "host")
    dn, do, dnn, dd, _ = core.build_bvh_arrays(verts, idx, qb, 0, 4, "device")
    assert dnn == hnn and dd == hd and np.array_equal(do, ho) and dn[:dnn].tobytes() == hn[:hnn].tobytes()
    assert (hn[:hnn]["nprims"] > 1).any()
    zeros = hn[:hnn]["bmin"][hn[:hnn]["bmin"] == 0]
    assert len(zeros) and np.signbit(zeros).any()  # a box that reaches a zero plane from both sides starts at -0.0


@pytest.mark.parametrize("n,kind,seed", [(1, "soup", 1), (2, "soup", 2), (3, "soup", 3), (4, "soup", 4), (5, "soup", 5), (63, "soup", 6), (64, "soup", 7),
                                         (65, "soup", 8), (128, "clusters", 9), (129, "soup", 10), (1000, "clusters", 11), (4097, "soup", 12),
                                         (20000, "line", 13), (100003, "clusters", 14), (300000, "soup", 15), (50000, "grid", 16)])
def test_random_triangle_soups(gpu, n, kind, seed):
    """Sizes around the single-thread threshold (64), the wave and tile sizes of the reduction kernels, and distributions
    that stress the split logic: uniform soups, tight clusters far apart (empty SAH buckets, huge coordinate ranges), points
    on a line (two degenerate axes), a regular grid (many equal centroids: ties in the bucket index and in the 4-element sort)."""
    rng = np.random.default_rng(seed)
    if kind == "soup":
        c = rng.uniform(-100, 100, (n, 1, 3))
        v = c + rng.normal(0, 1.5, (n, 3, 3))
    elif kind == "clusters":
        centres = rng.uniform(-1e4, 1e4, (7, 3))
        c = centres[rng.integers(0, 7, n)][:, None, :] + rng.normal(0, 0.01, (n, 1, 3))
        v = c + rng.normal(0, 0.001, (n, 3, 3))
    elif kind == "line":
        t = rng.uniform(-50, 50, (n, 1, 1))
        v = np.concatenate([t + rng.normal(0, 0.01, (n, 3, 1)), np.zeros((n, 3, 1)), np.full((n, 3, 1), 2.0)], axis=2)
    else:  # grid: integer lattice, many coincident and equal centroids
        g = rng.integers(0, 12, (n, 1, 3)).astype(np.float64)
        v = g + np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float64)[None]
    verts = np.ascontiguousarray(v.reshape(-1, 3), np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    qb = np.zeros((1, 6), np.float32)
    for max_prims in ((4,) if n > 5000 else (4, 1, 8)):
        hn, ho, hnn, hd, _ = core.build_bvh_arrays(verts, idx, qb, 0, max_prims, "host")
        dn, do, dnn, dd, _ = core.build_bvh_arrays(verts, idx, qb, 0, max_prims, "device")
        assert dnn == hnn and dd == hd, (n, kind, max_prims)
        assert np.array_equal(do[:n], ho[:n]) and dn[:dnn].tobytes() == hn[:hnn].tobytes(), (n, kind, max_prims)


@pytest.mark.parametrize("cfg,budget_s", [("C2", 0.1), ("C5", 0.25), ("C4", 0.25)])
def test_full_size_configs_equal_the_host_builder(gpu, cfg, budget_s):
    """BASELINE.json's scenes at full size: 1 000 012, ~8 M and 10 000 012 triangles."""
    prims, _ = scenes.config(cfg)
    acc, refined, qb = _inputs(prims)
    _same(acc, refined, qb)            # first call: includes one-off allocation effects
    dt, nn, depth = _same(acc, refined, qb)
    print("%s: %d triangles, %d nodes, depth %d: dr_bvh_build_device %.3f s (host pointers in and out)" % (cfg, len(refined), nn, depth, dt))
    assert dt < 1.5 * budget_s  # (measured: C2 0.03 s, C5 0.09 s, C4 0.14 s)
