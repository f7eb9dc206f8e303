"""Helpers shared by the tests (inputs only; the checker lives in oracle/)."""
import numpy as np

from dartray_amd import core, pbrt, scenes


def aggregate_test_rays(bmin, bmax, n, seed=1, hits=None):
    """Random rays after the reference's AggregateTestRenderer recipe
    (lib/renderers/aggregate_test_renderer.dart:42-118): origins in the world box expanded by its
    own extent, directions uniform on the sphere with a 1/32 chance of being axis aligned, tmin in
    {0, 1e-3}, a quarter of the origins placed on points of the scene surface (`hits`)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bmin = np.asarray(bmin, np.float64)
    bmax = np.asarray(bmax, np.float64)
    ext = bmax - bmin
    lo, hi = bmin - ext, bmax + ext
    o = lo + rng.random((n, 3)) * (hi - lo)
    pick = np.zeros(n, bool)
    if hits is not None and len(hits):
        pick = rng.random(n) < 0.25
        o[pick] = hits[rng.integers(0, len(hits), pick.sum())]
    z = 1.0 - 2.0 * rng.random(n)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    phi = 2.0 * np.pi * rng.random(n)
    d = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
    axis = rng.random(n) < (1.0 / 32.0)
    ax = np.zeros((n, 3))
    which = rng.integers(0, 3, n)
    ax[np.arange(n), which] = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    d[axis] = ax[axis]
    tmin = np.where(rng.random(n) < 0.5, 0.0, 1.0e-3)
    # Rays that start ON a surface use the epsilon real secondary rays carry (isect.rayEpsilon): with tmin == 0 a
    # t == 0 hit is accepted by Triangle.intersect but its flat leaf box fails `tmax > minDistance`
    # (bvh_accel.dart:471), a disagreement the reference's AggregateTestRenderer would merely log.
    tmin[pick] = 1.0e-3
    tmax = np.where(rng.random(n) < 0.25, rng.random(n) * np.linalg.norm(ext) * 2.0, np.inf)
    return o.astype(np.float32), d.astype(np.float32), tmin, tmax


def rel_err_image(gpu, ref):
    """SURVEY.md section 8d parity metric: per pixel max_c |gpu-ref| / max(max_c |ref|, 1e-6)."""
    denom = np.maximum(np.abs(ref).max(axis=-1), 1e-6)
    return np.abs(gpu - ref).max(axis=-1) / denom


def quadric_prims(seed=3, nspheres=24, ndisks=12):
    """Random full / clipped spheres and (annular, partial) disks under random affine transforms, mixed with
    the Cornell walls so that leaves hold triangles and quadrics side by side."""
    rng = np.random.Generator(np.random.PCG64(seed))
    prims = scenes.cornell_walls()
    for k in range(nspheres + ndisks):
        t = pbrt.Transform.Translate(*(rng.random(3) * 16 - 8))
        t = t * pbrt.Transform.Rotate(rng.random() * 360, *(rng.random(3) - 0.5))
        if k % 3 == 0:
            t = t * pbrt.Transform.Scale(*(0.5 + rng.random(3)))
        ro = bool(k % 5 == 0)
        if k < nspheres:
            r = 0.5 + 2 * rng.random()
            kw = {}
            if k % 2:
                kw = dict(z0=-r * rng.random(), z1=r * rng.random(), phiMax=90 + 270 * rng.random())
            shape = core.Sphere(t.m, t.mInv, ro, r, **kw)
        else:
            r = 0.5 + 2 * rng.random()
            shape = core.Disk(t.m, t.mInv, ro, height=rng.random() - 0.5, radius=r,
                              innerRadius=(r * 0.5 * rng.random() if k % 2 else 0.0),
                              phiMax=(360.0 if k % 4 else 200.0))
        prims.append(core.GeometricPrimitive(shape, core.MatteMaterial(tuple(0.2 + 0.6 * rng.random(3)))))
    return prims
