"""Fixed cost of a render: BASELINE configs[0] (C1: Cornell box, 2 triangles + 1 area light, DirectLighting, 64 x 64, 4 spp) and a few
other small renders through the public host API, wall time per render() after a warm-up (HBM-resident scene, film copied out)."""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
from dartray_amd import _abi, scenes

_abi.init(0)
for name, kw in (("C1", {}), ("C2", dict(xres=64, yres=64, spp=4, blob=(40, 20))), ("C2", dict(xres=256, yres=256, spp=16, blob=(200, 100))),
                 ("C2", dict(xres=256, yres=256, spp=64))):
    prims, mk = scenes.config(name, **kw)
    scene = scenes.make_scene(prims)
    r = mk()
    r.render(scene)
    r.render(scene)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = r.render(scene)
    dt = (time.perf_counter() - t0) / n
    st = r.last_stats
    samples = st["camera_samples"]
    print("%s %s: %.3f ms per render, %d samples, %.2f Msamples/s; kernels: closest %.3f any %.3f shade %.3f gen %.3f film %.3f ms (sum %.3f)" % (
        name, kw, dt * 1e3, samples, samples / dt / 1e6, st["closest_ms"], st["any_ms"], st["shade_ms"], st["gen_ms"], st["film_ms"],
        st["closest_ms"] + st["any_ms"] + st["shade_ms"] + st["gen_ms"] + st["film_ms"]))
