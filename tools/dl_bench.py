"""Time DirectLighting on the C2 scene (1024^2, 64 spp): the k_shade_direct path.  usage: python tools/dl_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dartray_amd import _abi, core, scenes
_abi.init(0)
prims, mk = scenes.config("C2", spp=64)
r = mk()
r.surfaceIntegrator = core.DirectLightingIntegrator(0, 5)
scene = scenes.make_scene(prims)
r.render(scene)
t0 = time.time()
for _ in range(3):
    r.render(scene)
dt = (time.time() - t0) / 3
st = r.last_stats
print("DL C2 64spp: %.1f ms/render, %.1f Msamples/s, shade %.1f ms" % (dt * 1e3, 1024 * 1024 * 64 / dt / 1e6, st.get("shade_ms", 0)))
