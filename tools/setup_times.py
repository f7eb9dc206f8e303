"""Where a one-shot render's set-up time goes: scene generation (numpy), BVHAccel (refine + dr_bvh_build_device + reorder),
dr_scene_create (validation, pair records, upload), first render, second render.  usage: setup_times.py C4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dartray_amd import _abi, scenes, core
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
_abi.init(0)
t = time.time(); prims, mk = scenes.config(cfg); t_gen = time.time() - t
r = mk()
t = time.time(); acc = core.BVHAccel(prims); t_bvh = time.time() - t
lights = acc.lights() + ([r.env] if getattr(r, "env", None) is not None else [])
scene = core.Scene(acc, lights)
t = time.time(); dev = scene._device(); t_create = time.time() - t
import torch
film = torch.zeros((r.camera.film.height, r.camera.film.width, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize(); t = time.time(); r.render_device(scene, film.data_ptr(), s); torch.cuda.synchronize(); t_first = time.time() - t
film.zero_(); torch.cuda.synchronize(); t = time.time(); r.render_device(scene, film.data_ptr(), s); torch.cuda.synchronize(); t_second = time.time() - t
print("%s: scene generation %.2f s | BVHAccel %.3f s (builder %s: %.0f ms) | dr_scene_create %.3f s | first render %.3f s | second render %.3f s"
      % (cfg, t_gen, t_bvh, acc.builder, acc.build_ms, t_create, t_first, t_second))
