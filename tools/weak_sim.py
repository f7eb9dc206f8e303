import sys, time
sys.path.insert(0, '.')
import torch
from dartray_amd import _abi, scenes, dist as drdist
_abi.init(0)
for world, res in ((1, 1024), (2, 1440), (4, 2048), (8, 2880)):
    prims, mk = scenes.config("C2", xres=res, yres=res, spp=256)
    r = drdist.shard(mk(), 0, world)
    scene = scenes.make_scene(prims)
    film = torch.zeros((res, res, 4), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        r.render_device(scene, film.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        r.render_device(scene, film.data_ptr(), s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    n = len(r.pixels()) * 256
    print("world %d res %d: rank-0 share %.3e samples, %.1f ms/step, %.1f Msamples/s per GPU" % (world, res, n, dt * 1e3, n / dt / 1e6))
    del scene, film
