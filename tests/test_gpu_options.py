"""The switches of the library through the C ABI (dr_set_option), and one process that renders several scenes in turn.

Round 4's dr_option handed out a pointer into ONE thread-local buffer: with STATE_LAYOUT and a second switch both set through
dr_set_option the forced layout was read from the pilot switch's value.  Environment variables hid it (getenv pointers are
stable), and every earlier test that combined switches used the environment.  These set them through the ABI only and ask
the library what the render actually ran with (dr_scene_last_render_info)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(code, timeout=600, **env_extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DARTRAY_") or k in ("DARTRAY_LIB", "DARTRAY_RCCL_LIB")}
    env.update(env_extra)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0 and "OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
    return res


def test_switches_set_through_the_c_abi_do_not_clobber_each_other():
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from dartray_amd import _abi, scenes\n"
        "_abi.init(0)\n"
        "lib = _abi.lib()\n"
        "def opt(name, value):\n"
        "    _abi.check(lib.dr_set_option(name.encode(), None if value is None else str(value).encode()))\n"
        "prims, mk = scenes.config('C2', xres=40, yres=40, spp=16, blob=(60, 30))\n"
        "r = mk(); scene = scenes.make_scene(prims); dev = scene._device()\n"
        "ref = r.render(scene).film\n"
        "info = dev.last_render_info(); assert info['state_layout'] == 64, info\n"
        # the round-4 failure: the layout switch first, then two more look-ups of other names (one with a LONGER value)
        "opt('STATE_LAYOUT', 4); opt('BATCH_BITS', 28); opt('VERBOSE', '0000000000000000000000000000000000000001')\n"
        "f = r.render(scene).film; info = dev.last_render_info()\n"
        "assert info['state_layout'] == 4, info\n"
        "assert np.array_equal(f, ref)\n"
        "opt('STATE_LAYOUT', 64); f = r.render(scene).film; assert dev.last_render_info()['state_layout'] == 64\n"
        "opt('STATE_LAYOUT', None); opt('BATCH_BITS', None); opt('VERBOSE', None)\n"
        # kernel switches: the pair kernels with the cold-state closest-hit variant
        "opt('TRACE_IMPL', 5); opt('OVERLAP_ANY', 0)\n"
        "f = r.render(scene).film; info = dev.last_render_info()\n"
        "assert (info['closest_kernel'], info['any_hit_kernel'], info['overlap_any']) == (5, 3, 0), info\n"
        "assert np.array_equal(f, ref)\n"
        "opt('OVERLAP_ANY', None); opt('TRACE_IMPL', 2); opt('TRACE_WG_PER_CU', 5)\n"
        "f = r.render(scene).film; info = dev.last_render_info()\n"
        "assert (info['closest_kernel'], info['any_hit_kernel'], info['trace_wg_per_cu'], info['overlap_any']) == (2, 2, 5, 1), info\n"
        "assert np.array_equal(f, ref)\n"
        "opt('TRACE_WG_PER_CU', None)\n"
        # names the header no longer lists are refused, like any unknown name
        "for n in (b'WORKSPACE', b'TREELET', b'TREELET_TOP', b'TREELET_ROUNDS', b'PIPELINES', b'ANY8', b'LAYOUT_PILOT', b'NO_SUCH_SWITCH'): assert lib.dr_set_option(n, b'1') != 0\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    _run(code)


def test_one_process_renders_several_scenes_in_turn():
    """One long-lived host, several scenes: four device scenes of ONE aggregate with different light lists go through its
    3-entry cache (eviction destroys a device scene at once), two aggregates of different sizes share nothing but the
    process; every scene keeps its own workspace, pilot choice and state layout.  Rendered alternately -- small, large,
    another integrator, the evicted one again -- every film equals the film of that scene rendered alone in a fresh
    process."""
    body = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from dartray_amd import _abi, core, scenes\n"
        "_abi.init(0)\n"
        "def scene_a():\n"
        "    prims, mk = scenes.config('C2', xres=64, yres=48, spp=32, blob=(60, 30)); return prims, mk(), None\n"
        "def scene_b():\n"
        "    prims, mk = scenes.config('C5', xres=80, yres=64, spp=64, yard=(6, 12), env_res=(128, 64)); r = mk(); return prims, r, r.env\n"
        "def scene_c():\n"
        "    prims, mk = scenes.config('C1'); return prims, mk(), None\n"
        "def scene_d():\n"
        "    prims, mk = scenes.config('C4', xres=56, yres=40, spp=16, hair=(40, 24)); return prims, mk(), None\n"
        "makers = {'a': scene_a, 'b': scene_b, 'c': scene_c, 'd': scene_d}\n" % (ROOT, os.path.join(ROOT, "tests")))
    alone = body + (
        "prims, r, env = makers[sys.argv[1]]()\n"
        "np.save(sys.argv[2], r.render(scenes.make_scene(prims, env)).film)\n"
        "print('OK')\n")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    paths = {}
    for k in "abcd":
        paths[k] = os.path.join(out, "film_alone_%s.npy" % k)
        env = {kk: v for kk, v in os.environ.items() if not kk.startswith("DARTRAY_")}
        env["DARTRAY_PILOT"] = "force"
        res = subprocess.run([sys.executable, "-c", alone, k, paths[k]], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
    together = body + (
        "built = {k: makers[k]() for k in 'abcd'}\n"
        "sc = {k: scenes.make_scene(built[k][0], built[k][2]) for k in 'abcd'}\n"
        "films = {}\n"
        "order = 'abcdabdcbadcc'\n"
        "for k in order:\n"
        "    f = built[k][1].render(sc[k]).film\n"
        "    if k in films: assert np.array_equal(films[k], f), k\n"
        "    films[k] = f\n"
        # the aggregate-level cache: four more light lists on scene b's aggregate (its area lights + an equal environment light
        # of their own each: the cache is keyed on the light objects), three entries -- the least recently used device scene is
        # destroyed at once, and a list that was evicted is uploaded and piloted again
        "agg = sc['b'].aggregate\n"
        "lists = [agg.lights() + [scenes.sky_env(128, 64)] for _ in range(4)]\n"
        "for l in lists + [lists[0], sc['b'].lights]:\n"
        "    f = built['b'][1].render(core.Scene(agg, l)).film; assert np.array_equal(f, films['b'])\n"
        "    assert len(agg._scenes) <= 3\n"
        "for k in 'abcd': assert np.array_equal(films[k], np.load(sys.argv[1] % k)), k\n"
        "info = {k: sc[k]._device().last_render_info() for k in 'abcd'}\n"
        "assert info['b']['state_layout'] in (4, 64) and info['a']['state_layout'] == 64, info\n"
        "print('OK')\n")
    res = subprocess.run([sys.executable, "-c", together, os.path.join(out, "film_alone_%s.npy")],
                         env=dict({kk: v for kk, v in os.environ.items() if not kk.startswith("DARTRAY_")}, DARTRAY_PILOT="force"),
                         capture_output=True, text=True, timeout=900)
    for p in paths.values():
        os.remove(p)
    assert res.returncode == 0 and "OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])


def test_pilot_reports_what_it_measured_and_any_hit_follows_the_closest_hit_family():
    """dr_scene_get_pilot / dr_scene_last_render_info after a forced pilot on a small scene: the candidates' ms per algorithmic GB are
    there (kernel 5 only when kernel 3 did not lose clearly), the calibration batches are counted, and the any-hit kernel is the
    closest-hit kernel's family (2 beside 2, 3 beside 3 / 5) unless its own batch disagreed by more than 15 %.  A second render of the
    scene runs no pilot.  Films are the oracle's either way (tests/test_gpu_render.py::test_traversal_pilot_*)."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from dartray_amd import _abi, scenes\n"
        "_abi.init(0)\n"
        "prims, mk = scenes.config('C2', xres=48, yres=40, spp=16, blob=(60, 30))\n"
        "r = mk(); scene = scenes.make_scene(prims); dev = scene._device()\n"
        "assert dev.pilot()['closest'][2] == 0.0 and dev.trace_kernels() == (0, 0)\n"
        "a = r.render(scene).film; info = dev.last_render_info(); p = dev.pilot(); k = dev.trace_kernels()\n"
        "print(info, p, k)\n"
        "assert info['pilot_batches'] in (3, 4) and p['closest'][2] > 0 and p['closest'][3] > 0 and p['any_hit'][2] > 0 and p['any_hit'][3] > 0\n"
        "assert (p['closest'][5] > 0) == (info['pilot_batches'] == 4)\n"
        "assert info['pilot_batches'] == 4 or p['closest'][3] > 1.05 * p['closest'][2]\n"
        "own, other = (p['any_hit'][3], p['any_hit'][2]) if k[0] != 2 else (p['any_hit'][2], p['any_hit'][3])\n"
        "family = 3 if k[0] != 2 else 2\n"
        # the any-hit rays' visit order: far child first (ids 6 / 7 = 2 / 3) iff the pilot's first batch was cheaper per ray that way (ratio below 0.97)
        "far = k[1] in (6, 7); base = k[1] - 4 if far else k[1]\n"
        "assert base == (family if not other < 0.85 * own else 5 - family), (k, p)\n"
        "assert p['far_first'] > 0 and far == (p['far_first'] < 0.97), (k, p)\n"
        "assert (info['closest_kernel'], info['any_hit_kernel']) == k\n"
        "b = r.render(scene).film; assert dev.last_render_info()['pilot_batches'] == 0 and np.array_equal(a, b)\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    _run(code, DARTRAY_PILOT="force")


def test_lazy_sample_generation_leaves_the_film_alone():
    """The device sampler shuffles bounce b's LD blocks only for the 64-pixel groups that still have a path alive at bounce b (the camera
    rays' hits marked by k_trace_pk, then each stage's output list): a view that is mostly sky, a view with no sky at all, 64 / 256 / 512
    samples per pixel (the three shuffle kernels), depth 1 (only bounce 0 and 1 exist) and depth 5 -- films equal DARTRAY_LAZY_GEN=0's and
    the oracle's."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "import oracle.binding as ob\n"
        "from dartray_amd import _abi, core, scenes\n"
        "_abi.init(0)\n"
        "lib = _abi.lib()\n"
        "frac = []\n"
        "def both(prims, r, env=None, oracle=True):\n"
        "    scene = scenes.make_scene(prims, env); dev = scene._device(); films = []; dev.reset_stats()\n"
        "    for on in (None, b'0'):\n"
        "        _abi.check(lib.dr_set_option(b'LAZY_GEN', on)); films.append(r.render(scene).film)\n"
        "        assert dev.last_render_info()['lazy_gen'] == (1 if on is None else 0), dev.last_render_info()\n"
        "        ss = dev.sampler_stats(); assert 0 < ss['generated'] <= ss['named'] and (on is None or ss['generated'] == ss['named']), ss\n"
        "        frac.append(ss['generated'] / ss['named']); dev.reset_stats()\n"
        "    _abi.check(lib.dr_set_option(b'LAZY_GEN', None))\n"
        "    assert np.array_equal(films[0], films[1])\n"
        "    if oracle:\n"
        "        osc = ob.OracleScene(prims, env=env) if env is not None else ob.OracleScene(prims)\n"
        "        assert np.array_equal(films[0], osc.render(ob.render_desc(r, sampler_mode=1))['film'])\n"
        "    return dev.last_render_info()\n"
        "for spp, res, depth, oracle in ((64, (64, 48), 5, True), (256, (40, 32), 1, True), (512, (64, 40), 3, False)):\n"
        "    prims5, mk5 = scenes.config('C5', xres=res[0], yres=res[1], spp=spp, yard=(4, 12), env_res=(64, 32)); r5 = mk5()\n"
        "    r5.surfaceIntegrator.maxDepth = depth; both(prims5, r5, r5.env, oracle)\n"
        "prims, mk = scenes.config('C2', xres=40, yres=32, spp=64, blob=(40, 20)); both(prims, mk())\n"
        # a handful of random shapes of the same thing: odd image sizes, 64 .. 256 spp, depth 1 .. 8, more or less sky
        "rng = np.random.Generator(np.random.PCG64(505))\n"
        "for _ in range(6):\n"
        "    spp = int(rng.choice([64, 128, 256])); res = (int(rng.integers(17, 50)), int(rng.integers(9, 40)))\n"
        "    prims5, mk5 = scenes.config('C5', xres=res[0], yres=res[1], spp=spp, yard=(int(rng.integers(2, 6)), int(rng.integers(4, 14))), env_res=(64, 32))\n"
        "    r5 = mk5(); r5.surfaceIntegrator.maxDepth = int(rng.integers(1, 9)); both(prims5, r5, r5.env, True)\n"
        # several batches whose pixel counts are no multiples of 64 (the last group of a batch is partial)
        "prims5, mk5 = scenes.config('C5', xres=70, yres=50, spp=64, yard=(4, 12), env_res=(64, 32)); r5 = mk5()\n"
        "_abi.check(lib.dr_set_option(b'BATCH_BITS', b'16'))\n"
        "info = both(prims5, r5, r5.env, True); assert info['batches'] > 2, info\n"
        "_abi.check(lib.dr_set_option(b'BATCH_BITS', None))\n"
        # the courtyard under the sky: whole groups of pixels see only sky (no bounce-0 blocks) and fewer still reach bounce 2
        "assert frac[0] < 0.9 and frac[1] == 1.0 and all(f == 1.0 for f in frac[1::2]), frac\n"
        "print('OK', frac)\n" % (ROOT, os.path.join(ROOT, "tests")))
    _run(code, timeout=900)


def test_coherent_camera_kernel_equals_the_per_lane_kernels():
    """k_trace_pk walks a tile's 64 camera rays with ONE stack (node and triangles loaded once per wave, every lane testing its own
    ray; stack entries carry the mask of the lanes that pushed them).  Hits, films and the visit counters must equal what the per-lane
    kernels give (DARTRAY_COHERENT_CAMERA=0) and the oracle's -- also where the rays of a tile do NOT share their direction signs (a
    camera looking straight down an axis: every tile around the image centre mixes four sign patterns; spp 4: a tile spans 16
    pixels), where every ray has zero direction components (orthographic camera: all lanes take the literal f64 slab test), and on a
    deep tree."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "import oracle.binding as ob\n"
        "from dartray_amd import _abi, core, scenes\n"
        "_abi.init(0)\n"
        "lib = _abi.lib()\n"
        "def both(prims, r, env=None):\n"
        "    scene = scenes.make_scene(prims, env); dev = scene._device(); films = []; stats = []\n"
        "    for on in (b'1', b'0'):\n"
        "        _abi.check(lib.dr_set_option(b'COHERENT_CAMERA', on))\n"
        "        out = r.render(scene); films.append(out.film); stats.append(r.last_stats)\n"
        "        info = dev.last_render_info(); assert info['coherent_camera'] == int(on), info\n"
        "        cs = dev.coherent_stats(); st_ = r.last_stats\n"   # the coherent kernel's share of the closest-hit totals: the camera rays, one launch per batch
        "        assert (cs['rays'], cs['launches']) == ((st_['camera_samples'], st_['batches']) if on == b'1' else (0, 0)), (cs, st_)\n"
        "        assert 0 <= cs['nodes'] <= st_['closest_nodes'] and 0 <= cs['tris'] <= st_['closest_tris'] and cs['ms'] <= st_['closest_ms'] + 1e-6\n"
        "    _abi.check(lib.dr_set_option(b'COHERENT_CAMERA', None))\n"
        "    assert np.array_equal(films[0], films[1])\n"
        "    keys = ('closest_nodes', 'any_nodes', 'closest_tris', 'any_tris', 'closest_rays', 'any_rays')\n"
        "    assert all(stats[0][k] == stats[1][k] for k in keys), (stats[0], stats[1])\n"
        "    osc = ob.OracleScene(prims, env=env) if env is not None else ob.OracleScene(prims); osc.counters(reset=True)\n"
        "    ref = osc.render(ob.render_desc(r, sampler_mode=1)); c = osc.counters()\n"
        "    assert np.array_equal(films[0], ref['film']) and all(stats[0][k] == c[k] for k in keys), (stats[0], c)\n"
        "for spp, blob in ((64, (60, 30)), (4, (24, 12)), (128, (90, 45))):\n"
        "    prims, mk = scenes.config('C2', xres=24, yres=20, spp=spp, blob=blob); both(prims, mk())\n"
        # the camera on the z axis looking at the origin: direction signs change in the middle of the image, inside tiles
        "prims, mk = scenes.config('C2', xres=9, yres=9, spp=8, blob=(40, 20)); both(prims, mk())\n"
        "prims, mk = scenes.config('C4', xres=20, yres=16, spp=64, hair=(60, 30)); both(prims, mk())\n"
        "prims5, mk5 = scenes.config('C5', xres=24, yres=16, spp=64, yard=(4, 12), env_res=(64, 32)); r5 = mk5(); both(prims5, r5, r5.env)\n"
        # orthographic camera: every ray is (0, 0, 1) in camera space -> zero components -> the literal test for every box
        "prims = scenes.cornell_prims(scenes.blob_prim(40, 20)); film = core.ImageFilm(20, 16)\n"
        "cam = core.OrthographicCamera.lookAt((0, 0, -35), (0, 0, 0), (0, 1, 0), film, screenWindow=[-11, 11, -8, 8])\n"
        "both(prims, core.SamplerRenderer(core.LowDiscrepancySampler(cam, 64), cam, core.PathIntegrator(3), core.EmissionIntegrator()))\n"
        "print('OK')\n" % (ROOT, os.path.join(ROOT, "tests")))
    _run(code, timeout=900)
