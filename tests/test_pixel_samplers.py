"""Pixel samplers (lib/pixel_samplers/*.dart): the ORDER in which the sampler window is walked.  It decides which
numbers of the task's serial RNG stream a pixel receives, i.e. the reference's exact image; the device's keyed streams
do not depend on it.  Host classes (independent Python port incl. its own Dart VM generator) vs the oracle's C++."""
import numpy as np
import pytest

from dartray_amd import core, pbrt, scenes


def _oracle_order(ob, kind, x, y, w, h, ts=32, rnd=True):
    out = np.zeros((w * h, 2), np.int32)
    ob.lib().orc_pixel_order(kind, x, y, w, h, ts, int(rnd), out.ctypes.data)
    return out


@pytest.mark.parametrize("win", [(0, 0, 65, 65), (0, 0, 100, 70), (3, 5, 31, 33), (0, 0, 1, 1), (0, 0, 32, 64)])
def test_orders_are_permutations_and_ports_agree(ob, win):
    x, y, w, h = win
    lin = core.LinearPixelSampler().setup(x, y, w, h)
    assert np.array_equal(lin, _oracle_order(ob, 0, x, y, w, h))
    assert tuple(lin[0]) == (x, y) and tuple(lin[-1]) == (x + w - 1, y + h - 1)
    for samp, kind in ((core.TilePixelSampler(), 1), (core.TilePixelSampler(16, False), 1), (core.RandomPixelSampler(), 2)):
        got = samp.setup(x, y, w, h)
        ref = _oracle_order(ob, kind, x, y, w, h, samp.tileSize, samp.randomize)
        assert np.array_equal(got, ref), type(samp).__name__
        assert sorted(map(tuple, got)) == sorted(map(tuple, lin))          # every pixel exactly once


def test_tile_order_known_structure():
    t = core.TilePixelSampler(32, False).setup(0, 0, 65, 40)
    assert tuple(t[0]) == (0, 0) and tuple(t[31]) == (31, 0) and tuple(t[32]) == (0, 1)   # a 32 x 32 tile row by row
    assert tuple(t[32 * 32]) == (32, 0)                                                     # then the next tile
    assert tuple(t[2 * 32 * 32]) == (64, 0) and tuple(t[2 * 32 * 32 + 1]) == (64, 1)        # the 1-pixel-wide last column
    shuffled = core.TilePixelSampler().setup(0, 0, 65, 40)
    assert tuple(shuffled[0]) != (0, 0) or not np.array_equal(shuffled, t)                  # RNG(5489) moves tiles
    # tiles stay contiguous: 3 x 2 tiles, each one run of the list
    tile_id = shuffled[:, 0] // 32 + 100 * (shuffled[:, 1] // 32)
    change = np.flatnonzero(np.diff(tile_id)) + 1
    assert len(change) == 3 * 2 - 1


def test_serial_image_depends_on_the_order_counter_image_does_not(ob):
    prims, mk = scenes.config("C2", xres=40, yres=40, spp=4, blob=(12, 6))
    films = {}
    for name, ps in (("linear", core.LinearPixelSampler()), ("tile", core.TilePixelSampler(16)), ("random", core.RandomPixelSampler())):
        r = mk()
        r.sampler.pixelSampler = ps
        osc = ob.OracleScene(prims)
        films[name] = (osc.render(ob.render_desc(r, sampler_mode=0))["film"], osc.render(ob.render_desc(r, sampler_mode=1))["film"])
    assert not np.array_equal(films["linear"][0], films["tile"][0]) and not np.array_equal(films["tile"][0], films["random"][0])
    assert np.array_equal(films["linear"][1], films["tile"][1]) and np.array_equal(films["linear"][1], films["random"][1])
    for a, b in (("linear", "tile"), ("linear", "random")):   # same expectation either way
        assert abs(films[a][0][..., :3].mean() - films[b][0][..., :3].mean()) < 0.05 * films[a][0][..., :3].mean()


def test_loader_default_is_the_reference_default():
    head = 'Film "image" "integer xresolution" [16] "integer yresolution" [16]\nWorldBegin\nShape "sphere"\nWorldEnd\n'
    r = pbrt.loads(head).rendererObject
    assert isinstance(r.sampler.pixelSampler, core.TilePixelSampler) and r.sampler.pixelSampler.tileSize == 32   # render_options.dart:29
    r = pbrt.loads('Pixels "linear"\n' + head).rendererObject
    assert type(r.sampler.pixelSampler) is core.LinearPixelSampler
    r = pbrt.loads('Pixels "tile" "integer tilesize" [8] "bool random" ["false"]\n' + head).rendererObject
    assert (r.sampler.pixelSampler.tileSize, r.sampler.pixelSampler.randomize) == (8, False)
