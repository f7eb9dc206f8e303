"""Pixel filters (lib/filters/*.dart): the host classes that fill DrFilm.filter_table against the oracle's C++
restatement of the same Dart formulas, plus hand-derived values.  CPU only; tests/test_gpu_filters.py renders."""
import ctypes as C
import math

import numpy as np
import pytest

from dartray_amd import core

FILTERS = [  # (oracle kind, p0, p1, host object)
    (0, 0.0, 0.0, core.BoxFilter(0.5, 0.5)),
    (0, 0.0, 0.0, core.BoxFilter(1.5, 0.75)),
    (1, 2.0, 0.0, core.GaussianFilter(2.0, 2.0, 2.0)),
    (1, 0.7, 0.0, core.GaussianFilter(3.0, 1.5, 0.7)),
    (2, 1.0 / 3.0, 1.0 / 3.0, core.MitchellFilter(1.0 / 3.0, 1.0 / 3.0, 2.0, 2.0)),
    (2, 0.0, 0.5, core.MitchellFilter(0.0, 0.5, 2.5, 1.5)),   # Catmull-Rom
    (3, 0.0, 0.0, core.TriangleFilter(2.0, 2.0)),
    (3, 0.0, 0.0, core.TriangleFilter(1.0, 3.0)),
    (4, 3.0, 0.0, core.LanczosSincFilter(4.0, 4.0, 3.0)),
    (4, 2.0, 0.0, core.LanczosSincFilter(3.0, 2.0, 2.0)),
]


@pytest.mark.parametrize("kind,p0,p1,f", FILTERS)
def test_host_filters_equal_the_oracle(ob, kind, p0, p1, f):
    l = ob.lib()
    rng = np.random.default_rng(kind * 7 + 1)
    pts = np.concatenate([rng.uniform(-1.2, 1.2, (200, 2)) * (f.xWidth, f.yWidth),
                          [[0.0, 0.0], [f.xWidth, f.yWidth], [1e-7, -1e-7], [-f.xWidth * 0.5, f.yWidth * 0.5]]])
    for x, y in pts:
        ref = l.orc_filter_evaluate(kind, f.xWidth, f.yWidth, p0, p1, float(x), float(y))
        got = f.evaluate(float(x), float(y))
        assert got == ref or abs(got - ref) <= 4e-16 * max(1.0, abs(ref)), (x, y, got, ref)  # exp / sin: libm vs libm
    # the 16 x 16 table of ImageFilm (image_film.dart:74-82) as it crosses the ABI
    t = np.zeros(256, np.float32)
    l.orc_filter_table(kind, f.xWidth, f.yWidth, p0, p1, t.ctypes.data)
    assert np.allclose(core.ImageFilm(8, 8, f).filterTable, t, rtol=2e-7, atol=0)


def test_known_values():
    assert core.TriangleFilter(2.0, 3.0).evaluate(0.0, 0.0) == 6.0
    assert core.TriangleFilter(2.0, 2.0).evaluate(2.5, 0.0) == 0.0
    m = core.MitchellFilter(1.0 / 3.0, 1.0 / 3.0, 2.0, 2.0)
    assert math.isclose(m.evaluate(0.0, 0.0), ((6 - 2 / 3) / 6) ** 2, rel_tol=1e-15)
    assert abs(m._mitchell1D(1.0)) < 1e-15                          # |2x| = 2: the cubic's outer zero
    assert m.evaluate(1.5, 0.0) < 0.0                               # the negative lobe
    g = core.GaussianFilter(2.0, 2.0, 2.0)
    assert g.evaluate(2.0, 0.0) == 0.0 and math.isclose(g.evaluate(0.0, 0.0), (1 - math.exp(-8.0)) ** 2, rel_tol=1e-15)
    s = core.LanczosSincFilter(4.0, 4.0, 3.0)
    assert s.evaluate(0.0, 0.0) == 1.0 and s.evaluate(4.1, 0.0) == 0.0
    x = 0.25 * math.pi
    assert math.isclose(s._sinc1D(0.25), math.sin(x) / x * math.sin(3 * x) / (3 * x), rel_tol=1e-15)


def test_plugin_names_and_sample_extent():
    # RegisterStandardPlugins (render_manager_interface.dart:55-59)
    for name, cls in (("box", core.BoxFilter), ("gaussian", core.GaussianFilter), ("sinc", core.LanczosSincFilter),
                      ("mitchell", core.MitchellFilter), ("triangle", core.TriangleFilter)):
        assert isinstance(core.Plugin.get("filter", name)(), cls)
    assert core.Plugin.get("filter", "sinc")().xWidth == 4.0 and core.Plugin.get("filter", "mitchell")().b == 1.0 / 3.0
    # ImageFilm.getSampleExtent grows with the filter (image_film.dart:247-252)
    assert core.ImageFilm(32, 16, core.GaussianFilter()).getSampleExtent() == (-2, 35, -2, 19)
