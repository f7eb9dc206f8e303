"""Known-answer tests that pin the CPU oracle to values derivable by hand from the reference
source.  The reference's own tests are empty, so these are the pins."""
import ctypes as C
import math

import numpy as np
import pytest


def test_van_der_corput_is_bit_reversal(ob):
    # montecarlo.dart:495-504 with scramble 0: radical inverse base 2
    l = ob.lib()
    exp = [0.0, 0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875]
    assert [l.orc_van_der_corput(n, 0) for n in range(8)] == exp
    # the clamp to ONE_MINUS_EPSILON: all 24 kept bits set
    assert l.orc_van_der_corput(0, 0xFFFFFFFF) == 0.9999999403953552


def test_sobol2_gray_code_matrix(ob):
    # montecarlo.dart:486-493: v = 1<<31; v ^= v>>1 => columns 0x80000000, 0xC0000000, 0xA0000000, 0xF0000000
    l = ob.lib()
    exp = [0.0, 0.5, 0.75, 0.25, 0.625, 0.125, 0.375, 0.875]
    assert [l.orc_sobol2(n, 0) for n in range(8)] == exp


def test_concentric_sample_disk_regions(ob):
    # montecarlo.dart:155-201
    l = ob.lib()

    def csd(u1, u2):
        a, b = C.c_double(), C.c_double()
        l.orc_concentric_sample_disk(u1, u2, C.byref(a), C.byref(b))
        return a.value, b.value

    assert csd(0.5, 0.5) == (0.0, 0.0)  # degeneracy at the origin
    x, y = csd(1.0, 0.5)  # region 1, sy == 0 -> theta = 8 + 0 -> angle 2*pi
    assert x == pytest.approx(1.0, abs=1e-15) and y == pytest.approx(0.0, abs=1e-15)
    x, y = csd(0.5, 1.0)  # region 2: r = 1, theta = 2 -> pi/2
    assert x == pytest.approx(0.0, abs=1e-15) and y == pytest.approx(1.0)
    x, y = csd(0.0, 0.5)  # region 3: r = 1, theta = 4 -> pi
    assert x == pytest.approx(-1.0) and y == pytest.approx(0.0, abs=1e-15)
    x, y = csd(0.5, 0.0)  # region 4: r = 1, theta = 6 -> 3pi/2
    assert x == pytest.approx(0.0, abs=1e-15) and y == pytest.approx(-1.0)
    # uniformity of area: radius**2 is uniform -> point (0.75, 0.5): r = 0.5 on the +x axis
    x, y = csd(0.75, 0.5)
    assert (x, y) == (0.5 * math.cos(8 * math.pi / 4), 0.5 * math.sin(8 * math.pi / 4))


def test_cosine_hemisphere_is_f32_rounded_unit_vector(ob):
    out = (C.c_double * 3)()
    ob.lib().orc_cosine_sample_hemisphere(0.3, 0.8, out)
    v = np.array(list(out))
    assert np.all(v == v.astype(np.float32).astype(np.float64))  # Vector stores are f32
    assert abs(np.linalg.norm(v) - 1.0) < 1e-6 and v[2] > 0


def test_power_heuristic(ob):
    l = ob.lib()
    assert l.orc_power_heuristic(1, 1.0, 1, 1.0) == 0.5
    assert l.orc_power_heuristic(1, 3.0, 1, 4.0) == 9.0 / 25.0
    assert l.orc_power_heuristic(1, 2.0, 1, 0.0) == 1.0


@pytest.mark.parametrize("w,h,count,grid", [(64, 64, 8, (4, 2)), (64, 64, 4, (2, 2)), (128, 32, 8, (8, 1)),
                                           (32, 128, 8, (2, 4)), (65, 65, 2, (2, 1))])
def test_get_sub_window_grid(ob, w, h, count, grid):
    # common.dart:52-73: halve nx while 2*w*ny < h*nx
    ext = np.zeros(4, np.int32)
    nx, ny = grid
    cover = np.zeros((h, w), np.int32)
    for num in range(count):
        ob.lib().orc_get_sub_window(w, h, num, count, ext.ctypes.data)
        x0, x1, y0, y1 = ext
        xo, yo = num % nx, num // nx
        assert x0 == math.floor(xo / nx * w) and y0 == math.floor(yo / ny * h)
        cover[y0:y1, x0:x1] += 1
    assert np.all(cover == 1)  # the tasks tile the window exactly once


def test_distribution1d(ob):
    # montecarlo.dart:25-52,82-92
    l = ob.lib()

    def dist(f, u):
        f = np.asarray(f, np.float64)
        u = np.asarray(u, np.float64)
        cdf = np.zeros(len(f) + 1, np.float32)
        fi = C.c_double()
        idx = np.zeros(len(u), np.int32)
        l.orc_distribution1d(f.ctypes.data, len(f), cdf.ctypes.data, C.byref(fi), u.ctypes.data, len(u), idx.ctypes.data)
        return cdf, fi.value, idx

    cdf, fi, idx = dist([1, 1], [0.0, 0.49, 0.5, 0.99])
    assert list(cdf) == [0.0, 0.5, 1.0] and fi == 1.0 and list(idx) == [0, 0, 1, 1]
    cdf, fi, idx = dist([1, 3], [0.0, 0.24, 0.25, 0.9])
    assert list(cdf) == [0.0, 0.25, 1.0] and fi == 2.0 and list(idx) == [0, 0, 1, 1]
    cdf, fi, idx = dist([0, 0], [0.3, 0.7])  # funcInt == 0 -> uniform
    assert list(cdf) == [0.0, 0.5, 1.0] and list(idx) == [0, 1]


def _tri_hit(ob, tri, o, d, tmin=0.0, tmax=np.inf, reverse=0):
    tri = np.asarray(tri, np.float32).reshape(9)
    ray = ob.make_rays([o], [d], tmin, tmax)
    out = np.zeros(9, np.float64)
    hit = ob.lib().orc_triangle_intersect(tri.ctypes.data, ray.ctypes.data, reverse, out.ctypes.data)
    hp = ob.lib().orc_triangle_intersectP(tri.ctypes.data, ray.ctypes.data)
    return hit, hp, out


UNIT = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]


def test_triangle_interior_hit(ob):
    # triangle.dart:44-160: ray down -z onto the unit triangle at (0.25, 0.25)
    hit, hp, out = _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, -1))
    assert hit == 1 and hp == 1
    t, b1, b2 = out[:3]
    assert (t, b1, b2) == (2.0, 0.25, 0.25)
    assert tuple(out[3:6]) == (0.25, 0.25, 0.0)
    # dpdu = p2-p1 = (1,0,0), dpdv = p3-p2 = (-1,1,0); nn = normalize(dpdu x dpdv) = (0,0,1)
    assert tuple(out[6:9]) == (0.0, 0.0, 1.0)
    hit, _, out = _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, -1), reverse=1)
    assert tuple(out[6:9]) == (-0.0, -0.0, -1.0)


def test_triangle_edge_rules(ob):
    # b1 + b2 == 1 exactly is accepted (reject only when > 1), b == 0 accepted (reject only when < 0)
    assert _tri_hit(ob, UNIT, (0.5, 0.5, 1.0), (0, 0, -1))[0] == 1
    assert _tri_hit(ob, UNIT, (0.0, 0.5, 1.0), (0, 0, -1))[0] == 1
    assert _tri_hit(ob, UNIT, (0.5, 0.0, 1.0), (0, 0, -1))[0] == 1
    assert _tri_hit(ob, UNIT, (0.75, 0.5, 1.0), (0, 0, -1))[0] == 0
    assert _tri_hit(ob, UNIT, (-0.25, 0.5, 1.0), (0, 0, -1))[0] == 0


def test_triangle_t_range_rules(ob):
    # t == tmax is a hit, t > tmax is not (triangle.dart:96); t < tmin is not
    assert _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, -1), tmax=2.0)[:2] == (1, 1)
    assert _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, -1), tmax=1.999999)[:2] == (0, 0)
    assert _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, -1), tmin=2.0)[:2] == (1, 1)
    assert _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, -1), tmin=2.000001)[:2] == (0, 0)
    # parallel ray: divisor == 0
    assert _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (1, 0, 0))[:2] == (0, 0)
    # behind the origin
    assert _tri_hit(ob, UNIT, (0.25, 0.25, 2.0), (0, 0, 1))[:2] == (0, 0)


def _slab(ob, bmin, bmax, o, d, tmin=0.0, tmax=np.inf):
    ray = ob.make_rays([o], [d], tmin, tmax)
    a = np.asarray(bmin, np.float32)
    b = np.asarray(bmax, np.float32)
    return ob.lib().orc_slab(a.ctypes.data, b.ctypes.data, ray.ctypes.data)


def test_slab_axis_parallel_and_nan(ob):
    # bvh_accel.dart:439-472
    box = ((-1, -1, -1), (1, 1, 1))
    assert _slab(ob, *box, (0, 0, -5), (0, 0, 1)) == 1       # through the middle: x/y slabs give -inf..inf
    assert _slab(ob, *box, (2, 0, -5), (0, 0, 1)) == 0       # outside in x: (-1-2)*inf = -inf, (1-2)*inf = -inf
    assert _slab(ob, *box, (0, 0, -5), (0, 0, -1)) == 0      # pointing away: tmax < minDistance
    assert _slab(ob, *box, (0, 0, -5), (0, 0, 1), tmax=3.9) == 0   # tmin = 4 >= maxDistance
    assert _slab(ob, *box, (0, 0, -5), (0, 0, 1), tmax=4.1) == 1
    # origin exactly on a slab plane with a zero direction component: 0 * inf = NaN.  Comparisons with
    # NaN are false, so a NaN on the y or z axis neither rejects nor tightens the interval ...
    assert _slab(ob, *box, (0, 1, -5), (0, 0, 1)) == 1
    assert _slab(ob, *box, (0, -1, -5), (0, 0, 1)) == 1
    # ... but tmin/tmax are INITIALISED from the x axis (bvh_accel.dart:441-442), so an x-axis NaN
    # survives to the final `tmin < maxDistance && tmax > minDistance` and the box is missed.
    assert _slab(ob, *box, (1, 0, -5), (0, 0, 1)) == 0
    assert _slab(ob, *box, (-1, 0, -5), (0, 0, 1)) == 0
    # -0.0 direction component: invDir = -inf, dirIsNeg = 1
    assert _slab(ob, *box, (0.5, 0, -5), (-0.0, 0, 1)) == 1


def test_dart_random_matches_independent_python_port(ob):
    """The Dart VM generator (multiply-with-carry, A = 0xffffda61) restated independently in Python
    integers; NB parity never depends on it (sample values are explicit inputs of the GPU path)."""
    M64 = (1 << 64) - 1

    def mix64(n):
        n = ((~n) + (n << 21)) & M64
        n ^= n >> 24
        n = (n * 265) & M64
        n ^= n >> 14
        n = (n * 21) & M64
        n ^= n >> 28
        n = (n + (n << 31)) & M64
        return n

    class R:
        def __init__(self, seed):
            h = mix64(seed & M64) or 0x5A17
            self.lo, self.hi = h & 0xffffffff, h >> 32
            for _ in range(4):
                self.step()

        def step(self):
            s = (0xffffda61 * self.lo + self.hi) & M64
            self.lo, self.hi = s & 0xffffffff, s >> 32

        def next_int(self, mx):
            if mx & (mx - 1) == 0:
                self.step()
                return self.lo & (mx - 1)
            while True:
                self.step()
                r = self.lo % mx
                if not (self.lo - r + mx > (1 << 32)):
                    return r

        def next_double(self):
            return (self.next_int(1 << 26) * float(1 << 27) + self.next_int(1 << 27)) / float(1 << 53)

    for seed in (0, 1, 5489, 123456789):
        n = 64
        u = np.zeros(n, np.uint32)
        f = np.zeros(n, np.float64)
        ob.lib().orc_dart_random(seed, n, u.ctypes.data, f.ctypes.data)
        a, b = R(seed), R(seed)
        assert [a.next_int(0xffffffff) for _ in range(n)] == list(u)
        assert [b.next_double() for _ in range(n)] == list(f)
        assert np.all(f >= 0) and np.all(f < 1)


def _ld(ob, mode, seed, pixel, spp, n1D, n2D):
    a = np.asarray(n1D, np.int32)
    b = np.asarray(n2D, np.int32)
    nf = 5 + int(a.sum()) + 2 * int(b.sum())
    out = np.zeros((spp, nf), np.float32)
    ob.lib().orc_ld_pixel_sample(mode, seed, pixel, spp, a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data)
    return out


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("spp", [4, 64, 256])
def test_ld_pixel_sample_is_stratified(ob, mode, spp):
    """(0,2)-sequence property (montecarlo.dart:407-551): every 1-D slot has exactly one value in each
    of the spp strata; every 2-D slot is stratified in both projections and in every elementary
    interval of area 1/spp -- independent of scrambles and shuffles."""
    n1D, n2D = [1] * 14, [1] * 9  # the PathIntegrator + EmissionIntegrator layout
    s = _ld(ob, mode, 7, 3, spp, n1D, n2D)
    assert s.shape == (spp, 37) and np.all(s >= 0) and np.all(s < 1)
    one_d = [4] + list(range(5, 19))
    for c in one_d:
        assert sorted(np.floor(s[:, c].astype(np.float64) * spp).astype(int)) == list(range(spp))
    for c in [0, 2] + list(range(19, 37, 2)):
        x, y = s[:, c].astype(np.float64), s[:, c + 1].astype(np.float64)
        assert sorted(np.floor(x * spp).astype(int)) == list(range(spp))
        assert sorted(np.floor(y * spp).astype(int)) == list(range(spp))
        k = int(math.log2(spp))
        for i in range(k + 1):
            nx, ny = 1 << i, 1 << (k - i)
            cell = np.floor(x * nx).astype(int) * ny + np.floor(y * ny).astype(int)
            assert sorted(cell) == list(range(spp))


def test_ld_pixel_sample_python_port(ob):
    """LDPixelSample restated independently in Python for a tiny case (serial mode)."""
    M64 = (1 << 64) - 1
    seed, spp = 42, 4
    u = np.zeros(4096, np.uint32)
    ob.lib().orc_dart_random(seed, len(u), u.ctypes.data, None)
    draws = iter(int(v) for v in u)

    def vdc(n, scr):
        n = int("{:032b}".format(n)[::-1], 2) ^ scr
        return np.float32(((n >> 8) & 0xffffff) / float(1 << 24))

    def sobol2(n, scr):
        v = 1 << 31
        while n:
            if n & 1:
                scr ^= v
            n >>= 1
            v ^= v >> 1
        return np.float32(((scr >> 8) & 0xffffff) / float(1 << 24))

    def shuffle(a, count, dims):
        for i in range(count):
            other = i + next(draws) % (count - i)
            for j in range(dims):
                a[dims * i + j], a[dims * other + j] = a[dims * other + j], a[dims * i + j]

    def block1():
        scr = next(draws)
        a = [vdc(i, scr) for i in range(spp)]
        for _ in range(spp):
            next(draws)  # Shuffle of one entry still draws
        shuffle(a, spp, 1)
        return a

    def block2():
        s0, s1 = next(draws), next(draws)
        a = []
        for i in range(spp):
            a += [vdc(i, s0), sobol2(i, s1)]
        for _ in range(spp):
            next(draws)
        shuffle(a, spp, 2)
        return a

    img, lens, tm = block2(), block2(), block1()
    o1 = [block1() for _ in range(2)]
    o2 = [block2() for _ in range(1)]
    exp = np.zeros((spp, 5 + 2 + 2), np.float32)
    for i in range(spp):
        exp[i] = [img[2 * i], img[2 * i + 1], lens[2 * i], lens[2 * i + 1], tm[i], o1[0][i], o1[1][i], o2[0][2 * i], o2[0][2 * i + 1]]
    got = _ld(ob, 0, seed, 0, spp, [1, 1], [1])
    assert np.array_equal(got, exp)
