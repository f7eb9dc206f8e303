"""PBRT-v2 scene-file front end for the path.

Host-side mirror of the reference's scene API and parser, restricted to the
directives whose plugins are on the device path:

  * `DartRay`     -- the API state machine of lib/dartray/dartray.dart:121-640
                    (CTM, graphics state, attribute / transform stacks,
                    named coordinate systems, render options, worldEnd);
  * `PbrtLexer` / `PbrtParser` -- lib/dartray/pbrt_lexer.dart:117-240 and
                    lib/dartray/pbrt_parser.dart:157-560;
  * `ParamSet`    -- typed lookups of lib/core/param_set.dart.

Everything here is host logic: it produces the same flattened world-space
primitive list / light list / camera the hand-built scenes in scenes.py
produce, and hands them to core.BVHAccel / core.SamplerRenderer (the C ABI).
Matrices follow the reference's numerics (Float32List storage, f64
expressions): matrix4x4.dart:193-343, transform.dart:83-86,110-129,214-331.

Plugins on the path: shapes trianglemesh (with N / S / uv), heightfield, sphere, disk;
materials matte (Lambertian / Oren-Nayar), plastic, mirror, glass; area lights
on any of those shapes, infinite lights (constant or .npy lat-long map), point,
spot and distant lights; perspective / orthographic / environment cameras, image film, box / gaussian / mitchell / triangle / sinc filters,
low-discrepancy sampler,
bvh accelerator, path and directlighting (strategies "all" and "one") integrators.

A directive that needs a plugin outside that list (other quadrics, measured /
metal / uber materials, textures, projection / goniometric lights, volumes,
instancing, animated transforms, image-file radiance maps,
non-LD samplers, the Metropolis renderer) raises `UnsupportedFeature` naming it
with file:line -- never a silent approximation.
"""
import gzip
import math
import os

import numpy as np

from . import core

__all__ = ["DartRay", "PbrtParser", "PbrtLexer", "ParamSet", "Transform", "UnsupportedFeature", "load", "loads"]


class UnsupportedFeature(NotImplementedError):
    """The scene uses a reference plugin that is not on the MI355X path."""


# ---------------------------------------------------------------------------
# Matrix4x4 / Transform (lib/core/matrix4x4.dart, lib/core/transform.dart)
# ---------------------------------------------------------------------------
def _f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32)


def _mat(rows):
    return _f32(rows).reshape(4, 4)


_mul = core._mul  # Matrix4x4.Mul (matrix4x4.dart:193-206): left-to-right f64 sums, f32 store


_inverse = core._inv  # Matrix4x4.invert (matrix4x4.dart:242-354): the reference's cofactor formula, f32 stores


class Transform:
    """lib/core/transform.dart:25-86: a matrix and its inverse, both f32."""

    __slots__ = ("m", "mInv")

    def __init__(self, m=None, mInv=None):
        self.m = np.eye(4, dtype=np.float32) if m is None else _mat(m)
        self.mInv = _inverse(self.m) if mInv is None else _mat(mInv)

    def __mul__(self, t2):  # transform.dart:83-86
        return Transform(_mul(self.m, t2.m), _mul(t2.mInv, self.mInv))

    def copy(self):
        return Transform(self.m.copy(), self.mInv.copy())

    @staticmethod
    def Inverse(t):  # transform.dart:58-60
        return Transform(t.mInv.copy(), t.m.copy())

    @staticmethod
    def Translate(dx, dy, dz):  # transform.dart:214-227 (delta is a Vector: f32 components)
        d = _f32([dx, dy, dz])
        m = np.eye(4, dtype=np.float32)
        mi = np.eye(4, dtype=np.float32)
        m[:3, 3] = d
        mi[:3, 3] = -d
        return Transform(m, mi)

    @staticmethod
    def Scale(x, y, z):  # transform.dart:229-241
        return Transform(np.diag(_f32([x, y, z, 1.0])), np.diag(_f32([1.0 / x, 1.0 / y, 1.0 / z, 1.0])))

    @staticmethod
    def Rotate(angle, ax, ay, az):  # transform.dart:276-303
        a = _f32([ax, ay, az]).astype(np.float64)
        a = _f32(a / math.sqrt(float(a @ a))).astype(np.float64)  # Vector.Normalize -> f32 Vector
        x, y, z = (float(v) for v in a)
        s = math.sin((math.pi / 180.0) * angle)  # Radians (common.dart:87-88)
        c = math.cos((math.pi / 180.0) * angle)
        m = _mat([[x * x + (1.0 - x * x) * c, x * y * (1.0 - c) - z * s, x * z * (1.0 - c) + y * s, 0.0],
                  [x * y * (1.0 - c) + z * s, y * y + (1.0 - y * y) * c, y * z * (1.0 - c) - x * s, 0.0],
                  [x * z * (1.0 - c) - y * s, y * z * (1.0 - c) + x * s, z * z + (1.0 - z * z) * c, 0.0],
                  [0.0, 0.0, 0.0, 1.0]])
        return Transform(m, m.T.copy())

    @staticmethod
    def LookAt(pos, look, up):  # transform.dart:305-331
        c2w = core.look_at(_f32(pos), _f32(look), _f32(up))
        return Transform(_inverse(c2w), c2w)

    def transformPoints(self, P):
        """Transform.transformPoint (transform.dart:110-129) over an [n,3] f32 array."""
        return core.transform_points(self.m, np.asarray(P, dtype=np.float32).reshape(-1, 3))


# ---------------------------------------------------------------------------
# ParamSet (lib/core/param_set.dart): typed name -> values with usage tracking
# ---------------------------------------------------------------------------
_KIND = {"float": "float", "integer": "int", "bool": "bool", "string": "string", "point": "point", "vector": "vector",
         "normal": "normal", "rgb": "spectrum", "color": "spectrum", "texture": "texture"}


class ParamSet:
    def __init__(self):
        self._p = {}    # (kind, name) -> values
        self._used = set()

    def add(self, ptype, name, values, where=""):
        if ptype in ("spectrum", "xyz", "blackbody"):
            raise UnsupportedFeature(f"{where}parameter type '{ptype}' (sampled / XYZ / blackbody spectra are converted "
                                     "through CIE tables that are not on the path)")
        kind = _KIND.get(ptype)
        if kind is None:
            raise ValueError(f"{where}unhandled parameter type '{ptype}'")  # pbrt_parser.dart:587-589
        if kind in ("float", "point", "vector", "normal", "spectrum"):
            vals = [float(v) for v in values]
        elif kind == "int":
            vals = [int(v) for v in values]
        elif kind == "bool":
            vals = [str(v).lower() == "true" for v in values]  # pbrt_parser.dart:572-583
        else:
            vals = [str(v) for v in values]
        self._p[(kind, name.lower())] = vals  # names are case-insensitive (param_set.dart:416-417)

    def _get(self, kind, name):
        name = name.lower()
        v = self._p.get((kind, name))
        if v is not None:
            self._used.add((kind, name))
        return v

    def has(self, kind, name):
        return (kind, name.lower()) in self._p

    def findOne(self, kind, name, default):
        v = self._get(kind, name)
        return v[0] if v and len(v) == 1 else default  # param_set.dart findOne*: single-valued entries only

    def findOneFloat(self, name, d):
        return self.findOne("float", name, d)

    def findOneInt(self, name, d):
        return self.findOne("int", name, d)

    def findOneBool(self, name, d):
        return self.findOne("bool", name, d)

    def findOneString(self, name, d):
        return self.findOne("string", name, d)

    def findFloat(self, name):
        return self._get("float", name)

    def findInt(self, name):
        return self._get("int", name)

    def findPoint(self, name, kind="point"):
        v = self._get(kind, name)
        return None if v is None else np.asarray(v[:len(v) // 3 * 3], dtype=np.float32).reshape(-1, 3)

    def findNormal(self, name):
        return self.findPoint(name, "normal")

    def findVector(self, name):
        return self.findPoint(name, "vector")

    def findOneSpectrum(self, name, d):
        v = self._get("spectrum", name)
        if v is None or len(v) != 3:
            return np.asarray(d, dtype=np.float32).reshape(3)
        return np.asarray(v, dtype=np.float32)  # RGBSpectrum.rgb: the three values, stored f32

    def unused(self):
        return sorted(k for k in self._p if k not in self._used)

    def names(self):
        return sorted(self._p)


# ---------------------------------------------------------------------------
# lexer / parser (lib/dartray/pbrt_lexer.dart, lib/dartray/pbrt_parser.dart)
# ---------------------------------------------------------------------------
class PbrtLexer:
    """Tokens: ('id', s) identifiers [a-zA-Z][a-zA-Z0-9._]*, ('str', s) single- or double-quoted strings,
    ('num', text) numbers, ('[',) and (']',); '#' comments run to the end of the line (pbrt_lexer.dart:117-240)."""

    def __init__(self, text, path="<string>"):
        self.stack = [[text, 0, path, 1]]

    def addInclude(self, text, path):  # pbrt_lexer.dart:82-84
        self.stack.append([text, 0, path, 1])

    @property
    def where(self):
        top = self.stack[-1] if self.stack else ["", 0, "<eof>", 0]
        return f"{top[2]}:{top[3]}: "

    def next(self):
        while self.stack:
            top = self.stack[-1]
            text, i = top[0], top[1]
            n = len(text)
            while i < n:
                ch = text[i]
                if ch == "\n":
                    top[3] += 1
                    i += 1
                elif ch in " \t\r":
                    i += 1
                elif ch == "#":
                    while i < n and text[i] not in "\r\n":
                        i += 1
                else:
                    break
            if i >= n:
                self.stack.pop()
                continue
            ch = text[i]
            if ch in "\"'":
                j = text.find(ch, i + 1)
                if j < 0:
                    j = n
                top[1] = j + 1
                top[3] += text.count("\n", i, j)
                return ("str", text[i + 1:j])
            if ch.isalpha():
                j = i + 1
                while j < n and (text[j].isalnum() or text[j] in "._"):
                    j += 1
                top[1] = j
                return ("id", text[i:j])
            if ch in "[]":
                top[1] = i + 1
                return (ch,)
            if ch.isdigit() or ch in "-+.":
                j = i + 1
                while j < n and (text[j].isdigit() or text[j] in ".eE" or (text[j] in "+-" and text[j - 1] in "eE")):
                    j += 1
                top[1] = j
                return ("num", text[i:j])
            top[1] = i + 1
            raise ValueError(f"{self.where}unexpected character {ch!r}")
        return ("eof",)


_NUMERIC_DIRECTIVES = {"lookat": 9, "rotate": 4, "scale": 3, "translate": 3, "transform": 16, "concattransform": 16,
                       "transformtimes": 2}


class PbrtParser:
    """pbrt_parser.dart:157-420: reads directives and calls the DartRay API."""

    def __init__(self, dartray):
        self.dartray = dartray

    # resource lookup (resource_manager.dart): files relative to the scene's directory, '.gz' decoded
    @staticmethod
    def _read(path):
        with open(path, "rb") as f:
            data = f.read()
        if path.endswith(".gz") or data[:2] == b"\x1f\x8b":
            data = gzip.decompress(data)
        return data.decode("utf-8", errors="replace")

    def parse(self, path):
        self.base = os.path.dirname(os.path.abspath(path))
        return self.parseString(self._read(path), path)

    def parseString(self, text, path="<string>", base=None):
        if base is not None:
            self.base = base
        lex = PbrtLexer(text, path)
        tok = lex.next()
        api = self.dartray
        while tok[0] != "eof":
            if tok[0] != "id":
                raise ValueError(f"{lex.where}expected a directive, found {tok!r}")
            name = tok[1]
            key = name.lower()
            where = lex.where
            if key == "include":  # pbrt_parser.dart:352-365
                tok = lex.next()
                if tok[0] != "str":
                    raise ValueError(f"{where}Include expects a file name")
                inc = os.path.join(getattr(self, "base", "."), tok[1])
                if not os.path.exists(inc):
                    raise FileNotFoundError(f"{where}missing include: {tok[1]}")
                lex.addInclude(self._read(inc), inc)
                tok = lex.next()
                continue
            if key == "activetransform":
                tok = lex.next()
                try:
                    api.activeTransform(tok[1] if len(tok) > 1 else "")
                except UnsupportedFeature as e:
                    raise UnsupportedFeature(f"{where}{e}") from None
                tok = lex.next()
                continue
            ident = cls = typ = None
            values = None
            tok = lex.next()
            if key == "texture":  # Texture "name" "type" "class"
                ident, tok = tok[1], lex.next()
                typ, tok = tok[1], lex.next()
                cls, tok = tok[1], lex.next()
            elif key == "makenamedmaterial":
                ident, tok = tok[1], lex.next()
            elif tok[0] == "str":
                typ, tok = tok[1], lex.next()
            elif tok[0] == "[":
                values = []
                tok = lex.next()
                while tok[0] not in ("]", "eof"):
                    values.append(float(tok[1]))
                    tok = lex.next()
                tok = lex.next()
            elif tok[0] == "num":
                values = []
                while tok[0] == "num":
                    values.append(float(tok[1]))
                    tok = lex.next()
            params = ParamSet()
            while tok[0] == "str":  # "type name" value | [ values ]
                decl = tok[1].split()
                if len(decl) != 2:
                    raise ValueError(f'{lex.where}expected parameter "type name", found "{tok[1]}"')
                tok = lex.next()
                if tok[0] == "[":
                    vals = []
                    tok = lex.next()
                    while tok[0] not in ("]", "eof"):
                        vals.append(tok[1])
                        tok = lex.next()
                else:
                    vals = [tok[1]]
                params.add(decl[0], decl[1], vals, where)
                tok = lex.next()
            self._dispatch(key, name, typ, ident, cls, values, params, where)
        return api

    def _dispatch(self, key, name, typ, ident, cls, v, ps, where):
        api = self.dartray
        need = _NUMERIC_DIRECTIVES.get(key)
        if need is not None and (v is None or len(v) != need):
            raise ValueError(f"{where}{name} requires {need} values")  # pbrt_parser.dart:214-330 (warns and skips)
        try:
            if key == "accelerator":
                api.accelerator(typ, ps)
            elif key == "arealightsource":
                api.areaLightSource(typ, ps)
            elif key == "attributebegin":
                api.attributeBegin()
            elif key == "attributeend":
                api.attributeEnd()
            elif key == "camera":
                api.camera(typ, ps)
            elif key == "coordinatesystem":
                api.coordinateSystem(typ)
            elif key == "coordsystransform":
                api.coordSysTransform(typ)
            elif key == "concattransform":  # file order is column-major (pbrt_parser.dart:214-217)
                api.concatTransform(_mat(v).T)
            elif key == "film":
                api.film(typ, ps)
            elif key in ("ident", "identity"):
                api.identity()
            elif key == "lightsource":
                api.lightSource(typ, ps)
            elif key == "lookat":
                api.lookAt(*v)
            elif key == "makenamedmaterial":
                api.makeNamedMaterial(ident, ps)
            elif key == "material":
                api.material(typ, ps)
            elif key == "namedmaterial":
                api.namedMaterial(typ)
            elif key == "objectbegin":
                api.objectBegin(typ)
            elif key == "objectend":
                api.objectEnd()
            elif key == "objectinstance":
                api.objectInstance(typ)
            elif key == "pixelfilter":
                api.pixelFilter(typ, ps)
            elif key == "renderer":
                api.renderer(typ, ps)
            elif key == "reverseorientation":
                api.reverseOrientation()
            elif key == "rotate":
                api.rotate(*v)
            elif key == "pixels":
                api.pixels(typ, ps)
            elif key == "sampler":
                api.sampler(typ, ps)
            elif key == "scale":
                api.scale(*v)
            elif key == "shape":
                api.shape(typ, ps)
            elif key == "surfaceintegrator":
                api.surfaceIntegrator(typ, ps)
            elif key == "texture":
                api.texture(ident, typ, cls, ps)
            elif key == "translate":
                api.translate(*v)
            elif key == "transform":
                api.transform(_mat(v).T)
            elif key == "transformbegin":
                api.transformBegin()
            elif key == "transformend":
                api.transformEnd()
            elif key == "transformtimes":
                api.transformTimes(*v)
            elif key == "volume":
                api.volume(typ, ps)
            elif key == "volumeintegrator":
                api.volumeIntegrator(typ, ps)
            elif key == "worldbegin":
                api.worldBegin()
            elif key == "worldend":
                api.worldEnd()
            else:
                raise ValueError(f"unhandled command {name}")  # pbrt_parser.dart:367-369
        except UnsupportedFeature as e:
            raise UnsupportedFeature(f"{where}{e}") from None


# ---------------------------------------------------------------------------
# the API state machine (lib/dartray/dartray.dart)
# ---------------------------------------------------------------------------
class _GraphicsState:  # graphics_state.dart:23-49
    def __init__(self, o=None):
        self.material = "matte" if o is None else o.material
        self.materialParams = ParamSet() if o is None else o.materialParams
        self.namedMaterials = {} if o is None else dict(o.namedMaterials)
        self.currentNamedMaterial = "" if o is None else o.currentNamedMaterial
        self.areaLight = "" if o is None else o.areaLight
        self.areaLightParams = ParamSet() if o is None else o.areaLightParams
        self.reverseOrientation = False if o is None else o.reverseOrientation


class DartRay:
    """The scene-description API (lib/dartray/dartray.dart:121-640) for the plugins on the path.

    `worldEnd()` builds `self.scene` (core.Scene over a core.BVHAccel) and `self.renderer`
    (core.SamplerRenderer); with `render=True` (the reference's behaviour, dartray.dart:562-585) it also
    renders on the device and leaves the result in `self.outputImage`.  `overrides` mirrors
    RenderOverrides (core/render_overrides.dart): 'xresolution', 'yresolution', 'pixelsamples', 'seed',
    'taskNum', 'taskCount'."""

    def __init__(self, render=True, overrides=None, warn=None):
        self.renderOnWorldEnd = render
        self.overrides = dict(overrides or {})
        self.warn = warn or (lambda msg: None)
        self.outputImage = None
        self.scene = None
        self.rendererObject = None
        self._reset_options()
        self._reset_world()

    # -- state ---------------------------------------------------------------
    def _reset_options(self):  # render_options.dart:24-40
        self.opt = dict(filterName="box", filterParams=ParamSet(), filmName="image", filmParams=ParamSet(),
                        samplerName="lowdiscrepancy", samplerParams=ParamSet(), acceleratorName="bvh",
                        acceleratorParams=ParamSet(), rendererName="sampler", rendererParams=ParamSet(),
                        surfaceIntegratorName="directlighting", surfaceIntegratorParams=ParamSet(),
                        volumeIntegratorName="emission", volumeIntegratorParams=ParamSet(),
                        cameraName="perspective", cameraParams=ParamSet(), cameraToWorld=Transform(),
                        pixelSamplerName="tile", pixelSamplerParams=ParamSet())
        self.primitives = []
        self.lights = []

    def _reset_world(self):
        self.ctm = Transform()
        self.gs = _GraphicsState()
        self._pushedGS, self._pushedCTM = [], []
        self.named = {}
        self.inWorld = False

    def _apply(self, t):
        self.ctm = self.ctm * t

    # -- transforms (dartray.dart:121-205) ------------------------------------
    def identity(self):
        self.ctm = Transform()

    def translate(self, dx, dy, dz):
        self._apply(Transform.Translate(dx, dy, dz))

    def scale(self, sx, sy, sz):
        self._apply(Transform.Scale(sx, sy, sz))

    def rotate(self, angle, dx, dy, dz):
        self._apply(Transform.Rotate(angle, dx, dy, dz))

    def lookAt(self, ex, ey, ez, lx, ly, lz, ux, uy, uz):
        self._apply(Transform.LookAt((ex, ey, ez), (lx, ly, lz), (ux, uy, uz)))

    def transform(self, m):
        self.ctm = Transform(m)

    def concatTransform(self, m):
        self._apply(Transform(m))

    def coordinateSystem(self, name):
        self.named[name] = self.ctm.copy()

    def coordSysTransform(self, name):
        if name in self.named:
            self.ctm = self.named[name].copy()
        else:
            self.warn(f"Couldn't find named coordinate system '{name}'")

    def activeTransform(self, which):  # dartray.dart:198-208
        if which != "All":
            raise UnsupportedFeature(f"ActiveTransform {which}: animated transforms (TransformedPrimitive) are not on the path")

    def transformTimes(self, start, end):
        self.transformStartTime, self.transformEndTime = start, end  # only read by animated shapes

    # -- render options (dartray.dart:210-259) --------------------------------
    def pixelFilter(self, name, ps):
        self.opt["filterName"], self.opt["filterParams"] = name, ps

    def film(self, name, ps):
        self.opt["filmName"], self.opt["filmParams"] = name, ps

    def pixels(self, name, ps):
        # pixel ORDER does not change the image in counter-stream mode; it is kept on the
        # sampler for the serial reference stream (oracle / host-buffer replays)
        self.opt["pixelSamplerName"] = name
        self.opt["pixelSamplerParams"] = ps

    def sampler(self, name, ps):
        self.opt["samplerName"], self.opt["samplerParams"] = name, ps

    def accelerator(self, name, ps):
        self.opt["acceleratorName"], self.opt["acceleratorParams"] = name, ps

    def surfaceIntegrator(self, name, ps):
        self.opt["surfaceIntegratorName"], self.opt["surfaceIntegratorParams"] = name, ps

    def volumeIntegrator(self, name, ps):
        self.opt["volumeIntegratorName"], self.opt["volumeIntegratorParams"] = name, ps

    def renderer(self, name, ps):
        self.opt["rendererName"], self.opt["rendererParams"] = name, ps

    def camera(self, name, ps):
        self.opt["cameraName"], self.opt["cameraParams"] = name, ps
        self.opt["cameraToWorld"] = Transform.Inverse(self.ctm)
        self.named["camera"] = self.opt["cameraToWorld"].copy()

    # -- world block (dartray.dart:262-300) -----------------------------------
    def worldBegin(self):
        self.ctm = Transform()
        self.named["world"] = self.ctm.copy()
        self.inWorld = True

    def attributeBegin(self):
        self._pushedGS.append(_GraphicsState(self.gs))
        self._pushedCTM.append((self.ctm.copy(), True))

    def attributeEnd(self):
        if not self._pushedGS:
            self.warn("Unmatched attributeEnd() encountered. Ignoring it.")
            return
        self.gs = self._pushedGS.pop()
        self.ctm = self._pushedCTM.pop()[0]

    def transformBegin(self):
        self._pushedCTM.append((self.ctm.copy(), False))

    def transformEnd(self):
        if not self._pushedCTM:
            self.warn("Unmatched pbrtTransformEnd() encountered. Ignoring it.")
            return
        self.ctm = self._pushedCTM.pop()[0]

    def reverseOrientation(self):
        self.gs.reverseOrientation = not self.gs.reverseOrientation

    # -- materials / lights / shapes (dartray.dart:302-470, 780-1100) ---------
    def texture(self, name, typ, texname, ps):
        raise UnsupportedFeature(f"Texture \"{name}\" \"{typ}\" \"{texname}\": only constant textures are on the path")

    def material(self, name, ps):
        self.gs.material, self.gs.materialParams, self.gs.currentNamedMaterial = name, ps, ""

    def makeNamedMaterial(self, name, ps):
        mat_type = ps.findOneString("type", "")
        if not mat_type:
            self.warn("No parameter string 'type' found in MakeNamedMaterial")
            return
        self.gs.namedMaterials[name] = self._makeMaterial(mat_type, ParamSet(), ps)

    def namedMaterial(self, name):
        self.gs.currentNamedMaterial = name

    def _makeMaterial(self, name, geomParams, matParams):
        """MatteMaterial.Create through TextureParams (matte_material.dart:67-72; texture_params.dart: the shape's
        own parameters are searched before the Material directive's)."""
        if name not in ("matte", "mirror", "glass", "plastic"):
            raise UnsupportedFeature(f"Material \"{name}\": only 'matte', 'plastic', 'mirror' and 'glass' are on the path "
                                     "(SURVEY.md section 8 row f4)")
        for ps in (geomParams, matParams):
            for tex in ("Kd", "sigma", "bumpmap", "Kr", "Kt", "index", "Ks", "roughness"):
                if ps.has("texture", tex):
                    raise UnsupportedFeature(f"{name} '{tex}' bound to a texture: only constant textures are on the path")

        def spectrum(pname, default):  # TextureParams.getSpectrumTexture: the shape's parameters first
            return geomParams.findOneSpectrum(pname, None) if geomParams.has("spectrum", pname) else \
                matParams.findOneSpectrum(pname, default)

        if name == "plastic":                 # plastic_material.dart:72-78
            rough = geomParams.findOneFloat("roughness", None) if geomParams.has("float", "roughness") else \
                matParams.findOneFloat("roughness", 0.1)
            return core.PlasticMaterial(spectrum("Kd", (0.25, 0.25, 0.25)), spectrum("Ks", (0.25, 0.25, 0.25)), rough)
        if name == "mirror":                  # mirror_material.dart:57-61
            return core.MirrorMaterial(spectrum("Kr", (0.9, 0.9, 0.9)))
        if name == "glass":                   # glass_material.dart:71-78
            index = geomParams.findOneFloat("index", None) if geomParams.has("float", "index") else \
                matParams.findOneFloat("index", 1.5)
            return core.GlassMaterial(spectrum("Kr", (1.0, 1.0, 1.0)), spectrum("Kt", (1.0, 1.0, 1.0)), index)
        kd = geomParams.findOneSpectrum("Kd", None) if geomParams.has("spectrum", "Kd") else \
            matParams.findOneSpectrum("Kd", (0.5, 0.5, 0.5))
        sigma = geomParams.findOneFloat("sigma", None) if geomParams.has("float", "sigma") else \
            matParams.findOneFloat("sigma", 0.0)
        return core.MatteMaterial(kd, sigma)

    def _createMaterial(self, shapeParams):  # dartray.dart:780-804
        gs = self.gs
        if gs.currentNamedMaterial and gs.currentNamedMaterial in gs.namedMaterials:
            return gs.namedMaterials[gs.currentNamedMaterial]
        return self._makeMaterial(gs.material, shapeParams, gs.materialParams)

    def lightSource(self, name, ps):  # dartray.dart:368-376
        if name == "point":                   # point_light.dart:99-105
            I = ps.findOneSpectrum("I", (1.0, 1.0, 1.0))
            sc = ps.findOneSpectrum("scale", (1.0, 1.0, 1.0))
            frm = ps.findPoint("from")
            frm = (0.0, 0.0, 0.0) if frm is None or len(frm) != 1 else tuple(float(v) for v in frm[0])
            l2w = Transform.Translate(*frm) * self.ctm
            self.lights.append(core.PointLight(l2w.m, (I.astype(np.float64) * sc.astype(np.float64)).astype(np.float32)))
            return
        if name == "distant":                 # distant_light.dart:91-98
            L = ps.findOneSpectrum("L", (1.0, 1.0, 1.0))
            sc = ps.findOneSpectrum("scale", (1.0, 1.0, 1.0))
            frm, to = ps.findPoint("from"), ps.findPoint("to")
            frm = np.zeros(3, np.float32) if frm is None or len(frm) != 1 else frm[0]
            to = np.array([0, 0, 1], np.float32) if to is None or len(to) != 1 else to[0]
            d = (frm.astype(np.float64) - to.astype(np.float64)).astype(np.float32)
            self.lights.append(core.DistantLight(self.ctm.m, (L.astype(np.float64) * sc.astype(np.float64)).astype(np.float32), d))
            return
        if name == "spot":                    # spot_light.dart:100-125
            I = ps.findOneSpectrum("I", (1.0, 1.0, 1.0))
            sc = ps.findOneSpectrum("scale", (1.0, 1.0, 1.0))
            coneangle = ps.findOneFloat("coneangle", 30.0)
            conedelta = ps.findOneFloat("conedeltaangle", 5.0)
            frm, to = ps.findPoint("from"), ps.findPoint("to")
            frm = np.zeros(3, np.float32) if frm is None or len(frm) != 1 else frm[0]
            to = np.array([0, 0, 1], np.float32) if to is None or len(to) != 1 else to[0]
            d = core._normalize((to.astype(np.float64) - frm.astype(np.float64)).astype(np.float32)).astype(np.float64)
            if abs(d[0]) > abs(d[1]):         # Vector.CoordinateSystem (vector.dart:198-214)
                inv = 1.0 / math.sqrt(d[0] * d[0] + d[2] * d[2])
                du = _f32([-d[2] * inv, 0.0, d[0] * inv]).astype(np.float64)
            else:
                inv = 1.0 / math.sqrt(d[1] * d[1] + d[2] * d[2])
                du = _f32([0.0, d[2] * inv, -d[1] * inv]).astype(np.float64)
            dv = _f32(np.cross(d, du)).astype(np.float64)
            dirToZ = Transform(_mat([[du[0], du[1], du[2], 0], [dv[0], dv[1], dv[2], 0], [d[0], d[1], d[2], 0], [0, 0, 0, 1]]))
            l2w = self.ctm * Transform.Translate(float(frm[0]), float(frm[1]), float(frm[2])) * Transform.Inverse(dirToZ)
            self.lights.append(core.SpotLight(l2w.m, (I.astype(np.float64) * sc.astype(np.float64)).astype(np.float32), coneangle,
                                              coneangle - conedelta, l2w.mInv))
            return
        if name != "infinite":
            raise UnsupportedFeature(f"LightSource \"{name}\": only 'point', 'spot', 'distant' and 'infinite' are on the path "
                                     "(besides area lights)")
        L = ps.findOneSpectrum("L", (1.0, 1.0, 1.0))          # infinite_area_light.dart:309-316
        sc = ps.findOneSpectrum("scale", (1.0, 1.0, 1.0))
        nsamples = ps.findOneInt("nsamples", 1)
        mapname = ps.findOneString("mapname", "")
        texels = None
        Lmul = (L.astype(np.float64) * sc.astype(np.float64)).astype(np.float32)
        if mapname:
            texels = self._radianceMap(mapname, Lmul)
        light = core.InfiniteAreaLight(self.ctm.m, Lmul, nsamples, texels)
        light.worldToLight = self.ctm.mInv.copy()  # Light keeps Transform.Inverse(l2w) (light.dart:30)
        self.lights.append(light)

    def _radianceMap(self, mapname, L):
        """Level-0 texels of the radiance MIPMap.  Image decoding is outside the path: `.npy` [H,W,3] f32
        lat-long maps are read directly; the texels are pre-multiplied by L as the reference does for
        file maps (infinite_area_light.dart:44-49), in addition to the factor _radiance() applies."""
        path = os.path.join(getattr(self, "base", "."), mapname)
        if not mapname.endswith(".npy"):
            raise UnsupportedFeature(f"LightSource \"infinite\" mapname \"{mapname}\": image decoders (EXR/TGA/PNG) are "
                                     "not on the path; supply the lat-long map as a .npy [H,W,3] float32 array")
        tex = np.load(path).astype(np.float32)
        return (tex.astype(np.float64) * L.astype(np.float64)).astype(np.float32)

    def areaLightSource(self, name, ps):
        self.gs.areaLight, self.gs.areaLightParams = name, ps

    def _makeShape(self, name, ps):
        """_makeShape (dartray.dart:1001-1060) for the shapes on the path; None drops the directive like the
        reference does for a malformed mesh."""
        o2w, ro = self.ctm, self.gs.reverseOrientation
        if name == "sphere":                  # sphere.dart:313-321
            radius = ps.findOneFloat("radius", 1.0)
            return core.Sphere(o2w.m, o2w.mInv, ro, radius, ps.findOneFloat("zmin", -radius),
                               ps.findOneFloat("zmax", radius), ps.findOneFloat("phimax", 360.0))
        if name == "disk":                    # disk.dart:157-165
            return core.Disk(o2w.m, o2w.mInv, ro, ps.findOneFloat("height", 0.0), ps.findOneFloat("radius", 1.0),
                             ps.findOneFloat("innerradius", 0.0), ps.findOneFloat("phimax", 360.0))
        if name == "heightfield":             # heightfield.dart:23-94: refines into ONE TriangleMesh with uvs
            nu, nv = ps.findOneInt("nu", -1), ps.findOneInt("nv", -1)
            Pz = ps.findFloat("Pz")
            if nu < 2 or nv < 2 or Pz is None or len(Pz) != nu * nv:
                raise ValueError("heightfield needs nu, nv >= 2 and nu * nv values of Pz")
            xs = (np.arange(nu, dtype=np.float64) / (nu - 1)).astype(np.float32)   # uvs[ui] = x / (nx - 1), a Float32List
            ys = (np.arange(nv, dtype=np.float64) / (nv - 1)).astype(np.float32)
            gy, gx = np.meshgrid(ys, xs, indexing="ij")
            uv = np.stack([gx, gy], axis=-1).reshape(-1, 2)
            P = np.concatenate([uv, np.asarray(Pz, np.float32).reshape(-1, 1)], axis=1)   # Point(uvs[ui], uvs[ui+1], z[pi])
            x, y = np.meshgrid(np.arange(nu - 1), np.arange(nv - 1), indexing="xy")
            x, y = x.reshape(-1), y.reshape(-1)
            v = lambda a, b: a + b * nu
            idx = np.stack([v(x, y), v(x + 1, y), v(x + 1, y + 1), v(x, y), v(x + 1, y + 1), v(x, y + 1)], axis=1).reshape(-1, 3)
            return core.TriangleMesh(idx.astype(np.uint32), self.ctm.transformPoints(P), ro, uvs=uv.reshape(-1),
                                     objectToWorld=self.ctm.m, worldToObject=self.ctm.mInv)
        if name != "trianglemesh":
            raise UnsupportedFeature(f"Shape \"{name}\": only 'trianglemesh', 'heightfield', 'sphere' and 'disk' are on the path "
                                     "(SURVEY.md section 8 row f4)")
        vi = ps.findInt("indices")            # triangle_mesh.dart:91-193
        P = ps.findPoint("P")
        if vi is None or P is None:
            return None
        if ps.has("texture", "alpha") or ps.findOneFloat("alpha", 1.0) == 0.0:
            raise UnsupportedFeature("trianglemesh 'alpha' (alpha textures) is not on the path")
        uvs = ps.findFloat("uv")
        if uvs is None:
            uvs = ps.findFloat("st")
        discard = ps.findOneBool("discarddegenerateUVs", False)
        if uvs is not None:                   # triangle_mesh.dart:114-126
            if len(uvs) < 2 * len(P):
                self.warn(f"Not enough of 'uv's for triangle mesh. Expected {2 * len(P)}, found {len(uvs)}.  Discarding.")
                uvs = None
            elif len(uvs) > 2 * len(P):
                self.warn("More 'uv's provided than will be used for triangle mesh.")
        S = ps.findVector("S")
        if S is not None and len(S) != len(P):
            self.warn("Number of 'S's for triangle mesh must match 'P's")
            S = None
        N = ps.findNormal("N")
        if N is not None and len(N) != len(P):
            self.warn("Number of 'N's for triangle mesh must match 'P's")
            N = None
        ntris = len(vi) // 3
        idx = np.asarray(vi[:3 * ntris], dtype=np.int64).reshape(-1, 3)
        if discard and uvs is not None and N is not None:   # triangle_mesh.dart:140-165
            uv = np.asarray(uvs[:2 * len(P)], np.float32).reshape(-1, 2)
            P64 = P.astype(np.float64)
            for vp in range(0, len(N) - len(N) % 3, 3):     # the reference walks nvi = N.length index slots
                if vp + 2 >= idx.size:
                    break
                a, b, c = (int(idx.reshape(-1)[vp + k]) for k in range(3))
                if max(a, b, c) >= len(P):
                    break
                e1 = (P64[a] - P64[b]).astype(np.float32).astype(np.float64)
                e2 = (P64[c] - P64[b]).astype(np.float32).astype(np.float64)
                if 0.5 * float(np.linalg.norm(np.cross(e1, e2).astype(np.float32).astype(np.float64))) < 1.0e-7:
                    continue
                if (uv[a] == uv[b]).all() or (uv[b] == uv[c]).all() or (uv[c] == uv[a]).all():
                    self.warn("Degenerate uv coordinates in triangle mesh.  Discarding all uvs.")
                    uvs = None
                    break
        if idx.size and (idx.min() < 0 or idx.max() >= len(P)):
            self.warn("trianglemesh has out of-bounds vertex index")  # triangle_mesh.dart:160-166: shape dropped
            return None
        return core.TriangleMesh(idx.astype(np.uint32), self.ctm.transformPoints(P), self.gs.reverseOrientation,
                                 n=N, s=S, uvs=None if uvs is None else np.asarray(uvs[:2 * len(P)], np.float32),
                                 objectToWorld=self.ctm.m, worldToObject=self.ctm.mInv)

    def shape(self, name, ps):  # dartray.dart:380-470
        mesh = self._makeShape(name, ps)
        if mesh is None:
            return
        mtl = self._createMaterial(ps)
        for k in ps.unused():
            self.warn(f"Parameter '{k[1]}' not used")  # ParamSet.reportUnused
        area = None
        if self.gs.areaLight:
            if self.gs.areaLight not in ("area", "diffuse"):
                raise UnsupportedFeature(f"AreaLightSource \"{self.gs.areaLight}\"")
            lp = self.gs.areaLightParams      # diffuse_area_light.dart:91-97
            L = lp.findOneSpectrum("L", (1.0, 1.0, 1.0))
            sc = lp.findOneSpectrum("scale", (1.0, 1.0, 1.0))
            area = core.DiffuseAreaLight((L.astype(np.float64) * sc.astype(np.float64)).astype(np.float32),
                                         lp.findOneInt("nsamples", 1), mesh)
        self.primitives.append(core.GeometricPrimitive(mesh, mtl, area))
        if area is not None:
            self.lights.append(area)

    def volume(self, name, ps):
        raise UnsupportedFeature(f"Volume \"{name}\": participating media are not on the path")

    def objectBegin(self, name):
        raise UnsupportedFeature("ObjectBegin: instancing (TransformedPrimitive over nested aggregates) is not on the path")

    def objectEnd(self):
        raise UnsupportedFeature("ObjectEnd: instancing is not on the path")

    def objectInstance(self, name):
        raise UnsupportedFeature("ObjectInstance: instancing is not on the path")

    # -- worldEnd (dartray.dart:549-780) --------------------------------------
    def _makeFilm(self):
        o = self.opt
        fp = o["filterParams"]
        name = o["filterName"]
        if name == "box":
            filt = core.BoxFilter(fp.findOneFloat("xwidth", 0.5), fp.findOneFloat("ywidth", 0.5))  # box_filter.dart:41-45
        elif name == "gaussian":   # gaussian_filter.dart:39-46
            filt = core.GaussianFilter(fp.findOneFloat("xwidth", 2.0), fp.findOneFloat("ywidth", 2.0), fp.findOneFloat("alpha", 2.0))
        elif name == "mitchell":   # mitchell_filter.dart:44-50
            xw, yw = fp.findOneFloat("xwidth", 2.0), fp.findOneFloat("ywidth", 2.0)
            filt = core.MitchellFilter(fp.findOneFloat("B", 1.0 / 3.0), fp.findOneFloat("C", 1.0 / 3.0), xw, yw)
        elif name == "triangle":   # triangle_filter.dart:32-37
            filt = core.TriangleFilter(fp.findOneFloat("xwidth", 2.0), fp.findOneFloat("ywidth", 2.0))
        elif name == "sinc":       # lanczos_sinc_filter.dart:47-53
            filt = core.LanczosSincFilter(fp.findOneFloat("xwidth", 4.0), fp.findOneFloat("ywidth", 4.0), fp.findOneFloat("tau", 3.0))
        else:
            raise UnsupportedFeature(f"PixelFilter \"{name}\"")
        if o["filmName"] != "image":
            raise UnsupportedFeature(f"Film \"{o['filmName']}\"")
        ps = o["filmParams"]                                   # image_film.dart:309-324
        xres = int(self.overrides.get("xresolution", ps.findOneInt("xresolution", 640)))
        yres = int(self.overrides.get("yresolution", ps.findOneInt("yresolution", 480)))
        ps.findOneString("filename", "")
        crop = ps.findFloat("cropwindow")
        if crop is None or len(crop) != 4:
            crop = [0.0, 1.0, 0.0, 1.0]
        return core.ImageFilm(xres, yres, filt, tuple(crop))

    def _makeCamera(self, film):
        o = self.opt
        if o["cameraName"] not in ("perspective", "orthographic", "environment"):
            raise UnsupportedFeature(f"Camera \"{o['cameraName']}\"")
        ps = o["cameraParams"]                                 # perspective_camera.dart:134-183
        sopen, sclose = ps.findOneFloat("shutteropen", 0.0), ps.findOneFloat("shutterclose", 1.0)
        if sclose < sopen:
            sopen, sclose = sclose, sopen
        lensr = ps.findOneFloat("lensradius", 0.0)
        focald = ps.findOneFloat("focaldistance", 1.0e30)
        frame = ps.findOneFloat("frameaspectratio", film.xResolution / film.yResolution)
        sw = ps.findFloat("screenwindow")
        if sw is not None and len(sw) == 4:
            screen = list(sw)
        elif frame > 1.0:
            screen = [-frame, frame, -1.0, 1.0]
        else:
            screen = [-1.0, 1.0, -1.0 / frame, 1.0 / frame]
        if o["cameraName"] == "orthographic":                  # orthographic_camera.dart:120-160
            return core.OrthographicCamera(o["cameraToWorld"].m, screen, sopen, sclose, lensr, focald, film)
        if o["cameraName"] == "environment":                   # environment_camera.dart:54-90
            return core.EnvironmentCamera(o["cameraToWorld"].m, sopen, sclose, film)
        fov = ps.findOneFloat("fov", 60.0)
        halffov = ps.findOneFloat("halffov", -1.0)
        if halffov > 0.0:
            fov = 2.0 * halffov
        return core.PerspectiveCamera(o["cameraToWorld"].m, screen, sopen, sclose, lensr, focald, fov, film)

    def _makeSurfaceIntegrator(self):
        o = self.opt
        name, ps = o["surfaceIntegratorName"], o["surfaceIntegratorParams"]
        if name == "path":
            return core.PathIntegrator(ps.findOneInt("maxdepth", 5))      # path_integrator.dart:133-136
        if name == "directlighting":                                       # direct_lighting_integrator.dart:98-111
            maxdepth = ps.findOneInt("maxdepth", 5)
            st = ps.findOneString("strategy", "all")
            # (an unknown strategy is a LogWarning and 'all' in the reference, :102-108)
            return core.DirectLightingIntegrator(1 if st == "one" else 0, maxdepth)
        raise UnsupportedFeature(f"SurfaceIntegrator \"{name}\": only 'path' and 'directlighting' are on the path")

    def makeRenderer(self, **renderer_kw):
        """_makeRenderer (dartray.dart:640-700): film, camera, sampler, integrators, SamplerRenderer."""
        o = self.opt
        if o["rendererName"] != "sampler":
            raise UnsupportedFeature(f"Renderer \"{o['rendererName']}\": only 'sampler' is on the path")
        if o["samplerName"] != "lowdiscrepancy":
            raise UnsupportedFeature(f"Sampler \"{o['samplerName']}\": only 'lowdiscrepancy' is on the path")
        film = self._makeFilm()
        camera = self._makeCamera(film)
        nsamp = int(self.overrides.get("pixelsamples", o["samplerParams"].findOneInt("pixelsamples", 4)))
        pname, pps = o["pixelSamplerName"], o["pixelSamplerParams"]          # dartray.dart:980-996
        if pname == "tile":                                                  # tile_pixel_sampler.dart:102-106
            pixels = core.TilePixelSampler(pps.findOneInt("tilesize", 32), pps.findOneBool("random", True))
        elif pname == "random":
            pixels = core.RandomPixelSampler()
        elif pname == "linear":
            pixels = core.LinearPixelSampler()
        else:
            raise UnsupportedFeature(f"Pixels \"{pname}\"")
        sampler = core.LowDiscrepancySampler(camera, nsamp, int(self.overrides.get("seed", 5489)), pixels)
        kw = dict(taskNum=self.overrides.get("taskNum", 0), taskCount=self.overrides.get("taskCount", 1))
        kw.update(renderer_kw)
        return core.SamplerRenderer(sampler, camera, self._makeSurfaceIntegrator(), core.EmissionIntegrator(), **kw)

    def makeScene(self):
        """_makeScene (dartray.dart:603-637): the 'bvh' aggregate over every primitive + the lights in file order."""
        o = self.opt
        if o["acceleratorName"] != "bvh":
            raise UnsupportedFeature(f"Accelerator \"{o['acceleratorName']}\": only 'bvh' is on the path")
        ps = o["acceleratorParams"]                            # bvh_accel.dart:474-482
        accel = core.BVHAccel(self.primitives, ps.findOneInt("maxnodeprims", 4), ps.findOneString("splitmethod", "sah"))
        return core.Scene(accel, self.lights)

    def worldEnd(self):
        self._pushedGS.clear()
        self._pushedCTM.clear()
        self.rendererObject = self.makeRenderer()
        self.scene = self.makeScene()
        self.sceneLights = list(self.lights)
        self.scenePrimitives = list(self.primitives)
        if self.renderOnWorldEnd:
            self.outputImage = self.rendererObject.render(self.scene)
        saved = self.opt
        self._reset_world()
        self.lastOptions = saved
        return self.outputImage

    # -- conveniences for tests / tools ---------------------------------------
    def pointLights(self):
        """Delta lights: [(PointLight | SpotLight | DistantLight, index of the first primitive whose area light follows it in Scene.lights, or None)]."""
        out = []
        for k, l in enumerate(self.sceneLights):
            if not isinstance(l, (core.PointLight, core.DistantLight)):
                continue
            before = None
            for i, gp in enumerate(self.scenePrimitives):
                if gp.areaLight is not None and any(gp.areaLight is a for a in self.sceneLights[k + 1:]):
                    before = i
                    break
            out.append((l, before))
        return out

    def envLight(self):
        """(InfiniteAreaLight or None, index of the first primitive whose area light follows it or None)."""
        env = [l for l in self.sceneLights if isinstance(l, core.InfiniteAreaLight)]
        if not env:
            return None, None
        if len(env) > 1:
            raise UnsupportedFeature("more than one infinite light")
        after = self.sceneLights[self.sceneLights.index(env[0]) + 1:]
        for i, gp in enumerate(self.scenePrimitives):
            if gp.areaLight is not None and any(gp.areaLight is a for a in after):
                return env[0], i
        return env[0], None


def load(path, render=False, overrides=None, warn=None):
    """Parse a .pbrt (or .pbrt.gz) file; returns the DartRay API object with `.scene`, `.rendererObject`,
    `.scenePrimitives`, `.sceneLights` (and `.outputImage` when render=True)."""
    api = DartRay(render=render, overrides=overrides, warn=warn)
    api.base = os.path.dirname(os.path.abspath(path))
    PbrtParser(api).parse(path)
    return api


def loads(text, base=".", render=False, overrides=None, warn=None):
    api = DartRay(render=render, overrides=overrides, warn=warn)
    api.base = base
    PbrtParser(api).parseString(text, "<string>", base)
    return api


def _write_pfm(path, rgb):
    """Portable float map (little endian, bottom row first)."""
    h, w = rgb.shape[:2]
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode())
        f.write(np.ascontiguousarray(rgb[::-1], dtype="<f4").tobytes())


def main(argv=None):
    """python -m dartray_amd.pbrt scene.pbrt [-o out.pfm|out.npy] [--spp N] [--xres W --yres H]"""
    import argparse
    import time
    ap = argparse.ArgumentParser(description="Render a PBRT-v2 scene file on the MI355X path")
    ap.add_argument("scene")
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--xres", type=int)
    ap.add_argument("--yres", type=int)
    a = ap.parse_args(argv)
    ov = {k: v for k, v in (("pixelsamples", a.spp), ("xresolution", a.xres), ("yresolution", a.yres)) if v}
    t0 = time.time()
    api = load(a.scene, render=False, overrides=ov, warn=lambda m: print("warning:", m))
    t1 = time.time()
    out = api.rendererObject.render(api.scene)
    t2 = time.time()
    st = api.rendererObject.last_stats
    print(f"parsed + built BVH in {t1 - t0:.2f} s; rendered {out.width}x{out.height} in {t2 - t1:.2f} s "
          f"({st['camera_samples'] / max(t2 - t1, 1e-9) / 1e6:.1f} Msamples/s)")
    if a.output:
        if a.output.endswith(".npy"):
            np.save(a.output, out.rgb)
        else:
            _write_pfm(a.output, out.rgb)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
