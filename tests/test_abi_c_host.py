"""The C ABI seen from a foreign host: tests/abi/c1_from_c.c is compiled with gcc against include/dartray_hip.h alone
(no Python, no HIP headers), links to the product library, builds BASELINE config C1 by hand and renders it with the
recorded serial sample stream of tests/golden/c1_serial.npz.  CPU suite: the header compiles as C11 and as C++17 with
its layout asserts, the program links, every declared symbol is exported, and the struct layouts of dartray_amd/_abi.py
(the ctypes host) agree with the header's.  GPU suite: the program's film equals the golden."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from dartray_amd import _abi

HEADER = os.path.join(ROOT, "include", "dartray_hip.h")
SRC = os.path.join(ROOT, "tests", "abi", "c1_from_c.c")


def _build(tmp_path):
    exe = str(tmp_path / "c1_from_c")
    libdir = os.path.join(ROOT, "dartray_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           os.path.join(libdir, "libdartray_hip.so"), "-Wl,-rpath," + libdir])
    return exe


def test_header_compiles_as_c11_and_cxx17_with_its_layout_asserts():
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", HEADER])


def test_c_host_links_against_the_header_and_library_only(hip, tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    # without its input blob the program stops before any device call
    assert subprocess.run([exe, str(tmp_path / "missing"), str(tmp_path / "out")]).returncode == 1


def test_every_function_the_header_declares_is_exported_and_bound(hip):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(dr_[a-z0-9_]+)\s*\(", text))
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)


def test_ctypes_layouts_equal_the_headers_static_asserts(hip):
    """DR_ABI_SIZE / DR_ABI_OFFSET lines of the header vs the ctypes Structures of the Python host."""
    text = open(HEADER).read()
    sizes = re.findall(r"^DR_ABI_SIZE\((\w+), (\d+)\);", text, flags=re.M)
    offsets = re.findall(r"^DR_ABI_OFFSET\((\w+), (\w+), (\d+)\);", text, flags=re.M)
    assert len(sizes) >= 14 and len(offsets) >= 40
    for name, n in sizes:
        assert C.sizeof(getattr(_abi, name)) == int(n), name
    for name, field, n in offsets:
        assert getattr(getattr(_abi, name), field).offset == int(n), (name, field)


def test_comm_entry_points_fail_loudly_without_a_communicator(hip):
    lib = hip.lib()
    assert lib.dr_comm_world() == 0 and lib.dr_comm_rank() == -1
    buf = (C.c_float * 4)()
    assert lib.dr_film_reduce(C.cast(buf, C.c_void_p), 1, 0, None) == -1  # DR_ERR_INVALID
    assert b"dr_comm_init" in lib.dr_last_error()
    assert lib.dr_comm_allreduce_f64(C.cast(buf, C.c_void_p), 1, 1, None) == -1
    ident = (C.c_char * 128)()
    assert lib.dr_comm_init(0, 0, C.cast(ident, C.c_void_p), 128) == -1  # world < 1
    assert lib.dr_comm_init(0, 1, C.cast(ident, C.c_void_p), 64) == -1   # short id
    assert lib.dr_comm_destroy() == 0                                    # nothing to destroy


@pytest.mark.gpu
def test_c1_rendered_by_the_c_host_equals_the_golden(gpu, tmp_path):
    from dartray_amd import scenes
    g = np.load(os.path.join(GOLDEN, "c1_serial.npz"))
    prims, mk = scenes.config("C1")
    r = mk()
    d, keep = r.describe()  # camera + film as the host objects hold them (what a Dart host reads off its Camera / Film)
    npix, spp, stride = len(g["pixel_xy"]), 4, g["sample_vec"].shape[1]
    film = r.camera.film
    blob = tmp_path / "in.blob"
    with open(blob, "wb") as f:
        f.write(np.array([npix, spp, stride, film.width, film.height], np.int32).tobytes())
        f.write(bytes(d.camera))
        f.write(bytes(d.film))
        f.write(np.ascontiguousarray(g["pixel_xy"], np.int32).tobytes())
        f.write(np.ascontiguousarray(g["sample_vec"], np.float32).tobytes())
    exe = _build(tmp_path)
    out = tmp_path / "out.blob"
    res = subprocess.run([exe, str(blob), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "c1_from_c: 16900 camera samples" in res.stdout, res.stdout
    raw = np.fromfile(out, np.float32)
    n4 = film.width * film.height * 4
    assert np.array_equal(raw[:n4].reshape(film.height, film.width, 4), g["film"])
    assert np.array_equal(raw[n4:].reshape(film.height, film.width, 3), g["rgb"])


def test_dart_binding_offsets_equal_the_c_layout(tmp_path):
    """integration/hip_sampler_renderer.dart writes the structs field by field at literal byte offsets (no Dart SDK
    exists here to run it): every SIZEOF_<Struct> / OFF_<Struct>_<field> constant of that file is checked against
    sizeof / offsetof of include/dartray_hip.h by a generated C program."""
    dart = open(os.path.join(ROOT, "integration", "hip_sampler_renderer.dart")).read()
    sizes = re.findall(r"^const int SIZEOF_(\w+) = (\d+);", dart, flags=re.M)
    offs = re.findall(r"^const int OFF_(Dr[A-Za-z]+)_(\w+) = (\d+);", dart, flags=re.M)
    assert len(sizes) >= 10 and len(offs) >= 80
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "dartray_hip.h"', 'int main(void) {', '  int bad = 0;']
    for name, n in sizes:
        src.append('  if (sizeof(%s) != %s) { printf("sizeof(%s) = %%zu, Dart says %s\\n", sizeof(%s)); bad = 1; }' % (name, n, name, n, name))
    for name, field, n in offs:
        src.append('  if (offsetof(%s, %s) != %s) { printf("offsetof(%s, %s) = %%zu, Dart says %s\\n", offsetof(%s, %s)); bad = 1; }'
                   % (name, field, n, name, field, n, name, field))
    src += ['  return bad;', '}']
    c = tmp_path / "dart_offsets.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "dart_offsets")
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(c), "-o", exe])
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout
    # every entry point the Dart file looks up exists in the header, and the layout version it checks at load is the header's
    header = open(HEADER).read()
    assert re.search(r"static const int ABI_VERSION = (\d+);", dart).group(1) == re.search(r"#define DR_ABI_VERSION (\d+)", header).group(1)
    assert "lookupFunction<_CommVoidC, _CommVoidD>('dr_abi_version')" in dart
    for sym in re.findall(r"lookupFunction<\w+, \w+>\('(dr_\w+)'\)", dart):
        assert re.search(r"\b%s\s*\(" % sym, header), sym


def _dart_brackets(text):
    """A lexer's view of a Dart file: comments and string literals (single / double / triple quoted, raw, with nested
    ${...} interpolation) skipped, every bracket matched.  Returns the number of top-level declarations' closing braces."""
    stack, i, n, closed_top = [], 0, len(text), 0
    pairs = {")": "(", "]": "[", "}": "{"}

    def skip_string(i, raw):
        q = text[i]
        triple = text[i:i + 3] == q * 3
        i += 3 if triple else 1
        while i < n:
            c = text[i]
            if not raw and c == "\\":
                i += 2
                continue
            if not raw and c == "$" and i + 1 < n and text[i + 1] == "{":
                depth, i = 1, i + 2
                while i < n and depth:
                    if text[i] in "'\"":
                        i = skip_string(i, False)
                        continue
                    depth += {"{": 1, "}": -1}.get(text[i], 0)
                    i += 1
                continue
            if triple and text[i:i + 3] == q * 3:
                return i + 3
            if not triple and c == q:
                return i + 1
            assert triple or c != "\n", "unterminated string literal near offset %d" % i
            i += 1
        raise AssertionError("unterminated string literal")

    while i < n:
        c = text[i]
        if text[i:i + 2] == "//":
            i = text.find("\n", i)
            i = n if i < 0 else i
        elif text[i:i + 2] == "/*":
            depth, i = 1, i + 2
            while depth:  # Dart block comments nest
                a, b = text.find("/*", i), text.find("*/", i)
                assert b >= 0, "unterminated block comment"
                if 0 <= a < b:
                    depth, i = depth + 1, a + 2
                else:
                    depth, i = depth - 1, b + 2
        elif c in "'\"":
            i = skip_string(i, i > 0 and text[i - 1] == "r" and (i < 2 or not (text[i - 2].isalnum() or text[i - 2] == "_")))
        elif c in "([{":
            stack.append((c, text.count("\n", 0, i) + 1))
            i += 1
        elif c in ")]}":
            assert stack and stack[-1][0] == pairs[c], "line %d: '%s' closes %s" % (text.count("\n", 0, i) + 1, c, stack[-1] if stack else "nothing")
            stack.pop()
            closed_top += (c == "}" and not stack)
            i += 1
        else:
            i += 1
    assert not stack, "unclosed %s" % (stack[-1],)
    return closed_top


@pytest.mark.parametrize("name", ["hip_sampler_renderer.dart", "hip_render_manager.dart"])
def test_dart_sources_are_lexically_well_formed(name):
    """No Dart SDK in the image: the least a reader without one can check of integration/*.dart beyond offsets and symbol names --
    strings and comments terminate, every bracket closes the bracket it should, statements end (no line of code ends in an operator
    followed by a closing brace), and the file declares what INTEGRATION.md says it does."""
    text = open(os.path.join(ROOT, "integration", name)).read()
    assert _dart_brackets(text) >= 2
    assert _dart_brackets("void f() { var s = 'a${g('}')}b'; /* x /* y */ } */ }") == 1
    with pytest.raises(AssertionError):
        _dart_brackets("void f() { g(1]; }")
    if name == "hip_sampler_renderer.dart":
        assert re.search(r"class\s+HipSamplerRenderer\s+extends\s+Renderer", text)
    else:
        assert "dr_render_sharded" in text or "HipSamplerRenderer" in text


_DART_SDK_NAMES = set("""String List Map Set Object Future Stream Uint8List Float32List Float64List Int32List Uint32List Int64List Uint64List
Pointer Void Int32 Int64 Uint8 Uint32 Uint64 Float Double DynamicLibrary Struct NativeFunction NativeType Process ProcessResult File
Directory Platform Duration Stopwatch Exception StateError ArgumentError Utf8 Allocator Completer Endian ByteData Isolate Random Int8
Uint16 Int16 Char Opaque Abi FileMode IOSink Iterable Function Type Null DateTime RangeError UnsupportedError FormatException Float32
Float64 addAll addr address asTypedList getUint32 little lookupFunction setAll setFloat32 setFloat64 setInt32 setInt64 setUint16
setUint32 setUint64 setUint8 toNativeUtf8 addStream asFloat32List asUint8List createTemp delayed delete exists exitCode getInt32 isAfter
lengthInBytes openWrite readAsString rename resolvedExecutable stderr systemTemp toFilePath writeAsBytes writeAsString""".split())


@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="the reference is only present in the build container")
def test_dart_integration_uses_only_names_the_reference_declares():
    """Without an SDK nothing can type-check integration/*.dart; this is the check a text search allows: every capitalised name
    the two files use is a class / function of the reference, of the Dart SDK, or their own, and every `.member` they access is a
    word of the reference's sources, an SDK member, or their own -- a misspelt field of Scene / BVHAccel / Camera would show here."""
    import glob
    ref = "\n".join(open(f, errors="ignore").read() for f in glob.glob("/root/reference/lib/**/*.dart", recursive=True))
    ref_words = set(re.findall(r"\b[A-Za-z_]\w*\b", ref))
    texts, own = {}, set()
    for name in ("hip_sampler_renderer.dart", "hip_render_manager.dart"):
        t = open(os.path.join(ROOT, "integration", name)).read()
        t = re.sub(r"//[^\n]*", "", t)
        t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
        t = re.sub(r"'(?:\\.|[^'\\\n])*'", "''", t)
        texts[name] = t
        own |= set(re.findall(r"\b(?:class|typedef)\s+(\w+)", t)) | set(re.findall(r"\b(?:final|var|int|double|bool|String)\s+(\w+)\s*[;=,)]", t))
        own |= set(re.findall(r"\bthis\.(\w+)", t)) | set(re.findall(r"\b(\w+)\s*\([^;{]*\)\s*(?:async\s*)?(?:=>|\{)", t)) | {"HipSamplerRenderer", "HipRenderManager"}
    for name, t in texts.items():
        used = set(re.findall(r"\b([A-Z][a-z]\w*)\b", t)) | set(re.findall(r"\.([a-z]\w+)\b", t))
        unknown = sorted(w for w in used if w not in ref_words and w not in _DART_SDK_NAMES and w not in own)
        assert not unknown, (name, unknown)
