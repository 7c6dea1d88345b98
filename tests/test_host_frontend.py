"""CPU-side checks of the host front end and of the C ABI surface (no GPU needed).

The host front end (C++: .nff ingest, ear clipping, octree build and flattening) and the oracle
(C: the same steps in the reference's recursive form) were written independently; they must
produce the same primitives, the same octree and the same camera, bit for bit.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracer_amd as rt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL_SCENES = ["tetra-3", "teapot-3", "balls", "refract", "glass", "glass2", "glass3", "glass4", "ball", "ball2", "2d",
              "teapot-1", "teapot-2", "tetra-5", "teapot", "tetra", "gears", "balls2"]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "raytracer_amd.h")).read()
    declared = set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", header))
    declared -= {"rt_material", "rt_light"}
    assert len(declared) >= 18
    lib = rt.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export " + name
    assert declared == set(rt._SIGNATURES), "binding and header disagree: %s" % (declared ^ set(rt._SIGNATURES))


def test_struct_sizes_match_the_header():
    assert C.sizeof(rt.Node) == 64
    assert C.sizeof(rt.TriRec) == 96
    assert C.sizeof(rt.SphRec) == 32
    assert C.sizeof(rt.Material) == 64
    assert C.sizeof(rt.Light) == 64


@pytest.mark.parametrize("name", ALL_SCENES)
def test_front_end_equals_oracle_structures(oracle_mod, name):
    path = rt.scene_path(name)
    s = rt.Scene.load(name)
    o = oracle_mod.OracleScene(path)
    d = s.desc
    assert (d.n_nodes, d.n_prims, d.n_lights) == (o.n_nodes, o.n_solids, o.n_lights)
    assert s.resolution == (o.resx, o.resy)
    assert d.lightcam_samples == o.lightcam_samples
    node4, cnt, ids = o.octree()
    nodes = s.nodes()
    assert np.array_equal(nodes["center"], node4[:, :3])
    assert np.array_equal(nodes["radius"], node4[:, 3])
    assert np.array_equal(nodes["tri_count"] + nodes["sph_count"], cnt)
    # pre-order layout with skip links: skip = index past the subtree
    assert nodes["skip"][0] == d.n_nodes
    assert (nodes["skip"] > np.arange(d.n_nodes)).all()
    tri = s.tri_records()
    sph, sprim = s.sph_records()
    off = 0
    for i in range(d.n_nodes):
        tb, tc, sb, sc = (int(nodes[k][i]) for k in ("tri_begin", "tri_count", "sph_begin", "sph_count"))
        mine = list(tri["prim"][tb:tb + tc]) + [int(x) & 0x7FFFFFFF for x in sprim[sb:sb + sc]]
        theirs = list(ids[off:off + cnt[i]])
        off += cnt[i]
        assert sorted(mine) == theirs
        assert list(tri["prim"][tb:tb + tc]) == sorted(tri["prim"][tb:tb + tc])   # ascending id inside a node
    cam, oc = s.camera(), o.camera()
    for k in ("u", "v", "w", "e"):
        assert np.array_equal(np.array(getattr(cam, k)), oc[k])
    for k in ("d", "l", "t", "pix_width"):
        assert getattr(cam, k) == oc[k]
    flags, mat, nrm = s.prim_arrays()
    mats = s.materials()
    step = max(1, d.n_prims // 300)
    for i in range(0, d.n_prims, step):
        so = o.solid(i)
        assert so["type"] == (flags[i] & rt.PRIM_SPHERE)
        assert np.array_equal(mats[mat[i]], so["material"])
        assert np.array_equal(s.prim_geometry(i), so["geom"])
        assert bool(flags[i] & rt.PRIM_HAS_VNORMALS) == bool(so["has_vn"])
        if so["has_vn"]:
            assert np.array_equal(nrm[i, 3:], so["vn"])
    # transparent-sphere tags follow the material
    for k in range(d.n_sph_slots):
        p = int(sprim[k]) & 0x7FFFFFFF
        assert bool(int(sprim[k]) & rt.SPH_TRANSPARENT) == (mats[mat[p]][6] > 0)
    assert np.array_equal(s.lights()[:, :7], o.lights())
    assert np.array_equal(np.array(d.background), o.background())
    o.close()
    s.close()


def test_triangle_records_are_the_reference_precompute():
    s = rt.Scene.load("teapot-3")
    tri = s.tri_records()
    for r in tri[:: max(1, len(tri) // 50)]:
        g = s.prim_geometry(int(r["prim"])).reshape(3, 3)
        assert np.array_equal(r["v0"], g[0])
        assert np.array_equal(r["ab"], g[0] - g[1])     # Triangle.h:42
        assert np.array_equal(r["ac"], g[0] - g[2])     # Triangle.h:43


def test_camera_quirks():
    s = rt.Scene.load("teapot-3")
    # aspect ratio is an integer division (Camera.h:73): 7680/4320 == 1, 100/200 == 0
    wide = s.camera(7680, 4320)
    sq = s.camera(7680, 7680)
    assert wide.t == sq.t and wide.l == sq.l and wide.pix_width == sq.pix_width
    tall = s.camera(100, 200)
    assert tall.t == -0.5 * tall.pix_width
    # PI is 3.14159 (Camera.h:13)
    import math
    half = wide.d * math.tan(45.0 / 2 * 3.14159 / 180.0)
    assert wide.pix_width == 2 * half / 7680


def _parse(argv):
    lib = rt.load()
    arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])
    n, a, p = C.c_int32(), C.c_double(), C.c_int32()
    inp, outp = C.create_string_buffer(512), C.create_string_buffer(512)
    rc = lib.rth_parse_args(len(argv), arr, C.byref(n), C.byref(a), C.byref(p), inp, 512, outp, 512)
    return rc, n.value, a.value, p.value, inp.value.decode(), outp.value.decode()


def test_argument_scanner_has_the_reference_behaviour():
    assert _parse(["drv", "-s", "4", "-a", "0.25", "-p", "teapot-3.nff", "out.ppm"]) == \
        (0, 4, 0.25, 1, "teapot-3.nff", "out.ppm")
    # ".nff" is appended when missing (Driver.cpp:771-783)
    assert _parse(["drv", "teapot-3", "out.ppm"])[4] == "teapot-3.nff"
    # any argument starting with p/s/a is taken as that flag (Driver.cpp:176-205):
    # an output called picture.ppm switches interpolated normals on
    assert _parse(["drv", "teapot-3.nff", "picture.ppm"])[3] == 1
    # an input called sphere.nff swallows the next argument as the sample count
    rc, n, _, _, inp, outp = _parse(["drv", "sphere.nff", "7.ppm"])
    assert (rc, n, inp, outp) == (0, 7, "sphere.nff", "7.ppm")
    assert _parse(["drv"])[0] != 0


def test_ppm_writer_truncates_like_the_reference(tmp_path):
    px = np.array([[[0.0, 0.999, 1.0], [254.999999, 255.0, 127.5]]])
    path = str(tmp_path / "x.ppm")
    rt.write_ppm(path, px)
    data = open(path, "rb").read()
    assert data == b"P6\n2 1\n255\n" + bytes([0, 0, 1, 254, 255, 127])
    rt.write_ppm(path, np.array([[[1, 2, 3]]], dtype=np.uint8))
    assert open(path, "rb").read() == b"P6\n1 1\n255\n\x01\x02\x03"


def test_missing_file_is_an_error():
    with pytest.raises(rt.RtError) as e:
        rt._check(rt.load().rth_scene_load(b"/nonexistent/scene.nff", C.byref(C.c_void_p())))
    assert e.value.code == -3


def test_row_band_partition_is_a_partition():
    s = rt.Scene.load("tetra-3")
    for resy, band, parts in [(100, 16, 8), (4320, 16, 8), (7, 3, 2), (512, 0, 1), (5, 16, 4)]:
        cam = s.camera(64, resy)
        seen = np.zeros(resy, dtype=int)
        for p in range(parts):
            rows = rt.part_row_indices(resy, band, parts, p)
            assert rt.part_rows(cam, band, parts, p) == len(rows)
            assert (np.diff(rows) > 0).all()
            seen[rows] += 1
        assert (seen == 1).all()


def test_no_gpu_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError):
        rt.Renderer(0)


HEADER = "b 0 0 0\nv\nfrom 3 3 3\nat 0 0 0\nup 0 0 1\nangle 40\nhither 1\nresolution 8 8\n"


def test_malformed_files_are_errors_or_empty_scenes_never_a_hang(tmp_path):
    def load(name, text):
        p = tmp_path / name
        p.write_bytes(text if isinstance(text, bytes) else text.encode())
        return rt.Scene.load(str(p))

    # a polygon that announces more vertices than the file has lines left (truncated or corrupt file)
    with pytest.raises(rt.RtError):
        load("trunc.nff", HEADER + "f 1 1 1 1 0 1 0 1\np 4\n0 0 0\n1 0 0\n")
    with pytest.raises(rt.RtError):
        load("huge.nff", HEADER + "p 200000000\n")
    # nothing usable: an empty scene with resolution 0 x 0 (rendering it is refused by rt_render)
    for name, text in (("empty.nff", ""), ("noise.nff", bytes(range(256)) * 8), ("half_view.nff", "v\nfrom 1 1 1\nat 0 0 0\n")):
        s = load(name, text)
        assert s.desc.n_prims == 0 and s.desc.n_nodes == 1
        s.close()
    # degenerate primitives are kept as the reference keeps them: a zero-area triangle, a zero-radius
    # sphere, a two-vertex "polygon" (dropped: fewer than three vertices), a polygon without an `f` line
    s = load("degenerate.nff", HEADER + "p 3\n0 0 0\n0 0 0\n0 0 0\nf 1 1 1 1 0 1 0 1\ns 0 0 0 0\np 2\n0 0 0\n1 1 1\n")
    assert s.desc.n_prims == 2
    assert s.desc.n_materials == 2          # the all-zero material of the polygon that had no `f` line, then the real one
    s.close()


def test_division_by_launch_constants():
    """The device divides sample positions by numbers fixed per launch with a multiply-high (Granlund and
    Montgomery's round-up method, `FastDiv` in csrc/device_records.hpp); the same arithmetic here, against //."""
    rng = np.random.default_rng(4)
    divisors = list(range(1, 70)) + [255, 256, 257, 960, 1000, 4095, 4096, 4097, 65535, 65536, 2 ** 20 + 7, 2 ** 28, 2 ** 31 - 1,
                                     2 ** 31, 2 ** 31 + 1, 2 ** 32 - 1]
    numbers = np.concatenate([np.arange(0, 300, dtype=np.uint64), rng.integers(0, 2 ** 32, 20000, dtype=np.uint64),
                              np.array([2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 31 - 1], dtype=np.uint64)])
    for d in divisors:
        if d < 2:
            q = numbers.copy()
        else:
            l = 0
            while (1 << l) < d:
                l += 1
            magic = (((1 << l) - d) << 32) // d + 1
            assert magic < 2 ** 32
            t = (numbers * np.uint64(magic)) >> np.uint64(32)
            q = (t + ((numbers - t) >> np.uint64(1))) >> np.uint64(l - 1)
        assert np.array_equal(q, numbers // np.uint64(d)), d
    for k in divisors:                          # multiples and their neighbours
        if k >= 2:
            n = np.array([m for m in (k, k - 1, k + 1, 3 * k, 3 * k - 1, (2 ** 32 - 1) // k * k, (2 ** 32 - 1) // k * k - 1) if 0 <= m < 2 ** 32],
                         dtype=np.uint64)
            l = 0
            while (1 << l) < k:
                l += 1
            magic = (((1 << l) - k) << 32) // k + 1
            t = (n * np.uint64(magic)) >> np.uint64(32)
            assert np.array_equal((t + ((n - t) >> np.uint64(1))) >> np.uint64(l - 1), n // np.uint64(k)), k


def test_public_header_is_plain_c(tmp_path):
    """include/raytracer_amd.h is the drop-in boundary: it must compile as C99 and as C++11 on its own, and a C
    program must link against the library with nothing but the header."""
    import shutil
    import subprocess
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no host compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_header.c"
    src.write_text('#include "raytracer_amd.h"\n'
                   '#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    rt_ctx *ctx = 0; rth_scene *sc = 0;\n'
                   '    if (rth_scene_load("/nonexistent.nff", &sc) == RT_OK) return 2;\n'
                   '    printf("%s|%d\\n", rt_last_error(), (int)sizeof(rt_node));\n'
                   '    (void)ctx; return 0;\n'
                   '}\n')
    inc = os.path.join(root, "include")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")], check=True)
    subprocess.run(["g++", "-std=c++11", "-pedantic", "-Wall", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")],
                   check=True)
    libdir = os.path.join(root, "raytracer_amd")
    exe = tmp_path / "use_header"
    subprocess.run(["gcc", str(tmp_path / "a.o"), "-o", str(exe), "-L", libdir, "-lraytracer_amd", "-Wl,-rpath," + libdir,
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip()
    msg, size = out.rsplit("|", 1)
    assert size == "64" and "nonexistent" in msg
