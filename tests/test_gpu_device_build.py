"""Hierarchies built on the device (csrc/lbvh_device.hpp) against the CPU oracle and against the host builder's.

The walks, the view grids and the self-contained blocks read the same node and leaf-record formats whoever built them, and
every hit is still decided by the reference's own tests on the same records, so a frame must not depend on the builder:
primary hit ids, ray counts and the 8-bit image identical, the double frame within 1e-9 of 255.  `device_build_min`
(primitives per hierarchy from which the device builds it; 50 000 by default) is lowered to 2 so that small scenes the
oracle renders in seconds go through the device builder too.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

from test_gpu_random_scenes import compare, fmt, random_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    for k, v in (("device_build_min", 50000), ("device_build_stack", 63), ("grid", 1), ("grid_min_samples", -1), ("local", -1)):
        r.set_option(k, v)
    r.close()


MODES = (("walks", {"grid": 0, "local": 0}), ("grids", {"grid": 1, "grid_min_samples": 0, "local": 0}), ("blocks", {"grid": 0, "local": 1}))


def frames_by_builder(gpu, path, flags=rt.FLAG_KEEP_ZERO_WEIGHT, res=None, **kw):
    s = rt.Scene.load(path)
    out = {}
    for builder, least in (("device", 2), ("host", 0)):
        gpu.set_option("device_build_min", least)
        for name, opts in MODES:
            for k, v in opts.items():
                gpu.set_option(k, v)
            gpu.upload(s)
            out[builder, name] = gpu.render(s.camera(*(res or (0, 0))), flags=flags, **kw)
    s.close()
    return out


@pytest.mark.parametrize("seed", range(16))
def test_random_scenes_through_device_built_hierarchies(gpu, tmp_path, seed):
    p = tmp_path / ("dev%d.nff" % seed)
    glassy = seed % 4 == 3
    p.write_text(random_scene(9000 + seed, transparent_spheres=glassy))
    flags = rt.FLAG_KEEP_ZERO_WEIGHT | (rt.FLAG_NO_LIGHTCAM if glassy else 0)
    frames = frames_by_builder(gpu, str(p), flags=flags, num_samples=1 + (seed % 5 == 1), seed=seed)
    o = ob.OracleScene(str(p))
    ref = o.render(threads=8, flags=ob.FLAG_NO_LIGHTCAM if glassy else 0, num_samples=1 + (seed % 5 == 1), seed=seed)
    o.close()
    for g in frames.values():
        compare(g, ref, 1e-9)


@pytest.mark.parametrize("scene,res", [("teapot-3", None), ("balls", (300, 300)), ("gears", (160, 160)), ("balls2", (160, 160)),
                                       ("tetra", (256, 256)), ("refract", (320, 200))])
def test_shipped_scenes_through_device_built_hierarchies(gpu, oracle_frame, scene, res):
    o = oracle_frame(scene, res=res)
    for g in frames_by_builder(gpu, scene, res=res).values():
        compare(g, o, 1e-9)


def test_a_hierarchy_too_deep_for_the_walks_is_built_again_on_the_host(gpu, oracle_frame):
    """`device_build_stack` lowers the depth the device builder's result may have to one no tree of 1500 triangles meets."""
    gpu.set_option("device_build_stack", 2)
    try:
        o = oracle_frame("teapot-3")
        for g in frames_by_builder(gpu, "teapot-3").values():
            compare(g, o, 1e-9)
    finally:
        gpu.set_option("device_build_stack", 63)


def large_scene(side, n_spheres, seed=5):
    """A height field of 2 * side^2 triangles with spheres sitting on it; the eye looks across it, two lights."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-8, 8, side + 1)
    x, y = np.meshgrid(xs, xs, indexing="ij")
    z = 0.6 * np.sin(1.3 * x) * np.cos(0.9 * y) + 0.05 * rng.normal(size=x.shape)
    out = ["b 0.2 0.3 0.5\nv\nfrom 9 -11 7\nat 0 0 0\nup 0 0 1\nangle 50\nhither 1\nresolution 128 128\n",
           "l 6 4 12\nl -9 -3 8\n", "f 0.7 0.6 0.4 0.8 0.2 20 0 1\n"]
    rows = []
    for i in range(side):
        for j in range(side):
            a = "%.17g %.17g %.17g" % (x[i, j], y[i, j], z[i, j])
            b = "%.17g %.17g %.17g" % (x[i + 1, j], y[i + 1, j], z[i + 1, j])
            c = "%.17g %.17g %.17g" % (x[i + 1, j + 1], y[i + 1, j + 1], z[i + 1, j + 1])
            d = "%.17g %.17g %.17g" % (x[i, j + 1], y[i, j + 1], z[i, j + 1])
            rows.append("p 3\n%s\n%s\n%s\np 3\n%s\n%s\n%s\n" % (a, b, c, a, c, d))
    out.append("".join(rows))
    out.append("f 0.9 0.3 0.3 0.6 0.4 40 0 1\n")
    cs = rng.uniform(-7.5, 7.5, (n_spheres, 2))
    out.append("".join("s %.17g %.17g %.17g %.17g\n" % (c[0], c[1], 0.75 + 0.1 * rng.random(), 0.01 + 0.03 * rng.random()) for c in cs))
    return "".join(out)


@pytest.mark.parametrize("scene,res", [("teapot-3", (512, 512)), ("balls", (400, 400)), ("teapot-3", (2304, 2304))])
def test_frames_walked_with_deep_stacks(gpu, scene, res):
    """A hierarchy whose walks need 39 stack entries or more does not fit the LDS of a compute unit full of waves: such frames
    run in blocks of 256 threads, as many to a compute unit as fit (`stack_min` makes a small scene run that way: the
    one-launch kernel at 512 x 512, a launch per pass from 4 M samples on).  Same frames as with the stacks the scene needs."""
    s = rt.Scene.load(scene)
    gpu.set_option("device_build_min", 50000)
    gpu.set_option("grid", 0)
    gpu.upload(s)
    cam = s.camera(*res)
    a = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    try:
        for entries in (40, 64):
            gpu.set_option("stack_min", entries)
            b = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
            assert np.array_equal(a["hit_id"], b["hit_id"]) and np.array_equal(a["rgb_u8"], b["rgb_u8"])
            assert np.abs(a["rgb_f64"] - b["rgb_f64"]).max() <= 1e-9
            assert a["stats"]["closest_rays"] == b["stats"]["closest_rays"] and a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"]
            assert b["stats"]["queue_overflow"] == 0
    finally:
        gpu.set_option("stack_min", 0)
        gpu.set_option("grid", 1)
    s.close()
