"""Scenes over the size from which the hierarchies are built on the device (50 000 primitives of a kind), with the product's own
choice of builder, against the CPU oracle and against the same scene with the host builder.

The oracle walks the reference's octree, whose leaves hold thousands of primitives in scenes this size (about 70 000 tests a
ray at 500 000 triangles), so it renders 64 x 64 samples; the larger frames compare the two builders' results with each other.
Upload and frame times are printed and recorded, not asserted.
"""
import time

import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

from test_gpu_device_build import large_scene
from test_gpu_random_scenes import compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    r.set_option("device_build_min", 50000)
    r.close()


@pytest.mark.parametrize("side,n_spheres", [(200, 60000), (500, 0)])
def test_large_scene_device_and_host_builders_give_the_oracles_frame(gpu, tmp_path, record_property, side, n_spheres):
    """80 000 triangles with 60 000 spheres on them, and a terrain of 500 000 triangles."""
    p = tmp_path / "large.nff"
    p.write_text(large_scene(side, n_spheres))
    s = rt.Scene.load(str(p))
    frames, times = {}, {}
    for builder, least in (("device", 50000), ("host", 0)):
        gpu.set_option("device_build_min", least)
        gpu.upload(s)
        t0 = time.perf_counter()
        gpu.upload(s)
        times[builder] = time.perf_counter() - t0
        frames[builder] = gpu.render(s.camera(64, 64), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        for res in (1024, 2048):
            gpu.render(s.camera(res, res))
            frames[builder, res] = gpu.render(s.camera(res, res), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    s.close()
    record_property("upload_s_device", times["device"])
    record_property("upload_s_host", times["host"])
    print("%d triangles + %d spheres: upload with the device builder %.1f ms, with the host builder %.1f ms; 2048^2 frame %.2f against %.2f ms"
          % (2 * side * side, n_spheres, 1e3 * times["device"], 1e3 * times["host"], frames["device", 2048]["stats"]["gpu_ms"],
             frames["host", 2048]["stats"]["gpu_ms"]))
    o = ob.OracleScene(str(p))
    o.set_resolution(64, 64)
    ref = o.render(threads=8)
    o.close()
    compare(frames["device"], ref, 1e-9)
    compare(frames["host"], ref, 1e-9)
    for res in (1024, 2048):
        a, b = frames["device", res], frames["host", res]
        assert np.array_equal(a["hit_id"], b["hit_id"]) and np.array_equal(a["rgb_u8"], b["rgb_u8"])
        assert np.abs(a["rgb_f64"] - b["rgb_f64"]).max() <= 1e-9
        assert a["stats"]["closest_rays"] == b["stats"]["closest_rays"] and a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"]
