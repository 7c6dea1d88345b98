"""The command-line host (`rt_driver`, the reference's `main`: Driver.cpp:91-216) end to end on the GPU.

`rt_driver glass.nff out.ppm` must write, byte for byte, the `hide.ppm` the reference's authors committed.
The driver runs as a child process, and children are only started while this process has not touched the GPU
itself -- hence the file name, which makes pytest collect it before the other GPU tests, and the guard below.
"""
import gzip
import os
import subprocess

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "raytracer_amd", "bin", "rt_driver")


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    magic, dims, maxval, rest = data.split(b"\n", 3)
    assert magic == b"P6" and maxval == b"255"
    w, h = (int(x) for x in dims.split())
    return np.frombuffer(rest, dtype=np.uint8).reshape(h, w, 3)


def test_driver_writes_the_reference_image_and_honours_its_flags(tmp_path):
    import torch
    if rt.Renderer.created or torch.cuda.is_initialized():
        pytest.skip("this process already uses the GPU; run this file first or on its own")
    assert os.path.exists(DRIVER), "build it with python -m raytracer_amd.build"
    glass_out, teapot_out = str(tmp_path / "glass.ppm"), str(tmp_path / "teapot.ppm")
    runs = [[DRIVER, rt.scene_path("glass"), glass_out],
            [DRIVER, "-s", "2", "-a", "0.1", "-p", rt.scene_path("teapot-3"), teapot_out],
            [DRIVER, str(tmp_path / "missing.nff"), str(tmp_path / "none.ppm")]]
    done = [subprocess.run(cmd, capture_output=True, text=True, timeout=110) for cmd in runs]
    assert done[0].returncode == 0, done[0].stderr
    assert done[1].returncode == 0, done[1].stderr
    assert done[2].returncode != 0 and not os.path.exists(runs[2][2])
    assert "Num samples = 2" in done[1].stdout and "Phong = 1" in done[1].stdout

    with gzip.open(os.path.join(ROOT, "tests", "golden", "hide.ppm.gz"), "rb") as f:
        golden = f.read()
    with open(glass_out, "rb") as f:
        assert f.read() == golden

    # the flagged run against the same options through the Python binding (this part does use the GPU here)
    img = read_ppm(teapot_out)
    s = rt.Scene.load("teapot-3")
    gpu = rt.Renderer(0)
    gpu.upload(s)
    fr = gpu.render(s.camera(), num_samples=2, aperture=0.1, flags=rt.FLAG_PHONG_NORMALS)
    assert np.array_equal(img, fr["rgb_u8"])
    gpu.close()
    s.close()


def test_pointer_octree_host_example(tmp_path):
    """examples/pointer_octree_host.cpp, the compile-tested form of the reference-side integration: a host that keeps a
    pointer octree and a primitive vector, flattens them (depth-first pre-order with skip links) and renders through the C ABI.  Its
    arrays must equal the front end's (exit code 0) and its image the command-line host's."""
    example = os.path.join(ROOT, "raytracer_amd", "bin", "pointer_octree_host")
    assert os.path.exists(example), "build it with python -m raytracer_amd.build"
    for scene in ("teapot-3", "glass3", "balls"):
        a, b = str(tmp_path / (scene + "_example.ppm")), str(tmp_path / (scene + "_driver.ppm"))
        r = subprocess.run([example, rt.scene_path(scene), a], capture_output=True, text=True, timeout=110)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "octree nodes flattened" in r.stdout
        r2 = subprocess.run([DRIVER, rt.scene_path(scene), b], capture_output=True, text=True, timeout=110)
        assert r2.returncode == 0, r2.stderr
        with open(a, "rb") as fa, open(b, "rb") as fb:
            assert fa.read() == fb.read()
