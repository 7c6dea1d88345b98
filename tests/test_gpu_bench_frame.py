"""The frames bench.py times are the frames that are checked here: each workload's 8-bit image against the CPU oracle
(whole frame for the scene files at their own size, rows spread over the frame for the 8K x 16 spp one) and its sha256
against the value pinned in tests/golden/golden.json, which bench.py reports its own frame against."""
import hashlib

import numpy as np
import pytest

import bench
import raytracer_amd as rt
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload", ["teapot3_8k16", "teapot3", "balls", "refract", "tetra3"])
def test_benchmark_frame_is_the_pinned_frame_and_equals_the_oracle(golden, workload):
    import os
    scene_name, resx, resy, N, _, _ = bench.WORKLOADS[workload]
    s = rt.Scene.load(scene_name)
    gpu = rt.Renderer(0)
    gpu.upload(s)
    cam = s.camera(resx, resy)
    g = gpu.render(cam, num_samples=N, want=("rgb_u8",))
    sha = hashlib.sha256(g["rgb_u8"].tobytes()).hexdigest()
    assert sha == golden["frame_sha256"][workload]
    o = ob.OracleScene(rt.scene_path(scene_name))
    if resx > 0:
        o.set_resolution(resx, resy)
    if cam.resx * cam.resy * N * N <= 1 << 20:
        rows = list(range(cam.resy))
    else:
        rows = sorted({int((k + 0.5) * cam.resy / 12) for k in range(12)})
    ref = o.render(num_samples=N, threads=min(16, os.cpu_count() or 1), rows=rows, want_hits=False, want_mask=False)
    assert np.array_equal(g["rgb_u8"][rows], ref["rgb_u8"][rows])
    o.close()
    gpu.close()
    s.close()
