"""A scene two orders of magnitude beyond the shipped ones: a 300 x 300 height field (180 k triangles that
share every interior edge) under 2000 spheres.  The hierarchy no longer fits the on-chip node cache, the
reference's octree puts thousands of primitives in front of every ray, and the device must still return the
oracle's primitive ids and image."""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def terrain(path, n, n_spheres):
    rng = np.random.default_rng(5)
    xs = np.linspace(-20, 20, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = 1.5 * np.sin(X * 0.7) * np.cos(Y * 0.5) + 0.06 * rng.normal(size=X.shape)
    V = np.stack([X, Y, Z], axis=2)
    out = ["b 0.2 0.3 0.5\nv\nfrom 30 24 18\nat 0 0 0\nup 0 0 1\nangle 45\nhither 1\nresolution 80 72\nl 30 10 40\nl -20 30 35\nl 5 -30 20\n",
           "f 0.4 0.7 0.3 0.8 0.3 12.5 0 1\n"]
    def v(i, j):
        return "%.17g %.17g %.17g" % tuple(V[i, j])
    for i in range(n):
        for j in range(n):
            a, b, c, d = v(i, j), v(i + 1, j), v(i + 1, j + 1), v(i, j + 1)
            out.append("p 3\n%s\n%s\n%s\np 3\n%s\n%s\n%s\n" % (a, b, c, a, c, d))
    out.append("f 0.9 0.8 0.7 0.5 0.6 40 0 1\n")
    for _ in range(n_spheres):
        c = rng.uniform(-18, 18, 2)
        out.append("s %.17g %.17g %.17g %.17g\n" % (c[0], c[1], 2.2 + rng.uniform(0, 3), rng.uniform(0.1, 0.5)))
    with open(path, "w") as f:
        f.write("".join(out))


def test_height_field_with_spheres(tmp_path):
    path = str(tmp_path / "terrain.nff")
    terrain(path, 300, 2000)
    s = rt.Scene.load(path)
    gpu = rt.Renderer(0)
    gpu.upload(s)
    g = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    o = ob.OracleScene(path)
    ref = o.render(threads=16)
    o.close()
    assert int((g["hit_id"] != ref["hit_id"]).sum()) == 0
    assert (g["hit_id"] >= 0).mean() > 0.5
    assert g["stats"]["closest_rays"] == ref["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]
    assert np.abs(g["rgb_f64"] - ref["rgb_f64"]).max() <= 1e-9
    assert np.array_equal(g["rgb_u8"], ref["rgb_u8"])
    # the same frame in four row bands
    cam = s.camera()
    for p in range(4):
        rows = [r for r in range(cam.resy) if (r // 8) % 4 == p]
        part = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT, band_rows=8, n_parts=4, part=p, want=("rgb_u8",))
        assert np.array_equal(part["rgb_u8"], g["rgb_u8"][rows])
    # the same frame through the large-frame configuration (one 1024-thread block per compute unit, deep stacks
    # beside a partial node and record cache in LDS), which a frame this small does not select by itself
    gpu.set_option("block_threads", 1024)
    try:
        big = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    finally:
        gpu.set_option("block_threads", 0)
    assert np.array_equal(big["hit_id"], g["hit_id"])
    assert np.array_equal(big["rgb_u8"], g["rgb_u8"])
    assert big["stats"]["closest_rays"] == g["stats"]["closest_rays"] and big["stats"]["shadow_rays"] == g["stats"]["shadow_rays"]
    gpu.close()
    s.close()
