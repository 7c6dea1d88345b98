"""GPU edge cases and full-size property checks, all through the C ABI.

Small hand-written scenes exercise the corners the shipped scenes do not reach (nothing to hit,
no lights, geometry outside the root cube, a camera outside the root sphere, odd sample counts);
the 8K x 16 spp benchmark frame is checked through properties that do not need a CPU render of
the whole frame (idempotence, row-band parts tiling the frame) plus an oracle render of a few rows.
"""
import ctypes as C
import os

import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

pytestmark = pytest.mark.gpu

HEADER = """b 0.1 0.2 0.3
v
from {frm}
at 0 0 0
up 0 0 1
angle 40
hither 1
resolution {res}
"""


def write_scene(tmp_path, name, body, frm="6 5 4", res="48 40"):
    p = tmp_path / (name + ".nff")
    p.write_text(HEADER.format(frm=frm, res=res) + body)
    return str(p)


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    r.close()


def both(gpu, path, flags=rt.FLAG_KEEP_ZERO_WEIGHT, oracle_flags=0, **kw):
    s = rt.Scene.load(path)
    gpu.upload(s)
    g = gpu.render(s.camera(), flags=flags, **kw)
    o = ob.OracleScene(path)
    ref = o.render(threads=4, flags=oracle_flags, num_samples=kw.get("num_samples", 1), aperture=kw.get("aperture", 0.0),
                   seed=kw.get("seed", 0))
    o.close()
    s.close()
    return g, ref


def check(g, ref, tol=1e-9):
    assert np.array_equal(g["hit_id"], ref["hit_id"])
    assert np.array_equal(g["rgb_u8"], ref["rgb_u8"])
    assert np.abs(g["rgb_f64"] - ref["rgb_f64"]).max() <= tol
    assert g["stats"]["closest_rays"] == ref["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]


def test_empty_scene_is_background(gpu, tmp_path):
    g, ref = both(gpu, write_scene(tmp_path, "empty", "l 1 2 3\n"))
    check(g, ref)
    assert (g["hit_id"] == -1).all()
    assert np.array_equal(g["rgb_u8"][0, 0], [int(0.1 * 255), int(0.2 * 255), int(0.3 * 255)])


def test_no_lights_means_black_surfaces(gpu, tmp_path):
    body = "f 1 0 0 1 0.5 10 0 1\ns 0 0 0 1\n"
    g, ref = both(gpu, write_scene(tmp_path, "nolight", body))
    check(g, ref)
    assert g["stats"]["shadow_rays"] == 0
    assert (g["hit_id"] == 0).any()


def test_single_triangle_and_single_sphere(gpu, tmp_path):
    body = "l 5 5 9\nf 0.2 0.9 0.3 0.8 0.3 20 0 1\np 3\n-2 -2 0\n2 -2 0\n0 2 0\nf 0.9 0.3 0.2 0.7 0.4 30 0 1\ns 0 0 1 0.7\n"
    g, ref = both(gpu, write_scene(tmp_path, "two", body))
    check(g, ref)


def test_sphere_reaching_outside_the_root_cube_uses_the_exact_walk(gpu, tmp_path):
    # centre + radius beyond the 100-unit root cube (Driver.cpp:231): the hierarchy shortcut is not
    # provably equivalent there, so the device must walk the octree like the reference
    body = ("l 20 30 60\nf 0.8 0.8 0.2 0.9 0.2 10 0 1\ns 0 0 -60 58\n"
            "f 0.2 0.4 0.9 0.6 0.5 40 0 1\ns 1 0 1 0.8\ns -1 1 0.5 0.5\ns 0 -1.5 0.6 0.6\ns 2 2 0.4 0.4\ns -2 -1 0.7 0.3\ns 1 2 1.5 0.35\n")
    g, ref = both(gpu, write_scene(tmp_path, "outside", body, frm="9 7 5"))
    check(g, ref)
    g2, _ = both(gpu, write_scene(tmp_path, "outside2", body, frm="9 7 5"), flags=rt.FLAG_KEEP_ZERO_WEIGHT | rt.FLAG_NO_CULL)
    assert np.array_equal(g2["rgb_u8"], g["rgb_u8"])


def test_camera_outside_the_root_sphere(gpu, tmp_path):
    # the root's own bounding-sphere test (Octree.h:1006) fails for rays that start outside and point away
    body = ("l 150 20 60\nl -30 -40 50\nf 0.9 0.9 0.9 0.5 0.6 30 0 1\ns 0 0 0 6\nf 0.5 0.5 0.9 0.8 0.3 5 0 1\n"
            "p 4\n-40 -40 -6\n40 -40 -6\n40 40 -6\n-40 40 -6\n")
    g, ref = both(gpu, write_scene(tmp_path, "far", body, frm="160 30 40"))
    check(g, ref)


def test_transparent_triangles_and_mixed_primitives(gpu, tmp_path):
    body = ("l 4 6 9\nl -5 2 7\nf 0.9 0.9 1.0 0.2 0.4 60 0.8 1.5\np 4\n-1 -1 0.5\n1 -1 0.5\n1 1 1.5\n-1 1 1.5\n"
            "f 0.8 0.3 0.3 0.7 0.5 20 0 1\ns 0 0 -0.5 0.8\ns 1.5 0.5 0.3 0.4\n"
            "f 0.3 0.8 0.3 0.9 0 5 0 1\np 4\n-5 -5 -1.4\n5 -5 -1.4\n5 5 -1.4\n-5 5 -1.4\n")
    g, ref = both(gpu, write_scene(tmp_path, "mixed", body))
    check(g, ref)
    assert g["stats"]["rays_per_generation"][2] > 0      # reflections of refractions are traced


def test_degenerate_primitives(gpu, tmp_path):
    # zero-area and needle triangles, a zero-radius sphere, coincident duplicates: nothing may trap, and whatever
    # the reference's arithmetic makes of them (NaN ratios fail every comparison) the device must make of them too
    body = ("l 4 6 9\nf 0.7 0.7 0.2 0.8 0.3 10 0 1\np 3\n0 0 0\n0 0 0\n0 0 0\np 3\n-1 0 0.2\n1 0 0.2\n3 0 0.2\n"
            "s 0.5 0.5 0.5 0\np 3\n-2 -2 0\n2 -2 0\n0 2 0\np 3\n-2 -2 0\n2 -2 0\n0 2 0\n"
            "s 0 0 1 0.5\ns 0 0 1 0.5\np 3\n0 0 0.4\n1e-9 0 0.4\n0 1e-9 0.4\n")
    g, ref = both(gpu, write_scene(tmp_path, "degenerate", body))
    check(g, ref)
    assert (g["hit_id"] >= 0).any()


def test_non_finite_and_far_away_coordinates(gpu, tmp_path):
    # a NaN vertex, an infinite one and a vertex a billion units away: the reference never hits the first two
    # (every comparison with NaN is false) and treats the third like any other; the device falls back to the
    # octree walk for such scenes and must agree
    body = ("l 4 6 9\nf 0.7 0.7 0.2 0.8 0.3 10 0 1\np 3\n-2 -2 0\n2 -2 0\n0 2 0\np 3\nnan 0 0\n1 0 1\n0 1 1\n"
            "p 3\n0 0 1\ninf 0 1\n0 1 1\np 3\n0 0 0.5\n1e9 0 0.5\n0 1 0.5\ns 0 0 1 0.4\n")
    g, ref = both(gpu, write_scene(tmp_path, "nonfinite", body))
    check(g, ref)
    assert (g["hit_id"] >= 0).any()


def test_lights_that_add_nothing_are_not_queried(gpu, tmp_path):
    # The kernels of small frames skip the shadow query of a light whose Phong terms at the hit are zero for sure (behind the
    # shading normal and mirrored away from the viewer).  Cases where "behind" is not "nothing": Shine 0 (pow(0, 0) is 1),
    # Kd 0, Ks 0, a coefficient too large to rule out infinity times zero, lights on both sides of thin geometry.
    lights = "l 5 5 9\nl -6 -3 -7\nl 0.2 0.1 -9\nl -8 6 0.5\n"
    shapes = ("p 3\n-3 -3 {z}\n3 -3 {z}\n0 3 {z}\ns 1.2 0.3 {zs} 0.6\ns -1.1 -0.4 {zs} 0.5\np 3\n-1 2 {z}\n-2.5 2.5 {zt}\n-1.5 3.5 {zt}\n")
    for name, mat in (("plain", "f 0.8 0.5 0.3 0.7 0.4 12 0 1"), ("shine0", "f 0.8 0.5 0.3 0.7 0.4 0 0 1"), ("kd0", "f 0.8 0.5 0.3 0 0.6 8 0 1"),
                      ("ks0", "f 0.8 0.5 0.3 0.9 0 8 0 1"), ("black", "f 0.8 0.5 0.3 0 0 8 0 1"), ("huge", "f 1e200 0.5 0.3 1e150 0.4 12 0 1"),
                      ("mirror", "f 0.8 0.5 0.3 0.3 0.9 30 0 1")):
        body = lights + mat + "\n" + shapes.format(z="0", zs="0.8", zt="1.5") + "f 0.2 0.6 0.9 0.6 0.5 3 0 1\n" + shapes.format(z="-2", zs="-1.2", zt="-3")
        path = write_scene(tmp_path, "dark_" + name, body, res="96 80")
        frames = []
        for opts in ({"local": 1, "fused": 0}, {"local": 0, "fused": 0, "pair": 1}, {"local": 0, "fused": 0, "pair": 0}):
            for k, v in opts.items():
                gpu.set_option(k, v)
            try:
                g, ref = both(gpu, path)
            finally:
                for k in opts:
                    gpu.set_option(k, -1)
            if name == "huge":      # the colour overflows on both sides: the bytes and the ray counts are what can be compared
                assert np.array_equal(g["rgb_u8"], ref["rgb_u8"]) and g["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]
            else:
                check(g, ref)
            frames.append(g)
        assert np.array_equal(frames[0]["rgb_u8"], frames[2]["rgb_u8"]) and np.array_equal(frames[1]["rgb_u8"], frames[2]["rgb_u8"])


def test_more_lights_and_materials_than_the_on_chip_tables_hold(gpu, tmp_path):
    # 18 lights and 70 materials: entries beyond the first 16 / 64 are read from memory instead of LDS
    rng = np.random.default_rng(5)
    lines = ["l %.3f %.3f %.3f" % tuple(rng.uniform(3, 9, 3)) for _ in range(18)]
    for k in range(70):
        r, g_, b = rng.uniform(0.2, 1, 3)
        lines.append("f %.3f %.3f %.3f 0.7 0.3 %d 0 1" % (r, g_, b, 5 + k))
        x, y = (k % 10) * 0.4 - 2, (k // 10) * 0.5 - 1.75
        lines.append("p 3\n%.3f %.3f 0\n%.3f %.3f 0\n%.3f %.3f 0" % (x, y, x + 0.38, y, x, y + 0.48))
    g, ref = both(gpu, write_scene(tmp_path, "many", "\n".join(lines) + "\n", res="64 64"))
    check(g, ref, tol=1e-8)
    assert len(np.unique(g["hit_id"])) > 60


def test_many_light_cameras(gpu, tmp_path, oracle_mod):
    # 40 transparent spheres x 3 lights = 120 (sphere, light) cameras, built ahead on the device side,
    # where the reference builds only those a shading point asks for: same pixels either way
    rng = np.random.default_rng(11)
    body = ["Z 24\n", "l 2 3 12\nl -6 1 10\nl 0 -7 11\n", "f 0.8 0.8 0.7 0.9 0.1 5 0 1\np 4\n-9 -9 0\n9 -9 0\n9 9 0\n-9 9 0\n",
            "f 0.9 0.9 1.0 0.1 0.3 40 0.85 1.4\n"]
    for _ in range(40):
        x, y = rng.uniform(-6, 6, 2)
        body.append("s %.17g %.17g %.17g %.17g\n" % (x, y, rng.uniform(1.2, 4.0), rng.uniform(0.3, 0.7)))
    path = write_scene(tmp_path, "manycams", "".join(body), frm="14 12 16", res="96 80")
    g, ref = both(gpu, path)
    check(g, ref, tol=1e-6)
    assert np.array_equal(g["lightcam_mask"], ref["lightcam_mask"])
    assert g["lightcam_mask"].sum() > 100
    assert ref["counters"]["lightcam_queries"] > 100


def test_odd_sample_count(gpu, tmp_path):
    body = "l 5 5 9\nf 0.2 0.9 0.3 0.8 0.3 20 0 1\np 3\n-2 -2 0\n2 -2 0\n0 2 0\nf 0.9 0.3 0.2 0.7 0.4 30 0 1\ns 0 0 1 0.7\n"
    g, ref = both(gpu, write_scene(tmp_path, "s3", body), num_samples=3, seed=11)
    check(g, ref)
    g, ref = both(gpu, write_scene(tmp_path, "s5", body), num_samples=5, aperture=0.2, seed=3)
    check(g, ref)


def test_malformed_scene_arrays_are_rejected_before_reaching_the_gpu(gpu):
    s = rt.Scene.load("tetra-3")
    good = s.desc
    lib = rt.load()

    def upload_with(patch):
        d = rt.SceneDesc()
        C.memmove(C.byref(d), C.byref(good), C.sizeof(rt.SceneDesc))
        patch(d)
        return lib.rt_upload_scene(gpu._ctx, C.byref(d))

    nodes = (rt.Node * good.n_nodes)()
    C.memmove(nodes, good.nodes, C.sizeof(rt.Node) * good.n_nodes)
    nodes[0].skip = good.n_nodes + 5

    def bad_skip(d):
        d.nodes = C.cast(nodes, C.POINTER(rt.Node))
    assert upload_with(bad_skip) == -1

    nodes2 = (rt.Node * good.n_nodes)()
    C.memmove(nodes2, good.nodes, C.sizeof(rt.Node) * good.n_nodes)
    for n in nodes2:
        if n.tri_count:
            n.tri_begin = good.n_tri_slots
            break

    def bad_range(d):
        d.nodes = C.cast(nodes2, C.POINTER(rt.Node))
    assert upload_with(bad_range) == -1

    def no_materials(d):
        d.n_materials = 0
    assert upload_with(no_materials) == -1
    assert b"scene" in lib.rt_last_error()
    gpu.upload(s)     # the context is still usable


def test_repeated_frames_are_identical(gpu):
    """The same mid-size frame many times over (primary pass handed out in units, paired passes, whole chunks): ray and hit
    counts, the hit map and the image repeat exactly.  A hand-over between waves that can go wrong shows here as a frame
    with a few chunks traced twice or never (tools/soak.py runs the long version of this over eight cases)."""
    import torch
    s = rt.Scene.load("teapot")
    gpu.upload(s)
    cam = s.camera(2048, 2048)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda:0")
    hit = torch.zeros((cam.resy, cam.resx), dtype=torch.int32, device="cuda:0")
    first = None
    for k in range(120):
        hit.fill_(-7)
        torch.cuda.synchronize()
        st = gpu.render_device(cam, rgb_u8_ptr=u8.data_ptr(), hit_id_ptr=hit.data_ptr(), num_samples=2, flags=rt.FLAG_PHONG_NORMALS)
        key = (tuple(st["rays_per_generation"]), tuple(st["hits_per_generation"]), st["shadow_rays"], st["queue_overflow"],
               int((hit == -7).sum()), int(hit.to(torch.int64).sum()), int(u8.to(torch.int64).sum()))
        if first is None:
            first = key
        assert key == first, k
    assert first[3] == 0 and first[4] == 0
    s.close()


def test_8k_frame_properties(gpu, oracle_mod):
    """teapot-3 at 7680x4320, 16 spp (the benchmark frame)."""
    s = rt.Scene.load("teapot-3")
    gpu.upload(s)
    cam = s.camera(7680, 4320)
    full = gpu.render(cam, num_samples=4, want=("rgb_u8",))
    again = gpu.render(cam, num_samples=4, want=("rgb_u8",))
    assert np.array_equal(full["rgb_u8"], again["rgb_u8"])          # idempotent
    assert full["stats"]["closest_rays"] == again["stats"]["closest_rays"]
    # eight row-band parts tile the frame exactly
    frame = np.zeros_like(full["rgb_u8"])
    rays = 0
    for part in range(8):
        rows = rt.part_row_indices(4320, 16, 8, part)
        piece = gpu.render(cam, num_samples=4, band_rows=16, n_parts=8, part=part, want=("rgb_u8",))
        frame[rows] = piece["rgb_u8"]
        rays += piece["stats"]["closest_rays"] + piece["stats"]["shadow_rays"]
    assert np.array_equal(frame, full["rgb_u8"])
    assert rays == full["stats"]["closest_rays"] + full["stats"]["shadow_rays"]
    # four rows through the teapot against the oracle
    o = oracle_mod.OracleScene(rt.scene_path("teapot-3"))
    o.set_resolution(7680, 4320)
    r0 = 2600
    ref = o.render(num_samples=4, threads=min(16, os.cpu_count() or 1), row0=r0, row1=r0 + 4, want_hits=False, want_mask=False)
    o.close()
    assert np.array_equal(full["rgb_u8"][r0:r0 + 4], ref["rgb_u8"][r0:r0 + 4])
    assert len(np.unique(full["rgb_u8"][r0].reshape(-1, 3), axis=0)) > 50   # the rows do cross the teapot
