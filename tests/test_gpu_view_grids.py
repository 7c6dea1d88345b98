"""The grids of what the eye and the lights can see (csrc/viewgrid.hpp, grid_query.hpp) against the CPU oracle.

A primary ray reads the list of its 8 x 8-pixel tile and a shadow ray the list of its direction's bin on the cube map
around its light, instead of walking the hierarchies; the lists must hold every primitive such a ray can meet.  What
is compared is what the walks are compared on: primary hit ids, ray counts, the 8-bit image (identical) and the double
frame (1e-9 of 255).  The scenes aim at what a conservative rasteriser gets wrong first: eyes and lights inside or next
to the geometry, primitives that pass behind the eye or through a cube map's edges and corners, slivers, spheres around
the eye or a light, coarse and fine grids, frames cut into row bands that do not line up with the tiles.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

from test_gpu_random_scenes import compare, fmt, gear_faces, height_field, loose_triangles, material, patches, polygons, spheres

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    for k, v in (("grid", 1), ("grid_res", 0), ("grid_min_samples", -1), ("local", -1), ("pair", -1)):
        r.set_option(k, v)
    r.close()


def close_quarters_scene(seed):
    """Geometry all around an eye that sits in the middle of it, wide angle, lights between the primitives."""
    rng = np.random.default_rng(seed)
    frm = rng.uniform(-1.5, 1.5, 3)
    at = frm + rng.normal(0, 1, 3)
    resx, resy = int(rng.integers(40, 120)), int(rng.integers(40, 120))
    out = ["b %s\nv\nfrom %s\nat %s\nup 0 0 1\nangle %.17g\nhither 1\nresolution %d %d\n"
           % (fmt(rng.uniform(0, 1, 3)), fmt(frm), fmt(at), rng.uniform(40, 130), resx, resy)]
    for _ in range(int(rng.integers(1, 5))):
        out.append("l %s\n" % fmt(rng.uniform(-3, 3, 3)))
    makers = [height_field, loose_triangles, polygons, patches, gear_faces]
    for _ in range(int(rng.integers(3, 8))):
        out.append(material(rng, transparent=rng.random() < 0.2))
        makers[int(rng.integers(0, len(makers)))](rng, out)
    out.append(material(rng, transparent=False))
    spheres(rng, out, int(rng.integers(2, 14)))
    if rng.random() < 0.5:      # a sphere around the eye, one around the first light
        out.append("s %s %.17g\n" % (fmt(frm + rng.normal(0, 0.1, 3)), rng.uniform(0.5, 6.0)))
    if rng.random() < 0.5:
        out.append(material(rng, transparent=False))
        out.append("p 4\n-30 -30 -2\n30 -30 -2\n30 30 -2\n-30 30 -2\n")      # passes under (and behind) the eye
    return "".join(out)


def render_modes(gpu, path, oracle_kw=None, **kw):
    """The frame with the grids forced on (eye and lights), in the launch-per-pass form and in self-contained blocks, and
    with no grids; and the oracle's."""
    s = rt.Scene.load(path)
    frames = {}
    for name, opts in (("grids, passes", {"grid": 1, "grid_min_samples": 0, "local": 0}),
                       ("grids, blocks", {"grid": 1, "grid_min_samples": 0, "local": 1}),
                       ("no grids", {"grid": 0, "local": 0})):
        for k, v in opts.items():
            gpu.set_option(k, v)
        gpu.upload(s)
        frames[name] = gpu.render(s.camera(), flags=rt.FLAG_KEEP_ZERO_WEIGHT, **kw)
    o = ob.OracleScene(path)
    ref = o.render(threads=8, num_samples=kw.get("num_samples", 1), aperture=kw.get("aperture", 0.0), seed=kw.get("seed", 0), **(oracle_kw or {}))
    o.close()
    s.close()
    return frames, ref


@pytest.mark.parametrize("seed", range(24))
def test_eye_and_lights_in_the_middle_of_the_geometry(gpu, tmp_path, seed):
    p = tmp_path / ("close%d.nff" % seed)
    p.write_text(close_quarters_scene(7000 + seed))
    gpu.set_option("grid_res", [0, 8, 32, 100][seed % 4])
    frames, ref = render_modes(gpu, str(p), num_samples=1 + (seed % 3 == 2), seed=seed)
    for name, g in frames.items():
        compare(g, ref, 1e-9)


def test_lists_hold_what_passes_behind_the_eye_and_through_cube_edges(gpu, tmp_path):
    """Hand-made: a floor that extends far behind the eye, a triangle through the eye's own plane, a sliver along a cube
    map's edge and one through its corner (as seen from the light), a light 0.02 from a wall, spheres around eye and light."""
    body = ("l 0.5 0.5 3\nl -2.98 0 1\nl 0 0 0.3\n"
            "f 0.8 0.8 0.7 0.9 0.1 5 0 1\np 4\n-60 -60 -1\n60 -60 -1\n60 60 -1\n-60 60 -1\n"
            "f 0.3 0.8 0.4 0.7 0.3 12 0 1\np 3\n-3 -4 -1\n-3 4 -1\n-3 0 6\n"                      # wall next to the second light
            "f 0.9 0.4 0.3 0.6 0.4 20 0 1\np 3\n5.9 4.9 4\n6.1 5.1 4\n-2 -2 -0.5\n"               # passes the eye closely
            "p 3\n0.5 0.5 0\n3.5 3.5 0\n3.5 3.5001 0.0001\n"                                      # sliver along x = y under light 1
            "p 3\n0.5 0.5 -0.5\n2.5 2.5 -2.5\n2.5001 2.5 -2.5\n"                                  # towards a cube corner of light 1
            "f 0.5 0.5 0.9 0.5 0.5 30 0 1\ns 6 5 4 0.8\ns 0.5 0.5 3 0.5\ns 0 0 0.3 2.5\ns 1 -1 0 0.6\n")
    head = "b 0.1 0.2 0.3\nv\nfrom 6 5 4\nat 0 0 0\nup 0 0 1\nangle 70\nhither 1\nresolution 96 72\n"
    p = tmp_path / "edges.nff"
    p.write_text(head + body)
    for res in (0, 6, 31):
        gpu.set_option("grid_res", res)
        frames, ref = render_modes(gpu, str(p))
        for g in frames.values():
            compare(g, ref, 1e-9)
    gpu.set_option("grid_res", 0)


@pytest.mark.parametrize("scene,res", [("teapot-3", None), ("balls", (300, 300)), ("refract", (320, 200)), ("gears", (200, 200)),
                                       ("tetra", (256, 256)), ("balls2", (160, 160))])
def test_shipped_scenes_through_the_grids(gpu, oracle_frame, scene, res):
    s = rt.Scene.load(scene)
    o = oracle_frame(scene, res=res)
    for opts in ({"grid": 1, "grid_min_samples": 0, "local": 0}, {"grid": 1, "grid_min_samples": 0, "local": 1}, {"grid": 3, "grid_min_samples": 0, "local": 0}):
        for k, v in opts.items():
            gpu.set_option(k, v)
        gpu.upload(s)
        g = gpu.render(s.camera(*(res or (0, 0))), flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        compare(g, o, 1e-9)
    s.close()


def test_row_bands_supersampling_and_a_moving_eye(gpu, oracle_frame):
    """Parts of a frame (bands of 16 rows, and of 5, which cut through the 8 x 8 tiles) read the grid of the whole frame;
    super-sampled pixels stay inside their tile; a lens with an aperture has no single eye and takes the walk; a second
    camera gets a grid of its own and the first one's is made again when it comes back."""
    gpu.set_option("grid", 1)
    gpu.set_option("grid_min_samples", 0)
    gpu.set_option("local", 0)
    s = rt.Scene.load("teapot-3")
    gpu.upload(s)
    res = (200, 150)
    o = oracle_frame("teapot-3", res=res)
    cam = s.camera(*res)
    for band, parts in ((16, 3), (5, 4)):
        frame = np.zeros_like(o["rgb_u8"])
        hits = np.zeros_like(o["hit_id"])
        for part in range(parts):
            rows = rt.part_row_indices(res[1], band, parts, part)
            piece = gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT, band_rows=band, n_parts=parts, part=part)
            frame[rows] = piece["rgb_u8"]
            hits[rows] = piece["hit_id"]
        assert np.array_equal(frame, o["rgb_u8"]) and np.array_equal(hits, o["hit_id"])
    for n, ap in ((2, 0.0), (3, 0.0), (2, 0.25)):
        g = gpu.render(cam, num_samples=n, aperture=ap, seed=9, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        compare(g, oracle_frame("teapot-3", res=res, num_samples=n, aperture=ap, seed=9), 1e-9)
    other = s.camera(120, 160)
    compare(gpu.render(other, flags=rt.FLAG_KEEP_ZERO_WEIGHT), oracle_frame("teapot-3", res=(120, 160)), 1e-9)
    compare(gpu.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT), o, 1e-9)
    s.close()
