"""Randomised scenes: the device against the CPU oracle on geometry nobody tuned for.

Every seed writes a small .nff file (meshes with shared edges, loose triangles, polygons that need ear
clipping, patches with vertex normals, opaque and transparent materials, spheres, one to five lights, a
random camera), renders it on both sides and asks for the same primitive ids, the same ray counts and the
same image.  The scenes are meant to hit what the shipped ones reach rarely: exact-t ties on shared edges,
primitives that straddle many octree cells, transparent triangles in front of spheres, lights inside the
geometry, shading points in the shadow of transparent spheres.
"""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    r.close()


def fmt(v):
    return " ".join("%.17g" % x for x in v)


def material(rng, transparent):
    col = rng.uniform(0.1, 1.0, 3)
    kd = rng.uniform(0.2, 1.0)
    ks = rng.choice([0.0, rng.uniform(0.1, 0.9)])
    shine = rng.choice([1.0, 3.0845, 20.0, 77.5])
    t = rng.uniform(0.3, 0.9) if transparent else 0.0
    ior = rng.uniform(1.05, 1.8)
    return "f %s %.17g %.17g %.17g %.17g %.17g\n" % (fmt(col), kd, ks, shine, t, ior)


def height_field(rng, out):
    """n x n grid of quads, each split in two triangles: every interior edge is shared by two primitives."""
    n = int(rng.integers(3, 7))
    origin = rng.uniform(-3, 3, 3)
    size = rng.uniform(1.5, 5.0)
    z = rng.uniform(-0.4, 0.4, (n + 1, n + 1))
    if rng.random() < 0.3:
        z[:] = 0.0                                   # exactly coplanar cells
    def vert(i, j):
        return origin + np.array([size * i / n, size * j / n, z[i, j]])
    for i in range(n):
        for j in range(n):
            a, b, c, d = vert(i, j), vert(i + 1, j), vert(i + 1, j + 1), vert(i, j + 1)
            out.append("p 3\n%s\n%s\n%s\n" % (fmt(a), fmt(b), fmt(c)))
            out.append("p 3\n%s\n%s\n%s\n" % (fmt(a), fmt(c), fmt(d)))


def loose_triangles(rng, out):
    for _ in range(int(rng.integers(3, 30))):
        c = rng.uniform(-4, 4, 3)
        s = rng.choice([0.05, 0.5, 3.0])
        out.append("p 3\n%s\n%s\n%s\n" % tuple(fmt(c + rng.normal(0, s, 3)) for _ in range(3)))


def polygons(rng, out):
    """Planar polygons with 4..7 vertices, convex or star-shaped: ear clipping decides the triangles."""
    for _ in range(int(rng.integers(1, 6))):
        k = int(rng.integers(4, 8))
        c = rng.uniform(-3, 3, 3)
        u = rng.normal(0, 1, 3)
        v = rng.normal(0, 1, 3)
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        if rng.random() < 0.5:
            ang = ang[::-1]
        rad = rng.uniform(0.5, 2.0, k) if rng.random() < 0.5 else np.full(k, rng.uniform(0.5, 2.0))
        pts = [c + r * (np.cos(a) * u + np.sin(a) * v) for a, r in zip(ang, rad)]
        out.append("p %d\n%s\n" % (k, "\n".join(fmt(p) for p in pts)))


def gear_faces(rng, out):
    """Many-cornered toothed outlines in an oblique plane: ear clipping turns each into a fan of long slivers,
    the case the hierarchy builder cuts into pieces."""
    for _ in range(int(rng.integers(1, 4))):
        k = 2 * int(rng.integers(6, 15))
        c = rng.uniform(-2.5, 2.5, 3)
        u = rng.normal(0, 1, 3)
        u /= np.linalg.norm(u)
        v = np.cross(u, rng.normal(0, 1, 3))
        v /= np.linalg.norm(v)
        r_out = rng.uniform(1.0, 3.0)
        r_in = r_out * rng.uniform(0.7, 0.95)
        ang = np.arange(k) * (2 * np.pi / k)
        if rng.random() < 0.5:
            ang = ang[::-1]
        pts = [c + (r_out if i % 2 else r_in) * (np.cos(a) * u + np.sin(a) * v) for i, a in enumerate(ang)]
        out.append("p %d\n%s\n" % (k, "\n".join(fmt(p) for p in pts)))


def patches(rng, out):
    for _ in range(int(rng.integers(1, 8))):
        c = rng.uniform(-3, 3, 3)
        rows = []
        for _ in range(3):
            p = c + rng.normal(0, 1.0, 3)
            nrm = rng.normal(0, 1, 3)
            rows.append(fmt(p) + " " + fmt(nrm))
        out.append("pp 3\n%s\n" % "\n".join(rows))


def spheres(rng, out, count):
    for _ in range(count):
        out.append("s %s %.17g\n" % (fmt(rng.uniform(-3.5, 3.5, 3)), rng.choice([0.05, 0.4, 1.2]) * rng.uniform(0.5, 1.5)))


def random_scene(seed, transparent_spheres, lightcam_samples=None):
    rng = np.random.default_rng(seed)
    frm = rng.normal(0, 1, 3)
    frm = frm / np.linalg.norm(frm) * rng.uniform(6, 14)
    if transparent_spheres:
        frm[2] = abs(frm[2]) + 2.0                     # above the floor, looking down at the caustics
    resx, resy = int(rng.integers(40, 97)), int(rng.integers(40, 97))
    head = ("b %s\nv\nfrom %s\nat %s\nup 0 0 1\nangle %.17g\nhither 1\nresolution %d %d\n"
            % (fmt(rng.uniform(0, 1, 3)), fmt(frm), fmt(rng.uniform(-1, 1, 3)), rng.uniform(25, 70), resx, resy))
    out = [head]
    if lightcam_samples is not None:
        out.append("Z %d\n" % lightcam_samples)
    lights = []
    for _ in range(int(rng.integers(1, 6))):
        where = rng.uniform(-10, 10, 3) if rng.random() < 0.8 else rng.uniform(-1, 1, 3)   # some lights inside the geometry
        if transparent_spheres and not lights:
            where[2] = rng.uniform(4, 10)              # the first one above the floor, for the caustics
        lights.append(where)
        out.append("l %s\n" % fmt(where))
    makers = [height_field, loose_triangles, polygons, patches, gear_faces]
    for _ in range(int(rng.integers(2, 7))):
        out.append(material(rng, transparent=rng.random() < 0.25))
        makers[int(rng.integers(0, len(makers)))](rng, out)
    out.append(material(rng, transparent=False))
    spheres(rng, out, int(rng.integers(0, 12)))
    if transparent_spheres:
        out.append(material(rng, transparent=True))
        for _ in range(int(rng.integers(1, 4))):     # on the way from the first light to a point of the floor
            foot = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), -4.5])
            c = lights[0] + rng.uniform(0.45, 0.8) * (foot - lights[0])
            out.append("s %s %.17g\n" % (fmt(c), rng.uniform(0.3, 1.0)))
    # a floor to catch shadows: always under transparent spheres, otherwise sometimes
    if transparent_spheres or rng.random() < 0.6:
        out.append(material(rng, transparent=False))
        out.append("p 4\n-8 -8 -4.5\n8 -8 -4.5\n8 8 -4.5\n-8 8 -4.5\n")
    return "".join(out)


def render_both(gpu, path, gpu_flags, oracle_flags, phong=False, **kw):
    s = rt.Scene.load(path)
    gpu.upload(s)
    g = gpu.render(s.camera(), flags=gpu_flags | (rt.FLAG_PHONG_NORMALS if phong else 0), **kw)
    o = ob.OracleScene(path, phong=phong)
    ref = o.render(threads=8, flags=oracle_flags, num_samples=kw.get("num_samples", 1), aperture=kw.get("aperture", 0.0),
                   seed=kw.get("seed", 0))
    o.close()
    s.close()
    return g, ref


def compare(g, ref, tol):
    mism = int((g["hit_id"] != ref["hit_id"]).sum())
    assert mism == 0, "%d primary hit ids differ" % mism
    assert g["stats"]["closest_rays"] == ref["counters"]["closest_rays"]
    assert g["stats"]["shadow_rays"] == ref["counters"]["shadow_rays"]
    assert g["stats"]["queue_overflow"] == 0
    d = np.abs(g["rgb_f64"] - ref["rgb_f64"]).max()
    assert d <= tol, "max |f64 difference| = %g" % d
    assert np.array_equal(g["rgb_u8"], ref["rgb_u8"])


@pytest.mark.parametrize("seed", range(40))
def test_opaque_spheres_and_triangle_soup(gpu, tmp_path, seed):
    p = tmp_path / ("soup%d.nff" % seed)
    p.write_text(random_scene(1000 + seed, transparent_spheres=False))
    g, ref = render_both(gpu, str(p), rt.FLAG_KEEP_ZERO_WEIGHT, 0, phong=bool(seed & 1))
    compare(g, ref, 1e-9)


@pytest.mark.parametrize("seed", range(6))
def test_supersampled_random_scene(gpu, tmp_path, seed):
    p = tmp_path / ("ss%d.nff" % seed)
    p.write_text(random_scene(2000 + seed, transparent_spheres=False))
    g, ref = render_both(gpu, str(p), rt.FLAG_KEEP_ZERO_WEIGHT, 0, num_samples=2 + seed % 3, aperture=0.1 * seed, seed=seed + 5)
    compare(g, ref, 1e-9)


@pytest.mark.parametrize("seed", range(16))
def test_transparent_spheres_shadow_order(gpu, tmp_path, seed):
    """With transparent spheres the answer of a shadow query depends on the order the reference visits cells in."""
    p = tmp_path / ("glassy%d.nff" % seed)
    p.write_text(random_scene(3000 + seed, transparent_spheres=True))
    g, ref = render_both(gpu, str(p), rt.FLAG_KEEP_ZERO_WEIGHT | rt.FLAG_NO_LIGHTCAM, ob.FLAG_NO_LIGHTCAM)
    compare(g, ref, 1e-9)
    assert np.array_equal(g["lightcam_mask"], ref["lightcam_mask"])


@pytest.mark.parametrize("seed", range(10))
def test_light_cameras_on_random_scenes(gpu, tmp_path, seed):
    p = tmp_path / ("caustic%d.nff" % seed)
    p.write_text(random_scene(4000 + seed, transparent_spheres=True, lightcam_samples=40 + 7 * seed))
    g, ref = render_both(gpu, str(p), rt.FLAG_KEEP_ZERO_WEIGHT, 0)
    compare(g, ref, 1e-6)
    assert np.array_equal(g["lightcam_mask"], ref["lightcam_mask"])
