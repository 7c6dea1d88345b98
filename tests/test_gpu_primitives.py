"""The device's primitive tests, one (primitive, ray) pair per GPU thread, against the oracle's restatement of
Triangle::intersect (Triangle.h:91-158), Sphere::intersect (Sphere.h:35-77) and the node bounding-sphere test
(Sphere.h:155-204): the returned t must be the same double, bit for bit."""
import numpy as np
import pytest

import raytracer_amd as rt
from oracle import binding as ob

pytestmark = pytest.mark.gpu
MIN_T, MAX_T = 0.01, 1e10
CUT = 5.0          # an upper limit in the middle of the hit distances of the random pairs


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    r.close()


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


def random_rays(rng, n, aim, spread):
    """Origins on a shell of radius ~5, directions towards `aim` perturbed by `spread`, unit length."""
    o = rng.normal(size=(n, 3))
    o = 5.0 * o / np.linalg.norm(o, axis=1, keepdims=True)
    d = aim - o + rng.normal(scale=spread, size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.hstack([o, d])


def test_triangle_tests_match_the_oracle_bit_for_bit(gpu):
    rng = np.random.default_rng(1)
    n = 60000
    tri = rng.uniform(-1, 1, size=(n, 9))
    rays = random_rays(rng, n, tri.reshape(n, 3, 3).mean(axis=1), 0.4)
    ref = ob.test_batch(0, tri, rays, MIN_T, MAX_T)
    assert 0.2 < (ref[:, 0] < MAX_T).mean() < 0.9           # a real mix of hits and misses
    plain = gpu.debug_intersect(0, tri, rays, MIN_T, MAX_T)
    fast = gpu.debug_intersect(1, tri, rays, MIN_T, MAX_T)
    assert same_bits(plain[:, 0], ref[:, 0])
    assert same_bits(fast[:, 0], ref[:, 0])


def borderline_rays(rng, tri, n):
    """Rays aimed at points ON the edges and vertices of their triangles (alpha = 0, beta = 0, alpha + beta = 1 up to
    rounding) and rays whose hit distance is minT or the upper limit up to rounding: where a decision taken without
    the quotient could differ from the reference's."""
    v = tri.reshape(n, 3, 3)
    w = rng.dirichlet([1, 1, 1], size=n)
    which = rng.integers(0, 6, size=n)
    w[which == 0, 0] = 0; w[which == 1, 1] = 0; w[which == 2, 2] = 0     # on an edge
    w[which == 3] = [1, 0, 0]                                              # a vertex
    w /= w.sum(axis=1, keepdims=True)
    target = np.einsum("nk,nkj->nj", w, v)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = rng.uniform(0.5, 9.0, size=n)
    dist[which == 4] = MIN_T                                               # t = minT up to rounding
    dist[which == 5] = CUT                                                 # t = the upper limit used below
    return np.hstack([target - d * dist[:, None], d])


def test_triangle_forms_without_quotients_decide_as_the_oracle(gpu):
    """The any-hit and nearest-hit forms decide from the determinants when that is safe: same decisions, and for the
    nearest-hit form the same t bit for bit, on random pairs and on pairs that sit on the decision boundaries."""
    rng = np.random.default_rng(11)
    n = 200000
    tri = rng.uniform(-1, 1, size=(n, 9))
    sets = {"random": random_rays(rng, n, tri.reshape(n, 3, 3).mean(axis=1), 0.4), "borderline": borderline_rays(rng, tri, n)}
    for name, rays in sets.items():
        for max_t in (MAX_T, CUT):
            ref = ob.test_batch(0, tri, rays, MIN_T, max_t)[:, 0]
            hit = ref < max_t
            assert 0.05 < hit.mean() < 0.95, (name, hit.mean())
            fast = gpu.debug_intersect(1, tri, rays, MIN_T, max_t)[:, 0]
            assert same_bits(fast, ref), (name, max_t)
            anyhit = gpu.debug_intersect(4, tri, rays, MIN_T, max_t)[:, 0]
            assert np.array_equal(anyhit == 0.0, hit) and np.array_equal(anyhit == max_t, ~hit), (name, max_t)
        # nearest-hit form with a cut-off: exact up to the cut-off, beyond it either the exact t or "nothing"
        full = ob.test_batch(0, tri, rays, MIN_T, MAX_T)[:, 0]
        cut = gpu.debug_intersect(5, tri, rays, MIN_T, CUT)[:, 0]
        near = full <= CUT
        assert same_bits(cut[near], full[near]), name
        assert np.all((cut[~near] == full[~near]) | (cut[~near] == MAX_T)), name


def test_triangle_edge_cases(gpu):
    tri = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]] * 8, dtype=np.float64)
    rays = np.array([
        [0, 0, 1, 1, 0, 0],            # parallel to the plane: detA = 0, NaN ratios, a miss
        [.25, .25, 2, 0, 0, -1],       # straight down onto the interior
        [.25, .25, .005, 0, 0, -1],    # nearer than minT
        [0, 0, 2, 0, 0, -1],           # through vertex v0 (alpha = beta = 0)
        [.5, .5, 2, 0, 0, -1],         # on the edge alpha + beta = 1
        [1, 0, 2, 0, 0, -1],           # through vertex v1
        [.25, .25, -2, 0, 0, 1],       # from behind: the test is two-sided
        [.25, .25, 2, 0, 0, -3],       # direction not of unit length
    ], dtype=np.float64)
    tri[7] *= 1e-3                      # a tiny triangle, scaled ray target
    rays[7, :2] *= 1e-3
    ref = ob.test_batch(0, tri, rays, MIN_T, MAX_T)
    for kind in (0, 1, 5):
        got = gpu.debug_intersect(kind, tri, rays, MIN_T, MAX_T)
        assert same_bits(got[:, 0], ref[:, 0]), (kind, got[:, 0], ref[:, 0])
    assert np.array_equal(gpu.debug_intersect(4, tri, rays, MIN_T, MAX_T)[:, 0] == 0.0, ref[:, 0] < MAX_T)
    assert ref[0, 0] == MAX_T and ref[1, 0] == 2.0 and ref[2, 0] == MAX_T and ref[6, 0] == 2.0
    # degenerate triangles (zero area, repeated vertices) never hit and never trap
    deg = np.array([[0, 0, 0, 1, 1, 1, 2, 2, 2], [1, 2, 3, 1, 2, 3, 1, 2, 3]], dtype=np.float64)
    drays = np.array([[0, 0, 5, 0.1, 0.2, -1], [0, 0, 5, 0.2, 0.4, -0.4]], dtype=np.float64)
    ref = ob.test_batch(0, deg, drays, MIN_T, MAX_T)
    for kind in (0, 1, 5):
        assert same_bits(gpu.debug_intersect(kind, deg, drays, MIN_T, MAX_T)[:, 0], ref[:, 0])
    # (the all-equal triangle yields t = NaN in the reference; a shadow query asks "t < maxT", which NaN is not)
    assert np.array_equal(gpu.debug_intersect(4, deg, drays, MIN_T, MAX_T)[:, 0] == 0.0, ref[:, 0] < MAX_T)


def test_sphere_tests_match_the_oracle_bit_for_bit(gpu):
    rng = np.random.default_rng(2)
    n = 60000
    sph = np.hstack([rng.uniform(-1, 1, size=(n, 3)), rng.uniform(0.005, 1.5, size=(n, 1))])
    rays = random_rays(rng, n, sph[:, :3], 0.6)
    # a share of origins inside their sphere (the far root is taken there, Sphere.h:58-62)
    inside = rng.random(n) < 0.15
    rays[inside, :3] = sph[inside, :3] + 0.3 * sph[inside, 3:4] * rng.uniform(-1, 1, size=(int(inside.sum()), 3))
    ref = ob.test_batch(2, sph, rays, MIN_T, MAX_T)
    got = gpu.debug_intersect(2, sph, rays, MIN_T, MAX_T)
    assert 0.2 < (ref[:, 0] < MAX_T).mean() < 0.95
    assert same_bits(got[:, 0], ref[:, 0])
    assert same_bits(got[:, 1:], ref[:, 1:])               # the hit record's normal as well


def test_node_bound_tests_match_the_oracle_bit_for_bit(gpu):
    rng = np.random.default_rng(3)
    n = 60000
    node = np.hstack([rng.uniform(-50, 50, size=(n, 3)), rng.uniform(0.01, 90, size=(n, 1))])
    o = rng.uniform(-60, 60, size=(n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.hstack([o, d])
    for sentinel in (1e10, 1e7):                            # closest-hit and shadow queries use different ones
        ref = ob.test_batch(3, node, rays, MIN_T, sentinel)
        got = gpu.debug_intersect(3, node, rays, MIN_T, sentinel)
        assert 0.1 < (ref[:, 0] < sentinel).mean() < 0.95
        assert same_bits(got[:, 0], ref[:, 0])
