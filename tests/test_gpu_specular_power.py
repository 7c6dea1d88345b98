"""The power of the specular term, pow(max(rdir.v, 0), Shine) (Solid.h:166-169), as the kernels form it (spec_pow in
csrc/shading.hpp: exp(s log a) with a logarithm and an exponential of their own) against the host's pow().

The bound the kernels' comment states: a relative error of (3 |s log a| + 2) 2^-53 (the logarithm's ulp and the product's
rounding, magnified by the exponential, and the exponential's own ulp).  Where the term can be seen at all (s log a > -36)
that is 1.3e-14 of it; the frames' own bar (1e-9 of 255 on the double frame, the 8-bit image identical) is tested by the parity
tests with every scene."""
import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    r = rt.Renderer(0)
    yield r
    r.close()


def device_pow(gpu, a, s):
    a, s = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(s, dtype=np.float64))
    geom = np.zeros((a.size, 4))
    geom[:, 0], geom[:, 1] = a.ravel(), s.ravel()
    return gpu.debug_intersect(6, geom, np.zeros((a.size, 6)), 0.0, 0.0)[:, 0].reshape(a.shape)


def test_within_the_stated_bound_of_pow(gpu):
    rng = np.random.default_rng(11)
    n = 400000
    # cosines: uniform, crowded towards 1 (the highlight), crowded towards 0, powers of two, subnormals
    a = np.concatenate([rng.random(n), 1.0 - rng.random(n) ** 4 * 0.1, rng.random(n) ** 8, 2.0 ** -rng.integers(0, 1074, n).astype(np.float64),
                        rng.random(n) * 5e-310])
    # exponents: the shipped scenes' (3.0827, 4.81884, 100000), and a spread over twelve decades
    s = np.concatenate([np.full(n, 3.0827), np.full(n, 4.81884), np.full(n, 100000.0), 10.0 ** rng.uniform(-6, 6, n), rng.uniform(0.5, 200, n)])
    got = device_pow(gpu, a, s)
    with np.errstate(all="ignore"):
        want = np.power(a, s)
        y = np.abs(s * np.log(a))
    seen = np.isfinite(want) & (want >= 2.3e-308)
    err = np.abs(got[seen] - want[seen]) / want[seen]
    assert np.all(err <= (3.0 * y[seen] + 2.0) * 2.0 ** -53), (err.max(), np.argmax(err))
    # where the term can be seen in a colour
    vis = seen & (y < 36)
    assert np.max(np.abs(got[vis] - want[vis]) / want[vis]) < 1.3e-14
    # results below the normal range: subnormal or zero as pow() gives them, to within the same bound or a step of the subnormals
    low = ~seen & np.isfinite(want)
    assert np.all(np.abs(got[low] - want[low]) <= (3.0 * y[low] + 2.0) * np.maximum(np.spacing(want[low]), 5e-324))


def test_special_values_follow_pow(gpu):
    inf, nan = np.inf, np.nan
    cases = [(0.0, 3.0), (0.0, 0.0), (0.0, -2.0), (1.0, 5.0), (1.0, inf), (1.0, nan), (1.0, -inf), (0.5, 0.0), (nan, 0.0), (nan, 2.0),
             (0.5, inf), (0.5, -inf), (2.0, inf), (2.0, -inf), (inf, 2.0), (inf, -2.0), (0.5, 1e300), (0.5, -1e300), (-0.0, 2.5), (0.5, nan),
             (1e-320, 0.5), (0.999999999, 1e12), (1.0000000001, 3.0), (0.25, 0.5), (0.5, 1074.0), (0.5, 1080.0), (0.5, -1023.5), (0.5, -1025.0)]
    a = np.array([c[0] for c in cases]); s = np.array([c[1] for c in cases])
    got = device_pow(gpu, a, s)
    with np.errstate(all="ignore"):
        want = np.power(a, s)
    for (x, y), g, w in zip(cases, got, want):
        if np.isnan(w):
            assert np.isnan(g), (x, y, g, w)
        elif np.isinf(w) or w == 0.0:
            assert g == w, (x, y, g, w)
        else:
            yl = 0.0 if (y == 0.0 or x == 1.0) else abs(y * np.log(x))
            assert abs(g - w) <= (3.0 * yl + 2.0) * np.spacing(w), (x, y, g, w)
