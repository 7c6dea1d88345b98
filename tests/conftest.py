import json
import os
import sys

import pytest

try:
    # PyTorch ships its own HIP runtime; when a test uses both, torch has to be the first to load one
    # (a process that initialised the system runtime through libraytracer_amd.so first leaves torch
    # without a visible GPU).
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import binding
    binding.load()
    return binding


_ORACLE_CACHE = {}


@pytest.fixture(scope="session")
def oracle_frame(oracle_mod):
    """Cached oracle renders keyed by (scene, phong, flags, resx, resy, N, aperture, seed)."""
    import raytracer_amd as rt

    def get(scene, phong=False, flags=0, res=None, num_samples=1, aperture=0.0, seed=0, lightcam_samples=None):
        key = (scene, phong, flags, res, num_samples, aperture, seed, lightcam_samples)
        if key not in _ORACLE_CACHE:
            o = oracle_mod.OracleScene(rt.scene_path(scene), phong=phong)
            if res is not None:
                o.set_resolution(*res)
            if lightcam_samples is not None:
                o.set_lightcam_samples(lightcam_samples)
            threads = min(16, os.cpu_count() or 1)
            _ORACLE_CACHE[key] = o.render(num_samples=num_samples, aperture=aperture, threads=threads, seed=seed,
                                          flags=flags)
            _ORACLE_CACHE[key]["lightcams"] = o.lightcams()
            o.close()
        return _ORACLE_CACHE[key]

    return get
