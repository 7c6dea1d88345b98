"""Where the upload of a large terrain goes (verbose marks), device against host builder, frames compared."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import raytracer_amd as rt
from test_gpu_device_build import large_scene

side, nsph = int(sys.argv[1]), int(sys.argv[2])
r = rt.Renderer(0)
path = "/tmp/large_%d.nff" % side
open(path, "w").write(large_scene(side, nsph))
t0 = time.perf_counter(); s = rt.Scene.load(path); print("parse + octree %.3f s" % (time.perf_counter() - t0), flush=True)
for builder, least in (("device", 2), ("host", 0)):
    r.set_option("device_build_min", least)
    r.set_option("verbose", 0)
    r.upload(s)
    t0 = time.perf_counter(); r.upload(s); t1 = time.perf_counter() - t0
    r.set_option("verbose", 1)
    r.upload(s)
    r.set_option("verbose", 0)
    ms = []
    for res in (1024, 4096):
        r.render(s.camera(res, res))
        ms.append(min(r.render(s.camera(res, res))["stats"]["gpu_ms"] for _ in range(3)))
    print("%s: upload %.1f ms; frames 1024^2 %.2f ms, 4096^2 %.2f ms" % (builder, 1e3 * t1, ms[0], ms[1]), flush=True)
