"""Upload and frame times of large scenes with the hierarchies built on the device and on the host."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import raytracer_amd as rt
from test_gpu_device_build import large_scene

r = rt.Renderer(0)
for side, nsph in ((200, 60000), (400, 100000), (700, 200000)):
    print("scene of side", side, flush=True)
    path = "/tmp/large_%d.nff" % side
    open(path, "w").write(large_scene(side, nsph))
    t0 = time.perf_counter(); s = rt.Scene.load(path); t_load = time.perf_counter() - t0
    row = []
    for builder, least in (("device", 2), ("host", 0)):
        r.set_option("device_build_min", least)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); r.upload(s); ts.append(time.perf_counter() - t0)
        ms = []
        for res in (1024, 4096):
            r.render(s.camera(res, res))
            ms.append(min(r.render(s.camera(res, res))["stats"]["gpu_ms"] for _ in range(3)))
        row.append((builder, min(ts), ms))
    print("%d triangles + %d spheres (parse + octree %.2f s): " % (2 * side * side, nsph, t_load) +
          "; ".join("%s upload %.3f s, frames 1024^2 %.2f ms 4096^2 %.2f ms" % (b, t, m[0], m[1]) for b, t, m in row), flush=True)
    s.close()
r.close()
