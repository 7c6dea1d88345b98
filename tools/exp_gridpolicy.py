"""Dense scenes: frame times with no grid, the lights' grids, the eye's grid, both, at several frame sizes."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import raytracer_amd as rt
from test_gpu_device_build import large_scene

r = rt.Renderer(0)
r.set_option("verbose", 1)
r.set_option("device_build_min", 0)
for side, nsph in ((100, 20000), (400, 100000)):
    path = "/tmp/large_%d.nff" % side
    open(path, "w").write(large_scene(side, nsph))
    s = rt.Scene.load(path)
    for res in (512, 1024, 2048, 4096):
        row = []
        for grid in (0, 2, 3, 1):
            r.set_option("grid", grid)
            r.set_option("grid_min_samples", 0)
            r.upload(s)
            r.render(s.camera(res, res))
            g = min((r.render(s.camera(res, res)) for _ in range(3)), key=lambda g: g["stats"]["gpu_ms"])
            row.append("grid %d: %.2f ms" % (grid, g["stats"]["gpu_ms"]))
        print("side %d res %d: " % (side, res) + "; ".join(row), flush=True)
    s.close()
