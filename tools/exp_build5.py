"""Mid-size frames of a large scene with a deep hierarchy: one launch of self-contained blocks against a launch per pass."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import raytracer_amd as rt
from test_gpu_device_build import large_scene
side, nsph = int(sys.argv[1]), int(sys.argv[2])
r = rt.Renderer(0)
path = "/tmp/large_%d.nff" % side
open(path, "w").write(large_scene(side, nsph))
s = rt.Scene.load(path)
for least in (2, 0):
    r.set_option("device_build_min", least)
    r.upload(s)
    for res in (512, 1024, 2048):
        row = []
        for opts in ({"local": -1}, {"local": 0}, {"local": 0, "block_threads": 1024}, {"local": 1, "local_threads": 512}):
            for k in ("local", "block_threads"): r.set_option(k, -1 if k == "local" else 0)
            r.set_option("local_threads", 1024)
            for k, v in opts.items(): r.set_option(k, v)
            r.render(s.camera(res, res))
            row.append("%s %.2f" % (opts, min(r.render(s.camera(res, res))["stats"]["gpu_ms"] for _ in range(3))))
        print("device_build_min %d, %d^2: " % (least, res) + " | ".join(row), flush=True)
