"""A million-primitive scene through the device builder with several leaf sizes: levels, stack need, frame times."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import raytracer_amd as rt
from test_gpu_device_build import large_scene
side, nsph = int(sys.argv[1]), int(sys.argv[2])
r = rt.Renderer(0)
path = "/tmp/large_%d.nff" % side
open(path, "w").write(large_scene(side, nsph))
s = rt.Scene.load(path)
for leaf in (0, 3, 4):
    r.set_option("leaf_max", leaf)
    r.set_option("device_build_min", 2)
    r.set_option("verbose", 1); r.upload(s); r.set_option("verbose", 0)
    ms = []
    for res in (1024, 4096):
        r.render(s.camera(res, res))
        ms.append(min(r.render(s.camera(res, res))["stats"]["gpu_ms"] for _ in range(3)))
    print("leaf_max %d: frames 1024^2 %.2f ms, 4096^2 %.2f ms" % (leaf, ms[0], ms[1]), flush=True)
