"""One large scene, verbose: what the device builder made of it, and whether the frames agree."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import raytracer_amd as rt
from test_gpu_device_build import large_scene

side, nsph = int(sys.argv[1]), int(sys.argv[2])
r = rt.Renderer(0)
r.set_option("verbose", 1)
path = "/tmp/large_%d.nff" % side
open(path, "w").write(large_scene(side, nsph))
s = rt.Scene.load(path)
fr = {}
for builder, least in (("device", 2), ("host", 0)):
    r.set_option("device_build_min", least)
    for grid in (1, 0):
        r.set_option("grid", grid)
        r.upload(s)
        r.render(s.camera(512, 512))
        g = r.render(s.camera(512, 512))
        fr[builder, grid] = g
        print(builder, "grid", grid, "frame %.2f ms" % g["stats"]["gpu_ms"], {k: g["stats"][k] for k in ("closest_rays", "shadow_rays", "primary_ms", "closest_ms", "shadow_ms")}, flush=True)
a = fr["device", 1]
for k, b in fr.items():
    print(k, "same ids", np.array_equal(a["hit_id"], b["hit_id"]), "same image", np.array_equal(a["rgb_u8"], b["rgb_u8"]), flush=True)
