import sys
sys.path.insert(0, "/root/repo")
import numpy as np, raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
g.set_option("direct", 0)
a = g.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT, num_samples=2)
print("direct 0", a["stats"]["rays_per_generation"], a["stats"]["hits_per_generation"], a["stats"]["queue_overflow"])
g.set_option("direct", 1)
try:
    b = g.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT, num_samples=2)
    print("direct 1", b["stats"]["rays_per_generation"], b["stats"]["hits_per_generation"], b["stats"]["queue_overflow"])
    print("same u8", np.array_equal(a["rgb_u8"], b["rgb_u8"]), "max f64 diff", np.abs(a["rgb_f64"] - b["rgb_f64"]).max(), "ids", np.array_equal(a["hit_id"], b["hit_id"]))
except Exception as e:
    print("direct 1 failed:", e)
