"""balls.nff with one glass sphere added: the shadow pass with the per-ray choice between the hierarchies and the
exact octree walk, against the exact walk for every ray (RT_FLAG_NO_CULL).  Light cameras off: only the walks are compared.
usage: python tools/exp_hybrid.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # before the library initialises the system HIP runtime
import numpy as np
import raytracer_amd as rt

text = open(rt.scene_path("balls")).read() + "\nf 0.9 0.9 1.0 0.1 0.3 40 0.85 1.4\ns 0.4 0.3 0.9 0.25\n"
path = "/tmp/balls_glass.nff"
open(path, "w").write(text)
s = rt.Scene.load(path)
g = rt.Renderer(0)
g.upload(s)
cam = s.camera(2048, 2048)
frames = {}
for label, flags in (("per-ray choice", rt.FLAG_NO_LIGHTCAM), ("exact walk only", rt.FLAG_NO_LIGHTCAM | rt.FLAG_NO_CULL)):
    for it in range(3):
        fr = g.render(cam, flags=flags | (rt.FLAG_TIMING if it == 2 else 0), want=("rgb_u8",))
    st = fr["stats"]
    frames[label] = fr["rgb_u8"]
    print("%-16s frame %.2f ms | primary %.2f queued %.2f shadow %.2f ms | %d shadow rays" % (
        label, st["gpu_ms"], st["primary_ms"], st["closest_ms"] - st["primary_ms"], st["shadow_ms"], st["shadow_rays"]))
print("images equal:", bool(np.array_equal(frames["per-ray choice"], frames["exact walk only"])))
