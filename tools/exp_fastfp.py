"""A build with FMA contraction allowed (tools/build_variant.sh fastfp -ffp-contract=fast) against the bar the exact
build is far inside of: primary hit ids identical except ties, >= 99.9 % of pixels within 1/255 per channel.
usage: RAYTRACER_AMD_LIB=build_variants/lib_fastfp.so python tools/exp_fastfp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from oracle import binding as ob
scenes = ["2d", "ball", "ball2", "balls", "balls2", "gears", "glass", "glass2", "glass3", "glass4", "refract", "teapot", "teapot-1",
          "teapot-2", "teapot-3", "tetra", "tetra-3", "tetra-5"]
small = {"gears": (192, 192), "balls2": (256, 256), "teapot": (256, 256), "tetra": (300, 300), "glass": (300, 300), "glass4": (256, 256),
         "ball2": (300, 300)}
g = rt.Renderer(0)
print("library", rt.LIB_PATH, "build", rt.build_id())
worst = 1.0
for name in scenes:
    s = rt.Scene.load(name); g.upload(s)
    res = small.get(name)
    cam = s.camera(*res) if res else s.camera()
    fr = g.render(cam)
    o = ob.OracleScene(rt.scene_path(name))
    if res: o.set_resolution(*res)
    ref = o.render(threads=min(16, os.cpu_count() or 1)); o.close()
    within = (np.abs(fr["rgb_u8"].astype(int) - ref["rgb_u8"].astype(int)).max(axis=2) <= 1).mean()
    hits = int((fr["hit_id"] != ref["hit_id"]).sum())
    worst = min(worst, within)
    print("%-9s %4dx%-4d  pixels within 1/255: %8.4f %%  bytes differing: %6d  primary hit ids differing: %d  max |f64 diff| %.3g" % (
        name, cam.resx, cam.resy, 100 * within, int((fr["rgb_u8"] != ref["rgb_u8"]).sum()), hits, np.abs(fr["rgb_f64"] - ref["rgb_f64"]).max()))
    s.close()
print("worst scene: %.4f %% within 1/255 (bar: 99.9 %%)" % (100 * worst))
