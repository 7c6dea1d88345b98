"""Repeats small frames through k_local and compares every run with the launch-per-pass frame: a race would show as
differing ray counts or pixels.  usage: python tools/exp_local_repeat.py [runs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = rt.Renderer(0)
for name, res, n in (("gears", (192, 160), 2), ("refract", (256, 256), 1), ("balls", (320, 320), 1), ("teapot-3", (512, 512), 1)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(*res)
    g.set_option("local", 0)
    ref = g.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
    g.set_option("local", 1)
    worst, bad_counts, bad_u8 = 0.0, 0, 0
    for threads in (1024, 256, 512):
        g.set_option("local_threads", threads)
        for _ in range(runs):
            f = g.render(cam, num_samples=n, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
            worst = max(worst, float(np.abs(f["rgb_f64"] - ref["rgb_f64"]).max()))
            bad_counts += f["stats"]["rays_per_generation"] != ref["stats"]["rays_per_generation"] or f["stats"]["shadow_rays"] != ref["stats"]["shadow_rays"]
            bad_u8 += int((f["rgb_u8"] != ref["rgb_u8"]).sum())
    g.set_option("local", -1); g.set_option("local_threads", 1024)
    print("%-9s %s x%d: %d runs, max |f64 difference| %.3g, runs with other ray counts %d, u8 values that differ %d" % (
        name, res, n, 3 * runs, worst, bad_counts, bad_u8), flush=True)
