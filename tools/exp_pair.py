"""Paired passes (shadow pass g + closest-hit pass g+1 in one launch) against a launch per pass.
usage: python tools/exp_pair.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
def timed(cam, N, n=5, **opts):
    for k, v in opts.items():
        g.set_option(k, v)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=N)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=N)
    return (time.perf_counter() - t0) / n * 1e3, st, u8.cpu().numpy()
s = rt.Scene.load("teapot-3"); g.upload(s); cam = s.camera(7680, 4320)
for lb in (24, 26, 28):
    g.set_batch(1 << lb)
    a = timed(cam, 4, 3, pair=0); b = timed(cam, 4, 3, pair=1)
    print("8K x 16 spp, batch 2^%d: separate %.2f ms, paired %.2f ms (%d batches); same image %s, same counts %s" % (
        lb, a[0], b[0], b[1]["batches"], np.array_equal(a[2], b[2]), a[1]["closest_rays"] == b[1]["closest_rays"] and a[1]["shadow_rays"] == b[1]["shadow_rays"]), flush=True)
g.set_batch(0)
for part in (2, 4, 8):
    pass
for name, res in (("teapot-3", 512), ("balls", 512), ("refract", 512), ("gears", 512), ("glass", 900), ("balls2", 512), ("teapot-3", 2048), ("balls", 2048)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(res, res)
    a = timed(cam, 1, 20, pair=0, fused=0); b = timed(cam, 1, 20, pair=1, fused=0)
    print("%-9s %4d: separate %.3f ms, paired %.3f ms  %.2fx; same image %s, same counts %s" % (
        name, res, a[0], b[0], a[0] / b[0], np.array_equal(a[2], b[2]), a[1]["closest_rays"] == b[1]["closest_rays"] and a[1]["shadow_rays"] == b[1]["shadow_rays"]), flush=True)
