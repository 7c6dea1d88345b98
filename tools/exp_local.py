"""Small frames as one launch of self-contained blocks (k_local) against a launch per pass.
usage: python tools/exp_local.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
def timed(cam, n=30, **opts):
    for k, v in opts.items():
        g.set_option(k, v)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    return (time.perf_counter() - t0) / n * 1e3, st, u8.cpu().numpy()
for name, res in (("teapot-3", 512), ("balls", 512), ("refract", 512), ("gears", 512), ("balls2", 512), ("teapot", 512),
                  ("teapot-3", 1024), ("balls", 1024), ("teapot-3", 2048), ("balls", 2048)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(res, res)
    a = timed(cam, local=0, fused=0)
    out = []
    for threads in (1024, 512, 256):
        b = timed(cam, local=1, fused=0, local_threads=threads)
        ok = np.array_equal(a[2], b[2]) and a[1]["rays_per_generation"] == b[1]["rays_per_generation"] and a[1]["shadow_rays"] == b[1]["shadow_rays"]
        out.append("%d threads %.3f ms (%.2fx)%s" % (threads, b[0], a[0] / b[0], "" if ok else " DIFFERENT"))
    print("%-9s %4d: launch per pass %.3f ms, self-contained blocks of %s" % (name, res, a[0], ", ".join(out)), flush=True)
