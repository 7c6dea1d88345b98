"""Self-contained blocks (k_local): how far a block's share of the primary chunks moves on per round.
usage: python tools/exp_local_rotate.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
def timed(cam, n=40, **opts):
    for k, v in opts.items():
        g.set_option(k, v)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    return (time.perf_counter() - t0) / n * 1e3, st, u8.cpu().numpy()
for name, rx, ry in (("teapot-3", 512, 512), ("balls", 512, 512), ("refract", 512, 512), ("gears", 512, 512), ("balls2", 512, 512),
                     ("teapot", 512, 512), ("teapot-3", 640, 480), ("balls", 640, 480), ("balls", 1024, 768), ("refract", 1024, 1024),
                     ("teapot-3", 1280, 720)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(rx, ry)
    a = timed(cam, local=1, fused=0, local_rotate=0)
    out = []
    for rot in (-1, 7, (rx + 7) // 8 // 4 + 1, ((rx + 7) // 8 // 4 + 1) | 1, (rx + 7) // 8 // 2 + 1, (rx + 7) // 8 // 8 + 1, 3 * ((rx + 7) // 8) // 4 + 1):
        b = timed(cam, local=1, fused=0, local_rotate=rot)
        ok = np.array_equal(a[2], b[2]) and a[1]["rays_per_generation"] == b[1]["rays_per_generation"] and a[1]["shadow_rays"] == b[1]["shadow_rays"]
        out.append("%d: %.3f%s" % (rot, b[0], "" if ok else " DIFFERENT"))
    print("%-9s %4dx%-4d: no rotation %.3f ms; %s" % (name, rx, ry, a[0], ", ".join(out)), flush=True)
