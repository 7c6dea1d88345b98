"""Self-contained blocks (k_local): primary chunks a block gets at least, i.e. how soon a small frame is spread over every compute unit.
usage: python tools/exp_local_grid.py"""
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
for name, rx, ry in (("teapot-3", 128, 128), ("teapot-3", 256, 256), ("balls", 256, 256), ("refract", 256, 256), ("teapot-3", 384, 384), ("balls", 384, 384),
                     ("teapot-3", 512, 512), ("teapot-3", 640, 480), ("balls", 640, 480), ("refract", 640, 480), ("gears", 640, 480), ("balls", 800, 600)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(rx, ry)
    a = timed(cam, local=0, fused=0)
    out = []
    for mc in (0, 8, 4, 2, 1):
        b = timed(cam, local=1, fused=0, local_min_chunks=mc)
        ok = np.array_equal(a[2], b[2]) and a[1]["rays_per_generation"] == b[1]["rays_per_generation"] and a[1]["shadow_rays"] == b[1]["shadow_rays"]
        out.append("%d: %.3f%s" % (mc, b[0], "" if ok else " DIFFERENT"))
    print("%-9s %4dx%-4d: launch per pass %.3f ms; at least so many chunks per block %s" % (name, rx, ry, a[0], ", ".join(out)), flush=True)
