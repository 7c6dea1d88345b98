"""How many generations of closest-hit passes push whole chunks of records (option whole_chunks), by workload.
usage: python tools/exp_whole_chunks.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
def timed(cam, spp, n, **opts):
    for k, v in opts.items():
        g.set_option(k, v)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        g.render_device(cam, num_samples=spp, rgb_u8_ptr=u8.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st = g.render_device(cam, num_samples=spp, rgb_u8_ptr=u8.data_ptr())
    return (time.perf_counter() - t0) / n * 1e3, st
for name, rx, ry, spp, n in (("teapot-3", 7680, 4320, 4, 5), ("teapot-3", 512, 512, 1, 40), ("balls", 512, 512, 1, 40), ("gears", 512, 512, 1, 30),
                             ("balls2", 512, 512, 1, 30), ("teapot-3", 2048, 2048, 1, 20), ("balls", 4096, 4096, 1, 8), ("gears", 4096, 4096, 1, 8),
                             ("balls2", 4096, 4096, 1, 4), ("teapot", 4096, 4096, 1, 8), ("refract", 4096, 4096, 1, 8)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(rx, ry)
    base = None
    out = []
    for w in (0, 1, 2, 3, 4, 6):
        t, st = timed(cam, spp, n, whole_chunks=w)
        key = (st["rays_per_generation"], st["hits_per_generation"], st["shadow_rays"], st["queue_overflow"])
        if base is None: base = key
        out.append("%d: %.3f%s" % (w, t, "" if key == base else " DIFFERENT"))
    g.set_option("whole_chunks", -1)
    print("%-9s %4dx%-4d %2d spp, ms by generations that push whole chunks: %s" % (name, rx, ry, spp * spp, ", ".join(out)), flush=True)
