"""Up to which frame size one launch of self-contained blocks (k_local) beats a launch per pass.
usage: python tools/exp_local_limit.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
def timed(cam, n=12, **opts):
    for k, v in opts.items():
        g.set_option(k, v)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    return (time.perf_counter() - t0) / n * 1e3, st, u8.cpu().numpy()
for name in ("teapot-3", "balls", "refract", "gears", "balls2", "teapot", "tetra"):
    s = rt.Scene.load(name); g.upload(s)
    out = []
    for rx, ry in ((1448, 1448), (2048, 2048), (2896, 2896), (4096, 4096)):
        cam = s.camera(rx, ry)
        a = timed(cam, local=0, fused=0)
        b = timed(cam, local=1, fused=0)
        ok = np.array_equal(a[2], b[2]) and a[1]["rays_per_generation"] == b[1]["rays_per_generation"] and b[1]["queue_overflow"] == 0
        out.append("%dx%d %.3f / %.3f%s" % (rx, ry, a[0], b[0], "" if ok else " DIFFERENT"))
    print("%-9s launch per pass / self-contained blocks, ms: %s" % (name, ", ".join(out)), flush=True)
