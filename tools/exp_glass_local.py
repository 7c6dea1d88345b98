import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import raytracer_amd as rt
g = rt.Renderer(0)
def timed(cam, n=20, **opts):
    for k, v in opts.items(): g.set_option(k, v)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3): g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
    return (time.perf_counter() - t0) / n * 1e3, st, u8.cpu().numpy()
for name in ("glass", "glass2", "glass4"):
    s = rt.Scene.load(name); g.upload(s)
    out = []
    for rx, ry in ((256, 256), (512, 512), (900, 900), (1448, 1448)):
        cam = s.camera(rx, ry)
        a = timed(cam, local=0, fused=0); b = timed(cam, local=1, fused=0)
        ok = np.array_equal(a[2], b[2]) and a[1]["rays_per_generation"] == b[1]["rays_per_generation"]
        out.append("%d: %.3f / %.3f%s" % (rx, a[0], b[0], "" if ok else " DIFFERENT"))
    print(name, "launch per pass / self-contained, ms:", ", ".join(out), flush=True)
