"""Time of one rank's share of the 8K x 16 spp frame for 1, 2, 4 and 8 parts (strong-scaling preview on one GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
whole = None
for parts in (1, 2, 4, 8):
    rows = rt.part_rows(cam, 16 if parts > 1 else 0, parts, 0)
    buf = torch.zeros((rows, 7680, 3), dtype=torch.uint8, device="cuda")
    kw = dict(rgb_u8_ptr=buf.data_ptr(), num_samples=4, band_rows=16 if parts > 1 else 0, n_parts=parts, part=0)
    g.render_device(cam, **kw)
    t0 = time.perf_counter()
    for _ in range(5):
        st = g.render_device(cam, **kw)
    dt = (time.perf_counter() - t0) / 5
    whole = whole or dt
    print("parts %d: %.2f ms per part (gpu %.2f ms), whole frame / parts = %.2f ms" % (parts, dt * 1e3, st["gpu_ms"], whole * 1e3 / parts), flush=True)
