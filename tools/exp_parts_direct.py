"""One part of 1, 4, 8 of the 8K x 16 spp frame with and without direct placement: pass times and batches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
for parts in (1, 4, 8):
    rows = rt.part_rows(cam, 16 if parts > 1 else 0, parts, 0)
    buf = torch.zeros((rows, 7680, 3), dtype=torch.uint8, device="cuda")
    for direct in (0, 1):
        g.set_option("direct", direct)
        kw = dict(rgb_u8_ptr=buf.data_ptr(), num_samples=4, band_rows=16 if parts > 1 else 0, n_parts=parts, part=0)
        g.render_device(cam, **kw)
        best = min(g.render_device(cam, **kw)["gpu_ms"] for _ in range(4))
        st = g.render_device(cam, flags=rt.FLAG_TIMING, **kw)
        print("parts %d direct %d: %.2f ms | primary %.2f queued %.2f shadow %.2f pair %.2f | batches %d" % (parts, direct, best, st["primary_ms"],
              st["closest_ms"] - st["primary_ms"], st["shadow_ms"], st["pair_ms"], st["batches"]), flush=True)
