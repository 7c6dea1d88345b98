"""Per-kernel times of the whole 8K x 16 spp frame and of one rank's eighth of it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
for parts in (1, 8):
    rows = rt.part_rows(cam, 16 if parts > 1 else 0, parts, 0)
    buf = torch.zeros((rows, 7680, 3), dtype=torch.uint8, device="cuda")
    kw = dict(rgb_u8_ptr=buf.data_ptr(), num_samples=4, band_rows=16 if parts > 1 else 0, n_parts=parts, part=0)
    g.render_device(cam, **kw)
    st = g.render_device(cam, flags=rt.FLAG_TIMING, **kw)
    print(parts, "gpu %.2f primary %.2f queued %.2f shadow %.2f other %.2f launches c%d s%d" % (st["gpu_ms"], st["primary_ms"], st["closest_ms"] - st["primary_ms"], st["shadow_ms"], st["other_ms"], st["closest_launches"], st["shadow_launches"]), st["rays_per_generation"])
