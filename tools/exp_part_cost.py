"""Cost per primary sample of the 8K x 16 spp frame: whole, in 66.8 M-sample batches, and as one rank's eighth
with several band heights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
buf = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
for label, kw in (("whole, 66.8M batches", dict(batch=66846720)), ("whole, default", dict(batch=0)),
                  ("part 0/8 bands 16", dict(batch=0, band_rows=16, n_parts=8, part=0)),
                  ("part 0/8 bands 8", dict(batch=0, band_rows=8, n_parts=8, part=0)),
                  ("part 0/8 bands 64", dict(batch=0, band_rows=64, n_parts=8, part=0)),
                  ("part 3/8 bands 16", dict(batch=0, band_rows=16, n_parts=8, part=3))):
    g.set_batch(kw.pop("batch"))
    a = dict(rgb_u8_ptr=buf.data_ptr(), num_samples=4, **kw)
    g.render_device(cam, **a)
    st = g.render_device(cam, flags=rt.FLAG_TIMING, **a)
    smp = st["rays_per_generation"][0]
    print("%-22s gpu %.2f ms | per G samples: frame %.1f primary %.2f queued %.2f shadow %.2f ms" % (
        label, st["gpu_ms"], st["gpu_ms"] / smp * 1e9, st["primary_ms"] / smp * 1e9, (st["closest_ms"] - st["primary_ms"]) / smp * 1e9, st["shadow_ms"] / smp * 1e9))
