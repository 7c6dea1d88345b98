"""Primary-pass time of the lower third of the 8K x 16 spp frame, rendered as a part, by number of batches.
usage: python tools/exp_part_batches.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
g.set_memory_budget(200 << 30)
for parts, part in ((3, 2), (3, 1), (3, 0), (1, 0)):
    rows = 4320 // parts
    u8 = torch.zeros((rows, 7680, 3), dtype=torch.uint8, device="cuda")
    n_chunks = (rows // 8) * (7680 // 8) * 16
    for nb in (1, 2, 3, 4, 8):
        per = -(-n_chunks // nb)
        per = -(-per // 16) * 16
        g.set_batch(per * 64)
        for _ in range(2):
            g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, band_rows=rows, n_parts=parts, part=part)
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, band_rows=rows, n_parts=parts, part=part, flags=rt.FLAG_TIMING)
        print("part %d of %d in %d batches: frame %.2f ms  primary %.2f  closest %.2f  shadow %.2f" % (
            part, parts, st["batches"], st["gpu_ms"], st["primary_ms"], st["closest_ms"] - st["primary_ms"], st["shadow_ms"]), flush=True)
