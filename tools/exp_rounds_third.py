"""EXPERIMENT: lower third of the 8K frame, 1 and 2 batches, by grid size (rounds of resident blocks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
g.set_memory_budget(200 << 30)
parts, part = 3, 2
rows = 4320 // parts
u8 = torch.zeros((rows, 7680, 3), dtype=torch.uint8, device="cuda")
for rounds in (1, 2, 4):
    g.set_option("rounds", rounds)
    for units in (0, 1):
        g.set_option("units", units)
        for nb in (1, 2, 3):
            g.set_batch(-(-43200 // nb) * 64 * 64)
            for _ in range(2):
                g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, band_rows=rows, n_parts=parts, part=part)
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, band_rows=rows, n_parts=parts, part=part, flags=rt.FLAG_TIMING)
            print("rounds %d units %d: %d batches, frame %.2f ms  primary %.2f  closest %.2f  shadow %.2f" % (
                rounds, units, st["batches"], st["gpu_ms"], st["primary_ms"], st["closest_ms"] - st["primary_ms"], st["shadow_ms"]), flush=True)
