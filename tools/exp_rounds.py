"""8K x 16 spp frame at a 16 GiB queue budget by the number of block rounds per pass.  usage: python tools/exp_rounds.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
u8 = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
for gib in (16, 96):
    for rounds in (1, 2, 3, 4, 8):
        g.set_memory_budget(gib << 30); g.set_option("rounds", rounds)
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, flags=rt.FLAG_TIMING)
        dt = (time.perf_counter() - t0) / 3
        print("budget %d GiB rounds %d: %.2f ms (%d batches) primary %.2f closest %.2f shadow %.2f" % (gib, rounds, dt * 1e3, st["batches"], st["primary_ms"], st["closest_ms"] - st["primary_ms"], st["shadow_ms"]), flush=True)
