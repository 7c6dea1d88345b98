"""8K x 16 spp frame time by number of batches (explicit batch sizes, 200 GiB budget).  usage: python tools/exp_batches.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
u8 = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
g.set_memory_budget(200 << 30)
n_chunks = (4320 // 8) * (7680 // 8) * 16
import numpy as np
ref = None
for nb, units in [(n, u) for n in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16) for u in (0, 1)]:
    per = -(-n_chunks // nb)
    per = -(-per // 16) * 16
    g.set_batch(per * 64)
    g.set_option("units", units)
    for _ in range(2):
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, flags=rt.FLAG_TIMING)
    dt = (time.perf_counter() - t0) / 3
    img = u8.cpu().numpy()
    if ref is None: ref = img
    print("units %d same image %s  %2d batches of %9d samples: %.2f ms  primary %.2f  closest %.2f  shadow %.2f  other %.2f" % (
        units, np.array_equal(ref, img), st["batches"], per * 64, dt * 1e3, st["primary_ms"], st["closest_ms"] - st["primary_ms"], st["shadow_ms"], st["other_ms"]), flush=True)
