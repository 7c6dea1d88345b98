"""8K x 16 spp frame time by queue memory budget, with the primary pass in fixed shares per block and in units drawn
while it runs.  usage: python tools/exp_budget.py [GiB ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
u8 = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
ref = None
for gib in [float(a) for a in sys.argv[1:]] or [8, 16, 24, 32, 48, 96]:
    for units in (0, 1):
        g.set_memory_budget(int(gib * (1 << 30))); g.set_option("units", units)
        for _ in range(3):
            g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
        dt = (time.perf_counter() - t0) / 3
        img = u8.cpu().numpy()
        if ref is None: ref = img
        print("budget %5.1f GiB  primary pass: %s: %.2f ms  (%d batches)  same image %s" % (
            gib, "units drawn while it runs" if units else "fixed shares            ", dt * 1e3, st["batches"], np.array_equal(img, ref)), flush=True)
