"""Light grids at several resolutions: build time at upload (verbose line on stderr) and frame time, per scene at 4096^2.
usage: python tools/exp_lightgrid_res.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
g.set_option("verbose", 1)
u8 = torch.zeros((4096, 4096, 3), dtype=torch.uint8, device="cuda")
for scene in ("teapot-3", "balls", "refract", "gears", "balls2", "tetra"):
    s = rt.Scene.load(scene)
    cam = s.camera(4096, 4096)
    for res in (256, 384, 512, 768):
        g.set_option("grid_res", res)
        t0 = time.perf_counter(); g.upload(s); t1 = time.perf_counter()
        best = 1e9
        for _ in range(4):
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=1)
            best = min(best, st["gpu_ms"])
        print("%-9s res %4d | upload %.2f ms | frame %.2f ms | sum %d" % (scene, res, (t1 - t0) * 1e3, best, int(u8.to(torch.int64).sum())), flush=True)
    s.close()
