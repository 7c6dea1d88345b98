"""Frame time by leaf size of the derived hierarchies.  usage: python tools/exp_leaf.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
g = rt.Renderer(0); g.set_memory_budget(96 << 30)
for name, res, N in (("teapot-3", (7680, 4320), 4), ("balls", (4096, 4096), 1), ("gears", (2048, 2048), 1), ("teapot-3", (512, 512), 1)):
    s = rt.Scene.load(name)
    for leaf in (1, 2, 3, 4):
        g.set_option("leaf_max", leaf); g.upload(s); cam = s.camera(*res)
        u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=N)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 3 if res[0] > 1000 else 30
        for _ in range(n):
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=N)
        print("%-9s %dx%d leaf %d: %.3f ms" % (name, res[0], res[1], leaf, (time.perf_counter() - t0) / n * 1e3), flush=True)
    s.close()
