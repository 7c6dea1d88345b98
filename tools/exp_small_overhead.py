"""What a small frame spends outside its traversal kernel: wall time per frame, GPU time between the frame's first and
last command, and the kernel itself.  usage: python tools/exp_small_overhead.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
for name in ("teapot-3", "balls", "refract", "tetra-3"):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(512, 512)
    u8 = torch.zeros((512, 512, 3), dtype=torch.uint8, device="cuda")
    for flags in (0, rt.FLAG_TIMING):
        for _ in range(5):
            g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), flags=flags)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 50
        acc = [0.0, 0.0]
        for _ in range(n):
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), flags=flags)
            acc[0] += st["gpu_ms"]; acc[1] += st["closest_ms"] + st["shadow_ms"] + st["pair_ms"]
        wall = (time.perf_counter() - t0) / n * 1e3
        print("%-9s timing flag %d: wall %.3f ms per frame, first to last command on the GPU %.3f ms, traversal kernels %.3f ms" % (
            name, 1 if flags else 0, wall, acc[0] / n, acc[1] / n), flush=True)
