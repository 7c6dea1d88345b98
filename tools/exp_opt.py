"""8K x 16 spp teapot frame (or EXP_SCENE/EXP_RX/EXP_RY/EXP_N) under settings of one option.
usage: python tools/exp_opt.py option value [value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, raytracer_amd as rt
scene, rx, ry, n = os.environ.get("EXP_SCENE", "teapot-3"), int(os.environ.get("EXP_RX", 7680)), int(os.environ.get("EXP_RY", 4320)), int(os.environ.get("EXP_N", 4))
s = rt.Scene.load(scene); g = rt.Renderer(0); cam = s.camera(rx, ry)
u8 = torch.zeros((ry, rx, 3), dtype=torch.uint8, device="cuda")
opt = sys.argv[1]
for v in sys.argv[2:]:
    g.set_option(opt, int(v)); g.upload(s)
    best = None
    for it in range(5):
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=n, flags=rt.FLAG_TIMING if it == 4 else 0)
        if it < 4 and (best is None or st["gpu_ms"] < best): best = st["gpu_ms"]
    print("%-16s = %-6s frame %7.2f ms | primary %6.2f queued %6.2f shadow %6.2f pair %6.2f | sum %d rays %d" % (opt, v, best, st["primary_ms"],
          st["closest_ms"] - st["primary_ms"], st["shadow_ms"], st["pair_ms"], int(u8.to(torch.int64).sum()), st["closest_rays"] + st["shadow_rays"]), flush=True)
