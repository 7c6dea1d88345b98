"""Frame times with and without the view grids (viewgrid.hpp): 8K x 16 spp teapot frame and the shipped scenes at 4096 x 4096.
usage: python tools/exp_grid.py [8k] [scene ...]       option grid: 0 none, 1 lights + eye, 2 lights only, 3 eye only"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt

args = sys.argv[1:] or ["8k", "teapot-3", "balls", "refract", "gears"]
modes = [int(m) for m in os.environ.get("GRID_MODES", "0,2,3,1").split(",")]
res_list = [int(r) for r in os.environ.get("GRID_RES", "256").split(",")]
g = rt.Renderer(0)
g.set_option("verbose", int(os.environ.get("VERBOSE", "1")))
for name in args:
    scene, (rx, ry), n = ("teapot-3", (7680, 4320), 4) if name == "8k" else (name, (4096, 4096), 1)
    s = rt.Scene.load(scene)
    buf = torch.zeros((ry, rx, 3), dtype=torch.uint8, device="cuda:0")
    hit = torch.zeros((ry, rx), dtype=torch.int32, device="cuda:0")
    ref = None
    for res in res_list:
        for mode in modes:
            if mode in (0, 3) and res != res_list[0]:
                continue
            g.set_option("grid", mode)
            g.set_option("grid_res", res)
            t0 = time.perf_counter()
            g.upload(s)
            t_up = time.perf_counter() - t0
            cam = s.camera(rx, ry)
            best = None
            for it in range(4):
                st = g.render_device(cam, rgb_u8_ptr=buf.data_ptr(), hit_id_ptr=hit.data_ptr(), num_samples=n, flags=rt.FLAG_TIMING if it == 3 else 0)
                best = st if best is None or st["gpu_ms"] < best["gpu_ms"] else best
            key = (hash(buf.cpu().numpy().tobytes()), hash(hit.cpu().numpy().tobytes()), st["closest_rays"], st["shadow_rays"])
            if ref is None:
                ref = key
            rays = st["closest_rays"] + st["shadow_rays"]
            print("%-9s grid %d res %4d | upload %6.1f ms | %7.2f ms %7.0f Mrays/s | primary %.2f queued %.2f shadow %.2f pair %.2f | same frame %s"
                  % (name, mode, res, t_up * 1e3, best["gpu_ms"], rays / best["gpu_ms"] / 1e3, st["primary_ms"], st["closest_ms"] - st["primary_ms"],
                     st["shadow_ms"], st["pair_ms"], key == ref), flush=True)
    s.close()
