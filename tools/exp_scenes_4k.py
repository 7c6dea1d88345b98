"""Throughput of the shipped scenes at a resolution that fills the chip (4096 x 4096, 1 spp).
usage: python tools/exp_scenes_4k.py [scene ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # before the library initialises the system HIP runtime
import raytracer_amd as rt

names = sys.argv[1:] or ["tetra-3", "teapot-3", "teapot", "tetra", "balls", "balls2", "refract", "gears", "glass"]
g = rt.Renderer(0)
for name in names:
    s = rt.Scene.load(name)
    g.upload(s)
    cam = s.camera(4096, 4096)
    buf = torch.zeros((4096, 4096, 3), dtype=torch.uint8, device="cuda:0")     # the frame stays on the device
    best = None
    for it in range(3):
        st = g.render_device(cam, rgb_u8_ptr=buf.data_ptr(), flags=rt.FLAG_TIMING if it == 2 else 0)
        best = st if best is None or st["gpu_ms"] < best["gpu_ms"] else best
    rays = st["closest_rays"] + st["shadow_rays"]
    print("%-9s %5d prims | %7.2f ms  %7.0f Mrays/s (%d rays) | primary %.1f queued %.1f shadow %.1f ms"
          % (name, s.desc.n_prims, best["gpu_ms"], rays / best["gpu_ms"] / 1e3, rays, st["primary_ms"],
             st["closest_ms"] - st["primary_ms"], st["shadow_ms"]), flush=True)
    s.close()
