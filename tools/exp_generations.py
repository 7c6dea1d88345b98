"""Rays and hits per generation of a workload.  usage: python tools/exp_generations.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, raytracer_amd as rt, bench
w = sys.argv[1] if len(sys.argv) > 1 else "teapot3_8k16"
scene, rx, ry, n, _, _ = bench.WORKLOADS[w]
s = rt.Scene.load(scene); g = rt.Renderer(0); g.upload(s); cam = s.camera(rx, ry)
u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=n)
print(w, "rays per generation", st["rays_per_generation"], "hits", st["hits_per_generation"], "shadow rays", st["shadow_rays"])
