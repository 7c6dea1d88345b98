"""Rays and hits per generation of a frame (statistics of one render).  usage: python tools/exp_generation_counts.py [scene rx ry n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
scene, rx, ry, n = (sys.argv[1:5] + ["teapot-3", "7680", "4320", "4"][len(sys.argv) - 1:])[:4]
rx, ry, n = int(rx), int(ry), int(n)
s = rt.Scene.load(scene); g = rt.Renderer(0); g.upload(s); cam = s.camera(rx, ry)
u8 = torch.zeros((ry, rx, 3), dtype=torch.uint8, device="cuda")
st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=n)
print("rays per generation", st["rays_per_generation"])
print("hits per generation", st["hits_per_generation"])
print("closest", st["closest_rays"], "shadow", st["shadow_rays"], "batches", st["batches"])
