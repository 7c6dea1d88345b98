"""Rays and hits per generation of the ray tree, for the bench scenes: how much of a frame the late generations are.
usage: python tools/exp_generations.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
g = rt.Renderer(0)
for name, res, N in (("teapot-3", (7680, 4320), 4), ("teapot-3", (512, 512), 1), ("balls", (512, 512), 1), ("refract", (512, 512), 1),
                     ("gears", (512, 512), 1), ("balls2", (512, 512), 1), ("glass", (900, 900), 1)):
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(*res)
    u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
    st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=N)
    r, h = st["rays_per_generation"], st["hits_per_generation"]
    print("%-9s %5dx%-5d rays %s  hits %s  share of rays after generation 1: %.1f %%, after 2: %.1f %%" % (
        name, res[0], res[1], r, h, 100.0 * sum(r[2:]) / sum(r), 100.0 * sum(r[3:]) / sum(r)), flush=True)
