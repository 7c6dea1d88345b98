"""The one-launch kernel against the wavefront pipeline on the same frames: outputs must agree, times are printed.
usage: python tools/exp_fused.py [scene:res ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
cases = sys.argv[1:] or ["teapot-3:512", "balls:512", "refract:512", "tetra-3:512", "gears:512", "glass:900", "balls2:512", "glass3:512"]
g = rt.Renderer(0)
for case in cases:
    name, res = case.split(":"); res = int(res)
    s = rt.Scene.load(name); g.upload(s); cam = s.camera(res, res)
    u8 = torch.zeros((res, res, 3), dtype=torch.uint8, device="cuda")
    out = {}
    for mode in (0, 1):
        g.set_option("fused", mode)
        fr = g.render(cam, flags=rt.FLAG_KEEP_ZERO_WEIGHT)
        for _ in range(3):
            g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr())
        dt = (time.perf_counter() - t0) / n
        out[mode] = (fr, dt, st)
    a, b = out[0][0], out[1][0]
    same = (np.array_equal(a["rgb_u8"], b["rgb_u8"]), np.array_equal(a["hit_id"], b["hit_id"]), float(np.abs(a["rgb_f64"] - b["rgb_f64"]).max()),
            np.array_equal(a["lightcam_mask"], b["lightcam_mask"]),
            a["stats"]["closest_rays"] == b["stats"]["closest_rays"], a["stats"]["shadow_rays"] == b["stats"]["shadow_rays"],
            a["stats"]["rays_per_generation"] == b["stats"]["rays_per_generation"], a["stats"]["hits_per_generation"] == b["stats"]["hits_per_generation"])
    rays = out[0][2]["closest_rays"] + out[0][2]["shadow_rays"]
    print("%-10s %4d  wavefront %.3f ms (gpu %.3f)  fused %.3f ms (gpu %.3f)  %.2fx  %.0f -> %.0f Mrays/s  same: %s" % (
        name, res, out[0][1] * 1e3, out[0][2]["gpu_ms"], out[1][1] * 1e3, out[1][2]["gpu_ms"], out[0][1] / out[1][1],
        rays / out[0][1] / 1e6, rays / out[1][1] / 1e6, same), flush=True)
    s.close()
g.close()
