"""Per-kernel-class times of one workload (HIP events around every launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt, bench
w = sys.argv[1] if len(sys.argv) > 1 else "teapot3_8k16"
scene_name, resx, resy, N, _, _ = bench.WORKLOADS[w]
s = rt.Scene.load(scene_name); g = rt.Renderer(0); g.upload(s)
cam = s.camera(resx, resy)
buf = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda")
g.render_device(cam, rgb_u8_ptr=buf.data_ptr(), num_samples=N)
tot = {}
for _ in range(3):
    st = g.render_device(cam, rgb_u8_ptr=buf.data_ptr(), num_samples=N, flags=rt.FLAG_TIMING)
    for k in ("gpu_ms", "closest_ms", "shadow_ms", "other_ms"):
        tot[k] = tot.get(k, 0) + st[k] / 3
print(w, os.environ.get("RAYTRACER_AMD_LIB", "base").split("/")[-1], {k: round(v, 2) for k, v in tot.items()}, "launches", st["closest_launches"], st["shadow_launches"])
