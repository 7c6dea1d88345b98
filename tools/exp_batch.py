"""Frame time of the 8K x 16 spp workload for several primary batch sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s)
cam = s.camera(7680, 4320)
buf = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
for b in [int(x) for x in sys.argv[1:]] or [32 << 20, 64 << 20, 128 << 20, 256 << 20, 540 << 20]:
    g.set_batch(b)
    g.render_device(cam, rgb_u8_ptr=buf.data_ptr(), num_samples=4)
    t0 = time.perf_counter()
    for _ in range(3):
        st = g.render_device(cam, rgb_u8_ptr=buf.data_ptr(), num_samples=4)
    dt = (time.perf_counter() - t0) / 3
    print("batch %4d M samples: %.2f ms/frame, %d batches, %.1f Mrays/s" % (b >> 20, dt * 1e3, st["batches"], (st["closest_rays"] + st["shadow_rays"]) / dt / 1e6), flush=True)
