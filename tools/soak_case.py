"""One case of the soak run under several option sets: frames, and how many differed from the first in ray counts or image.
usage: python tools/soak_case.py scene resx resy N flags seconds"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
name, rx, ry, N, flags, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
g = rt.Renderer(0)
s = rt.Scene.load(name); g.upload(s); cam = s.camera(rx, ry)
u8 = torch.zeros((cam.resy, cam.resx, 3), dtype=torch.uint8, device="cuda:0")
hit = torch.zeros((cam.resy, cam.resx), dtype=torch.int32, device="cuda:0")
for opts in ({}, {"pair": 0}):
    for k, v in opts.items(): g.set_option(k, v)
    first = None; n = 0; bad = 0; t0 = time.time()
    while time.time() - t0 < budget:
        hit.fill_(-7); torch.cuda.synchronize()
        st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), hit_id_ptr=hit.data_ptr(), num_samples=N, flags=flags)
        key = (tuple(st["rays_per_generation"]), tuple(st["hits_per_generation"]), st["shadow_rays"], st["queue_overflow"], int((hit == -7).sum()), int(hit.to(torch.int64).sum()))
        if first is None: first = key
        elif key != first:
            bad += 1
            if bad <= 3: print("   differs:", key, "first:", first, flush=True)
        n += 1
    print("%s: %d frames, %d differ" % (opts, n, bad), flush=True)
    for k in opts: g.set_option(k, -1 if k != "units" else 1)
