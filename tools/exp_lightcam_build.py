"""Light-camera set-up: device builder against host builder (same frames), and the set-up time.
usage: python tools/exp_lightcam_build.py [scene ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import raytracer_amd as rt
for name in sys.argv[1:] or ["glass", "glass2", "glass3", "glass4", "ball"]:
    s = rt.Scene.load(name)
    out = {}
    for host in (1, 0):
        g = rt.Renderer(0); g.set_option("lightcam_host", host); g.set_option("verbose", 0)
        g.upload(s)
        t0 = time.perf_counter(); fr = g.render(s.camera()); t1 = time.perf_counter()
        g.upload(s)
        t2 = time.perf_counter(); fr2 = g.render(s.camera()); t3 = time.perf_counter()
        fr3 = g.render(s.camera()); t4 = time.perf_counter()
        out[host] = (fr, t3 - t2, t4 - t3)
        g.close()
    a, b = out[1][0], out[0][0]
    print("%-7s first frame (set-up + render): host builder %.1f ms, device builder %.1f ms; later frames %.2f ms; same image %s, max f64 diff %.3g, lightcam rays %d / %d" % (
        name, out[1][1] * 1e3, out[0][1] * 1e3, out[0][2] * 1e3, np.array_equal(a["rgb_u8"], b["rgb_u8"]), np.abs(a["rgb_f64"] - b["rgb_f64"]).max(),
        a["stats"]["lightcam_rays"], b["stats"]["lightcam_rays"]), flush=True)
    s.close()
