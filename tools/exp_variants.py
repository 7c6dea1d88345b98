"""Frame and pass times of the 8K x 16 spp teapot frame (or another workload) for several builds of the library.
usage: python tools/exp_variants.py lib1.so lib2.so ...   (each in a process of its own)"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import os, sys
sys.path.insert(0, os.path.dirname(%r))
import torch, raytracer_amd as rt
scene, rx, ry, n = os.environ.get("EXP_SCENE", "teapot-3"), int(os.environ.get("EXP_RX", 7680)), int(os.environ.get("EXP_RY", 4320)), int(os.environ.get("EXP_N", 4))
s = rt.Scene.load(scene); g = rt.Renderer(0)
for kv in os.environ.get("EXP_OPTS", "").split():
    g.set_option(kv.split("=")[0], int(kv.split("=")[1]))
g.upload(s); cam = s.camera(rx, ry)
u8 = torch.zeros((ry, rx, 3), dtype=torch.uint8, device="cuda")
best = None
for it in range(5):
    st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=n, flags=rt.FLAG_TIMING if it == 4 else 0)
    if it < 4 and (best is None or st["gpu_ms"] < best): best = st["gpu_ms"]
print("%%-44s frame %%7.2f ms | primary %%6.2f queued %%6.2f shadow %%6.2f pair %%6.2f | sum %%d" %% (os.environ.get("RAYTRACER_AMD_LIB", "product") + " " + os.environ.get("EXP_OPTS", ""), best, st["primary_ms"],
      st["closest_ms"] - st["primary_ms"], st["shadow_ms"], st["pair_ms"], int(u8.to(torch.int64).sum())))
''' % here
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "product":
        env["RAYTRACER_AMD_LIB"] = lib
    subprocess.run([sys.executable, "-c", code], env=env, check=False)
