"""Scale check: set-up time and frame time on a generated terrain of n*n*2 triangles plus 3000 spheres
(tests/test_gpu_large_scene.py checks the same kind of scene against the CPU checker).
usage: python tools/exp_big_scene.py [n] [resx resy spp]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
big = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160, 2)
rng = np.random.default_rng(5)
xs = np.linspace(-20, 20, n + 1)
X, Y = np.meshgrid(xs, xs, indexing="ij")
Z = 1.5 * np.sin(X * 0.7) * np.cos(Y * 0.5) + 0.06 * rng.normal(size=X.shape)
out = ["b 0.2 0.3 0.5\nv\nfrom 30 24 18\nat 0 0 0\nup 0 0 1\nangle 45\nhither 1\nresolution 96 96\nl 30 10 40\nl -20 30 35\nl 5 -30 20\n",
       "f 0.4 0.7 0.3 0.8 0.3 12.5 0 1\n"]
def v(i, j):
    return "%.17g %.17g %.17g" % (X[i, j], Y[i, j], Z[i, j])
for i in range(n):
    for j in range(n):
        out.append("p 3\n%s\n%s\n%s\np 3\n%s\n%s\n%s\n" % (v(i, j), v(i + 1, j), v(i + 1, j + 1), v(i, j), v(i + 1, j + 1), v(i, j + 1)))
out.append("f 0.9 0.8 0.7 0.5 0.6 40 0 1\n")
for k in range(3000):
    c = rng.uniform(-18, 18, 2)
    out.append("s %.17g %.17g %.17g %.17g\n" % (c[0], c[1], 2.2 + rng.uniform(0, 3), rng.uniform(0.1, 0.5)))
path = "/tmp/terrain_%d.nff" % n
open(path, "w").write("".join(out))

t = time.time(); s = rt.Scene.load(path); t_front = time.time() - t
g = rt.Renderer(0)
t = time.time(); g.upload(s); t_up = time.time() - t
print("triangles %d | front end (parse, triangulate, octree) %.2f s | upload incl. hierarchy build %.2f s" % (2 * n * n, t_front, t_up), flush=True)
cam = s.camera(big[0], big[1])
for it in range(3):
    fr = g.render(cam, num_samples=big[2], want=("rgb_u8",), flags=rt.FLAG_TIMING if it == 2 else 0)
    st = fr["stats"]
    rays = st["closest_rays"] + st["shadow_rays"]
    print("%dx%d %d spp: %.2f ms, %.0f Mrays/s (%d rays) | primary %.1f closest %.1f shadow %.1f ms" % (big[0], big[1], big[2] ** 2, st["gpu_ms"], rays / st["gpu_ms"] / 1e3, rays, st["primary_ms"], st["closest_ms"], st["shadow_ms"]))
if "count" in os.environ.get("RAYTRACER_AMD_LIB", ""):
    import ctypes as C
    fn = C.CDLL(rt.LIB_PATH).rt_debug_counters
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 48)()
    fn(buf)
    fr = g.render(cam, num_samples=big[2], want=("rgb_u8",))
    fn(buf)
    names = ["rays", "lane_nodes", "wave_nodes", "lane_prims", "wave_prims", "lane_leaves", "wave_rounds"]
    for gi, gname in enumerate(("primary", "queued", "shadow")):
        v = dict(zip(names, list(buf[gi * 8:gi * 8 + 7])))
        r = max(v["rays"], 1)
        print("%-8s rays %11d | per ray: nodes %.2f leaves %.2f prims %.2f | per 64 rays the wave ran: node steps %.1f prim tests %.1f rounds %.1f"
              % (gname, v["rays"], v["lane_nodes"] / r, v["lane_leaves"] / r, v["lane_prims"] / r,
                 v["wave_nodes"] * 64 / r, v["wave_prims"] * 64 / r, v["wave_rounds"] * 64 / r))
    print(fr["stats"])
