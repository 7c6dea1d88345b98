"""Where a pass of a small frame spends its time: per-wave clock stamps of one launch (a -DRT_STAMP build).
usage: RAYTRACER_AMD_LIB=build_variants/lib_stamp.so python tools/exp_stamps.py [scene] [res]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import raytracer_amd as rt
scene = sys.argv[1] if len(sys.argv) > 1 else "teapot-3"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = rt.load()
s = rt.Scene.load(scene); g = rt.Renderer(0); g.upload(s); cam = s.camera(res, res)
W, S = 16384, 12
names = ["start", "map", "filled", "rec", "walked", "geom", "accum", "reserved", "pushed", "end"]
fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
g.set_option("fused", 1 if fused else 0)
for _ in range(3):
    g.render(cam, want=("rgb_u8",))
for sel in ([16] if fused else [0, 1, 2, 5, 8, 9, 10, 13]):
    lib.rt_debug_stamp_select(sel)
    buf = (C.c_ulonglong * (W * S))()
    lib.rt_debug_stamps(buf, W * S)          # clears
    st = g.render(cam, want=("rgb_u8",))["stats"]
    lib.rt_debug_stamps(buf, W * S)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(W, S).astype(np.int64)
    live = a[:, 0] > 0
    if not live.any():
        print("sel", sel, "no waves"); continue
    t0 = a[live, 0].min()
    worked = live & (a[:, 3] > 0)
    out = ["%s gen %d: %d waves started, %d worked; start spread %.1f us; launch span %.1f us" % (
        "closest" if sel < 8 else ("shadow" if sel < 16 else "fused"), sel % 8, live.sum(), worked.sum(), (a[live, 0].max() - t0) / 100.0,
        (a[live][:, [0, 9]].max() - t0) / 100.0)]
    if worked.any():
        w = a[worked]
        seg = []
        for k in range(1, 10):
            if (w[:, k] > 0).all() and (w[:, k - 1] > 0).all():
                seg.append("%s %.1f" % (names[k], np.median(w[:, k] - w[:, k - 1]) / 100.0))
            elif (w[:, k] > 0).any():
                ok = (w[:, k] > 0)
                prev = np.where(w[:, k - 1] > 0, w[:, k - 1], w[:, max(k - 2, 0)])
                seg.append("%s~%.1f" % (names[k], np.median((w[:, k] - prev)[ok]) / 100.0))
        out.append("  median us per segment (first chunk of a wave): " + ", ".join(seg))
        out.append("  first chunk done after (median, max) %.1f %.1f us; wave end (median, max) %.1f %.1f us" % (
            np.median(w[:, 8] - t0) / 100.0, (w[:, 8] - t0).max() / 100.0, np.median(w[:, 9] - t0) / 100.0, (w[:, 9] - t0).max() / 100.0))
    print("\n".join(out))
if fused:
    w = a[a[:, 10] > 0]
    t0 = w[:, 0].min()
    dur = (w[:, 10] - w[:, 3]) / 100.0
    order = np.argsort(dur)
    for name, idx in (("median", order[len(order) // 2]), ("p90", order[int(len(order) * 0.9)]), ("slowest", order[-1])):
        r = w[idx]
        print("  %s first chunk: iteration 1 starts %.1f, walked %.1f, geometry %.1f, lights %.1f, iteration 2 starts %.1f, chunk ends %.1f us" % (
            (name,) + tuple((int(r[k]) - int(r[3])) / 100.0 if r[k] > 0 else -1 for k in (4, 5, 6, 7, 9, 10))))
print(st["gpu_ms"], "ms frame;", st["rays_per_generation"], st["hits_per_generation"])
