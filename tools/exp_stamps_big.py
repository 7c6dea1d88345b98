"""Where the chunks of a large frame's passes spend their time: per-wave clock stamps of the n-th chunk of every wave
(a -DRT_STAMP build).  usage: RAYTRACER_AMD_LIB=build_variants/lib_stamp.so python tools/exp_stamps_big.py [nth ...]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
lib = rt.load()
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
u8 = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
W, S = 16384, 12
names = ["start", "map", "filled", "rec", "queried", "geom", "accum", "reserved", "pushed", "end"]
for _ in range(2):
    g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
for nth in [int(a) for a in sys.argv[1:]] or [40, 200]:
    for sel, label in ((0, "primary"), (8, "shadow gen 0"), (9, "shadow gen 1")):
        lib.rt_debug_stamp_select(sel + 256 * nth)
        buf = (C.c_ulonglong * (W * S))()
        lib.rt_debug_stamps(buf, W * S)          # clears
        g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
        lib.rt_debug_stamps(buf, W * S)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(W, S).astype(np.int64)
        ok = (a[:, 10] > 0) & (a[:, 8] > 0)
        if not ok.any():
            print(label, "nth", nth, ": no wave got that far"); continue
        w = a[ok]
        seg = []
        last = 10
        for k in range(3, 9):
            have = w[:, k] > 0
            if have.all():
                d = (w[:, k] - w[:, last]) / 2400.0
                seg.append("%s %.2f (p90 %.2f)" % (names[k], np.median(d), np.percentile(d, 90)))
                last = k
        tot = (w[:, 8] - w[:, 10]) / 2400.0
        print("%-14s chunk %4d of %5d waves: us (shader cycles / 2400) per segment: %s | whole chunk median %.2f p90 %.2f" % (label, nth, ok.sum(), ", ".join(seg), np.median(tot), np.percentile(tot, 90)), flush=True)
