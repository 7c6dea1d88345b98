"""How evenly the primary pass of a large batch ends: per-block end times of one launch (a -DRT_STAMP build).
usage: RAYTRACER_AMD_LIB=build_variants/lib_stamp.so python tools/exp_primary_balance.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
lib = rt.load()
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
g.set_memory_budget(200 << 30)
W, S = 16384, 12
for parts in (3,):
    rows = 4320 // parts
    for part in range(parts):
        u8 = torch.zeros((rows, 7680, 3), dtype=torch.uint8, device="cuda")
        for sel in (0, 8):
            lib.rt_debug_stamp_select(sel)
            buf = (C.c_ulonglong * (W * S))()
            lib.rt_debug_stamps(buf, W * S)          # clears
            st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4, band_rows=rows, n_parts=parts, part=part, flags=rt.FLAG_TIMING)
            lib.rt_debug_stamps(buf, W * S)
            a = np.frombuffer(buf, dtype=np.uint64).reshape(W, S).astype(np.int64)
            live = a[:, 0] > 0
            t0 = a[live, 0].min()
            end = (a[live, 9] - t0) / 100.0                      # us
            blocks = end[: len(end) // 16 * 16].reshape(-1, 16).max(axis=1)
            begin = ((a[live, 0] - t0) / 100.0)[: len(end) // 16 * 16].reshape(-1, 16).min(axis=1)
            half = len(blocks) // 2
            for name, sl in (("blocks of the first round ", slice(0, half)), ("blocks of the second round", slice(half, None))):
                d = blocks[sl] - begin[sl]
                print("    %s: start %.0f..%.0f us, end %.0f..%.0f us, lifetime min %.0f median %.0f max %.0f us" % (
                    name, begin[sl].min(), begin[sl].max(), blocks[sl].min(), blocks[sl].max(), d.min(), np.median(d), d.max()))
            print("%d parts, part %d, %s pass: %d batches, launch %.0f us; a block's last wave ends at min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f us "
                  "(mean %.0f: %.1f %% of the slots idle at the end)" % (
                      parts, part, "primary" if sel == 0 else "shadow 0", st["batches"], end.max(), blocks.min(), np.percentile(blocks, 10), np.median(blocks),
                      np.percentile(blocks, 90), blocks.max(), blocks.mean(), 100.0 * (1 - blocks.mean() / blocks.max())), flush=True)
