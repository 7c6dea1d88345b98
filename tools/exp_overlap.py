"""8K x 16 spp frame time by batch size, with the shadow passes on the same stream or on a second one.
usage: python tools/exp_overlap.py [batch_log2 ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
u8 = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
for lb in [int(a) for a in sys.argv[1:]] or [24, 25, 26, 28]:
    for ov in (0, 1):
        for blk in (1024, 256):
            g.set_batch(1 << lb); g.set_option("overlap", ov); g.set_option("block_threads", blk)
            try:
                g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
                dt = (time.perf_counter() - t0) / 3
                print("batch 2^%d  overlap %d  block %4d: %.2f ms  (%d batches)" % (lb, ov, blk, dt * 1e3, st["batches"]), flush=True)
            except rt.RtError as e:
                print("batch 2^%d overlap %d block %d: %s" % (lb, ov, blk, e))
