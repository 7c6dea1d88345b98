"""One 8K x 16 spp frame each with 2, 3 and 4 batches, for a kernel trace (per-launch durations of the primary pass).
usage: rocprofv3 --kernel-trace -d DIR -o trace --output-format csv -- python3 tools/exp_batches_trace.py; then
       python3 tools/exp_batches_trace.py --read DIR"""
import sys, os, glob, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for name in ("k_closest<true, true>", "k_closest<false, true>", "k_shadow<true>"):
        prim = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if name in r["Kernel_Name"]]
        print(name, "launches, ms:", " ".join("%.2f" % x for x in prim), " sum %.2f" % sum(prim))
    sys.exit(0)
import torch
import raytracer_amd as rt
s = rt.Scene.load("teapot-3"); g = rt.Renderer(0); g.upload(s); cam = s.camera(7680, 4320)
u8 = torch.zeros((4320, 7680, 3), dtype=torch.uint8, device="cuda")
g.set_memory_budget(200 << 30)
n_chunks = (4320 // 8) * (7680 // 8) * 16
for nb in (1, 2, 3, 4, 6):
    per = -(-n_chunks // nb)
    per = -(-per // 16) * 16
    g.set_batch(per * 64)
    st = g.render_device(cam, rgb_u8_ptr=u8.data_ptr(), num_samples=4)
    print(nb, st["batches"], st["gpu_ms"], flush=True)
