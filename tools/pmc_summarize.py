"""Sums rocprofv3 --pmc CSV output per kernel name and counter.

Writes gpurun_out/pmc_<tag>_summary.txt (readable) and gpurun_out/pmc_<tag>.json (per kernel: dispatches,
counter totals, and the derived per-launch HBM bytes: FETCH_SIZE and WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE tallies the 128-byte requests of wide loads at 64 bytes, so read bytes are doubled)."""
import csv, glob, json, os, subprocess, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sorted(glob.glob("gpurun_out/pmc_%s_*" % tag)):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[(k, os.path.basename(d))].add(row["Dispatch_Id"])
out = []
js = {}
for k in sorted(agg):
    if not (k.startswith("void k_") or k.startswith("k_")):
        continue
    n = max(len(v) for (kk, _), v in calls.items() if kk == k)
    out.append("%s  dispatches=%d" % (k, n))
    c = agg[k]
    e = {"dispatches": n, "counters": dict(c)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024 / n
        e["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 / n
    if c.get("TCC_REQ_sum"):
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"]
    if c.get("SQ_WAVE_CYCLES"):
        e["valu_active_share_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
        e["wait_share_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
    js[k.replace("void ", "")] = e
    for c in sorted(agg[k]):
        out.append("    %-34s %.6g" % (c, agg[k][c]))
# the build the counters belong to: bench.py refuses counters of another build
import raytracer_amd as rt
meta = {"build_id": rt.build_id(), "workload": workload}
try:
    meta["git"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    meta["git"] = None
js["_meta"] = meta
out.insert(0, "build %s  workload %s" % (meta["build_id"], workload))
txt = "\n".join(out)
print(txt)
open("gpurun_out/pmc_%s_summary.txt" % tag, "w").write(txt + "\n")
json.dump(js, open("gpurun_out/pmc_%s.json" % tag, "w"), indent=1, sort_keys=True)
