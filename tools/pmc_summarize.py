"""Sums rocprofv3 --pmc CSV output per kernel name and counter."""
import csv, glob, os, sys, collections
tag = sys.argv[1]
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
for k in sorted(agg):
    if not (k.startswith("void k_") or k.startswith("k_")):
        continue
    n = max(len(v) for (kk, _), v in calls.items() if kk == k)
    out.append("%s  dispatches=%d" % (k, n))
    for c in sorted(agg[k]):
        out.append("    %-34s %.6g" % (c, agg[k][c]))
txt = "\n".join(out)
print(txt)
open("gpurun_out/pmc_%s_summary.txt" % tag, "w").write(txt + "\n")
