"""Per-kernel and per-position summary of a rocprofv3 kernel trace CSV (one process)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("span %.3f ms, sum of kernel durations %.3f ms, kernels %d" % ((t1 - t0) / 1e6, busy / 1e6, len(rows)))
gaps = []
for a, b in zip(rows, rows[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
gaps.sort()
print("gaps: median %.1f us, p90 %.1f us, sum %.3f ms" % (gaps[len(gaps)//2] / 1e3, gaps[int(len(gaps)*.9)] / 1e3, sum(g for g in gaps if g > 0) / 1e6))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0, 0])
    a[0] += 1; a[1] += d; a[2] = max(a[2], d)
for k, (n, tot, mx) in agg.items():
    print("%-28s n=%4d total %9.3f ms  avg %8.1f us  max %8.1f us" % (k, n, tot / 1e6, tot / n / 1e3, mx / 1e3))
if len(sys.argv) > 2:
    for r in rows[: int(sys.argv[2])]:
        print("%-28s %9.1f us  vgpr %s lds %s grid %s" % (r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"]))
