#!/usr/bin/env python3
"""Static instruction census of one kernel in the compiler's assembly, per basic block.

usage: tools/isa_blocks.py file.s 'k_shadow<true>' [--min N]
For every label-delimited block: number of vector, scalar, LDS and memory instructions, the FP64 share of the
vector ones, and where its branches go -- enough to find the loops (node step, leaf test) and what runs once per chunk.
"""
import re
import subprocess
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            if dem.split("(")[0].replace("void ", "") == want:
                start = i
                break
    if start is None:
        sys.exit("kernel not found")
    blocks, cur = [], {"name": "entry", "ins": []}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "ins": []}
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    tot = {"v": 0, "s": 0}
    print("%-12s %5s %5s %5s %4s %4s %4s  branches" % ("block", "valu", "f64", "salu", "lds", "vmem", "smem"))
    for b in blocks:
        v = [x for x in b["ins"] if x.startswith("v_")]
        f64 = [x for x in v if "_f64" in x.split()[0]]
        s = [x for x in b["ins"] if x.startswith("s_") and not x.startswith(("s_load", "s_buffer", "s_waitcnt", "s_cbranch", "s_branch", "s_nop"))]
        lds = [x for x in b["ins"] if x.startswith("ds_")]
        vm = [x for x in b["ins"] if x.startswith(("global_", "buffer_", "flat_", "scratch_"))]
        sm = [x for x in b["ins"] if x.startswith(("s_load", "s_buffer"))]
        br = [x.split()[-1] for x in b["ins"] if x.startswith(("s_cbranch", "s_branch"))]
        tot["v"] += len(v); tot["s"] += len(s)
        if len(b["ins"]) >= min_n:
            print("%-12s %5d %5d %5d %4d %4d %4d  %s" % (b["name"], len(v), len(f64), len(s), len(lds), len(vm), len(sm), " ".join(br)))
    print("total valu %d salu %d blocks %d" % (tot["v"], tot["s"], len(blocks)))


if __name__ == "__main__":
    main()
