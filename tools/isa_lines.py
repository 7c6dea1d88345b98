#!/usr/bin/env python3
"""Static vector-instruction count of one kernel per source line.

usage: isa_lines.py <file.s> <kernel substring> [--top N]

The assembly must carry line tables (hipcc -gline-tables-only -S).  Every v_* instruction is booked on
the source line of the last .loc directive before it; inlined code is booked where it was written
(the callee's line), so a helper used three times shows the sum of its three copies.
"""
import re
import sys
from collections import Counter


def main():
    path, want = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 60
    files = {}
    per_line = Counter()
    per_file = Counter()
    inside = False
    cur = (0, 0)
    total = 0
    for line in open(path):
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if s.startswith(".type") and "@function" in s:
            inside = False
        if not inside:
            m = re.match(r"^([A-Za-z_][\w$.]*):", s)
            if m:
                name = m.group(1)
                try:
                    import subprocess
                    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
                except OSError:
                    dem = name
                inside = want in dem
            continue
        if s.startswith(".loc"):
            p = s.split()
            cur = (int(p[1]), int(p[2]))
            continue
        if s.startswith("s_endpgm"):
            inside = False
            continue
        if s.startswith("v_"):
            per_line[cur] += 1
            per_file[cur[0]] += 1
            total += 1
    print("total vector instructions %d" % total)
    for f, n in per_file.most_common():
        print("  %-28s %5d" % (files.get(f, str(f)), n))
    print()
    for (f, l), n in per_line.most_common(top):
        print("%-28s:%-5d %5d" % (files.get(f, str(f)), l, n))


if __name__ == "__main__":
    main()
