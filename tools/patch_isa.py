"""Re-encodes instructions in the compiler's gfx950 assembly whose short encoding issues more slowly than the long one.

v_cndmask_b32 in the VOP2 encoding (mask implicitly in VCC) occupies a SIMD for 16 cycles per wave-instruction on
MI355X, in the VOP3 encoding (same operands, mask named explicitly, VCC included) for 4.3 (tools/micro/valu_rates.hip).
The compiler prefers the short form wherever the mask is in VCC.  VOP3 takes no 32-bit literal, so selects whose
first source is a literal stay as they are.
usage: patch_isa.py in.s out.s"""
import re
import sys

INLINE = re.compile(r"^(v\d+|s\d+|-?\d+|-?0\.5|-?1\.0|-?2\.0|-?4\.0|vcc_lo|vcc_hi|exec_lo|exec_hi|m0)$")
pat = re.compile(r"^(\s*)v_cndmask_b32_e32\s+(v\d+),\s*([^,]+),\s*(v\d+),\s*vcc(\s*(;.*)?)$")
n = kept = 0
out = []
for line in open(sys.argv[1]):
    m = pat.match(line.rstrip("\n"))
    if m:
        src0 = m.group(3).strip()
        ok = bool(INLINE.match(src0))
        if ok and re.match(r"^-?\d+$", src0):
            ok = -16 <= int(src0) <= 64
        if ok:
            line = "%sv_cndmask_b32_e64 %s, %s, %s, vcc%s\n" % (m.group(1), m.group(2), src0, m.group(4), m.group(5) or "")
            n += 1
        else:
            kept += 1
    out.append(line)
open(sys.argv[2], "w").writelines(out)
print("patch_isa: %d v_cndmask_b32 re-encoded as VOP3, %d left (literal operand)" % (n, kept))
