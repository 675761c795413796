"""Check the hand-pinned VGPRs of kernels_scan.hip: outside the inline-asm blocks no instruction of a scan kernel
(or of the device functions in that object) may name a VGPR at or above TA_PIN_BASE.

    python scripts/check_pinned.py path/to/kernels_scan-hip-amdgcn-amd-amdhsa-gfx950.s [base]
"""
import re, sys

path = sys.argv[1]
base = int(sys.argv[2]) if len(sys.argv) > 2 else 104
bad = []
in_asm = False
func = None
reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
for ln, line in enumerate(open(path), 1):
    t = line.strip()
    if t.startswith(";;#ASMSTART"):
        in_asm = True; continue
    if t.startswith(";;#ASMEND"):
        in_asm = False; continue
    m = re.match(r"^(_Z\w+):", line)
    if m:
        func = m.group(1); continue
    if in_asm or not t or t.startswith(";") or t.startswith("."):
        continue
    # the edge kernels (last template argument true) issue no hand-pinned loads: any register is theirs
    if func and re.search(r"scan_kernelI\w*Lb1EEEvNS_9SweepArgs", func):
        continue
    code = t.split(";")[0]
    for m in reg.finditer(code):
        hi = int(m.group(1)) if m.group(1) else int(m.group(3))
        if hi >= base:
            bad.append((ln, func, code))
            break
if bad:
    for b in bad[:20]:
        print("line %d in %s: %s" % b)
    sys.exit("%d instruction(s) outside the asm blocks touch v%d or above" % (len(bad), base))
print("ok: no compiler-allocated VGPR at or above v%d in %s" % (base, path))
