"""Static instruction mix of one kernel in a saved assembly file (scratch/variants/NAME/kernels_scan-hip-amdgcn-amd-amdhsa-gfx950.s):
    python scripts/asm_mix.py FILE [mangled-name-substring]"""
import collections, sys
asm = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else "scan_two_rows_kernelIjLi4ELi2ELb1E"
start = [i for i in range(len(asm)) if asm.startswith("_ZN2ta", i) and asm[i:i + 200].split(":")[0].find(want) >= 0 and asm[i - 1] == "\n"][0]
end = asm.index(".Lfunc_end", start)
cats = collections.Counter()
for l in asm[start:end].split("\n"):
    t = l.strip()
    if not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    cats["VALU" if op.startswith("v_") else "BRANCH" if op.startswith(("s_cbranch", "s_branch")) else "WAIT" if op.startswith("s_waitcnt")
         else "SMEM" if op.startswith(("s_load", "s_buffer")) else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_")
         else "VMEM"] += 1
print(sum(cats.values()), dict(cats))
