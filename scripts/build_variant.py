"""Build a VARIANT of libtissue_scan.so with extra -D flags (experiments, ablations, stamps) into scratch/:

    python scripts/build_variant.py NAME -DTA_STAMPS [-DTA_FCAP=128 ...]     ->  scratch/libNAME.so

Run anything against it with TISSUE_SCAN_LIB=$PWD/scratch/libNAME.so (the product never loads a variant by itself).
`--no-pin-check` skips the check that compiler-allocated registers stay clear of the hand-pinned ones (ablations whose
results are wrong by construction do not need it; anything timed AND compared must keep it)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tissue_analysis_amd import build as B      # noqa: E402

name = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D") or a.startswith("-m") or a.startswith("-f")]
out_dir = os.path.join(ROOT, "scratch", "variants", name)
os.makedirs(out_dir, exist_ok=True)
hipcc = B._hipcc()
procs, objs = [], []
for src in B.SOURCES:
    o = os.path.join(out_dir, src.replace(".hip", ".o"))
    objs.append(o)
    extra = [] if "--no-extra-flags" in sys.argv else B.EXTRA_FLAGS.get(src, [])        # (e.g. machine LICM back on)
    cmd = [hipcc] + B.FLAGS + extra + defs + ["-c", os.path.join(B.CSRC, src), "-o", o]
    if src == "kernels_scan.hip":
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=out_dir)))
for src, p in procs:
    text = p.communicate()[0].decode(errors="replace")
    if p.returncode:
        sys.exit("hipcc failed on %s:\n%s" % (src, text))
    if src == "kernels_scan.hip":
        open(os.path.join(out_dir, "resource_usage.txt"), "w").write(text)
if "--no-pin-check" not in sys.argv:
    saved = (B.OBJDIR, B.FLAGS)
    B.OBJDIR, B.FLAGS = out_dir, B.FLAGS + defs
    try:
        B._check_pinned(hipcc)
    finally:
        B.OBJDIR, B.FLAGS = saved
lib = os.path.join(ROOT, "scratch", "lib%s.so" % name)
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
