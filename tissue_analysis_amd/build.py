"""Build recipe for libtissue_scan.so (hipcc, gfx950 only, in-tree).

    python -m tissue_analysis_amd.build [--force] [--save-temps]

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libtissue_scan.so")
SOURCES = ["ta_api.hip", "kernels_basic.hip", "kernels_scan.hip", "kernels_walls.hip", "kernels_wallsort.hip", "kernels_wallmedian.hip", "kernels_census.hip",
           "kernels_pairsort.hip"]
HEADERS = ["ta_device.h", "ta_kernels.h", "ta_sweep_common.h", "ta_sweep_switches.h", "ta_pin_tables.inc", os.path.join("..", "..", "include", "tissue_scan.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-DTA_BUILD"]
# Per-source extra flags.  (The sweep kernels live on a hand-set VGPR budget -- the plane in flight is pinned above it -- and
# machine LICM, which cannot see that, hoists constant materialisations for the drains' LDS atomics out of the plane loop:
# a build in which that breaks the budget is refused by _check_pinned; `-mllvm -disable-machine-licm` for kernels_scan.hip
# is the way out that was needed for one of round 4's variants and measured 0 - 2 % slower.  Not needed by the code as it is.)
EXTRA_FLAGS = {}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _check_pinned(hipcc, verbose=False):
    """kernels_scan.hip keeps the plane in flight in 21 VGPRs it names itself, above an amdgpu_num_vgpr budget
    that the compiler treats as a request, not a limit: refuse a build in which compiler-allocated code of an
    interior kernel (or of a device function they call) reaches them."""
    import re
    asm = os.path.join(OBJDIR, "kernels_scan.check.s")
    cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get("kernels_scan.hip", []) + ["--cuda-device-only", "-S", os.path.join(CSRC, "kernels_scan.hip"), "-o", asm]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=OBJDIR)
    bases = {}
    for line in open(os.path.join(CSRC, "kernels_scan.hip")):
        m = re.match(r"#define TA_PIN_(ADJ8_PAD|ADJ8|ADJ2_PAD|ADJ_PAD|MOM_PAD|ADJ2|ADJ|MOM) (\d+)", line)
        if m:
            bases[m.group(1)] = int(m.group(2))
    reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    in_asm, func, bad = False, "", []
    for ln, line in enumerate(open(asm), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        else:
            m = re.match(r"^(_Z\w+):", line)
            if m:
                func = m.group(1)
                continue
            if in_asm or not t or t[0] in ";.":
                continue
            if re.search(r"scan_(noadj_|wide_)?kernelI\w*Lb1EEEvNS_9SweepArgs", func):
                continue              # edge kernels issue no hand-pinned loads: any register is theirs
            if "scan_wide_pad_kernel" in func:
                ranges = [bases["ADJ8_PAD"], bases["ADJ8_PAD"] + 4]      # (25 registers: two overlapping windows of 21)
            elif "scan_wide_kernel" in func:
                ranges = [bases["ADJ8"], bases["ADJ8"] + 4]
            elif "scan_noadj_pad_kernel" in func:
                ranges = [bases["MOM_PAD"]]
            elif "scan_pad2_kernel" in func:
                ranges = [bases["ADJ2_PAD"]]
            elif "scan_pad_kernel" in func:
                ranges = [bases["ADJ_PAD"]]
            elif "scan_noadj_kernel" in func:
                ranges = [bases["MOM"]]
            elif "scan_two_rows_kernel" in func:
                ranges = [bases["ADJ2"], bases["ADJ2"] + 6]    # (+6: the second landing zone of TA_PLANES_IN_FLIGHT=2, v[96:108])
            elif "scan_kernel" in func:
                ranges = [bases["ADJ"]]
            else:
                ranges = list(bases.values())      # a device function: callable from any of them
            for m in reg.finditer(t.split(";")[0]):
                lo = int(m.group(1) or m.group(2)); hi = int(m.group(1) or m.group(3))
                if any(hi >= b and lo < b + 21 for b in ranges):      # (21 pinned with adjacency, 16 without: the wider check is safe)
                    bad.append("%s:%d: %s" % (func, ln, t))
                    break
    if bad or len(bases) != 8:
        raise RuntimeError("compiler-allocated VGPRs reach the hand-pinned registers in kernels_scan.hip:\n  %s"
                           % "\n  ".join(bad[:10]))


def build(force=False, save_temps=False, verbose=False):
    """Compile every HIP source for gfx950 and link libtissue_scan.so. Returns its path."""
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    procs, objs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if save_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                                cwd=OBJDIR)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode(errors="replace")
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s:\n%s\n" % (src, out))
        elif out.strip() and (verbose or save_temps):
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    stamp = os.path.join(OBJDIR, "kernels_scan.check.ok")       # (a failed check must not be skipped by the next call)
    if _stale(stamp, [os.path.join(OBJDIR, "kernels_scan.o")]):
        _check_pinned(hipcc, verbose)
        open(stamp, "w").write("ok\n")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


ASAN_DIR = os.path.join(OBJDIR, "asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libtissue_scan_asan.so")


def sanitizer_runtime():
    """Path of the AddressSanitizer runtime hipcc's clang links with -shared-libsan (to LD_PRELOAD under python)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
    return subprocess.check_output([clang, "-print-file-name=libclang_rt.asan-x86_64.so"]).decode().strip()


def build_sanitized(force=False, verbose=False):
    """HOST side of the same three sources under -fsanitize=address,undefined (the device side cannot be
    instrumented on this pool and is compiled as usual): the library the CPU suite loads to walk the C ABI's
    argument checks and failure paths.  Never the product build."""
    hipcc = _hipcc()
    os.makedirs(ASAN_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    flags = [f for f in FLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer",
                                                "-fsanitize=address,undefined", "-Wno-option-ignored"]
    procs, objs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(ASAN_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ASAN_DIR)))
    for src, p in procs:
        out = p.communicate()[0].decode(errors="replace")
        if p.returncode != 0:
            raise RuntimeError("hipcc (sanitized) failed on %s:\n%s" % (src, out))
    if force or procs or _stale(ASAN_LIB, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined",
                               "-shared-libsan", "-o", ASAN_LIB] + objs)
    return ASAN_LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_sanitized(force="--force" in sys.argv, verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True))
