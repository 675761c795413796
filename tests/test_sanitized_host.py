"""Host hardening: the C oracle and the host side of the C ABI under AddressSanitizer +
UndefinedBehaviorSanitizer, in this CPU container.  The device side cannot be instrumented on this pool; what runs
here is every entry point's argument validation, ta_ctx_create's failure paths (no GPU -> TA_ENODEVICE after partial
construction) and the C restatement on small / ragged / degenerate volumes.  The sanitizer runtime has to be the
first library of the process, hence the child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sanitized():
    from tissue_analysis_amd import build as ta_build
    from oracle import onepass_c
    try:
        rt = ta_build.sanitizer_runtime()
        if not os.path.exists(rt):
            pytest.skip("no AddressSanitizer runtime in this toolchain")
        lib = ta_build.build_sanitized()
        clang = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(rt))))), "bin", "clang")
        if not os.path.exists(clang):
            clang = "/opt/rocm/lib/llvm/bin/clang"
        oracle_lib = onepass_c.build_sanitized(clang)
    except (OSError, subprocess.CalledProcessError, RuntimeError) as e:
        pytest.fail("sanitized build failed: %s" % e)
    return rt, lib, oracle_lib


def test_c_abi_validation_and_c_oracle_are_clean_under_asan_ubsan(sanitized):
    rt, lib, oracle_lib = sanitized
    env = dict(os.environ, LD_PRELOAD=rt, TISSUE_SCAN_LIB=lib, ONEPASS_ORACLE_LIB=oracle_lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",     # (CPython itself never frees everything)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitized_child.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0, out[-2000:] + err[-4000:]
    assert "SANITIZED-OK" in out
    assert "Sanitizer" not in err and "runtime error" not in err, err[-4000:]


def test_failed_ctx_create_leaks_nothing(sanitized, tmp_path):
    """tests/native/ctx_create_leak.c against the sanitized library with LeakSanitizer on (a C program: the leak
    report of a Python process would be CPython's own)."""
    rt, lib, _ = sanitized
    clang = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(rt))))), "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    exe = str(tmp_path / "ctx_create_leak")
    subprocess.check_call([clang, "-g", "-O1", "-fsanitize=address,undefined", "-shared-libsan",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "ctx_create_leak.c"),
                           "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.dirname(rt)])
    p = subprocess.run([exe], env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-3000:]
    assert "LeakSanitizer" not in err and "AddressSanitizer" not in err, err[-3000:]
