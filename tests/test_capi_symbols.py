"""The C-ABI library loads here (no GPU) and exports every symbol include/tissue_scan.h declares;
without a GPU the product fails loudly instead of falling back to a CPU path."""
import os
import re

import pytest

from tissue_analysis_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tissue_scan.h")).read()
    return sorted(set(re.findall(r"TA_API\s+(?:const\s+char\s*\*|int)\s+(ta_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.ta_version() == _capi.ABI_VERSION == 5


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _capi.load()
    assert lib.ta_ctx_destroy(None) == _capi.TA_OK
    assert lib.ta_extract(None, 31, 10) == _capi.TA_EINVAL
    assert b"NULL" in lib.ta_last_error()


def test_no_cpu_fallback_when_there_is_no_gpu():
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    from tissue_analysis_amd import SpatialImageAnalysis
    with pytest.raises(_capi.TissueScanError) as e:
        SpatialImageAnalysis(np.ones((4, 4, 4), dtype=np.uint16), background=1)
    assert e.value.code == _capi.TA_ENODEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tissue_analysis_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_pinned_register_tables_are_the_generators_output():
    """csrc/ta_pin_tables.inc (the register numbers the plane in flight lands in, literals inside inline asm) is generated:
    the committed file must be what scripts/gen_pin_tables.py writes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.call([sys.executable, os.path.join(root, "scripts", "gen_pin_tables.py"), "--check"]) == 0
