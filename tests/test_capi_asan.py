"""CPU: host-side AddressSanitizer build of the C ABI + negative-argument sweep over every entry point (SURVEY 5 sanitizer
row).  No GPU: every call must be refused by its VNQA_CHECK_ARG prologue before any HIP call."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to build the ASan variant")
def test_every_entry_point_rejects_zeroed_arguments_under_host_asan():
    from videonavqa_amd import _lib as L
    from videonavqa_amd.build import asan_runtime, build
    lib = build(variant="asan", verbose=False)          # stamped: rebuilt only when a source changed
    rt = asan_runtime()
    assert os.path.exists(lib) and os.path.exists(rt), (lib, rt)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", VNQA_LIB=lib, HIP_VISIBLE_DEVICES="",
               CUDA_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native", "abi_negative.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [x for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert lines, (r.stdout[-500:], r.stderr[-2000:])
    d = json.loads(lines[-1])
    assert d["entry_points"] == len(L.exported_symbols())          # the sweep covered the whole header
    assert d["offenders"] == [], d["offenders"]
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0
    # spot checks of the contract: status entry points answer VNQA_ERR_INVALID_ARG (-1) for NULL and zeroed descriptors
    assert d["codes"]["vnqa_conv2d_igemm_fwd"] == [-1, -1]
    assert d["codes"]["vnqa_clip_adam_step"][0] == -1 if "vnqa_clip_adam_step" in d["codes"] else True
