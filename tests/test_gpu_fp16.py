"""GPU: the fp16-storage build of the library (libvnqa_hip_f16.so — the same kernel sources with IEEE fp16 as the 16-bit
format, csrc/vnqa_common.h) under the SAME tests as the bf16 build.

One 16-bit storage format per process, so the suite is re-run in a child pytest process with
VNQA_TEST_LOW_PRECISION=fp16: every test parametrised over (fp32, LOW) / LOW_DTYPE then builds fp16 tensors, fp16 models and
loads the fp16 library: kernel-level tests against PyTorch fp32 references, the reference goldens, the fused epilogues, the
trainer, and the full-size parity of the headline configuration (fp16 logits within 1.1e-3 of the exact-f32 precision —
measured 0.81e-3 with the coherently rounded stem — with the loss-scaled backward's flat gradient within 2 %; measured 0.6 %)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, timeout):
    if os.environ.get("VNQA_TEST_LOW_PRECISION") == "fp16":
        pytest.skip("already inside the fp16 child run")
    env = dict(os.environ, VNQA_TEST_LOW_PRECISION="fp16", VNQA_HALF="f16")      # (the fp32-only tests of the child load the f16 build too)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = "\n".join(r.stdout.strip().splitlines()[-25:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-1500:]
    return tail


def test_kernel_and_model_suites_on_the_fp16_storage_build():
    tail = _run(["tests/test_gpu_conv.py", "tests/test_gpu_fused_epilogue.py", "tests/test_gpu_glue.py", "tests/test_gpu_models.py",
                 "tests/test_gpu_trainer.py", "tests/test_weight_import.py", "tests/test_gpu_edge_cases.py", "tests/test_gpu_conv3d.py"], 1500)
    assert " passed" in tail, tail


def test_full_size_parity_of_the_headline_config_on_the_fp16_storage_build():
    tail = _run(["tests/test_gpu_fullsize.py", "-k", "config4_film_attn or stem_vs_torch"], 1500)
    assert " passed" in tail, tail


def test_split_tensors_and_the_fp16h_precision_on_the_fp16_build():
    """precision='fp16h' (tests/test_gpu_fp16h.py): the dual-output epilogue, the three-product conv on split tensors, the split
    weight gradient, the goldens, and north star's 1e-3 at full size on 4 weight seeds x 12 minibatches + smooth data."""
    tail = _run(["tests/test_gpu_fp16h.py"], 2400)
    assert " passed" in tail and "skipped" not in tail.splitlines()[-1], tail
