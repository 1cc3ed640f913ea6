"""CPU: host-side pieces of bench.py — FLOP accounting, the CPU-baseline child, the self-spawning `--gpus N` guard."""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
ENV = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")


def test_flop_accounting_matches_survey_8d():
    sys.path.insert(0, ROOT)
    import bench as Bn
    assert abs(Bn.stem_flops_per_frame(224, 224) / 1e9 - 37.167) < 2e-3
    assert abs(Bn.stem_flops_per_frame(160, 208) / 1e9 - 24.652) < 2e-3
    fwd, fb = Bn.trunk_flops_per_frame(196, 512, 512, 1, 128)
    assert abs(fwd / 1e9 - 1.978) < 2e-3
    clip = 35 * (Bn.stem_flops_per_frame(224, 224) + fb)
    assert abs(clip / 1e9 - 1476.2) < 0.2
    # the composed conv11 . conv12 pair executes fewer FLOPs than the reference formulation, never more
    assert Bn.stem_executed_flops_per_frame(224, 224, True) < Bn.stem_flops_per_frame(224, 224)
    assert Bn.stem_executed_flops_per_frame(224, 224, False) == Bn.stem_flops_per_frame(224, 224)


def test_cpu_baseline_child_prints_cumulative_lines():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-batch", "2",
                        "--frames", "3", "--height", "32", "--width", "48", "--cpu-steps", "2"], env=ENV,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
    assert len(lines) == 2
    for i, leg in enumerate(lines, 1):
        assert leg["kind"] == "port" and leg["unit"] == "clips/s" and leg["value"] > 0 and leg["cores"] >= 1
        assert "%d timed step(s) after 1 warm-up" % i in leg["sample"] and "2 clips x 3 frames 32x48" in leg["sample"]


def test_gpus_n_without_launcher_refuses_when_gpus_are_missing():
    """`bench.py --gpus 2` starts its own ranks; with fewer GPUs than ranks it must say so instead of silently
    running world=1 and printing n_gpus: 1 (missing #3)."""
    env = {k: v for k, v in ENV.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only 0 GPU(s) visible" in r.stderr and "{" not in r.stdout
