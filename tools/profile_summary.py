#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace result database into the per-round summary committed under profiles/:
    python tools/profile_summary.py gpurun_out/prof_r01 --steps 25 --round 1 \
        --bench gpurun_out/bench.json --bench-nooverlap gpurun_out/bench_noov.json --profiled gpurun_out/prof_bench.json
writes profiles/rNN_kernel_stats.csv (all kernels) and profiles/rNN_kernel_stats.md (top kernels + headline lines)."""
import argparse
import csv
import glob
import json
import os
import sqlite3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    with open(path) as f:
        lines = [l for l in f.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--steps", type=int, required=True, help="training steps inside the profiled run (warm-up + timed)")
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--bench")
    ap.add_argument("--bench-nooverlap")
    ap.add_argument("--profiled")
    ap.add_argument("--command", default="python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-parity --no-cpu-baseline")
    a = ap.parse_args()
    if a.trace_dir.endswith(".csv"):      # re-summarise an earlier run from its committed CSV (same columns as written below)
        with open(a.trace_dir) as f:
            rows = [(r["Name"], int(r["Calls"]), int(r["TotalDurationNs"]), int(r["MinNs"]), int(r["MaxNs"])) for r in csv.DictReader(f)]
        return summarise(a, rows)
    db = sorted(glob.glob(os.path.join(a.trace_dir, "**", "*_results.db"), recursive=True))[-1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d "
                     "join %s s on d.kernel_id=s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col)).fetchall()
    return summarise(a, rows)


# kernels of model CONSTRUCTION (they run once per process, not in a step): the fp64 composition of conv11 . conv12 and the
# calibration pass of the second-order weight rounding (patch moments: fp64 GEMMs over unfolded patches; rocSOLVER factorisation;
# the library's sequential rounding kernel)
ONCE = ("second_order_round_kernel", "rocsolver", "_DB_", "double", "im2col_kernel", "trsm", "potf2", "index_elementwise_kernel",
        "direct_copy_kernel", "triu_tril", "eye_", "randperm", "flip_kernel")


def summarise(a, rows):
    once_rows = [r for r in rows if any(k in r[0] for k in ONCE)]
    rows_all = rows
    rows = [r for r in rows if r not in once_rows]
    total = float(sum(r[2] for r in rows))
    tag = "r%02d" % a.round
    out_csv = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        tot_all = float(sum(r[2] for r in rows_all))
        for n, cnt, tot, mn, mx in rows_all:
            w.writerow([n, cnt, tot, round(tot / cnt, 1), round(100.0 * tot / tot_all, 3), mn, mx])
    md = ["# Round %d — rocprofv3 --kernel-trace --stats of `%s`" % (a.round, a.command), "",
          "MI355X (gfx950), 1 GPU, precision %s, B=8 clips x 35 frames x 224x224, side-stream stem pipeline on."
          % (load(a.bench).get("parity", {}).get("precision") or load(a.bench)["dtype"].split(" ")[0]),
          "%d steps profiled.  Raw CSV: `profiles/%s_kernel_stats.csv`; PMC HBM traffic of the dominant kernel: "
          "`profiles/%s_pmc_traffic.json`." % (a.steps, tag, tag), ""]

    def line(label, path):
        d = load(path)
        r, cfg = d["roofline"], d["config"]
        s = ("* %s: **%.1f clips/s, %.3f ms/step**, roofline.achieved %.1f TFLOP/s = %.1f%% of %d (avg launch %.4f ms); "
             "whole stem alone %.2f ms = %.1f%% MFMA utilisation; host enqueue %.2f ms/step"
             % (label, d["value"], d["ms_per_step"], r["achieved"], 100 * r["frac"], r["peak"], r["avg_launch_ms"],
                cfg["stem_alone_ms"], 100 * cfg["stem_alone_mfma_util"], cfg.get("host_enqueue_ms_per_step", float("nan"))))
        if "cpu_baseline" in d:
            s += "; cpu_baseline %.4f clips/s on %d threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
        return s
    if a.profiled:
        md.append(line("profiled run", a.profiled))
    if a.bench:
        md.append(line("un-profiled default run (`python bench.py`, profiles/%s_bench.json)" % tag, a.bench))
    if a.bench_nooverlap:
        md.append(line("same without the side-stream pipeline (`--no-overlap`, the stem kernel alone on the chip)", a.bench_nooverlap))
    md += ["", "The average duration of the `conv_ps_kernel<28, 2, 1>` row (the patch-stationary kernel's 5x5 instantiation: the composed",
           "conv11.conv12 — `conv_igemm_kernel<..., 256, 256, 2, 4, 1, 2>` where VNQA_COMPOSED_PS=0 or the geometry has no whole 2-D tiles) is the",
           "number `roofline.avg_launch_ms` must agree with (kernel durations are inflated when the trunk co-runs; the stem-alone",
           "passes bench.py runs after the timed region for `stem_alone_ms` are in this trace too).", "",
           "| kernel | calls/step | ms/step | avg µs | % GPU time |", "|---|---|---|---|---|"]
    for n, cnt, tot, mn, mx in rows[:40]:
        md.append("| `%s` | %.1f | %.3f | %.1f | %.1f |" % (n[:110], cnt / a.steps, tot / a.steps / 1e6, tot / cnt / 1e3,
                                                          100.0 * tot / total))
    md.append("")
    md.append("Sum over these kernels: %.3f ms/step of GPU time (two streams overlap, so this exceeds the wall time per step)."
              % (total / a.steps / 1e6))
    md += ["", "## One-time work of model construction (NOT in the table above, NOT in a step)", "",
           "The fp64 composition of conv11 and conv12 into the 5x5 kernel (stem.py `_compose_pair`) and the calibration pass of the "
           "second-order weight rounding (stem.py `second_order_round`: fp64 patch moments, one rocSOLVER factorisation + triangular solve "
           "and one `second_order_round_kernel` launch per layer), per stem built in the traced process (the bench builds several: the "
           "training stem, the parity legs'):", "", "| kernel | calls | total ms |", "|---|---|---|"]
    for n, cnt, tot, mn, mx in sorted(once_rows, key=lambda r: -r[2])[:12]:
        md.append("| `%s` | %d | %.2f |" % (n[:110], cnt, tot / 1e6))
    md.append("| all %d one-time kernel symbols | %d | %.2f |" % (len(once_rows), sum(r[1] for r in once_rows), sum(r[2] for r in once_rows) / 1e6))
    # framework kernels left in the step (everything that is not this library's): ATen elementwise / index / reduce kernels,
    # rocBLAS (Cijk_*) GEMMs, runtime copies
    fw = [(n, cnt, tot) for n, cnt, tot, mn, mx in rows if n.startswith("void at::") or n.startswith("Cijk_") or "rocclr" in n
          or n.startswith("at::") or "at::native" in n]
    md += ["", "## Framework (ATen / rocBLAS / runtime) kernels still launched", "",
           "| kernel | calls/step | us/step | % GPU time |", "|---|---|---|---|"]
    for n, cnt, tot in fw:
        md.append("| `%s` | %.2f | %.1f | %.3f |" % (n[:100], cnt / a.steps, tot / a.steps / 1e3, 100.0 * tot / total))
    md.append("")
    steady = fw
    md.append("Framework kernels (the one-time construction work is listed above, not here): "
              "%.1f launches/step, %.1f us/step = %.2f %% of the summed GPU kernel time — this still includes the fills / copies of "
              "model construction and of bench.py's warm-up allocation (the whole process is traced); the per-step list of the "
              "steady state is `profiles/%s_trunk_timeline.txt` (rocprofv3 trace of one training step's trunk chain: ~20 framework "
              "launches, ~100 us of its 4.6 ms)."
              % (sum(c for _, c, _ in steady) / a.steps, sum(t for _, _, t in steady) / a.steps / 1e3,
                 100.0 * sum(t for _, _, t in steady) / total, tag))
    with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.md"), "w") as f:
        f.write("\n".join(md) + "\n")
    print("wrote", out_csv)


if __name__ == "__main__":
    main()
