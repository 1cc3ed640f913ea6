#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv of tools/bench_cnn3d.py + its bench lines -> the config-2 profile page (markdown on stdout)."""
import csv, json, sys

f, bench, bench_generic, rnd = sys.argv[1:5]
d, dg = json.load(open(bench)), json.load(open(bench_generic))
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
steps = 13.0                                   # 3 warm-up + 10 timed
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
B = 32
conv2 = 2.0 * B * 16 * 56 * 56 * 64 * 128 * 27 / 1e9          # GFLOP of one pass over conv2 (fwd = dgrad = wgrad)
conv3 = 2.0 * B * 4 * 14 * 14 * 128 * 128 * 27 / 1e9
conv1 = 2.0 * B * 16 * 112 * 112 * 3 * 64 * 27 / 1e9
PEAK = 2500.0
print("# Round %s — BASELINE config 2: `v_only_cnn3d`, bs = 32, 16x3x112x112 clips, fwd + bwd + Adam (`tools/bench_cnn3d.py`)\n" % rnd)
print("* fused 16-bit path (csrc/cnn3d.hip + 27-tap igemm + small-channel wgrad): **%.0f clips/s, %.2f ms/step** un-profiled"
      % (d["clips_per_s"], d["ms_per_step"]))
print("* generic path (`VNQA_CNN3D_GENERIC=1`: padded-channel igemm / wgrad for every conv, stock BatchNorm3d / MaxPool3d — the round-2 "
      "bring-up): %.0f clips/s, %.2f ms/step" % (dg["clips_per_s"], dg["ms_per_step"]))
print("* algorithmic conv FLOPs per step: conv1 3 x %.0f GF, conv2 3 x %.0f GF, conv3a 3 x %.0f GF = %.2f TFLOP; whole step = %.0f TFLOP/s "
      "(%.1f %% of the dense bf16 peak) — the step is %.2f ms of kernels, of which the three conv2 GEMMs are the MFMA-bound part and the "
      "rest is HBM-bound passes over the 0.2 - 1 GB activations\n"
      % (conv1, conv2, conv3, 3 * (conv1 + conv2 + conv3) / 1e3, 3 * (conv1 + conv2 + conv3) / d["ms_per_step"], 
         100 * 3 * (conv1 + conv2 + conv3) / d["ms_per_step"] / PEAK, tot))


def find(sub):
    return [r for r in rows if sub in r["Name"]]


print("## roofline of the three conv2 GEMMs (64 -> 128 channels on 32 x 16 x 56 x 56 positions, %.0f GFLOP each)\n" % conv2)
print("| pass | kernel | avg launch µs (largest launch) | TFLOP/s | fraction of 2.5 PF |\n|---|---|---|---|---|")
for label, sub in (("forward", "conv_igemm_kernel<unsigned short, 512, 128"), ("dgrad", "conv_igemm_kernel<unsigned short, 256, 64"),
                   ("wgrad", "conv_wgrad_kernel<unsigned short, true>")):
    for r in find(sub):
        mx = float(r.get("MaxNs", r["AverageNs"])) / 1e3
        tf = conv2 / mx * 1e3                      # GFLOP / us = PFLOP/s
        print("| %s | `%s` | %.0f | %.0f | %.3f |" % (label, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60], mx,
                                                      tf, tf / PEAK))
print("\n(the same kernels also serve conv3a's 22-GFLOP passes; the largest launch of each name is conv2's)\n")
foreign = [r for r in rows if r["Name"].lstrip().startswith(("void at::native", "at::native", "Cijk_", "MIOpen", "void at::cuda"))]
ft = sum(float(r["TotalDurationNs"]) for r in foreign) / 1e6 / steps
print("kernels not from this library (Adam's `multi_tensor_apply`, fills, the CE loss): %.3f ms/step = %.1f %% of kernel time\n" % (ft, 100 * ft / tot))
print("## kernels by time (%.2f ms/step of kernel time, %d launches/step)\n" % (tot, sum(int(r["Calls"]) for r in rows) / steps))
print("| kernel | calls/step | ms/step | avg µs |\n|---|---|---|---|")
for r in rows[:24]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "h16")
    print("| `%s` | %.1f | %.3f | %.1f |" % (n[:90], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
