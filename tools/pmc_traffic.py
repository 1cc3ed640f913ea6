#!/usr/bin/env python3
"""HBM bytes per launch of the frozen stem's C_out = 512 kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcB_FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcB_WRITE_SIZE -- python3 bench.py ... (same)
    python tools/pmc_traffic.py gpurun_out/pmcB_FETCH_SIZE gpurun_out/pmcB_WRITE_SIZE > profiles/rNN_pmc_traffic.json
Units/corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact.
bench.py reads kernels[<dominant kernel>].hbm_bytes_per_launch for roofline.traffic."""
import csv
import glob
import json
import sys

KERNELS = {"conv_igemm_kernel": ("conv_igemm_kernel<unsigned short, 256, 256, 2, 4, 1, 2>",
                                 "frozen-stem igemm: the composed conv11.conv12 (5x5, 128 -> 512 on 56x56 maps, pool)"),
           "conv_ps_kernel<28,5x5>": ("conv_ps_kernel<28, 2, 1>",
                                      "patch-stationary 5x5 conv on 8 x 28-pixel tiles: the composed conv11.conv12 (128 -> 512 on 56x56 maps, pool)"),
           "conv_ps_kernel<28>": ("conv_ps_kernel<28, 1,", "patch-stationary 3x3 conv: conv21, conv22 (28x28 maps, conv22 pooled)"),
           "conv_ps_kernel<14>": ("conv_ps_kernel<14,", "patch-stationary 3x3 conv: conv31, conv32 (14x14 maps)")}
# unique bytes one 280-frame launch must move (padded 16-bit inputs + weights + outputs), averaged over the layers a symbol serves
N = 280
ALGO = {"conv_igemm_kernel": N * 60 * 60 * 128 * 2 + 512 * 25 * 128 * 2 + N * 30 * 30 * 512 * 2,
        "conv_ps_kernel<28,5x5>": N * 60 * 60 * 128 * 2 + 512 * 25 * 128 * 2 + N * 30 * 30 * 512 * 2,
        "conv_ps_kernel<28>": (2 * (N * 30 * 30 * 512 * 2 + 512 * 9 * 512 * 2) + N * 30 * 30 * 512 * 2 + N * 16 * 16 * 512 * 2) / 2.0,
        "conv_ps_kernel<14>": (2 * (N * 16 * 16 * 512 * 2 + 512 * 9 * 512 * 2) + 2 * N * 16 * 16 * 512 * 2) / 2.0}


def per_dispatch(d, counter, pat):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
            out.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return out


res = {"command": sys.argv[3] if len(sys.argv) > 3 else
       "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
       "--repeats 1 --no-parity --no-cpu-baseline --no-overlap (two separate passes)",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM); "
                     "WRITE_SIZE exact; both in KiB", "kernels": {}}
for key, (pat, what) in KERNELS.items():
    fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE", pat), per_dispatch(sys.argv[2], "WRITE_SIZE", pat)
    if key == "conv_igemm_kernel":      # the composed conv's launches only (the symbol also serves the small border-edge convs)
        fetch = {g: v for g, v in fetch.items() if g >= 1000000}
        write = {g: v for g, v in write.items() if g >= 1000000}
    n = sum(len(v) for v in fetch.values())
    if n == 0:
        continue
    f_avg = sum(sum(v) for v in fetch.values()) / n
    w_avg = sum(sum(v) for v in write.values()) / max(sum(len(v) for v in write.values()), 1)
    total = int((2 * f_avg + w_avg) * 1024)
    res["kernels"][key] = {
        "what": what, "dispatches": n, "FETCH_SIZE_KiB_raw_avg": round(f_avg, 1), "WRITE_SIZE_KiB_avg": round(w_avg, 1),
        "hbm_read_bytes_per_launch": int(2 * f_avg * 1024), "hbm_write_bytes_per_launch": int(w_avg * 1024),
        "hbm_bytes_per_launch": total, "algorithmic_bytes_per_launch": int(ALGO[key]),
        "traffic_over_algorithmic": round(total / ALGO[key], 2),
        "per_grid": {str(g): {"launches": len(fetch[g]), "FETCH_SIZE_KiB": sum(fetch[g]) / len(fetch[g]),
                              "WRITE_SIZE_KiB": sum(write.get(g, [0])) / max(len(write.get(g, [0])), 1)} for g in sorted(fetch, reverse=True)}}
if "conv_igemm_kernel" in res["kernels"]:
    res["hbm_bytes_per_launch"] = res["kernels"]["conv_igemm_kernel"]["hbm_bytes_per_launch"]      # (round 1-2 readers)
print(json.dumps(res, indent=1))
