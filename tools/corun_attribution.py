#!/usr/bin/env python3
"""From two rocprofv3 --kernel-trace databases of the SAME bench.py workload — one overlapped (stem on its stream beside the trunk),
one with --no-overlap (every kernel alone on the chip) — where the overlapped step's time goes:

  * every kernel's ALONE duration (the serial run) and its CHIP SHARE: min(1, workgroups / (256 CUs x workgroups that fit a CU)),
    the latter from the dispatch's LDS bytes, VGPR count and workgroup size;
  * stem alone S = sum of the stem kernels' alone durations; trunk chip-time C = sum over the trunk's kernels of alone duration x share
    (what the trunk takes from the chip: a latency-bound 8-workgroup LSTM chain takes 3 %, a 980-workgroup conv everything);
  * work conservation says step >= S + C; what the measured step adds beyond that is interference proper (L2 / issue / dispatch).

  python tools/corun_attribution.py <overlapped trace dir> <no-overlap trace dir> [steps back from the end, default 2]"""
import glob
import re
import sqlite3
import sys

CUS, LDS_CU, VGPR_SIMD = 256, 160 * 1024, 512


def step_rows(path, back):
    db = sorted(glob.glob(path + '/**/*_results.db', recursive=True))[-1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    pick = lambda names, pool, pre: next((pre + n for n in names if n in pool), "0")
    gx, gy, gz = (pick(["grid_size_" + a, "grid_" + a], cols, "d.") for a in "xyz")
    wx, wy, wz = (pick(["workgroup_size_" + a, "workgroup_" + a], cols, "d.") for a in "xyz")
    lds = pick(["lds_block_size", "group_segment_size", "lds_size"], cols, "d.")
    if lds == "0":
        lds = pick(["group_segment_size", "lds_size"], scol, "s.")
    vg = pick(["arch_vgpr_count", "vgpr_count"], scol, "s.")
    ag = pick(["accum_vgpr_count", "agpr_count"], scol, "s.")
    rows = c.execute("select d.start, d.end, s.kernel_name, d.%s, %s, %s, %s, %s, %s, %s, %s, %s, %s from %s d join %s s on d.kernel_id=s.id "
                     "order by d.start" % (q, gx, gy, gz, wx, wy, wz, lds, vg, ag, kd, ks)).fetchall()
    adam = [i for i, r in enumerate(rows) if "clip_adam" in r[2]]
    return rows[adam[-back - 1] + 1:adam[-back] + 1]


def short(n):
    n = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n)
    return re.sub(r"^_ZN2at6native\d*", "at::", n)[:64]


def share(r):
    _, _, _, _, gx, gy, gz, wx, wy, wz, lds, vg, ag = r
    wg_threads = max(1, (wx or 1) * (wy or 1) * (wz or 1))
    n_wg = max(1, ((gx or 1) * (gy or 1) * (gz or 1)) // wg_threads)
    waves = (wg_threads + 63) // 64
    regs = max(1, (vg or 0) + (ag or 0))
    waves_per_simd = max(1, min(8, VGPR_SIMD // max(regs, 64)))
    by_regs = max(1, (4 * waves_per_simd) // waves)
    by_lds = max(1, LDS_CU // lds) if lds else 32
    per_cu = max(1, min(by_regs, by_lds, 32 // waves if waves <= 32 else 1))
    return min(1.0, n_wg / float(CUS * per_cu)), n_wg


def main():
    over, serial = sys.argv[1], sys.argv[2]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ov, se = step_rows(over, back), step_rows(serial, back)
    # the stem's queue in the overlapped run = the queue of the fused first conv; every kernel name found there is a stem kernel
    stem_q = next(r[3] for r in ov if "conv_first" in r[2] or "clip_to_nhwc4" in r[2])
    stem_names = set(r[2] for r in ov if r[3] == stem_q)
    t0 = ov[0][0]
    step_ms = (ov[-1][1] - t0) / 1e6
    S = sum(r[1] - r[0] for r in se if r[2] in stem_names) / 1e6
    S_over = sum(r[1] - r[0] for r in ov if r[3] == stem_q) / 1e6
    chain, chip, agg = 0.0, 0.0, {}
    for r in se:
        if r[2] in stem_names:
            continue
        d = (r[1] - r[0]) / 1e6
        f, n_wg = share(r)
        chain += d
        chip += d * f
        a = agg.setdefault(short(r[2]), [0, 0.0, 0.0, f, n_wg])
        a[0] += 1
        a[1] += d
        a[2] += d * f
    serial_ms = (se[-1][1] - se[0][0]) / 1e6
    print("overlapped step %.3f ms (%d kernels); serial step %.3f ms (%d kernels)" % (step_ms, len(ov), serial_ms, len(se)))
    print("stem alone S = %.3f ms (its queue is busy %.3f ms in the overlapped step: x %.2f)" % (S, S_over, S_over / S))
    print("trunk chain alone = %.3f ms of kernel time, of which chip-time C = %.3f ms (alone duration x chip share)" % (chain, chip))
    print("work conservation: step >= S + C = %.3f ms; measured %.3f ms: %.3f ms beyond it (%.1f %%)" %
          (S + chip, step_ms, step_ms - S - chip, 100 * (step_ms - S - chip) / step_ms))
    # the stem's kernels: alone (serial run) vs beside the trunk (overlapped run), by kernel name
    al, ov_ = {}, {}
    for r in se:
        if r[2] in stem_names:
            a = al.setdefault(short(r[2]), [0, 0.0]); a[0] += 1; a[1] += (r[1] - r[0]) / 1e6
    for r in ov:
        if r[3] == stem_q:
            a = ov_.setdefault(short(r[2]), [0, 0.0]); a[0] += 1; a[1] += (r[1] - r[0]) / 1e6
    print("\n%-66s %5s %9s %10s %8s" % ("stem kernel", "calls", "alone ms", "co-run ms", "stretch"))
    for k, a in sorted(al.items(), key=lambda kv: -kv[1][1])[:14]:
        o = ov_.get(k, [0, 0.0])
        print("%-66s %5d %9.3f %10.3f %8.2f" % (k, a[0], a[1], o[1], o[1] / a[1] if a[1] > 0 else 0.0))
    print("\n%-66s %5s %9s %9s %6s %7s" % ("trunk kernel (serial run)", "calls", "alone ms", "chip ms", "share", "WGs"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:24]:
        print("%-66s %5d %9.3f %9.3f %6.2f %7d" % (k, a[0], a[1], a[2], a[3], a[4]))


if __name__ == "__main__":
    main()
