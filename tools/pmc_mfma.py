#!/usr/bin/env python3
"""MFMA utilisation, issue/wait split and LDS bank conflicts per kernel from two rocprofv3 --pmc passes over the frozen stem:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/pmcM -- python3 tools/stem_only.py --iters 5
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmcS -- python3 tools/stem_only.py --iters 5
    python tools/pmc_mfma.py /tmp/pmcM /tmp/pmcS > profiles/r02_pmc_mfma.json
Formulas (MI355X_MICROARCH.md, rocprofv3 PMC slots / DVFS): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs, so the
kernel's active cycles = GUI/8 and the effective clock = GUI/8/duration; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs, so
mfma_util = MFMA_BUSY / (GUI/8 * 256 CUs * 4 SIMDs); the SQ wait counters are quad-cycles and are only used as ratios."""
import collections
import csv
import glob
import json
import re
import sys

CUS, SIMDS = 256, 4


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("unsigned short", "h16").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)[:96]


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return acc, dur


def avg(v):
    return sum(v) / len(v) if v else None


def main():
    m, m_dur = load(sys.argv[1])
    s, _ = load(sys.argv[2])
    rows = []
    for key, c in m.items():
        gui = avg(c.get("GRBM_GUI_ACTIVE"))
        busy = avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES"))
        if not gui or busy is None:
            continue
        t = avg(list(m_dur[key].values())) if m_dur.get(key) else None
        row = {"kernel": key[0], "grid": key[1], "launches": len(c["GRBM_GUI_ACTIVE"]),
               "avg_ms_profiled": round(t * 1e3, 4) if t else None,
               "active_cycles": int(gui / 8), "effective_clock_GHz": round(gui / 8 / t * 1e-9, 3) if t else None,
               "mfma_busy_cycles_all_simds": int(busy),
               "mfma_util": round(busy / (gui / 8 * CUS * SIMDS), 4)}
        cu_busy = avg(c.get("SQ_BUSY_CU_CYCLES"))
        if cu_busy:
            row["cu_busy_frac"] = round(cu_busy / (gui / 8 * CUS), 4) if cu_busy < gui * CUS else round(cu_busy / (gui * CUS), 4)
        sc = s.get(key)
        if sc:
            wave = avg(sc.get("SQ_WAVE_CYCLES"))
            if wave:
                for src, dst in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"),
                                 ("SQ_ACTIVE_INST_ANY", "active_inst_frac")):
                    v = avg(sc.get(src))
                    if v is not None:
                        row[dst] = round(v / wave, 4)
            idx, conf = avg(sc.get("SQ_LDS_IDX_ACTIVE")), avg(sc.get("SQ_LDS_BANK_CONFLICT"))
            if idx:
                row["lds_bank_conflict_frac"] = round(conf / idx, 4)
        rows.append(row)
    rows.sort(key=lambda r: -(r["avg_ms_profiled"] or 0) * r["launches"])
    print(json.dumps({
        "command": "two rocprofv3 --kernel-trace --pmc passes over `python3 tools/stem_only.py --iters 5` (see tools/pmc_mfma.py)",
        "formulas": "active_cycles = GRBM_GUI_ACTIVE/8 (summed over 8 XCDs); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (active_cycles*256*4); "
                    "effective_clock = active_cycles / duration; wait/issue/active fractions are of SQ_WAVE_CYCLES; "
                    "lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
        "note": "profiled passes run at a lower clock than un-profiled ones (guide: DVFS give-back); durations here are the profiled ones",
        "kernels": rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]}, indent=1))


if __name__ == "__main__":
    main()
