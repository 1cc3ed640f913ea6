#!/usr/bin/env python3
"""One line per bench run of tools/h2d_power.sh: clips/s and the mean of rocm-smi's samples taken while the GPU was busy (> 90 % use)."""
import json
import re
import sys

label, smi, bench = sys.argv[1:4]
rows = []
for line in open(smi):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    card = d.get("card0") or next((v for k, v in d.items() if k.startswith("card")), None)
    if not card:
        continue
    row = {}
    for k, v in card.items():
        m = re.search(r"-?\d+(\.\d+)?", str(v))
        if not m:
            continue
        x = float(m.group(0))
        kl = k.lower()
        if "power" in kl:
            row["power_w"] = x
        elif "sclk" in kl and "level" not in kl or kl.startswith("sclk clock speed"):
            row["sclk_mhz"] = x
        elif "mclk" in kl and "level" not in kl or kl.startswith("mclk clock speed"):
            row["mclk_mhz"] = x
        elif "fclk" in kl and "level" not in kl:
            row["fclk_mhz"] = x
        elif "socclk" in kl and "level" not in kl:
            row["socclk_mhz"] = x
        elif "gpu use" in kl:
            row["use"] = x
    rows.append(row)
busy = [r for r in rows if r.get("use", 0) > 90]
b = json.loads(open(bench).read().strip().splitlines()[-1])
mean = lambda k: (sum(r[k] for r in busy if k in r) / max(1, sum(1 for r in busy if k in r)))
print("%-12s %7.1f clips/s %7.3f ms/step  stem alone %.3f ms | %3d busy samples of %3d: power %6.1f W  sclk %6.1f MHz  mclk %6.1f  fclk %6.1f  socclk %6.1f"
      % (label, b["value"], b["ms_per_step"], b["config"].get("stem_alone_ms", float("nan")), len(busy), len(rows), mean("power_w"), mean("sclk_mhz"),
         mean("mclk_mhz"), mean("fclk_mhz"), mean("socclk_mhz")))
if rows and not busy:
    print("   (no busy samples; first sample keys: %s)" % sorted(rows[0]))
elif not rows:
    print("   (rocm-smi gave no JSON: %s)" % open(smi).read()[:300].replace("\n", " "))
