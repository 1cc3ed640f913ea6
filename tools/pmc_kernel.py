#!/usr/bin/env python3
"""Average the PMC counters of one kernel from a rocprofv3 --pmc ... --output-format csv run:
    python tools/pmc_kernel.py <dir> <kernel-name-substring>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    raise SystemExit("no counter_collection.csv under " + sys.argv[1])
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print("%-24s n=%d avg=%.1f" % (k, len(v), sum(v) / len(v)))
