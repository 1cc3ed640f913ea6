#!/usr/bin/env python3
"""Print name / calls / avg us / total ms of a rocprofv3 kernel_stats.csv (optionally only names matching a substring)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or any(k in r["Name"] for k in sys.argv[2:]):
        print("%-110s %5s calls  avg %9.1f us  total %8.3f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  float(r["TotalDurationNs"]) / 1e6))
