#!/usr/bin/env python3
"""Print the per-kernel table of a `rocprofv3 --kernel-trace --stats --output-format csv` run: tools/kstats.py <dir> [rows]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:rows]:
    print("%-60s %6s calls  avg %9.3f ms  total %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
