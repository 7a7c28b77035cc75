#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 *kernel_stats.csv: python tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    print("%-64s calls=%6s avg_us=%9.2f total_ms=%9.2f pct=%s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
