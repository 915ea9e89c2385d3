#!/usr/bin/env python3
"""Print the per-kernel rows of a rocprofv3 --stats output directory (development tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f'{r["Name"][:70]:70s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s} %')
