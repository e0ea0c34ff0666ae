#!/usr/bin/env python3
"""Print the kernel_stats csv files rocprofv3 --stats wrote under a directory (name, calls, average us, % of time)."""
import csv
import glob
import sys

for f in sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)):
    print("#", f)
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
