#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv and prints, for the LAST forward pass in it, every kernel's duration and the idle gap
in front of it (start minus the previous kernel's end) -- where a forward's time goes that is not kernel time.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 tools/run_model.py --model mobilenet_v2 --batch 64
    python tools/ktrace_gaps.py gpurun_out/kt <kernels per forward>
"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
per = int(sys.argv[2]) if len(sys.argv) > 2 else 60
last = rows[-per:]
tot_k = tot_g = 0.0
prev_end = None
for st, en, name in last:
    gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
    dur = (en - st) / 1e3
    tot_k += dur
    tot_g += max(gap, 0.0)
    short = name.replace("void mv::", "").split("(")[0][:70]
    print(f"{short:72s} {dur:8.1f} us   gap before {gap:7.1f} us")
    prev_end = en
print(f"{len(last)} kernels: kernel time {tot_k:.1f} us, gaps {tot_g:.1f} us, span {(last[-1][1] - last[0][0]) / 1e3:.1f} us")
