#!/bin/bash
# usage: tools/pmc_pass.sh <op> <outdir> <counter> [<counter> ...]   (on the GPU box; one rocprofv3 --pmc pass)
set -e
op=$1; out=$2; shift 2
export TMPDIR=/tmp
R=$PWD
mkdir -p "$R/$out"
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$out" -- python3 "$R/tools/run_op.py" --op "$op" > "$R/$out/run.log" 2>&1
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mv::" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, c), v in sorted(d.items()):
    print(f"{k:62s} {c:32s} {sum(v)/len(v):18.1f}  (n={len(v)})")
if dur: print(f"avg kernel us under PMC: {sum(dur)/len(dur):.1f}")
PY
