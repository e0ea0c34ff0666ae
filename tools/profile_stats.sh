#!/bin/bash
# usage: tools/profile_stats.sh <tag> <op> [<op> ...]   (on the GPU box)
# rocprofv3 --kernel-trace --stats pass only (per-kernel average duration) for operators of tools/run_op.py.
set -e
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
for op in "$@"; do
  out=gpurun_out/prof_${tag}_${op}/stats
  mkdir -p "$R/$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out" -- python3 "$R/tools/run_op.py" --op "$op" --iters ${ITERS:-60} > "$R/$out/run.log" 2>&1
  {
    echo "# rocprofv3 --kernel-trace --stats -- python3 tools/run_op.py --op $op --iters 60"
    python3 "$R/tools/kstats.py" "$R/$out" | grep "mv::" || true
    echo "# Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev (the first launches run before the clocks have ramped)"
    grep -h "mv::" "$R/$out"/*/*kernel_stats.csv || true
  } > "$R/gpurun_out/${tag}_${op}_stats.txt"
  cat "$R/gpurun_out/${tag}_${op}_stats.txt"
done
