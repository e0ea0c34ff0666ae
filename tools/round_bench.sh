#!/bin/bash
# usage: tools/round_bench.sh <tag>   (on the GPU box)
# The driver's command (python bench.py) once plainly and once under rocprofv3 --kernel-trace --stats, so that the
# HIP-event figures of the JSON line (headline + configs) can be read against the profiler's per-kernel averages
# of the same commit.  Outputs under gpurun_out/<tag>_*.
set -e
tag=$1
R=$PWD
export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
python3 "$R/bench.py" > "$R/gpurun_out/${tag}_bench.json" 2> "$R/gpurun_out/${tag}_bench.err"
tail -c 600 "$R/gpurun_out/${tag}_bench.json"; echo
out="$R/gpurun_out/${tag}_prof_bench"
rm -rf "$out"; mkdir -p "$out"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$R/bench.py" --no-cpu-baseline \
    > "$R/gpurun_out/${tag}_bench_under_rocprofv3.json" 2> "$out/run.err"
cd "$R"
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline   (same commit as ${tag}_bench.json)"
  echo "# Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev"
  grep -h "mv::" "$out"/*/*kernel_stats.csv || true
} > "$R/gpurun_out/${tag}_bench_kernel_stats.csv"
cat "$R/gpurun_out/${tag}_bench_kernel_stats.csv"
