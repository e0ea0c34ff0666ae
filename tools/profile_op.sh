#!/bin/bash
# usage: tools/profile_op.sh <tag> <op>      (on the GPU box)
# rocprofv3 evidence for one operator of tools/run_op.py: a --kernel-trace --stats pass (per-kernel average duration), then
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass; SQ counters in their own).  The program goes directly
# after `--`.  Summaries land in gpurun_out/<tag>_<op>_{stats,pmc}.txt -- copy the ones to keep into profiles/.
set -e
tag=$1; op=$2
export TMPDIR=/tmp
R=$PWD
out=gpurun_out/prof_${tag}_${op}
mkdir -p "$R/$out/stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/stats" -- python3 "$R/tools/run_op.py" --op "$op" --iters 12 > "$R/$out/stats/run.log" 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 tools/run_op.py --op $op --iters 12"
  python3 "$R/tools/kstats.py" "$R/$out/stats" | grep -v "at::\|Cijk\|elementwise\|distribution" || true
} > "$R/gpurun_out/${tag}_${op}_stats.txt"
{
  echo "# rocprofv3 --pmc <group> --kernel-trace -- python3 tools/run_op.py --op $op   (one group per pass)"
  echo "# FETCH_SIZE / WRITE_SIZE in KiB; gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 for 16-B/lane streaming reads"
  bash "$R/tools/pmc_pass.sh" "$op" "$out/fetch" FETCH_SIZE
  bash "$R/tools/pmc_pass.sh" "$op" "$out/write" WRITE_SIZE
  bash "$R/tools/pmc_pass.sh" "$op" "$out/sq1" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
  bash "$R/tools/pmc_pass.sh" "$op" "$out/sq2" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
  bash "$R/tools/pmc_pass.sh" "$op" "$out/grbm" GRBM_GUI_ACTIVE
} > "$R/gpurun_out/${tag}_${op}_pmc.txt" 2>&1
echo "profiled $op"
