#!/bin/bash
# Instruction-cache counters of the pass kernels (GPU box); one --pmc set per rocprofv3 run.
# usage: tools/pmc_icache.sh <outdir under gpurun_out> [pass_bench args...]
set -e
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 tools/pass_bench.py --spp 16 "$@" > "$OUT/pass$i.log" 2>&1
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.csv"
