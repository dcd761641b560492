#!/bin/bash
# Interleaved comparison of built library sets on ONE GPU box with an option sweep for each:
#   build the variants into _ab/lib<X>/ (make LIBDIR=_ab/libX EXTRA_DEFS=...), then
#   gpurun -- './tools/ab_sweep.sh "B C" "test_224 atrium250k" --sweep trace_min_active=40,48,56,64'
set -e
variants=$1; scenes=$2; shift 2
out=gpurun_out/ab; mkdir -p $out; : > $out/ab_sweep.log
for sc in $scenes; do
  for v in $variants; do
    echo "== variant $v, $sc" >> $out/ab_sweep.log
    HYDRA_AMD_LIB_DIR=$PWD/_ab/lib$v python tools/pass_bench.py --scene $sc --spp 64 --in-flight 64 "$@" >> $out/ab_sweep.log 2>&1
  done
done
cat $out/ab_sweep.log
