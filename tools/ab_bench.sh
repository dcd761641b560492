#!/bin/bash
# Interleaved A/B of two built library sets on ONE GPU box (box-to-box spread is larger than most kernel changes).
#   build variant A, `mkdir -p _ab/libA && cp hydracore_amd/lib/*.so _ab/libA/`, the same for B, then
#   gpurun -- './tools/ab_bench.sh [scene ...]'      (_ab/ is git-ignored but travels with the snapshot)
set -e
scenes=${@:-test_224 atrium250k}
out=gpurun_out/ab; mkdir -p $out
for rep in 1 2 3; do
  for v in A B; do
    for sc in $scenes; do
      HYDRA_AMD_LIB_DIR=$PWD/_ab/lib$v python tools/pass_bench.py --scene $sc --spp 16 2>&1 | tail -n 1 | sed "s/^/$v $sc /" >> $out/ab.log
    done
  done
done
cat $out/ab.log
