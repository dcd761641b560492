#!/bin/bash
# LDS triangle pool of the persistent traversal kernels (option top_tris_in_lds / HYDRA_HIP_TOP_TRIS): whole-pass stage times per pool size
export TMPDIR=/tmp
for sc in test_224 atrium250k; do
  for t in 0 4 8 16; do
    printf "%-12s top_tris_in_lds=%-2s " $sc $t
    HYDRA_HIP_TOP_TRIS=$t python tools/pass_bench.py --scene $sc --spp 64 --sweep samples_in_flight=64 2>&1 | tail -1
  done
done
