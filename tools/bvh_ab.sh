#!/bin/bash
# A/B of the tree builders on one scene (GPU box): host binned SAH vs the device builds (LBVH, PLOC at several radii); whole-pass rates of tools/pass_bench.py
#   tools/bvh_ab.sh [scene] [spp]
SCENE=${1:-atrium250k}; SPP=${2:-8}
echo "== host SAH"; python tools/pass_bench.py --scene $SCENE --spp $SPP | tail -1
echo "== GPU LBVH"; HYDRA_GPU_BVH=0 HYDRA_GPU_BVH_METHOD=lbvh python tools/pass_bench.py --scene $SCENE --spp $SPP | tail -1
for r in 8 16 32 64 100 128; do echo "== GPU PLOC radius $r"; HYDRA_GPU_BVH=0 HYDRA_GPU_BVH_METHOD=ploc HYDRA_GPU_BVH_RADIUS=$r python tools/pass_bench.py --scene $SCENE --spp $SPP | tail -1; done
