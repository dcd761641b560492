#!/bin/bash
# Round profile set (GPU box): GPU tests, the bench line (with its live PMC child runs), rocprofv3 kernel stats of the same command,
# the other scenes.  usage: tools/final_profile.sh <outdir under gpurun_out>
OUT=${1:-gpurun_out/final}
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
timeout -k 10 600 python bench.py --pmc-out $OUT/pmc_live > $OUT/bench_256spp.json.log 2>&1; tail -1 $OUT/bench_256spp.json.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 2 --no-cpu-baseline --no-pmc --no-extra > $OUT/bench_under_rocprof_2x64spp.log 2>&1; tail -1 $OUT/bench_under_rocprof_2x64spp.log | cut -c1-160
for sc in atrium250k_sky atrium250k_glass atrium250k_nmap atrium250k_cutouts tests/golden/scenes/test_42; do
  n=$(basename $sc)
  timeout -k 10 600 python bench.py --no-cpu-baseline --no-extra --scene $sc --pmc-out $OUT/pmc_live_$n > $OUT/bench_${n}_256spp.json.log 2>&1; tail -1 $OUT/bench_${n}_256spp.json.log | cut -c1-200
done
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_2x64spp.csv
find $OUT/prof -name "*domain_stats.csv" | head -1 | xargs -I{} cp {} $OUT/domain_stats_bench_2x64spp.csv
rm -rf $OUT/prof
# row f3: throughput of the MMLT path and its kernel breakdown
timeout -k 10 300 python tools/mmlt_bench.py > $OUT/mmlt_bench_test_42_1080p_1M_chains.log 2>&1; tail -1 $OUT/mmlt_bench_test_42_1080p_1M_chains.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mmlt -- python3 tools/mmlt_bench.py --passes 8 > $OUT/mmlt_bench_under_rocprof.log 2>&1
find $OUT/prof_mmlt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_mmlt_8_passes.csv
rm -rf $OUT/prof_mmlt
# row f4: IHWLayer::EvalGBuffer at 1080p
timeout -k 10 200 python tools/gbuffer_bench.py > $OUT/gbuffer_bench_test_224_1080p.log 2>&1; tail -1 $OUT/gbuffer_bench_test_224_1080p.log | cut -c1-200
timeout -k 10 200 python tools/gbuffer_bench.py --scene atrium250k > $OUT/gbuffer_bench_atrium250k_1080p.log 2>&1; tail -1 $OUT/gbuffer_bench_atrium250k_1080p.log | cut -c1-200
ls $OUT
# the extra configs of the bench line under the kernel trace (configs[2] atrium250k, configs[4] MMLT on test_42)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_extra -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_with_extra_configs_under_rocprof.log 2>&1
find $OUT/prof_extra -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_with_extra_configs.csv
rm -rf $OUT/prof_extra
# device tree builders A/B
timeout -k 10 400 bash tools/bvh_ab.sh atrium250k 16 2>&1 | grep -v "^make\|^atrium" > $OUT/bvh_ab_atrium250k.log; tail -2 $OUT/bvh_ab_atrium250k.log
ls $OUT
# row f4: procedural textures at BASELINE configs[2]'s size (k_proctex next to k_bounce<ALL | PROCTEX>)
timeout -k 10 300 python tools/pass_bench.py --scene atrium250k_proctex --spp 64 --in-flight 64 > $OUT/pass_atrium250k_proctex.log 2>&1; tail -1 $OUT/pass_atrium250k_proctex.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_pt -- python3 tools/pass_bench.py --scene atrium250k_proctex --spp 64 --in-flight 64 > $OUT/pass_atrium250k_proctex_under_rocprof.log 2>&1
find $OUT/prof_pt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_atrium250k_proctex.csv
rm -rf $OUT/prof_pt
ls $OUT
