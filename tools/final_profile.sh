set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log | cut -c1-160
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final3 -- python3 bench.py --steps 2 --no-cpu-baseline > gpurun_out/bench_under_rocprof_final3.log 2>&1; tail -1 gpurun_out/bench_under_rocprof_final3.log | cut -c1-160
bash tools/pmc_traffic.sh gpurun_out/pmc_final3 > gpurun_out/pmc_final3.log 2>&1; tail -12 gpurun_out/pmc_final3.log
timeout -k 10 300 python bench.py --no-cpu-baseline --scene atrium250k > gpurun_out/bench_final_atrium.log 2>&1; tail -1 gpurun_out/bench_final_atrium.log | cut -c90-140
timeout -k 10 300 python bench.py --no-cpu-baseline --scene atrium250k_sky > gpurun_out/bench_final_atrium_sky.log 2>&1; tail -1 gpurun_out/bench_final_atrium_sky.log | cut -c90-140
timeout -k 10 300 python bench.py --no-cpu-baseline --scene atrium250k_glass > gpurun_out/bench_final_atrium_glass.log 2>&1; tail -1 gpurun_out/bench_final_atrium_glass.log | cut -c90-140
