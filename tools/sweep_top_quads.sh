# variant libraries built with: make LIBDIR=hydracore_amd/lib_<name> EXTRA_DEFS="-DHK_TOP_QUADS=.. -DHK_TOP_STRIDE=.. -DHK_LDS_DEPTH=.. -DHK_TRACE_BLOCK=.."
B="timeout -k 10 200 python bench.py --no-cpu-baseline"
for v in "$@"; do
  name=${v%%:*}; bpc=${v#*:}
  HYDRA_HIP_BLOCKS_PER_CU=$bpc HYDRA_AMD_LIB_DIR=$PWD/hydracore_amd/lib_$name $B > gpurun_out/bench_${name}_bpc$bpc.log 2>&1 || exit 1
  HYDRA_HIP_BLOCKS_PER_CU=$bpc HYDRA_AMD_LIB_DIR=$PWD/hydracore_amd/lib_$name $B --scene atrium250k > gpurun_out/bench_${name}_bpc${bpc}_atrium.log 2>&1 || exit 1
  for f in gpurun_out/bench_${name}_bpc$bpc.log gpurun_out/bench_${name}_bpc${bpc}_atrium.log; do echo $f; tail -1 $f | cut -c90-125; done
done
