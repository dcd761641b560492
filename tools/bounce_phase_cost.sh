# k_bounce with phases left out (HK_EXP_BOUNCE_SKIP: 1 = light pick/sample/eval, 2 = BSDF sampling, 4 = surface reconstruction); results invalid,
# only the bounce-0 launch (identical inputs in every variant) is compared.  Variant libraries: make LIBDIR=hydracore_amd/lib_skipN EXTRA_DEFS=-DHK_EXP_BOUNCE_SKIP=N
for scene in "" "--scene atrium250k"; do
for v in full skip1 skip2 skip3 skip7; do
  if [ $v = full ]; then L=$PWD/hydracore_amd/lib; else L=$PWD/hydracore_amd/lib_$v; fi
  HYDRA_AMD_LIB_DIR=$L timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1 --warmup 1 $scene > gpurun_out/bounce_phase_$v.log 2>&1 || { tail -3 gpurun_out/bounce_phase_$v.log; exit 1; }
  python - $v "$scene" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/bounce_phase_%s.log'%sys.argv[1]) if x.startswith('{')][-1]
j=json.loads(l)
print("%-22s %-6s k_bounce at bounce 0: %7.3f ms   (all bounces %8.1f ms)" % (sys.argv[2] or "test_224", sys.argv[1], j["bounce_kernel_ms_per_bounce"][0], j["stage_ms"]["bounce_hit_light_bsdf"]))
PY
done; done
