for v in lib lib_bb64 lib_bb128 lib_bb512; do
  echo "== $v"
  HYDRA_AMD_LIB_DIR=$PWD/hydracore_amd/$v python tools/pass_bench.py --spp 64 --sweep samples_in_flight=64 || exit 1
  HYDRA_AMD_LIB_DIR=$PWD/hydracore_amd/$v python tools/pass_bench.py --scene atrium250k --spp 64 --sweep samples_in_flight=64 | tail -1 || exit 1
done
