#!/bin/bash
# HBM traffic of the dominant kernel for bench.py's roofline.traffic (GPU box).  One counter group per rocprofv3 run
# (--pmc only, no trace domains), as MI355X_MICROARCH.md prescribes; FETCH_SIZE is doubled for 16-B-per-lane loads.
# usage: tools/pmc_traffic.sh <outdir under gpurun_out>   ->  <outdir>/pmc_traffic.json (+ pmc_summary.csv)
set -e
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > "$OUT/pass$i.log" 2>&1
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.csv"
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = {}
for r in csv.DictReader(open(out + "/pmc_summary.csv")):
    rows[(r["kernel"], r["counter"])] = (int(r["launches"]), float(r["avg_per_launch"]))
k = "k_trace_dyn<false, false>"
fetch_kb, write_kb = rows[(k, "FETCH_SIZE")][1], rows[(k, "WRITE_SIZE")][1]
hit, miss = rows[(k, "TCC_HIT_sum")][1], rows[(k, "TCC_MISS_sum")][1]
acc_fetch = rows.get(("k_accumulate", "FETCH_SIZE"), (0, 0.0))[1]
json.dump({
    "workload_key": "test_224|1920x1080|d8|spp64",
    "kernel": k,
    "source": "tools/pmc_traffic.sh: rocprofv3 --pmc, one counter group per run (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum), "
              "python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline; %d launches averaged (all bounces)" % rows[(k, "FETCH_SIZE")][0],
    "k_trace_FETCH_SIZE_KB_per_launch_raw": fetch_kb,
    "k_trace_WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests of 16-B-per-lane loads as 64 B (MI355X_MICROARCH.md, HBM section) => read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact",
    "calibration_k_accumulate_FETCH_SIZE_KB": acc_fetch,
    "k_trace_hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "k_trace_L2_hit_rate": hit / (hit + miss) if hit + miss > 0 else None,
}, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/pmc_traffic.json").read())
PY
