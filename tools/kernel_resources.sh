#!/bin/bash
# per-kernel register / spill / LDS figures of one translation unit of libhydra_hip.so from the compiler (no GPU needed):
#   tools/kernel_resources.sh <tu: trace | bounce_lean | bounce_classic | bounce_nmap | bounce_all | bounce_all45 | mmlt_lean | mmlt_all | main> [filter regex] [extra hipcc flags...]
TU=${1:-trace}; shift || true
F=${1:-.}; shift || true
SRC=hydracore_amd/csrc/hk_inst_$TU.hip
[ "$TU" = main ] && SRC=hydracore_amd/csrc/hydra_hip.hip
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -c $SRC -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-58s VGPR %4s spill %3s SGPR %4s scratch %5s B/lane  occupancy %2s waves/SIMD  LDS %6s B" % (name[:58], r.get("VGPRs"), r.get("VGPR Spill", r.get("VGPRs Spill")), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
' | grep -E "$F"
rm -f /tmp/kres_$$.o
