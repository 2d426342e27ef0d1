#!/bin/bash
# Counter passes over the binning of a stretched grid (run on the GPU box): bash tools/pmc_binning.sh <nodes> <tag>
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1e8}
OUT=$REPO/gpurun_out/pmc_bin_${2:-a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  d="$OUT/$(echo $pass | cut -d' ' -f1)"
  echo "[pmc] $pass"
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d "$d" -o p -- python3 "$REPO/tools/time_binning.py" stretched:$N 10 2 > "$d.log" 2>&1
  python3 "$REPO/tools/pmc_kernels.py" "$d" ag_bin
  find "$d" -name '*counter_collection.csv' -size +4M -delete
done
