#!/bin/bash
# Kernel-trace and HBM-counter passes over the benchmark command (run on the GPU box).
#   bash tools/profile_round.sh <tag> [extra bench.py arguments ...]    -> gpurun_out/prof_<tag>/{trace,fetch,write,sq}
set -eo pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
shift || true
ARGS="--steps ${PROFILE_STEPS:-5} --warmup 2 --no-cpu-baseline --no-host-tensors --profile off $*"
LABEL=${PROFILE_LABEL:-o1280_r7}
echo "[profile] kernel trace"; 
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
echo "[profile] FETCH_SIZE pass"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 "$REPO/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
echo "[profile] WRITE_SIZE pass"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 "$REPO/bench.py" $ARGS > "$OUT/write.log" 2>&1
echo "[profile] SQ counters pass"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/sq" -o sq -- python3 "$REPO/bench.py" $ARGS > "$OUT/sq.log" 2>&1
python3 "$REPO/tools/pmc_kernels.py" "$OUT/sq" > "$OUT/sq_counters_per_kernel.txt"
# keep what travels back small: the per-dispatch counter CSVs are summarised here
python3 "$REPO/tools/pmc_summary.py" --fetch "$OUT/fetch" --write "$OUT/write" --workload "$LABEL" --out "$OUT/hbm_traffic.json"
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
find "$OUT" -name '*kernel_trace.csv' -size +16M -delete
echo "[profile] done"
