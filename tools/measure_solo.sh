#!/bin/bash
# Per-kernel evidence with the path-slot pool as ONE group (kernels never overlap, so rocprofv3 durations and PMC counters are
# clean per-kernel figures):   tools/measure_solo.sh <tag> [bench.py args, e.g. --scene scenes/lucy_standin.scene --width 3840 ...]
#   pass 0: rocprofv3 --kernel-trace --stats                 (durations)
#   pass 1/2: --pmc FETCH_SIZE / WRITE_SIZE                  (HBM-side bytes; separate passes, MI355X_MICROARCH.md section HBM)
#   pass 3/4: --pmc VALU counter groups                      (lane utilisation, issue rate)
# The program goes directly after `--`; PMC passes carry only --kernel-trace.  Output: gpurun_out/<tag>/ and gpurun_out/<tag>.json.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
BENCH="python3 $ROOT/bench.py --solo --steps 1 --warmup 1 --no-cpu-baseline $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -3 "$OUT/stats.log"; }
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/pass$i" -- $BENCH > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/pass$i.log"; }
done
ARGS="$*"
python3 "$ROOT/tools/summarize_solo.py" "$OUT" "$ROOT/gpurun_out/$TAG.json" "${ARGS//$ROOT\//}"   # repository-relative paths in the record
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
cp "$OUT"/stats/*/*kernel_stats.csv "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" 2>/dev/null
cat "$ROOT/gpurun_out/$TAG.json"
