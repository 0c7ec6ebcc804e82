#!/bin/bash
# On the GPU box: tools/calibration/run.sh  ->  gpurun_out/fetch_calibration/{run.log, pass1, pass2}, summary on stdout
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_calibration
BIN=$ROOT/tools/calibration/fetch_calibration
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
timeout -k 10 120 "$BIN" > "$OUT/run.log" 2>&1 || { echo "plain run failed"; tail -3 "$OUT/run.log"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pass1" -- "$BIN" > "$OUT/pass1.log" 2>&1 || { echo "pass 1 failed"; tail -3 "$OUT/pass1.log"; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/pass2" -- "$BIN" > "$OUT/pass2.log" 2>&1 || { echo "pass 2 failed"; tail -3 "$OUT/pass2.log"; }
python3 "$ROOT/tools/calibration/summarize.py" "$OUT" > "$ROOT/gpurun_out/fetch_size_calibration.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
cat "$ROOT/gpurun_out/fetch_size_calibration.txt"
