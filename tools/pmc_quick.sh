#!/bin/bash
# A few PMC passes over one short bench run, summarised per kernel:
#   tools/pmc_quick.sh <tag> "<bench args>" "<counters of pass 1>" "<counters of pass 2>" ...      (environment knobs are inherited)
# The program goes directly after `--`; passes carry only --kernel-trace besides --pmc.
TAG=$1; ARGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
cd /tmp
i=0
for CNT in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline $ARGS > "$OUT/pass$i.log" 2>&1 || { echo "pass $i ($CNT) failed"; tail -3 "$OUT/pass$i.log"; }
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$ROOT/gpurun_out/$TAG.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
