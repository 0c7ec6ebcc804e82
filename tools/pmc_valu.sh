#!/bin/bash
# VALU-side PMC counters of the traversal kernels with the pool as one group (kernels never overlap), for A/B runs:
#   tools/pmc_valu.sh <outdir> "<ENV=value ...>" [bench args]
# Two passes (counter groups that fit together), then tools/pmc_summary.py + the derived lane-utilisation / issue figures.
set -u
OUT=$1; KV=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$ROOT/$OUT"
cd /tmp
i=0
for CNT in \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" \
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  env $KV timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$ROOT/$OUT/pass$i" -- python3 "$ROOT/bench.py" --solo --spp 32 --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$ROOT/$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$ROOT/$OUT/pass$i.log"; }
done
python3 "$ROOT/tools/pmc_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/summary.txt" 2>&1
python3 - "$ROOT/$OUT/summary.txt" <<'PY'
import re, sys
cur = None; tot = {}
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+)", line)
    if m: cur = m.group(1); tot[cur] = {}; continue
    m = re.match(r"\s+(\S+)\s+total (\S+)", line)
    if m and cur: tot[cur][m.group(1)] = float(m.group(2))
for k, c in tot.items():
    if "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"] > 0:
        util = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
        issue = c["SQ_INSTS_VALU"] * 4.0 / (c.get("GRBM_GUI_ACTIVE", 0) * 128.0) if c.get("GRBM_GUI_ACTIVE") else float("nan")
        print("%-60s lane utilisation %.3f  VALU issue %.3f  VALU insts %.4g  SALU insts %.4g" % (k[:60], util, issue, c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU", 0)))
PY
