#!/bin/bash
# A/B bench of one library under different environment knobs, in one GPU session:
#   tools/ab_env.sh "<bench args>" "VAR=value VAR2=value" "VAR=other" ...      ("-" = no knob)
ARGS=$1; shift
python3 -c "
from scenes.gen_assets import ensure_assets, ensure_large_asset
ensure_assets(); ensure_large_asset('torus_knot_871200.ply')"
for kv in "$@"; do
  echo "== [$kv]  [$ARGS]"
  if [ "$kv" = "-" ]; then kv=""; fi
  env $kv timeout -k 10 300 python bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/ab_env.err | python3 -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['kernel_ms_per_step'])" || tail -5 gpurun_out/ab_env.err
done
