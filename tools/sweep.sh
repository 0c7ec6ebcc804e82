#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   (runs bench.py with VAR=value, prints value + per-kernel ms)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$VAR=$v', d['value'], d['kernel_ms_per_step'], d['roofline']['frac'])"
done
