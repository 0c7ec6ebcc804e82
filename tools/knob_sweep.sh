#!/bin/bash
# Bench line of config 2 under one library knob at a time (csrc/host/knobs.h): tools/knob_sweep.sh "<bench args>" NAME=v1,v2,... [NAME=...]
ARGS=$1; shift
run() { timeout -k 10 300 python bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%8.1f Msamples/s  %7.2f ms/step  %s' % (d['value'], d['ms_per_step'], d['kernel_ms_per_step']))"; }
echo "defaults:"; run
for spec in "$@"; do
  name=${spec%%=*}
  for v in $(echo ${spec#*=} | tr ',' ' '); do
    echo "$name=$v:"
    env $name=$v bash -c "$(declare -f run); ARGS='$ARGS'; run"
  done
done
