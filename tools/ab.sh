#!/bin/bash
# A/B bench of library variants on the GPU box: tools/ab.sh <spp> <name...>   ("base" = the in-tree library)
SPP=$1; shift
for n in "$@"; do
  if [ "$n" = base ]; then unset PTR_HIP_LIBRARY; else export PTR_HIP_LIBRARY=$PWD/variants/libptr_$n.so; fi
  echo "== $n"
  timeout -k 10 200 python bench.py --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['kernel_ms_per_step'], d['roofline']['avg_launch_ms'], 'extend alg bytes/launch', d['roofline']['alg_bytes_per_launch'], 'B/sample', d['roofline']['bytes_per_sample'])"
done
