#!/bin/bash
# A/B bench of library variants on the GPU box: tools/ab.sh "<bench args>" <name...>   ("base" = the in-tree library)
# e.g. tools/ab.sh "--spp 64" base head ; tools/ab.sh "--scene scenes/knot_glass.scene --depth 16 --spp 32" base head
ARGS=$1; shift
python3 - $ARGS <<'PY'
import re, sys
from scenes.gen_assets import ensure_assets, ensure_large_asset
ensure_assets(); ensure_large_asset('torus_knot_871200.ply')
args = sys.argv[1:]
if '--scene' in args:   # the large generated meshes the scene names (file names ending in _<triangles>.ply)
    for a in re.findall(r'assets/(\w+_\d{6,}\.ply)', open(args[args.index('--scene') + 1]).read()):
        ensure_large_asset(a)
PY
for n in "$@"; do
  if [ "$n" = base ]; then unset PTR_HIP_LIBRARY; else export PTR_HIP_LIBRARY=$PWD/variants/libptr_$n.so; fi
  echo "== $n  [$ARGS]"
  timeout -k 10 300 python bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['kernel_ms_per_step'])"
done
