#!/bin/bash
# The measurement pass behind profiles/: run on the GPU box (through gpurun) from the repository root, in up to three calls
# (each fits gpurun's 20-minute limit):
#   bash tools/refresh_profiles.sh <tag> solo      per-kernel records with the pool as ONE group (config 2 and config 5), then the
#                                                  bench line that quotes them, then rocprofv3 --stats of the default command
#   bash tools/refresh_profiles.sh <tag> configs   tools/full_configs.py (parity + throughput at full size, BVH build times)
#   bash tools/refresh_profiles.sh <tag> rest      strong-scaling probe, bench lines of configs 1/3/4/5, Metal variants, stream probe
# Everything lands in gpurun_out/ with the tag in its name; back in the container `python tools/collect_profiles.py <tag>`
# copies the summaries into profiles/.  One GPU process at a time; --pmc passes carry only --kernel-trace.
TAG=${1:-refresh}
WHAT=${2:-solo}
ROUND=${ROUND:-r3}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
CFG5="--scene $R/scenes/lucy_standin.scene --width 3840 --height 2160 --depth 12 --spp 16"
case $WHAT in
solo)
  python3 -c "
from scenes.gen_assets import ensure_assets, ensure_large_asset
ensure_assets(); [ensure_large_asset(a) for a in ('torus_knot_871200.ply', 'lucy_standin_28005128.ply', 'blob_1002528.ply')]"
  # (bench.py quotes the solo records of this round: they are copied into place before the bench line is taken)
  bash tools/measure_solo.sh solo_cfg2_$TAG > gpurun_out/solo_cfg2_$TAG.log 2>&1 && cp gpurun_out/solo_cfg2_$TAG.json profiles/${ROUND}_solo_cfg2.json
  bash tools/measure_solo.sh solo_cfg4_$TAG --scene $R/scenes/knot_glass.scene --depth 16 --spp 128 > gpurun_out/solo_cfg4_$TAG.log 2>&1 && cp gpurun_out/solo_cfg4_$TAG.json profiles/${ROUND}_solo_cfg4.json
  PTR_VERBOSE=build bash tools/measure_solo.sh solo_cfg5_$TAG $CFG5 > gpurun_out/solo_cfg5_$TAG.log 2>&1 && cp gpurun_out/solo_cfg5_$TAG.json profiles/${ROUND}_solo_cfg5.json
  timeout -k 10 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
  cut -c1-300 gpurun_out/bench_$TAG.json
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1)
  find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
  ;;
configs)
  PTR_VERBOSE=build timeout -k 10 1000 python tools/full_configs.py --configs 2,3,4,5 --out gpurun_out/full_configs_$TAG.json > gpurun_out/full_configs_$TAG.log 2>&1
  tail -5 gpurun_out/full_configs_$TAG.log
  ;;
rest)
  POOLS=0 timeout -k 10 200 python tools/strong_scaling_probe.py 2>&1 | grep pool > gpurun_out/strong_probe_$TAG.log
  cat gpurun_out/strong_probe_$TAG.log
  python3 -c "
from scenes.gen_assets import ensure_assets, ensure_large_asset
ensure_assets(); [ensure_large_asset(a) for a in ('torus_knot_871200.ply', 'lucy_standin_28005128.ply', 'blob_1002528.ply')]"
  for cfg in "1 --scene scenes/cornell.scene --width 512 --height 512 --depth 4 --spp 64" "3 --scene scenes/helmet_env.scene --depth 8 --spp 256" \
             "4 --scene scenes/knot_glass.scene --depth 16 --spp 128" "5 --scene scenes/lucy_standin.scene --width 3840 --height 2160 --depth 12 --spp 32"; do
    set -- $cfg; id=$1; shift
    timeout -k 10 400 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg$id.json 2>/dev/null
  done
  timeout -k 10 500 python tools/metal_variants.py --out gpurun_out/metal_variants_$TAG.json > gpurun_out/metal_variants_$TAG.log 2>&1
  timeout -k 10 300 python tools/depth2_probe.py > gpurun_out/depth2_probe_$TAG.txt 2>&1
  tail -3 gpurun_out/depth2_probe_$TAG.txt
  ;;
esac
echo done $WHAT
