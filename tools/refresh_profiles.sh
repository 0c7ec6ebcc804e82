#!/bin/bash
# The measurement pass behind profiles/: run on the GPU box (through gpurun) from the repository root.
#   bash tools/refresh_profiles.sh <tag>        -> everything lands in gpurun_out/ with the tag in its name
# then, back in the container:  python tools/collect_profiles.py <tag>   copies the summaries into profiles/.
# One GPU process at a time; the --pmc passes carry only --kernel-trace (no other trace domains).
TAG=${1:-refresh}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 700 python tools/full_configs.py --configs 2,3,4,5 --out gpurun_out/full_configs_$TAG.json > gpurun_out/full_configs_$TAG.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cut -c1-200 gpurun_out/bench_$TAG.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1)
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$TAG/pass_$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1)
done
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.txt
find gpurun_out/prof_$TAG gpurun_out/pmc_$TAG -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_$TAG -name "*counter_collection.csv" -delete
POOLS=16777216 timeout -k 10 200 python tools/strong_scaling_probe.py 2>&1 | grep pool > gpurun_out/strong_probe_$TAG.log
cat gpurun_out/strong_probe_$TAG.log
for cfg in "1 --scene scenes/cornell.scene --width 512 --height 512 --depth 4 --spp 64" "3 --scene scenes/helmet_env.scene --depth 8 --spp 256" \
           "4 --scene scenes/knot_glass.scene --depth 16 --spp 128" "5 --scene scenes/lucy_standin.scene --width 3840 --height 2160 --depth 12 --spp 32"; do
  set -- $cfg; id=$1; shift
  timeout -k 10 400 python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_cfg$id.json 2>/dev/null
done
echo done
