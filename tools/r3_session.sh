set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r3_tests.log 2>&1; tail -3 gpurun_out/r3_tests.log
PTR_TEST_VARIANT=aos python -m pytest tests -m gpu -q -x -k "scheduling or cornell_image or partition or multi_device or frames_rendered or edge_cases" > gpurun_out/r3_tests_aos.log 2>&1; tail -3 gpurun_out/r3_tests_aos.log
python3 -c "
from scenes.gen_assets import ensure_assets, ensure_large_asset
ensure_assets(); [ensure_large_asset(a) for a in ('lucy_standin_28005128.ply', 'blob_1002528.ply')]"
CFG5="--scene scenes/lucy_standin.scene --width 3840 --height 2160 --depth 12 --spp 32"
{
for lib in base aos; do
  if [ $lib = base ]; then unset PTR_HIP_LIBRARY; else export PTR_HIP_LIBRARY=$PWD/variants/libptr_$lib.so; fi
  echo "#### library: $lib"
  bash tools/ab_env.sh "" "-" "PTR_SHADE_SORT=1"
  bash tools/ab_env.sh "$CFG5" "-" "PTR_SHADE_SORT=1"
  bash tools/ab_env.sh "--solo" "-" "PTR_SHADE_SORT=1"
done; } > gpurun_out/r3_ab_aos_sort.txt 2>&1
cat gpurun_out/r3_ab_aos_sort.txt
