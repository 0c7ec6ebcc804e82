set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r3_tests.log 2>&1; tail -3 gpurun_out/r3_tests.log
{ bash tools/ab.sh "" before base before base
bash tools/ab.sh "--scene scenes/knot_glass.scene --depth 16 --spp 128" before base
bash tools/ab.sh "--scene scenes/lucy_standin.scene --width 3840 --height 2160 --depth 12 --spp 32" before base
bash tools/ab.sh "--scene scenes/helmet_env.scene --depth 8 --spp 256" before base; } > gpurun_out/r3_ab_small_shade_connect.txt 2>&1
cat gpurun_out/r3_ab_small_shade_connect.txt
