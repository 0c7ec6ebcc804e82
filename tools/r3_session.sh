set -o pipefail
cd $GRAFT_REPO_ROOT
PTR_POOL_GROUPS=1 python tools/launch_timeline.py --parts 8 > gpurun_out/r3_timeline_p8_g1.txt 2>&1
python tools/launch_timeline.py --parts 8 --child > gpurun_out/r3_timeline_p8_g4.out 2> gpurun_out/r3_timeline_p8_g4.err
python tools/launch_timeline.py --parts 8 > gpurun_out/r3_timeline_p8_g4.txt 2>&1
cat gpurun_out/r3_timeline_p8_g1.txt; cat gpurun_out/r3_timeline_p8_g4.txt; grep "^\[poll\]" gpurun_out/r3_timeline_p8_g4.err | head -60
