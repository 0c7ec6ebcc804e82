set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r3_tests.log 2>&1; tail -5 gpurun_out/r3_tests.log
bash tools/measure_solo.sh r3_solo_cfg2 > gpurun_out/r3_solo_cfg2.log 2>&1 && cp gpurun_out/r3_solo_cfg2.json profiles/r3_solo_cfg2.json
timeout -k 10 300 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; cut -c1-300 gpurun_out/r3_bench.json; tail -3 gpurun_out/r3_bench.err
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --single-device --spp 32 --steps 2 --warmup 1 > gpurun_out/r3_bench_2rank_rehearsal.json 2> gpurun_out/r3_bench_2rank_rehearsal.err; cut -c1-700 gpurun_out/r3_bench_2rank_rehearsal.json; tail -3 gpurun_out/r3_bench_2rank_rehearsal.err
