set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r3_settle_tests.log 2>&1; tail -5 gpurun_out/r3_settle_tests.log
