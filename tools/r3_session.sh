set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -k "surface_records" 2>&1 | tail -3
