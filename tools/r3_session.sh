set -o pipefail
cd $GRAFT_REPO_ROOT
python3 -c "
from scenes.gen_assets import ensure_assets, ensure_large_asset
ensure_assets(); [ensure_large_asset(a) for a in ('torus_knot_871200.ply', 'lucy_standin_28005128.ply', 'blob_1002528.ply')]"
{ echo "# tools/steps_probe.py (counting build, PTR_VERBOSE=steps, PTR_POOL_GROUPS=1): where a k_shade visit's time goes";
for sc in cornell_mesh lucy_standin knot_glass helmet_env; do echo "== scenes/$sc.scene"; PTR_POOL_GROUPS=1 timeout -k 10 300 python tools/steps_probe.py scenes/$sc.scene 16 2>&1 | grep -v "^\[upload\|^\[bvh\|^\[geometry"; done; } > gpurun_out/r3_steps_probe.txt 2>&1
cat gpurun_out/r3_steps_probe.txt
