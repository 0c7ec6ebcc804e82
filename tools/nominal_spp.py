#!/usr/bin/env python3
"""BASELINE configs 2-5 at their nominal sample counts (256 / 1024 / 2048 / 4096 spp), one frame each on one MI355X.
Frames whose per-sample accumulators exceed the memory budget are rendered in several passes (hip_backend.cpp: renderBands).

  python tools/nominal_spp.py [--configs 2,3,4,5] [--out gpurun_out/nominal_spp.json]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    2: ("cornell_mesh.scene", [], 256),
    3: ("helmet_env.scene", [], 1024),
    4: ("knot_glass.scene", ["torus_knot_871200.ply"], 2048),
    5: ("lucy_standin.scene", ["lucy_standin_28005128.ply", "blob_1002528.ply"], 4096),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "nominal_spp.json"))
    args = ap.parse_args()
    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    from scenes.gen_assets import ensure_assets, ensure_large_asset

    ensure_assets()
    report = {}
    for cid in [int(c) for c in args.configs.split(",")]:
        scene, assets, spp = CONFIGS[cid]
        for a in assets:
            ensure_large_asset(a)
        host = pt.HostScene.load(os.path.join(ROOT, "scenes", scene), os.path.join(ROOT, "scenes"))
        dev = pt.DeviceScene(host.desc, 0, keepalive=host)
        s = host.settings_for(seed=1337)
        dev.render_image(s, 1)
        img, st = dev.render_image(s, spp)
        row = {"scene": scene, "resolution": [s.width, s.height], "max_depth": s.maxDepth, "spp": spp, "render_s": round(st.totalSeconds, 3),
               "msamples_per_s": round(s.width * s.height * spp / st.totalSeconds / 1e6, 1), "finite": bool(np.isfinite(img).all()),
               "mean": float(img.mean())}
        print("config", cid, row, flush=True)
        report[str(cid)] = row
        dev.close()
        del host
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
