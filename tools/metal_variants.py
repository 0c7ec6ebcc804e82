#!/usr/bin/env python3
"""Metal-semantics variants of BASELINE configs 3, 4 and 5 on one MI355X: throughput at full size and parity against the
oracle's restatement of the same Metal lines on a strip of the frame (same protocol as tools/full_configs.py).

  python tools/metal_variants.py [--out gpurun_out/metal_variants.json]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VARIANTS = {
    "3_metal_pbr": dict(scene="helmet_env.scene", assets=[], semantics=127, gpu_spp=256, parity_spp=64, strip=128),
    "4_absorbing_glass": dict(scene="knot_glass_absorbing.scene", assets=["torus_knot_871200.ply"], semantics=127, gpu_spp=256, parity_spp=64, strip=128),
    "5_separable_sss": dict(scene="lucy_standin_sss.scene", assets=["lucy_standin_28005128.ply", "blob_1002528.ply"], semantics=127, gpu_spp=32,
                            parity_spp=16, strip=64),
}


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "metal_variants.json"))
    args = ap.parse_args()
    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    import oracle_lib as ol
    from scenes.gen_assets import ensure_assets, ensure_large_asset

    ensure_assets()
    report = {}
    lum = np.array([0.2126, 0.7152, 0.0722])
    for name, cfg in VARIANTS.items():
        for a in cfg["assets"]:
            ensure_large_asset(a)
        host = pt.HostScene.load(os.path.join(ROOT, "scenes", cfg["scene"]), os.path.join(ROOT, "scenes"))
        dev = pt.DeviceScene(host.desc, 0, keepalive=host)
        s = host.settings_for(seed=1337, metalSemantics=cfg["semantics"])
        s0 = host.settings_for(seed=1337, metalSemantics=0)
        w, h = s.width, s.height
        row = {"scene": cfg["scene"], "resolution": [w, h], "max_depth": s.maxDepth, "metalSemantics": cfg["semantics"], "sssMode": s.sssMode}
        dev.render_image(s, 1)
        img, st = dev.render_image(s, cfg["gpu_spp"])
        row["gpu_spp"] = cfg["gpu_spp"]
        row["gpu_msamples_per_s"] = round(w * h * cfg["gpu_spp"] / st.totalSeconds / 1e6, 1)
        _, st0 = dev.render_image(s0, cfg["gpu_spp"])
        row["gpu_msamples_per_s_embree_semantics_same_scene"] = round(w * h * cfg["gpu_spp"] / st0.totalSeconds / 1e6, 1)
        osc = ol.OracleScene(host)
        y0 = max(0, ((h - cfg["strip"]) // 2 // 16) * 16)
        y1 = min(h, y0 + cfg["strip"])
        n = cfg["parity_spp"]
        g1, _ = dev.render_image(s, 1)
        o1, _, _ = osc.render(s, 1, threads=0, rows=(y0, y1))
        rel = np.abs(g1[y0:y1] - o1[y0:y1]) / (np.abs(o1[y0:y1]) + 1e-2)
        row["fraction_pixels_within_1e-3_at_1spp"] = round(float((rel.max(axis=2) <= 1e-3).mean()), 4)
        gN = dev.render_image(s, n)[0]
        oN, secs, _ = osc.render(s, n, threads=0, rows=(y0, y1))
        s2 = s.copy()
        s2.seed = 1338
        oM, _, _ = osc.render(s2, n, threads=0, rows=(y0, y1))
        noise, err = rmse(oN[y0:y1], oM[y0:y1]), rmse(gN[y0:y1], oN[y0:y1])
        ratio = float((gN[y0:y1] @ lum).mean() / max((oN[y0:y1] @ lum).mean(), 1e-30))
        row.update({"parity_rows": [y0, y1], "parity_spp": n, "noise_floor_N": noise, "rmse_vs_oracle": err, "rmse_over_N": round(err / max(noise, 1e-30), 4),
                    "mean_luminance_ratio": round(ratio, 5), "pass": bool(err <= 1.25 * noise and abs(ratio - 1.0) <= 0.005),
                    "oracle_msamples_per_s": round(w * (y1 - y0) * n / max(secs, 1e-9) / 1e6, 3)})
        # the variant really differs from the Embree-parity render of the same scene
        row["rmse_vs_embree_semantics"] = rmse(gN[y0:y1], dev.render_image(s0, n)[0][y0:y1])
        print(name, row, flush=True)
        report[name] = row
        osc.close()
        dev.close()
        del host
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
