#!/usr/bin/env python3
"""Full-size runs of the BASELINE configurations on one MI355X, with the CPU oracle beside them.

For each configuration: scene load + BVH build / upload time, GPU render throughput at the configuration's own
resolution and depth, and parity against the oracle on a strip of the full-resolution frame (deterministic-stream
fraction at 1 spp; RMSE against the oracle's own seed-to-seed noise floor N at `parity_spp`; mean-luminance ratio).
Config 2 additionally runs the whole protocol of SURVEY.md section 8(d) on the full frame.

  python tools/full_configs.py [--configs 2,3,4,5] [--out gpurun_out/full_configs.json]

The oracle is test infrastructure (oracle/README.md); it is used here only as the checker.
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

CONFIGS = {
    2: dict(scene="cornell_mesh.scene", assets=[], gpu_spp=256, parity_spp=256, strip=None, nominal_spp=256),
    3: dict(scene="helmet_env.scene", assets=[], gpu_spp=1024, parity_spp=64, strip=128, nominal_spp=1024),
    4: dict(scene="knot_glass.scene", assets=["torus_knot_871200.ply"], gpu_spp=512, parity_spp=64, strip=128, nominal_spp=2048),
    5: dict(scene="lucy_standin.scene", assets=["lucy_standin_28005128.ply", "blob_1002528.ply"], gpu_spp=64, parity_spp=16, strip=64,
            nominal_spp=4096),
}


def rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "full_configs.json"))
    ap.add_argument("--skip-oracle", action="store_true")
    args = ap.parse_args()

    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    import oracle_lib as ol
    from scenes.gen_assets import ensure_assets, ensure_large_asset

    ensure_assets()
    report = {}
    lum = np.array([0.2126, 0.7152, 0.0722])
    for cid in [int(c) for c in args.configs.split(",")]:
        cfg = CONFIGS[cid]
        row = {"scene": cfg["scene"]}
        t0 = time.time()
        for a in cfg["assets"]:
            ensure_large_asset(a)
        row["asset_generation_s"] = round(time.time() - t0, 2)
        t0 = time.time()
        host = pt.HostScene.load(os.path.join(ROOT, "scenes", cfg["scene"]), os.path.join(ROOT, "scenes"))
        row["scene_load_s"] = round(time.time() - t0, 2)
        t0 = time.time()
        dev = pt.DeviceScene(host.desc, 0, keepalive=host)
        row["bvh_build_upload_s"] = round(time.time() - t0, 2)
        row["scene_info"] = dev.info()
        s = host.settings_for(seed=1337)
        w, h = s.width, s.height
        row["resolution"] = [w, h]
        row["max_depth"] = s.maxDepth
        print("config", cid, row, flush=True)

        # throughput at the configuration's own size (whole frame), after a short warm-up render
        dev.render_image(s, 1)
        t0 = time.time()
        img, st = dev.render_image(s, cfg["gpu_spp"])
        wall = time.time() - t0
        row["gpu_spp"] = cfg["gpu_spp"]
        row["gpu_render_s"] = round(st.totalSeconds, 3)
        row["gpu_wall_s_incl_readback"] = round(wall, 3)
        row["gpu_msamples_per_s"] = round(w * h * cfg["gpu_spp"] / st.totalSeconds / 1e6, 1)
        row["gpu_kernel_ms"] = {"extend": round(st.traceKernelMs, 1), "shade": round(st.shadeKernelMs, 1), "connect": round(st.shadowKernelMs, 1)}
        row["nominal_spp_projected_s"] = round(st.totalSeconds * cfg["nominal_spp"] / cfg["gpu_spp"], 1)
        row["image_finite"] = bool(np.isfinite(img).all())
        row["image_mean"] = float(img.mean())
        print("config", cid, "gpu", row["gpu_msamples_per_s"], "Msamples/s", flush=True)

        if not args.skip_oracle:
            t0 = time.time()
            osc = ol.OracleScene(host)
            row["oracle_build_s"] = round(time.time() - t0, 2)
            strip = cfg["strip"]
            if strip is None:
                y0, y1 = 0, h
            else:
                y0 = max(0, ((h - strip) // 2 // 16) * 16)
                y1 = min(h, y0 + strip)
            row["parity_rows"] = [y0, y1]
            # deterministic-stream check at 1 spp
            g1, _ = dev.render_image(s, 1)
            o1, secs, _ = osc.render(s, 1, threads=0, rows=(y0, y1))
            rel = np.abs(g1[y0:y1] - o1[y0:y1]) / (np.abs(o1[y0:y1]) + 1e-2)
            row["fraction_pixels_within_1e-3_at_1spp"] = round(float((rel.max(axis=2) <= 1e-3).mean()), 4)
            row["oracle_msamples_per_s"] = round(w * (y1 - y0) / max(secs, 1e-9) / 1e6, 3)
            # RMSE protocol at parity_spp
            n = cfg["parity_spp"]
            gN = img if n == cfg["gpu_spp"] else dev.render_image(s, n)[0]
            oN, secsN, _ = osc.render(s, n, threads=0, rows=(y0, y1))
            s2 = s.copy()
            s2.seed = 1338
            oM, _, _ = osc.render(s2, n, threads=0, rows=(y0, y1))
            noise = rmse(oN[y0:y1], oM[y0:y1])
            err = rmse(gN[y0:y1], oN[y0:y1])
            ratio = float((gN[y0:y1] @ lum).mean() / max((oN[y0:y1] @ lum).mean(), 1e-30))
            row["parity_spp"] = n
            row["noise_floor_N"] = noise
            row["rmse_vs_oracle"] = err
            row["rmse_over_N"] = round(err / max(noise, 1e-30), 4)
            row["mean_luminance_ratio"] = round(ratio, 5)
            row["pass"] = bool(err <= 1.25 * noise and abs(ratio - 1.0) <= 0.005)
            row["oracle_render_s"] = round(secsN, 1)
            row["oracle_threads"] = os.cpu_count()
            row["oracle_msamples_per_s"] = round(w * (y1 - y0) * n / max(secsN, 1e-9) / 1e6, 3)
            osc.close()
            print("config", cid, "parity", {k: row[k] for k in ("fraction_pixels_within_1e-3_at_1spp", "rmse_over_N", "mean_luminance_ratio", "pass")},
                  flush=True)
        dev.close()
        del host
        report[str(cid)] = row
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
