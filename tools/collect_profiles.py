#!/usr/bin/env python3
"""Copy the summaries of a tools/refresh_profiles.sh pass (gpurun_out/*_<tag>*) into profiles/ under the round's names.

  python tools/collect_profiles.py <tag> [--round 2]

Only what exists is copied, so the three calls of refresh_profiles.sh (solo / configs / rest) can be collected as they finish.
"""
import argparse
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", type=int, default=3)
    args = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(ROOT, "profiles")
    r = "r%d_" % args.round
    t = args.tag
    done = []

    def copy(src, dst):
        if os.path.exists(src):
            shutil.copy(src, os.path.join(prof, dst))
            done.append(dst)

    for cfg in ("cfg2", "cfg4", "cfg5"):
        copy(os.path.join(out, "solo_%s_%s.json" % (cfg, t)), r + "solo_%s.json" % cfg)
        copy(os.path.join(out, "solo_%s_%s_kernel_stats.csv" % (cfg, t)), r + "kernel_stats_solo_%s.csv" % cfg)
    bench = os.path.join(out, "bench_%s.json" % t)
    if os.path.exists(bench) and os.path.getsize(bench) > 0:
        json.dump(last_json_line(bench), open(os.path.join(prof, r + "bench.json"), "w"), indent=1)
        done.append(r + "bench.json")
    stats = glob.glob(os.path.join(out, "prof_%s" % t, "*", "*_kernel_stats.csv"))
    if stats:
        copy(stats[0], r + "kernel_stats.csv")
    copy(os.path.join(out, "full_configs_%s.json" % t), r + "full_configs.json")
    copy(os.path.join(out, "strong_probe_%s.log" % t), r + "strong_scaling_probe.txt")
    copy(os.path.join(out, "metal_variants_%s.json" % t), r + "metal_variants.json")
    copy(os.path.join(out, "depth2_probe_%s.txt" % t), r + "depth2_probe.txt")
    cfg = {}
    for i in (1, 3, 4, 5):
        p = os.path.join(out, "bench_%s_cfg%d.json" % (t, i))
        if os.path.exists(p) and os.path.getsize(p) > 0:
            cfg[str(i)] = last_json_line(p)
    if cfg:
        json.dump(cfg, open(os.path.join(prof, r + "bench_configs.json"), "w"), indent=1)
        done.append(r + "bench_configs.json")
    print("copied:", ", ".join(done))


if __name__ == "__main__":
    main()
