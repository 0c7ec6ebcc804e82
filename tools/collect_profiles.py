#!/usr/bin/env python3
"""Copy the summaries of a tools/refresh_profiles.sh pass (gpurun_out/*_<tag>*) into profiles/ under the round's names.

  python tools/collect_profiles.py <tag> [--round 1]
"""
import argparse
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--round", type=int, default=1)
    args = ap.parse_args()
    out = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(ROOT, "profiles")
    r = "r%d_" % args.round
    t = args.tag
    bench = last_json_line(os.path.join(out, "bench_%s.json" % t))
    json.dump(bench, open(os.path.join(prof, r + "bench.json"), "w"), indent=1)
    shutil.copy(glob.glob(os.path.join(out, "prof_%s" % t, "*", "*_kernel_stats.csv"))[0], os.path.join(prof, r + "kernel_stats.csv"))
    shutil.copy(os.path.join(out, "pmc_%s" % t, "summary.txt"), os.path.join(prof, r + "pmc_hbm_traffic.txt"))
    shutil.copy(os.path.join(out, "strong_probe_%s.log" % t), os.path.join(prof, r + "strong_scaling_probe.txt"))
    shutil.copy(os.path.join(out, "full_configs_%s.json" % t), os.path.join(prof, r + "full_configs.json"))
    cfg = {}
    for i in (1, 3, 4, 5):
        p = os.path.join(out, "bench_%s_cfg%d.json" % (t, i))
        if os.path.exists(p):
            cfg[str(i)] = last_json_line(p)
    json.dump(cfg, open(os.path.join(prof, r + "bench_configs.json"), "w"), indent=1)
    # FETCH_SIZE / WRITE_SIZE of k_extend<false, false> per dispatch (KB) -> what bench.py quotes as roofline.traffic
    text = open(os.path.join(prof, r + "pmc_hbm_traffic.txt")).read()
    m = re.search(r"== k_extend<false, false>\s+FETCH_SIZE\s+total \S+\s+per-dispatch (\S+)\s+\((\d+) dispatches\)\s+WRITE_SIZE\s+total \S+\s+per-dispatch (\S+)", text)
    tp = os.path.join(prof, r + "hbm_traffic.json")
    traffic = json.load(open(tp))
    fetch, disp, write = float(m.group(1)), int(m.group(2)), float(m.group(3))
    traffic.update({"dispatches": disp, "FETCH_SIZE_KB_per_launch": round(fetch), "WRITE_SIZE_KB_per_launch": round(write),
                    "k_extend_hbm_bytes_per_launch": int((2 * round(fetch) + round(write)) * 1024)})
    json.dump(traffic, open(tp, "w"), indent=1)
    print("bench", bench["value"], "Msamples/s; k_extend HBM bytes per launch", traffic["k_extend_hbm_bytes_per_launch"])
    print(open(os.path.join(prof, r + "strong_scaling_probe.txt")).read())


if __name__ == "__main__":
    main()
