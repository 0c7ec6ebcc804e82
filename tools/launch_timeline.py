#!/usr/bin/env python3
"""Per-launch timeline of one render: when every k_extend / k_shade / k_connect launch started and ended.

  PTR_POOL_GROUPS=1 python tools/launch_timeline.py [--parts 8] [--spp 256] [--scene scenes/cornell_mesh.scene]

Renders partition 0 of `--parts` (what one rank of that many does) with PTR_VERBOSE=polls,launches set, so the library prints
one "[launch] kind K start S end E" line per kernel (kind 0 extend, 1 shade, 2 connect; ms from the first launch) to
stderr, then summarises them per iteration.  With one pool group the three kernels of an iteration are consecutive; with
several groups the lines of the groups interleave.
"""
import argparse
import importlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import torch
    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    bands = importlib.import_module("metal-pathtracer-arm64_amd.bands")
    host = pt.HostScene.load(args.scene, os.path.join(ROOT, "scenes"))
    s = host.settings_for(seed=1337)
    scene = pt.DeviceScene(host.desc, 0, keepalive=host)
    rows = bands.max_band_count(s.height, args.parts) * bands.BAND_ROWS
    out = torch.zeros((rows, s.width, 3), dtype=torch.float32, device="cuda")
    scene.render_device(s, args.spp, out.data_ptr(), 0, 0, args.parts, want_stats=False)   # warm-up
    os.environ["PTR_VERBOSE"] = "polls,launches"
    st = scene.render_device(s, args.spp, out.data_ptr(), 0, 0, args.parts, want_stats=True)
    print("total %.3f ms" % (st.totalSeconds * 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes", "cornell_mesh.scene"))
    ap.add_argument("--dump", default=None, help="write the raw [launch] lines to this file")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        child(args)
        return
    r = subprocess.run([sys.executable, __file__, "--child", "--parts", str(args.parts), "--spp", str(args.spp), "--scene", args.scene],
                       capture_output=True, text=True)
    if args.dump:
        with open(args.dump, "w") as f:
            f.write("".join(l + "\n" for l in r.stderr.splitlines() if l.startswith("[launch]")))
    rows = [re.findall(r"[-\d.]+", l) for l in r.stderr.splitlines() if l.startswith("[launch]")]
    print(r.stdout.strip(), "| %d launches" % len(rows))
    if os.environ.get("PTR_POOL_GROUPS") == "1":
        for i in range(0, len(rows) - 2, 3):
            e, s, c = rows[i], rows[i + 1], rows[i + 2]
            print("it %3d start %8.2f  extend %.3f  shade %.3f  connect %.3f  iteration %.3f" %
                  (i // 3, float(e[1]), float(e[3]), float(s[3]), float(c[3]), float(c[2]) - float(e[1])))
    else:
        # several groups: how much of the frame had at least one / exactly k kernels running
        ev = sorted([(float(x[1]), 1) for x in rows] + [(float(x[2]), -1) for x in rows])
        busy = {}
        level, last = 0, ev[0][0]
        for t, d in ev:
            busy[level] = busy.get(level, 0.0) + (t - last)
            level, last = level + d, t
        end = max(float(x[2]) for x in rows)
        print("last launch ended at %.2f ms; time with k kernels in flight: %s" %
              (end, ", ".join("%d: %.2f ms" % (k, v) for k, v in sorted(busy.items()))))
        names = {0: "extend", 1: "shade", 2: "connect"}
        for kind in (0, 1, 2):
            d = [float(x[3]) for x in rows if int(x[0]) == kind]
            print("  %-8s %4d launches, sum %.2f ms, mean %.3f ms" % (names[kind], len(d), sum(d), sum(d) / max(len(d), 1)))


if __name__ == "__main__":
    main()
