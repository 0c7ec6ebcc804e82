#!/usr/bin/env python3
"""Single-GPU estimate of strong scaling: time the render of partition 0 of P (what one rank of P does) for several
pool sizes and compare with the full frame.  efficiency(P) = T(1) / (P * T_part(P)); the gather is not included."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pt = importlib.import_module("metal-pathtracer-arm64_amd")
bands = importlib.import_module("metal-pathtracer-arm64_amd.bands")
host = pt.HostScene.load(os.path.join(ROOT, "scenes", "cornell_mesh.scene"), os.path.join(ROOT, "scenes"))
s = host.settings_for(width=1920, height=1080, max_depth=8, seed=1337)
spp = int(os.environ.get("SPP", "256"))
dev = torch.device("cuda", 0)
for pool in [int(x) for x in os.environ.get("POOLS", "0").split(",")]:
    if pool:
        os.environ["PTR_POOL_SLOTS"] = str(pool)
    scene = pt.DeviceScene(host.desc, 0, keepalive=host)
    base = None
    for parts in [int(x) for x in os.environ.get("PARTS", "1,2,4,8").split(",")]:
        rows = bands.max_band_count(1080, parts) * bands.BAND_ROWS
        out = torch.zeros((rows, 1920, 3), dtype=torch.float32, device=dev)
        scene.render_device(s, spp, out.data_ptr(), 0, 0, parts, want_stats=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            scene.render_device(s, spp, out.data_ptr(), 0, 0, parts, want_stats=False)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps
        if base is None:
            base = t * parts
        st = scene.render_device(s, spp, out.data_ptr(), 0, 0, parts, want_stats=True)   # the library's own clock (first launch .. resolve)
        print("pool %9d  parts %d  part-0 render %.2f ms  efficiency %.3f  (%.0f Msamples/s projected)  inside the library %.2f ms" %
              (pool, parts, t * 1e3, base / (parts * t), 1920 * 1080 * spp / t / 1e6 * 1.0, st.totalSeconds * 1e3), flush=True)
    scene.close()
