#!/usr/bin/env python3
"""First-contact GPU check: ray-level parity, small image parity, and a timing probe (run through gpurun)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("metal-pathtracer-arm64_amd")
import oracle_lib as ol
from scenes.gen_assets import ensure_assets

ensure_assets()
print("devices", pt.device_count(), flush=True)

host = pt.HostScene.load(os.path.join(ROOT, "tests/golden/cornell_small_mesh.scene"), os.path.join(ROOT, "scenes"))
s = host.settings_for(width=64, height=64)
dev = pt.DeviceScene(host.desc, 0, keepalive=host)
print("bvh", dev.info(), flush=True)
osc = ol.OracleScene(host)

# ray parity
rng = np.random.default_rng(1)
n = 20000
org = rng.uniform(20, 530, size=(n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, np.full((n, 1), np.inf, np.float32)], axis=1).astype(np.float32)
g, st = dev.trace_rays(rays)
o = osc.trace_rays(rays)
hit_g, hit_o = g["t"] >= 0, o["t"] >= 0
print("hit agreement", (hit_g == hit_o).mean(), "hits", hit_o.mean())
both = hit_g & hit_o
print("max |dt|/t", np.max(np.abs(g["t"][both] - o["t"][both]) / o["t"][both]), "prim agree", (g["primIndex"][both] == o["primIndex"][both]).mean(),
      "type agree", (g["primType"][both] == o["primType"][both]).mean())
print("nodes/ray", st.nodesVisited / n, "prims/ray", st.leafPrimTests / n, flush=True)

for spp in (1, 16):
    img, stats = dev.render_image(s, spp, count=True)
    ref, secs, cnt = osc.render(s, spp, threads=0, count=True)
    rel = np.abs(img - ref) / (np.abs(ref) + 1e-2)
    print("spp", spp, "within1e-3", (rel.max(axis=2) <= 1e-3).mean(), "mean", img.mean(), ref.mean(), "gpu s", stats.totalSeconds, "cpu s", secs)
    print("  gpu counters", {k: v for k, v in stats.as_dict().items() if isinstance(v, int)})
    print("  cpu counters", cnt, flush=True)

# config 2 timing probe
host2 = pt.HostScene.load(os.path.join(ROOT, "scenes/cornell_mesh.scene"), os.path.join(ROOT, "scenes"))
s2 = host2.settings_for()
t0 = time.time()
dev2 = pt.DeviceScene(host2.desc, 0, keepalive=host2)
print("config2 upload", time.time() - t0, dev2.info(), flush=True)
for spp in (4, 16):
    img, stats = dev2.render_image(s2, spp, count=False)
    ms = stats.samples / stats.totalSeconds / 1e6
    print("config2 spp", spp, "sec", stats.totalSeconds, "Msamples/s", ms, "trace ms", stats.traceKernelMs, "shade", stats.shadeKernelMs, "connect", stats.shadowKernelMs,
          "launches", stats.traceLaunches, "mean", img.mean(), flush=True)
pt.write_image(os.path.join(ROOT, "gpurun_out", "config2_16spp.pfm"), img, "pfm")
