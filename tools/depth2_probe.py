#!/usr/bin/env python3
"""Deterministic-stream criterion of SURVEY.md section 8(d) on every parity scene: fraction of pixels within 1e-3
relative of the oracle at 1 spp, for depth 2 and for the scene's own depth.  Prints one line per scene/depth."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pt = importlib.import_module("metal-pathtracer-arm64_amd")
import oracle_lib as ol
from scenes.gen_assets import ensure_assets, ensure_large_asset

SCENES = os.path.join(ROOT, "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = [
    (os.path.join(GOLDEN, "smoke.scene"), 64, 64),
    (os.path.join(GOLDEN, "cornell_small_mesh.scene"), 128, 128),
    (os.path.join(GOLDEN, "materials.scene"), 192, 128),
    (os.path.join(GOLDEN, "env_materials.scene"), 192, 128),
    (os.path.join(GOLDEN, "lucy_small.scene"), 160, 90),
    (os.path.join(SCENES, "cornell.scene"), 512, 512),
    (os.path.join(SCENES, "cornell_mesh.scene"), 480, 270),
    (os.path.join(SCENES, "helmet_env.scene"), 480, 270),
    (os.path.join(SCENES, "knot_glass.scene"), 480, 270),
]


def main():
    ensure_assets()
    for a in ("torus_knot_871200.ply", "blob_125000.ply"):
        ensure_large_asset(a)
    for path, w, h in CASES:
        host = pt.HostScene.load(path, SCENES)
        dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
        base = host.settings_for(width=w, height=h, seed=1337)
        for depth in sorted({2, 4, int(base.maxDepth)}):
            s = host.settings_for(width=w, height=h, max_depth=depth, seed=1337)
            g, gsig = dev.render_signatures(s)
            o, osig, marginal = osc.render_signatures(s, threads=32)
            plain, _ = dev.render_image(s, 1)
            assert np.array_equal(plain, g), "the counting build renders another image"
            rel = np.abs(g - o) / (np.abs(o) + 1e-2)
            bad = rel.max(axis=2) > 1e-3
            nee = ((gsig ^ osig) & 0xFFFF) != 0
            prim = ((gsig ^ osig) >> 16) != 0
            s2 = s.copy()
            s2.debugShadowSlack = 1e-3
            g2, _ = dev.render_image(s2, 1)
            o2, _, _ = osc.render(s2, 1, threads=32)
            bad2 = (np.abs(g2 - o2) / (np.abs(o2) + 1e-2)).max(axis=2) > 1e-3
            print("%-28s %4dx%-4d depth %2d  within 1e-3: %.4f (%d differ: %d shadow-decision [%d marginal], %d other-primitive, %d same-signature)"
                  "  sig mismatch among agreeing %d  | slack 1e-3: %.4f" %
                  (os.path.basename(path), w, h, depth, 1.0 - bad.mean(), bad.sum(), (bad & nee & ~prim).sum(), (bad & nee & ~prim & marginal).sum(),
                   (bad & prim).sum(), (bad & ~nee & ~prim).sum(), (~bad & (nee | prim)).sum(), 1.0 - bad2.mean()), flush=True)
        dev.close()
        osc.close()


if __name__ == "__main__":
    main()
