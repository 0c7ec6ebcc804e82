#!/usr/bin/env python3
"""Host-side preparation time of a scene (no GPU work): parse + load, then world-space bake + BVH build + leaf-order arrays through
ptr_debug_scene_geometry, with the builder's own phase timings (PTR_VERBOSE=build).   python tools/build_time.py [scene] [repeats]"""
import importlib
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PTR_VERBOSE"] = "build"
pt = importlib.import_module("metal-pathtracer-arm64_amd")
from scenes.gen_assets import ensure_assets, ensure_large_asset

scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes", "lucy_standin.scene")
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ensure_assets()
for a in re.findall(r"assets/(\w+_\d{6,}\.ply)", open(scene).read()):
    ensure_large_asset(a)
t0 = time.time()
host = pt.HostScene.load(scene, os.path.join(ROOT, "scenes"))
print("parse + load %.2f s; %d hardware threads" % (time.time() - t0, os.cpu_count()), flush=True)
for _ in range(repeats):
    t0 = time.time()
    g = pt.debug_scene_geometry(host.desc)
    print("bake + BVH + leaf order %.2f s: %d nodes, depth %d, SAH cost %.2f, %d oversize triangles, 32 B nodes usable: %d" % (
        time.time() - t0, g["nodes"], g["max_depth"], g["sah_cost_milli"] / 1000.0, g["oversize"], g["quantized_usable"]), flush=True)
