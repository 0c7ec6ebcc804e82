#!/usr/bin/env python3
"""Lane-utilisation bookkeeping of k_extend (counting build, library env PTR_VERBOSE=steps) for one scene:
    python tools/steps_probe.py [scene] [spp]        (knobs such as PTR_POOL_GROUPS come from the environment)"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PTR_VERBOSE"] = "steps"
pt = importlib.import_module("metal-pathtracer-arm64_amd")
from scenes.gen_assets import ensure_assets

ensure_assets()
scene = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scenes", "cornell_mesh.scene")
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
host = pt.HostScene.load(scene, os.path.join(ROOT, "scenes"))
dev = pt.DeviceScene(host.desc, 0, keepalive=host)
s = host.settings_for(seed=1337)
dev.render_image(s, 4)
img, st = dev.render_image(s, spp, count=True)
print("counting build: %.1f Msamples/s; extend %.1f ms shade %.1f ms connect %.1f ms; rays/sample %.2f nodes/ray %.1f prims/ray %.2f" % (
    s.width * s.height * spp / st.totalSeconds / 1e6, st.traceKernelMs, st.shadeKernelMs, st.shadowKernelMs,
    st.extendRays / (s.width * s.height * spp), st.extendNodesVisited / max(st.extendRays, 1), st.extendLeafPrimTests / max(st.extendRays, 1)))
img, st = dev.render_image(s, spp)
print("timed build:    %.1f Msamples/s; extend %.1f ms shade %.1f ms connect %.1f ms" % (
    s.width * s.height * spp / st.totalSeconds / 1e6, st.traceKernelMs, st.shadeKernelMs, st.shadowKernelMs))
