#!/usr/bin/env python3
"""What a closest-hit walk costs per ray with two-, four- and eight-wide nodes (1, 2, 3 binary levels collapsed per step; four-wide nodes
whose children are chosen by box area, the way BuildWideNodes ships them) on the BASELINE
scenes' own trees: node steps, box tests, primitive tests (ptr_debug_walk_counts: host only, no GPU).  DESIGN.md section 4.3c.

  python tools/wide_walk_counts.py [--rays 200000] > profiles/r3_wide_walk_counts.txt
"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("metal-pathtracer-arm64_amd")


def rays_for(host, n, seed):
    """Half camera rays (coherent), half rays from random points of the scene's box in random directions (the bounces)."""
    d = host.desc
    lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
    for i in range(d.meshCount):
        m = d.meshes[i]
        p = np.ctypeslib.as_array(m.positions, shape=(m.vertexCount, 3))
        M = np.array(list(m.localToWorld), dtype=np.float64).reshape(4, 4)
        w = p @ M[:3, :3].T + M[:3, 3]
        if abs(np.linalg.det(M[:3, :3])) < 1e-12:   # (column-major stores: try the transpose)
            w = p
        lo, hi = np.minimum(lo, w.min(0)), np.maximum(hi, w.max(0))
    for i in range(d.rectCount):
        r = d.rects[i]
        c, u, v = np.array(list(r.corner)[:3]), np.array(list(r.edgeU)[:3]), np.array(list(r.edgeV)[:3])
        for q in (c, c + u, c + v, c + u + v):
            lo, hi = np.minimum(lo, q), np.maximum(hi, q)
    rng = np.random.default_rng(seed)
    org = rng.uniform(lo, hi, size=(n, 3))
    dirs = rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    rays = np.concatenate([org, np.full((n, 1), 1e-4), dirs, np.full((n, 1), np.inf)], axis=1).astype(np.float32)
    return np.ascontiguousarray(rays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=200000)
    args = ap.parse_args()
    from scenes.gen_assets import ensure_assets, ensure_large_asset
    ensure_assets()
    lib = pt.load_library()
    lib.ptr_debug_walk_counts.argtypes = [C.POINTER(pt.PtrSceneDesc), C.POINTER(C.c_float), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    print("# tools/wide_walk_counts.py: closest-hit walks of %d random rays (origins uniform in the scene's box, directions uniform) on the host,"
          % args.rays)
    print("# children in order of entry distance; per ray: node steps | box tests | primitive tests")
    for name, assets in (("cornell_mesh.scene", []), ("knot_glass.scene", ["torus_knot_871200.ply"]), ("helmet_env.scene", [])):
        for a in assets:
            ensure_large_asset(a)
        host = pt.HostScene.load(os.path.join(ROOT, "scenes", name), os.path.join(ROOT, "scenes"))
        rays = rays_for(host, args.rays, 5)
        print("== scenes/%s" % name)
        base = None
        for levels in (1, 2, 3, 5):   # 4 = four-wide, children chosen by box area (not a level count)
            out = (C.c_uint64 * 4)()
            err = C.create_string_buffer(512)
            rc = lib.ptr_debug_walk_counts(C.byref(host.desc), rays.ctypes.data_as(C.POINTER(C.c_float)), rays.shape[0], levels, out, err, len(err))
            if rc != 0:
                raise RuntimeError(err.value.decode())
            steps, boxes, prims, hits = [int(v) / rays.shape[0] for v in out]
            base = base or (steps, boxes)
            label = {1: "2-wide", 2: "4-wide by level", 3: "8-wide by level", 5: "4-wide by area (shipped)"}[levels]
            print("  %-26s %6.2f steps | %6.2f box tests | %5.2f primitive tests   (hit fraction %.3f; steps x%.2f, box tests x%.2f of the binary walk)"
                  % (label + ":", steps, boxes, prims, hits, steps / base[0], boxes / base[1]))


if __name__ == "__main__":
    main()
