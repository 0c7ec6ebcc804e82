#!/usr/bin/env python3
"""Generates the golden output vectors under tests/golden/vectors/ from the oracle (oracle/liboracle.so).

  python tests/golden/make_goldens.py            # rewrite every vector
  python tests/golden/make_goldens.py --check    # regenerate in memory and compare with the committed files

The vectors pin the oracle across rounds (tests/test_golden.py, CPU) and give the HIP path a committed target that does
not need the oracle at test time (tests/test_gpu_golden.py).  Inputs: the scenes in tests/golden/*.scene; no reference
source text is stored, only inputs and the oracle's outputs.
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(HERE, "vectors")

# (name, scene file, width, height, depth, spp, seed, settings overrides)
IMAGES = [
    ("smoke_64x64_d4_4spp_seed1337", "smoke.scene", 64, 64, 4, 4, 1337, {}),
    ("cornell_64x64_d4_1spp_seed1337", "cornell_small_mesh.scene", 64, 64, 4, 1, 1337, {}),
    ("cornell_64x64_d4_32spp_seed1337", "cornell_small_mesh.scene", 64, 64, 4, 32, 1337, {}),
    ("cornell_64x64_d4_32spp_seed1338", "cornell_small_mesh.scene", 64, 64, 4, 32, 1338, {}),
    ("materials_96x64_d6_1spp_seed1337", "materials.scene", 96, 64, 6, 1, 1337, {}),
    ("materials_96x64_d6_16spp_seed1337", "materials.scene", 96, 64, 6, 16, 1337, {}),
    ("materials_96x64_d6_16spp_seed1338", "materials.scene", 96, 64, 6, 16, 1338, {}),
    ("env_materials_96x64_d6_1spp_seed1337", "env_materials.scene", 96, 64, 6, 1, 1337, {}),
]


def write_pfm(path, img):
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img[::-1], dtype="<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(x) for x in f.readline().split()]
        scale = float(f.readline())
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data[::-1].astype(np.float32)


def bsdf_inputs(n, seed):
    rng = np.random.default_rng(seed)
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.05
    wo = wo / np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rng.normal(size=(n, 3))
    wi[:, 2] = np.abs(wi[:, 2]) + 0.05
    wi = wi / np.linalg.norm(wi, axis=1, keepdims=True)
    pos = rng.uniform(-1, 1, size=(n, 3))
    normal = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    return pos.astype(np.float32), normal.astype(np.float32), wo.astype(np.float32), wi.astype(np.float32), states


def generate():
    pt = importlib.import_module("metal-pathtracer-arm64_amd")
    import oracle_lib as ol

    out = {"images": {}, "kat": {}}
    for name, scene, w, h, depth, spp, seed, over in IMAGES:
        host = pt.HostScene.load(os.path.join(HERE, scene), os.path.join(ROOT, "scenes"))
        s = host.settings_for(width=w, height=h, max_depth=depth, seed=seed, **over)
        img, _, counters = ol.OracleScene(host).render(s, spp, threads=0, count=True)
        out["images"][name] = (img, {k: int(v) for k, v in counters.items()})
    kat = out["kat"]
    kat["rng_hash"] = {str(x): int(ol.rng_hash(x)) for x in (0, 1, 1337, 0x9E3779B9, 0xFFFFFFFF)}
    fl, st = ol.rng_floats(1337, 8)
    kat["rng_floats_seed1337"] = {"floats": [float(v) for v in fl], "state": int(st)}
    host = pt.HostScene.load(os.path.join(HERE, "materials.scene"), os.path.join(ROOT, "scenes"))
    s = host.settings_for(width=96, height=64, max_depth=6, seed=1337)
    kat["camera_materials_96x64"] = [float(v) for v in ol.build_camera(s)]
    xys = np.array([[0, 0, 0], [95, 63, 0], [48, 32, 3], [7, 50, 11]], dtype=np.uint32)
    rays, states = ol.camera_rays(s, xys)
    kat["camera_rays"] = {"xys": xys.tolist(), "rays": rays.astype(np.float64).round(7).tolist(), "states": [int(v) for v in states]}
    pos, normal, wo, wi, states = bsdf_inputs(16, 7)
    d = host.desc
    bs = {}
    for i in range(d.materialCount):
        m = d.materials[i]
        ev = ol.eval_bsdf(m, s, np.concatenate([pos, normal, wo, wi], axis=1))
        sm, st2 = ol.sample_bsdf(m, s, np.concatenate([pos, normal, wo], axis=1), np.ones(16, np.uint32), states)
        bs[str(i)] = {"type": int(m.typeEta[0]), "eval": ev.astype(np.float64).tolist(), "sample": sm.astype(np.float64).tolist(),
                      "states": [int(v) for v in st2]}
    kat["bsdf_materials_scene"] = bs
    osc = ol.OracleScene(pt.HostScene.load(os.path.join(HERE, "cornell_small_mesh.scene"), os.path.join(ROOT, "scenes")))
    kat["cornell_small_mesh_scene_info"] = {k: int(v) for k, v in osc.info().items()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    out = generate()
    if args.check:
        bad = []
        for name, (img, counters) in out["images"].items():
            ref = read_pfm(os.path.join(OUT, name + ".pfm"))
            if not np.array_equal(ref, img):
                bad.append((name, float(np.abs(ref - img).max())))
        kat = json.load(open(os.path.join(OUT, "kat.json")))
        if json.loads(json.dumps(out["kat"])) != kat["kat"]:
            bad.append(("kat.json", None))
        print("differences:", bad if bad else "none")
        sys.exit(1 if bad else 0)
    os.makedirs(OUT, exist_ok=True)
    meta = {}
    for name, (img, counters) in out["images"].items():
        write_pfm(os.path.join(OUT, name + ".pfm"), img)
        meta[name] = counters
    json.dump({"generator": "tests/golden/make_goldens.py", "images": {n: {"scene": sc, "width": w, "height": h, "depth": dp, "spp": sp, "seed": sd,
                                                                         "counters": meta[n]}
                                                                     for n, sc, w, h, dp, sp, sd, _ in IMAGES},
               "kat": out["kat"]}, open(os.path.join(OUT, "kat.json"), "w"), indent=1)
    print("wrote", len(out["images"]), "images and kat.json to", OUT)


if __name__ == "__main__":
    main()
