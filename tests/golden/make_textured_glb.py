#!/usr/bin/env python3
"""Writes tests/golden/textured.glb: a small textured glTF for the Metal metallic-roughness model's texture path.

Two primitives over one vertex buffer (a 48 x 24 lat-long sphere with TEXCOORD_0, TEXCOORD_1 and analytic TANGENTs) plus a
ground quad WITHOUT tangents (the loader generates MikkTSpace tangents for it, like the reference):
  * "shell"  (upper sphere): base colour PNG (sRGB RGBA, alpha used by alphaMode MASK), metallic-roughness + occlusion in one
                             ORM PNG, normal map PNG, KHR_texture_transform on the base colour, mirrored-repeat sampler
  * "glow"   (lower sphere): emissive baseline JPEG (4:2:0) on TEXCOORD_1, alphaMode BLEND with a baseColorFactor alpha of 0.6
  * "ground" (quad)        : tiled base colour + the same normal map through a clamp-to-edge sampler, no vertex tangents
Images are generated procedurally (no RNG) and embedded as buffer views.  Needs Pillow for the encoders only; the committed
.glb is what the tests read (the library decodes PNG / JPEG itself)."""
import io
import json
import math
import os
import struct

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def _png(arr, mode):
    buf = io.BytesIO()
    Image.fromarray(arr, mode).save(buf, "PNG")
    return buf.getvalue()


def main(path=os.path.join(HERE, "textured.glb")):
    n_lon, n_lat = 48, 24
    pos, nrm, uv0, uv1, tan = [], [], [], [], []
    for i in range(n_lat + 1):
        theta = math.pi * i / n_lat
        for j in range(n_lon + 1):
            phi = 2.0 * math.pi * j / n_lon
            d = (math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi))
            pos.append(d)
            nrm.append(d)
            uv0.append((j / n_lon, i / n_lat))
            uv1.append((2.0 * j / n_lon, 0.5 + 0.5 * math.cos(theta)))
            tan.append((-math.sin(phi), 0.0, math.cos(phi), 1.0))       # d(position)/d(u), right-handed
    faces_top, faces_bottom = [], []
    for i in range(n_lat):
        for j in range(n_lon):
            a = i * (n_lon + 1) + j
            b, c, d = a + 1, a + n_lon + 1, a + n_lon + 2
            (faces_top if i < n_lat * 0.6 else faces_bottom).extend([(a, b, c), (b, d, c)])
    base = len(pos)
    for x, z, u, v in ((-3, -3, 0, 0), (3, -3, 4, 0), (-3, 3, 0, 4), (3, 3, 4, 4)):
        pos.append((x, -1.0, z))
        nrm.append((0.0, 1.0, 0.0))
        uv0.append((u, v))
        uv1.append((u * 0.25, v * 0.25))
        tan.append((0.0, 0.0, 0.0, 1.0))
    faces_ground = [(base, base + 2, base + 1), (base + 1, base + 2, base + 3)]
    pos, nrm = np.array(pos, "<f4"), np.array(nrm, "<f4")
    uv0, uv1, tan = np.array(uv0, "<f4"), np.array(uv1, "<f4"), np.array(tan, "<f4")

    # ---- images
    yy, xx = np.mgrid[0:128, 0:256]
    checker = ((xx // 16 + yy // 16) % 2).astype(np.float64)
    base_rgba = np.zeros((128, 256, 4), np.uint8)
    base_rgba[..., 0] = (60 + 180 * checker).astype(np.uint8)
    base_rgba[..., 1] = (200 - 120 * checker + 30 * np.sin(xx / 9.0)).clip(0, 255).astype(np.uint8)
    base_rgba[..., 2] = (90 + 100 * np.cos(yy / 11.0) ** 2).astype(np.uint8)
    base_rgba[..., 3] = np.where(((xx // 32) % 4 == 3) & ((yy // 32) % 2 == 1), 40, 255).astype(np.uint8)      # holes for MASK
    orm = np.zeros((64, 64, 3), np.uint8)
    oy, ox = np.mgrid[0:64, 0:64]
    orm[..., 0] = (140 + 100 * np.sin(ox / 5.0) * np.sin(oy / 7.0)).astype(np.uint8)                         # occlusion
    orm[..., 1] = (40 + 180 * (ox / 63.0)).astype(np.uint8)                                                  # roughness
    orm[..., 2] = np.where((oy // 8) % 2 == 0, 230, 20).astype(np.uint8)                                     # metallic
    ny, nx = np.mgrid[0:96, 0:96]
    bump = np.stack([0.6 * np.sin(nx / 4.0), 0.6 * np.cos(ny / 6.0), np.ones_like(nx, dtype=np.float64)], axis=-1)
    bump /= np.linalg.norm(bump, axis=-1, keepdims=True)
    normal_map = ((bump * 0.5 + 0.5) * 255.0 + 0.5).astype(np.uint8)
    ey, ex = np.mgrid[0:64, 0:128]
    glow = np.zeros((64, 128, 3), np.uint8)
    glow[..., 0] = (255 * np.exp(-((ex % 32 - 16) ** 2 + (ey % 32 - 16) ** 2) / 60.0)).astype(np.uint8)
    glow[..., 1] = (glow[..., 0] * 0.55).astype(np.uint8)
    glow[..., 2] = (glow[..., 0] * 0.15).astype(np.uint8)
    jpg = io.BytesIO()
    Image.fromarray(glow, "RGB").save(jpg, "JPEG", quality=88, subsampling=2)
    tiles = np.zeros((32, 32, 3), np.uint8)
    ty, tx = np.mgrid[0:32, 0:32]
    tiles[...] = np.where(((tx < 2) | (ty < 2))[..., None], (40, 40, 45), (170, 165, 150))
    images = [_png(base_rgba, "RGBA"), _png(orm, "RGB"), _png(normal_map, "RGB"), jpg.getvalue(), _png(tiles, "RGB")]
    mimes = ["image/png", "image/png", "image/png", "image/jpeg", "image/png"]

    chunks, views, accessors = [], [], []
    offset = 0

    def add_view(data, target=None):
        nonlocal offset
        pad = (4 - offset % 4) % 4
        if pad:
            chunks.append(b"\0" * pad)
            offset += pad
        v = {"buffer": 0, "byteOffset": offset, "byteLength": len(data)}
        if target:
            v["target"] = target
        views.append(v)
        chunks.append(data)
        offset += len(data)
        return len(views) - 1

    def add_accessor(arr, kind, ctype=5126, target=34962, minmax=False):
        a = {"bufferView": add_view(arr.tobytes(), target), "componentType": ctype, "count": int(arr.shape[0] if arr.ndim > 1 else arr.size), "type": kind}
        if minmax:
            a["min"] = [float(x) for x in arr.min(axis=0)]
            a["max"] = [float(x) for x in arr.max(axis=0)]
        accessors.append(a)
        return len(accessors) - 1

    a_pos = add_accessor(pos, "VEC3", minmax=True)
    a_nrm = add_accessor(nrm, "VEC3")
    a_uv0 = add_accessor(uv0, "VEC2")
    a_uv1 = add_accessor(uv1, "VEC2")
    a_tan = add_accessor(tan, "VEC4")
    prims = []
    for faces, material, with_tangent in ((faces_top, 0, True), (faces_bottom, 1, True), (faces_ground, 2, False)):
        idx = np.array(faces, "<u2").ravel()
        a_idx = add_accessor(idx, "SCALAR", ctype=5123, target=34963)
        attrs = {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv0, "TEXCOORD_1": a_uv1}
        if with_tangent:
            attrs["TANGENT"] = a_tan
        prims.append({"attributes": attrs, "indices": a_idx, "material": material, "mode": 4})
    image_defs = [{"bufferView": add_view(data), "mimeType": mime} for data, mime in zip(images, mimes)]
    doc = {
        "asset": {"version": "2.0", "generator": "tests/golden/make_textured_glb.py"},
        "scene": 0, "scenes": [{"nodes": [0]}],
        "nodes": [{"name": "object", "mesh": 0, "scale": [1.5, 1.5, 1.5], "translation": [0.0, 1.5, 0.0]}],
        "meshes": [{"name": "textured", "primitives": prims}],
        "samplers": [{"wrapS": 33648, "wrapT": 10497, "magFilter": 9729}, {"wrapS": 33071, "wrapT": 33071}, {"magFilter": 9728}],
        "images": image_defs,
        "textures": [{"source": 0, "sampler": 0}, {"source": 1}, {"source": 2}, {"source": 3}, {"source": 4}, {"source": 2, "sampler": 1}],
        "materials": [
            {"name": "shell", "alphaMode": "MASK", "alphaCutoff": 0.5, "doubleSided": True,
             "pbrMetallicRoughness": {"baseColorFactor": [1.0, 0.95, 0.9, 1.0], "metallicFactor": 1.0, "roughnessFactor": 0.8,
                                      "baseColorTexture": {"index": 0, "extensions": {"KHR_texture_transform": {"offset": [0.1, 0.0], "scale": [2.0, 1.0], "rotation": 0.2}}},
                                      "metallicRoughnessTexture": {"index": 1}},
             "normalTexture": {"index": 2, "scale": 1.0}, "occlusionTexture": {"index": 1, "strength": 0.8}},
            {"name": "glow", "alphaMode": "BLEND",
             "pbrMetallicRoughness": {"baseColorFactor": [0.3, 0.3, 0.35, 0.6], "metallicFactor": 0.0, "roughnessFactor": 0.5},
             "emissiveFactor": [1.0, 1.0, 1.0], "emissiveTexture": {"index": 3, "texCoord": 1},
             "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0}}},
            {"name": "ground", "pbrMetallicRoughness": {"baseColorTexture": {"index": 4}, "metallicFactor": 0.0, "roughnessFactor": 0.7},
             "normalTexture": {"index": 5, "scale": 0.7}}],
        "extensionsUsed": ["KHR_texture_transform", "KHR_materials_emissive_strength"],
        "bufferViews": views, "accessors": accessors,
    }
    binary = b"".join(chunks)
    binary += b"\0" * ((4 - len(binary) % 4) % 4)
    doc["buffers"] = [{"byteLength": len(binary)}]
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((4 - len(js) % 4) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(binary)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A))
        f.write(js)
        f.write(struct.pack("<II", len(binary), 0x004E4942))
        f.write(binary)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
