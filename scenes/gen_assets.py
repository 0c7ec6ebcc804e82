#!/usr/bin/env python3
"""Deterministic synthetic assets for the BASELINE configs (no RNG, closed-form geometry).

blob OBJ: lat-long grid of N x N quads on a radius-100 sphere displaced by r*(1 + 0.15*sin(7*theta)*sin(5*phi)),
with analytic `vn` normals (avoids reference quirk Q3: OBJ vertices without normals shade as +Y).
N = 188 gives 2*188*188 = 70,688 triangles.
"""
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def blob_obj(path: str, n: int = 188, radius: float = 100.0) -> int:
    def radius_at(theta, phi):
        return radius * (1.0 + 0.15 * math.sin(7.0 * theta) * math.sin(5.0 * phi))

    def point(theta, phi):
        r = radius_at(theta, phi)
        return (r * math.sin(theta) * math.cos(phi), r * math.cos(theta), r * math.sin(theta) * math.sin(phi))

    def normal(theta, phi):
        r = radius_at(theta, phi)
        dr_dt = radius * 0.15 * 7.0 * math.cos(7.0 * theta) * math.sin(5.0 * phi)
        dr_dp = radius * 0.15 * 5.0 * math.sin(7.0 * theta) * math.cos(5.0 * phi)
        st, ct, sp, cp = math.sin(theta), math.cos(theta), math.sin(phi), math.cos(phi)
        d = (st * cp, ct, st * sp)
        pt = (dr_dt * d[0] + r * ct * cp, dr_dt * d[1] - r * st, dr_dt * d[2] + r * ct * sp)
        pp = (dr_dp * d[0] - r * st * sp, dr_dp * d[1], dr_dp * d[2] + r * st * cp)
        nx = pp[1] * pt[2] - pp[2] * pt[1]
        ny = pp[2] * pt[0] - pp[0] * pt[2]
        nz = pp[0] * pt[1] - pp[1] * pt[0]
        length = math.sqrt(nx * nx + ny * ny + nz * nz)
        if length < 1e-9:
            return d
        nx, ny, nz = nx / length, ny / length, nz / length
        if nx * d[0] + ny * d[1] + nz * d[2] < 0.0:
            nx, ny, nz = -nx, -ny, -nz
        return (nx, ny, nz)

    lines = ["# lat-long displaced sphere, %d x %d quads" % (n, n)]
    for i in range(n + 1):
        theta = math.pi * i / n
        for j in range(n + 1):
            phi = 2.0 * math.pi * j / n
            p = point(theta, phi)
            lines.append("v %.8g %.8g %.8g" % p)
    for i in range(n + 1):
        theta = math.pi * i / n
        for j in range(n + 1):
            phi = 2.0 * math.pi * j / n
            lines.append("vn %.8g %.8g %.8g" % normal(theta, phi))
    tris = 0
    for i in range(n):
        for j in range(n):
            a = i * (n + 1) + j + 1
            b = a + 1
            c = a + (n + 1)
            d = c + 1
            # counter-clockwise seen from outside
            lines.append("f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c))
            lines.append("f %d//%d %d//%d %d//%d" % (b, b, d, d, c, c))
            tris += 2
    with open(path, "w") as f:
        f.write("\n".join(lines))
        f.write("\n")
    return tris


def sky_hdr(path: str, width: int, height: int) -> None:
    """Synthetic equirectangular environment (Radiance RGBE, flat scanlines): blue-to-white sky gradient,
    a dim ground, and three Gaussian suns of very different peak radiance at fixed (theta, phi)."""
    import numpy as np

    suns = [(0.9, 1.0, 5.0e4, 0.03, (1.0, 0.9, 0.7)), (1.2, 3.6, 8.0e2, 0.08, (0.7, 0.8, 1.0)), (0.5, 5.2, 60.0, 0.2, (1.0, 0.5, 0.3))]
    theta = (np.pi * (np.arange(height) + 0.5) / height)[:, None]
    phi = (2.0 * np.pi * (np.arange(width) + 0.5) / width)[None, :]
    up = np.cos(theta) + 0.0 * phi
    t = np.clip(up, 0.0, 1.0)
    sky = np.stack([0.9 * (1 - t) + 0.25 * t, 0.95 * (1 - t) + 0.45 * t, 1.0 * (1 - t) + 0.9 * t], axis=-1)
    ground = np.broadcast_to(np.array([0.12, 0.10, 0.08]), sky.shape)
    rgb = np.where((up > 0.0)[..., None], sky, ground).astype(np.float64)
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) + 0.0 * phi, np.sin(theta) * np.sin(phi)], axis=-1)
    for st, sp, peak, sigma, tint in suns:
        sd = np.array([math.sin(st) * math.cos(sp), math.cos(st), math.sin(st) * math.sin(sp)])
        ang = np.arccos(np.clip(d @ sd, -1.0, 1.0))
        rgb = rgb + (peak * np.exp(-0.5 * (ang / sigma) ** 2))[..., None] * np.array(tint)
    m = rgb.max(axis=-1)
    mant, e = np.frexp(m)
    scale = np.where(m < 1e-32, 0.0, mant * 256.0 / np.maximum(m, 1e-300))
    out = np.zeros((height, width, 4), dtype=np.uint8)
    out[..., :3] = np.clip((rgb * scale[..., None]).astype(np.int64), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m < 1e-32, 0, e + 128).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (height, width))
        f.write(out.tobytes())


# ---------------------------------------------------------------------------------------------------------
# Large procedural meshes (numpy): parametric grids written as binary little-endian PLY with vertex normals.
def _grid_faces(nu: int, nv: int, wrap_u: bool, wrap_v: bool):
    """Two triangles per quad of an (nu x nv)-quad parametric grid; vertex (i, j) has index i * cols + j."""
    import numpy as np

    cols = nv if wrap_v else nv + 1
    rows = nu if wrap_u else nu + 1
    i = np.arange(nu, dtype=np.int64)[:, None]
    j = np.arange(nv, dtype=np.int64)[None, :]
    i1 = (i + 1) % rows if wrap_u else i + 1
    j1 = (j + 1) % cols if wrap_v else j + 1
    a = (i * cols + j).ravel()
    b = (i * cols + j1).ravel()
    c = (i1 * cols + j).ravel()
    d = (i1 * cols + j1).ravel()
    faces = np.empty((a.size * 2, 3), dtype=np.int32)
    faces[0::2] = np.stack([a, b, c], axis=1)
    faces[1::2] = np.stack([b, d, c], axis=1)
    return faces


def _grid_normals(pos, wrap_u: bool, wrap_v: bool, outward_hint=None):
    """Unit normals of a parametric grid from central differences of the position array [rows, cols, 3]."""
    import numpy as np

    def diff(a, axis, wrap):
        if wrap:
            return np.roll(a, -1, axis=axis) - np.roll(a, 1, axis=axis)
        return np.gradient(a, axis=axis)

    du = diff(pos, 0, wrap_u)
    dv = diff(pos, 1, wrap_v)
    n = np.cross(dv, du)
    length = np.linalg.norm(n, axis=-1, keepdims=True)
    fallback = pos / np.maximum(np.linalg.norm(pos, axis=-1, keepdims=True), 1e-30) if outward_hint is None else outward_hint
    n = np.where(length > 1e-12, n / np.maximum(length, 1e-30), fallback)
    flip = (n * fallback).sum(axis=-1, keepdims=True) < 0.0
    return np.where(flip, -n, n)


def write_ply(path: str, pos, nrm, faces) -> int:
    import numpy as np

    pos = np.asarray(pos, dtype="<f4").reshape(-1, 3)
    nrm = np.asarray(nrm, dtype="<f4").reshape(-1, 3)
    verts = np.empty(pos.shape[0], dtype=[("p", "<f4", 3), ("n", "<f4", 3)])
    verts["p"] = pos
    verts["n"] = nrm
    rec = np.empty(faces.shape[0], dtype=[("k", "u1"), ("i", "<i4", 3)])
    rec["k"] = 3
    rec["i"] = faces
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "property float nx\nproperty float ny\nproperty float nz\nelement face %d\n"
                 "property list uchar int vertex_indices\nend_header\n" % (pos.shape[0], faces.shape[0])).encode())
        f.write(verts.tobytes())
        f.write(rec.tobytes())
    return int(faces.shape[0])


def displaced_sphere(n: int, radius: float, octaves):
    """Lat-long grid of n x n quads (2*n*n triangles incl. the degenerate pole fans, like blob_obj);
    r = radius * (1 + sum a*sin(k1*theta)*sin(k2*phi))."""
    import numpy as np

    theta = (np.pi * np.arange(n + 1) / n)[:, None]
    phi = (2.0 * np.pi * np.arange(n + 1) / n)[None, :]
    r = np.ones((n + 1, n + 1))
    for amp, k1, k2 in octaves:
        r = r + amp * np.sin(k1 * theta) * np.sin(k2 * phi)
    r = r * radius
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta) + 0.0 * phi, np.sin(theta) * np.sin(phi)], axis=-1)
    pos = d * r[..., None]
    nrm = _grid_normals(pos, False, False, outward_hint=d)
    return pos.reshape(-1, 3), nrm.reshape(-1, 3), _grid_faces(n, n, False, False)


def torus_knot(nu: int = 1320, nv: int = 330, p: int = 2, q: int = 3, big: float = 100.0, small: float = 40.0, tube: float = 16.0):
    """Closed genus-1 tube around a (p, q) torus knot: nu x nv quads, both directions periodic -> 2*nu*nv triangles
    (1320 x 330 -> 871,200: the config-4 stand-in for the Stanford dragon)."""
    import numpy as np

    u = 2.0 * np.pi * np.arange(nu) / nu
    centre = np.stack([(big + small * np.cos(q * u)) * np.cos(p * u), small * np.sin(q * u), (big + small * np.cos(q * u)) * np.sin(p * u)], axis=-1)
    tangent = np.roll(centre, -1, axis=0) - np.roll(centre, 1, axis=0)
    tangent /= np.linalg.norm(tangent, axis=-1, keepdims=True)
    # frame: project a fixed reference (the curve's radial direction) off the tangent -> continuous and closed
    ref = centre * np.array([1.0, 0.0, 1.0])
    ref /= np.maximum(np.linalg.norm(ref, axis=-1, keepdims=True), 1e-12)
    nrm1 = ref - tangent * (ref * tangent).sum(axis=-1, keepdims=True)
    nrm1 /= np.linalg.norm(nrm1, axis=-1, keepdims=True)
    nrm2 = np.cross(tangent, nrm1)
    v = (2.0 * np.pi * np.arange(nv) / nv)[None, :, None]
    ring = np.cos(v) * nrm1[:, None, :] + np.sin(v) * nrm2[:, None, :]
    pos = centre[:, None, :] + tube * ring
    nrm = _grid_normals(pos, True, True, outward_hint=ring)
    return pos.reshape(-1, 3), nrm.reshape(-1, 3), _grid_faces(nu, nv, True, True)


def ensure_large_asset(name: str, verbose: bool = False) -> str:
    """Generate one of the large stand-in meshes on demand (they are far too big to keep in the repository)."""
    assets = os.path.join(HERE, "assets")
    os.makedirs(assets, exist_ok=True)
    path = os.path.join(assets, name)
    if os.path.exists(path):
        return path
    tmp = path + ".tmp%d" % os.getpid()
    if name == "torus_knot_871200.ply":
        tris = write_ply(tmp, *torus_knot())
    elif name == "lucy_standin_28005128.ply":
        tris = write_ply(tmp, *displaced_sphere(3742, 100.0, [(0.15, 7, 5), (0.05, 23, 17), (0.015, 61, 47), (0.004, 173, 131)]))
    elif name == "blob_1002528.ply":
        tris = write_ply(tmp, *displaced_sphere(708, 100.0, [(0.12, 5, 4), (0.03, 19, 13)]))
    elif name == "blob_125000.ply":
        tris = write_ply(tmp, *displaced_sphere(250, 100.0, [(0.15, 7, 5), (0.04, 23, 17)]))
    else:
        raise ValueError("unknown large asset " + name)
    os.replace(tmp, path)
    if verbose:
        print("wrote", path, tris, "triangles")
    return path


# ---------------------------------------------------------------------------------------------------------
# Config-3 stand-in for DamagedHelmet: a .glb with three primitives and factor-only PBR materials.
def helmet_glb(path: str, n: int = 152) -> int:
    """Displaced sphere of 2*n*n triangles (n = 152 -> 46,208) split by latitude into three primitives:
    a metal shell (metallic 1, roughness 0.3), a dielectric band (metallic 0, roughness 0.6) and an emissive strip
    (KHR_materials_emissive_strength).  Geometry sits under two nested nodes (scale, then rotation+translation) so
    the loader's hierarchy code is exercised; indices are uint32, uint16-sized ones are exercised by the unit tests."""
    import json
    import struct

    import numpy as np

    pos, nrm, faces = displaced_sphere(n, 1.0, [(0.12, 6, 4), (0.03, 17, 11)])
    pos = pos.astype("<f4")
    nrm = nrm.astype("<f4")
    rows = faces.shape[0] // (2 * n)            # = n latitude bands of 2*n triangles each
    band = np.repeat(np.arange(rows), 2 * n)
    groups = [band < int(0.55 * n), (band >= int(0.55 * n)) & (band < int(0.62 * n)), band >= int(0.62 * n)]
    chunks = [pos.tobytes(), nrm.tobytes()]
    views = [{"buffer": 0, "byteOffset": 0, "byteLength": pos.nbytes, "target": 34962},
             {"buffer": 0, "byteOffset": pos.nbytes, "byteLength": nrm.nbytes, "target": 34962}]
    accessors = [{"bufferView": 0, "componentType": 5126, "count": int(pos.shape[0]), "type": "VEC3",
                  "min": [float(x) for x in pos.min(axis=0)], "max": [float(x) for x in pos.max(axis=0)]},
                 {"bufferView": 1, "componentType": 5126, "count": int(nrm.shape[0]), "type": "VEC3"}]
    offset = pos.nbytes + nrm.nbytes
    prims = []
    # order: strip (material 2) is the middle group
    for g, material in zip(groups, (0, 2, 1)):
        idx = faces[g].astype("<u4").ravel()
        chunks.append(idx.tobytes())
        views.append({"buffer": 0, "byteOffset": offset, "byteLength": idx.nbytes, "target": 34963})
        accessors.append({"bufferView": len(views) - 1, "componentType": 5125, "count": int(idx.size), "type": "SCALAR"})
        prims.append({"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": len(accessors) - 1, "material": material, "mode": 4})
        offset += idx.nbytes
    doc = {
        "asset": {"version": "2.0", "generator": "scenes/gen_assets.py"},
        "scene": 0,
        "scenes": [{"nodes": [0]}],
        "nodes": [{"name": "root", "scale": [100.0, 100.0, 100.0], "children": [1]},
                  {"name": "helmet", "mesh": 0, "rotation": [0.0, 0.38268343, 0.0, 0.92387953], "translation": [0.0, 0.1, 0.0]}],
        "meshes": [{"name": "shell", "primitives": prims}],
        "materials": [
            {"name": "metal", "pbrMetallicRoughness": {"baseColorFactor": [0.85, 0.78, 0.6, 1.0], "metallicFactor": 1.0, "roughnessFactor": 0.3}},
            {"name": "paint", "pbrMetallicRoughness": {"baseColorFactor": [0.25, 0.35, 0.6, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.6}},
            {"name": "strip", "pbrMetallicRoughness": {"baseColorFactor": [0.02, 0.02, 0.02, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9},
             "emissiveFactor": [1.0, 0.45, 0.1], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 6.0}}}],
        "extensionsUsed": ["KHR_materials_emissive_strength"],
        "buffers": [{"byteLength": offset}],
        "bufferViews": views,
        "accessors": accessors,
    }
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((4 - len(js) % 4) % 4)
    binary = b"".join(chunks)
    binary += b"\0" * ((4 - len(binary) % 4) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(binary)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A))
        f.write(js)
        f.write(struct.pack("<II", len(binary), 0x004E4942))
        f.write(binary)
    return int(faces.shape[0])


def ensure_assets(verbose: bool = False) -> None:
    assets = os.path.join(HERE, "assets")
    os.makedirs(assets, exist_ok=True)
    blob = os.path.join(assets, "blob_70688.obj")
    if not os.path.exists(blob):
        tris = blob_obj(blob, 188)
        if verbose:
            print("wrote", blob, tris, "triangles")
    small = os.path.join(assets, "blob_1152.obj")
    if not os.path.exists(small):
        tris = blob_obj(small, 24)
        if verbose:
            print("wrote", small, tris, "triangles")
    helmet = os.path.join(assets, "helmet_standin.glb")
    if not os.path.exists(helmet):
        tris = helmet_glb(helmet)
        if verbose:
            print("wrote", helmet, tris, "triangles")
    for name, w, h in (("sky_96x48.hdr", 96, 48), ("sky_1024x512.hdr", 1024, 512), ("sky_2048x1024.hdr", 2048, 1024)):
        path = os.path.join(assets, name)
        if not os.path.exists(path):
            sky_hdr(path, w, h)
            if verbose:
                print("wrote", path)


if __name__ == "__main__":
    ensure_assets(verbose=True)
    sys.exit(0)
