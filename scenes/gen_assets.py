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
    import struct

    suns = [(0.9, 1.0, 5.0e4, 0.03, (1.0, 0.9, 0.7)), (1.2, 3.6, 8.0e2, 0.08, (0.7, 0.8, 1.0)), (0.5, 5.2, 60.0, 0.2, (1.0, 0.5, 0.3))]
    rows = []
    for y in range(height):
        theta = math.pi * (y + 0.5) / height
        row = bytearray()
        for x in range(width):
            phi = 2.0 * math.pi * (x + 0.5) / width
            up = math.cos(theta)
            if up > 0.0:
                t = up
                rgb = [0.9 * (1 - t) + 0.25 * t, 0.95 * (1 - t) + 0.45 * t, 1.0 * (1 - t) + 0.9 * t]
            else:
                rgb = [0.12, 0.10, 0.08]
            d = (math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi))
            for st, sp, peak, sigma, tint in suns:
                sd = (math.sin(st) * math.cos(sp), math.cos(st), math.sin(st) * math.sin(sp))
                cosang = max(-1.0, min(1.0, d[0] * sd[0] + d[1] * sd[1] + d[2] * sd[2]))
                ang = math.acos(cosang)
                g = peak * math.exp(-0.5 * (ang / sigma) ** 2)
                rgb = [rgb[i] + g * tint[i] for i in range(3)]
            m = max(rgb)
            if m < 1e-32:
                row += bytes((0, 0, 0, 0))
            else:
                mant, e = math.frexp(m)
                scale = mant * 256.0 / m
                row += bytes((int(rgb[0] * scale), int(rgb[1] * scale), int(rgb[2] * scale), e + 128))
        rows.append(bytes(row))
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (height, width))
        for r in rows:
            f.write(r)


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
    for name, w, h in (("sky_96x48.hdr", 96, 48), ("sky_1024x512.hdr", 1024, 512)):
        path = os.path.join(assets, name)
        if not os.path.exists(path):
            sky_hdr(path, w, h)
            if verbose:
                print("wrote", path)


if __name__ == "__main__":
    ensure_assets(verbose=True)
    sys.exit(0)
