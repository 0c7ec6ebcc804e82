"""The host loaders against the reference's own third-party code, compiled from /root/reference into oracle/_ref by oracle/Makefile:
 * csrc/host/tangent_space.cpp (own MikkTSpace implementation) against external/MikkTSpace/mikktspace.c: tangent and sign per corner;
 * csrc/host/mesh_loaders.cpp (own OBJ / PLY readers) against external/tinyobjloader and external/tinyply with the reference's
   vertex / index assembly (src/renderer/SceneManager.mm:96-209, 223-518): positions, normals and indices bit-equal."""
import ctypes as C
import importlib
import os
import struct

import numpy as np
import pytest

import ref_loaders as ref

pt = importlib.import_module("metal-pathtracer-arm64_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref libraries not built (no reference tree)")


def own_tangents(p, n, t):
    lib = pt.load_library()
    lib.ptr_debug_generate_tangents.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_uint64, C.POINTER(C.c_float)]
    p, n, t = (np.ascontiguousarray(a, np.float32) for a in (p, n, t))
    out = np.zeros((p.shape[0], 4), np.float32)
    fp = C.POINTER(C.c_float)
    assert lib.ptr_debug_generate_tangents(p.ctypes.data_as(fp), n.ctypes.data_as(fp), t.ctypes.data_as(fp), p.shape[0] // 3, out.ctypes.data_as(fp)) == 0
    return out


def soup(positions, normals, uvs, faces):
    """indexed mesh -> triangle soup (one entry per corner), what the reference hands its tangent library"""
    f = np.asarray(faces, np.int64).reshape(-1)
    return positions[f].astype(np.float32), normals[f].astype(np.float32), uvs[f].astype(np.float32)


def unit(v):
    return (v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-30)).astype(np.float32)


def sphere(rows, cols, bump=0.0, mirror=False):
    """lat-long sphere with a texture seam (the first and last column share positions, not texture coordinates) and poles (degenerate
    quads: their triangles have two coinciding positions); mirror: the u coordinate runs backwards on the lower half"""
    th = np.linspace(0.0, np.pi, rows + 1)[:, None]
    ph = np.linspace(0.0, 2.0 * np.pi, cols + 1)[None, :]
    r = 1.0 + bump * np.sin(5 * th) * np.sin(3 * ph)
    pos = np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th) * np.ones_like(ph), r * np.sin(th) * np.sin(ph)], axis=-1).reshape(-1, 3).astype(np.float32)
    pos[cols::cols + 1] = pos[0::cols + 1]          # close the seam exactly
    nrm = unit(pos.astype(np.float64))
    u = np.broadcast_to(ph / (2.0 * np.pi), (rows + 1, cols + 1)).copy()
    v = np.broadcast_to(th / np.pi, (rows + 1, cols + 1)).copy()
    if mirror:
        u[rows // 2:] = 1.0 - u[rows // 2:]
    uv = np.stack([u, v], axis=-1).reshape(-1, 2).astype(np.float32)
    faces = []
    for i in range(rows):
        for j in range(cols):
            a, b, c, d = i * (cols + 1) + j, i * (cols + 1) + j + 1, (i + 1) * (cols + 1) + j, (i + 1) * (cols + 1) + j + 1
            faces += [[a, c, b], [b, c, d]]
    return pos, nrm, uv, np.array(faces)


def check(p, n, t, what):
    want, ok = ref.mikktspace(p, n, t)
    assert ok, what
    got = own_tangents(p, n, t)
    assert np.array_equal(got[:, 3], want[:, 3]), (what, "signs", int((got[:, 3] != want[:, 3]).sum()))
    equal = float((got[:, :3] == want[:, :3]).all(axis=1).mean())
    worst = float(np.abs(got[:, :3] - want[:, :3]).max())
    assert worst <= 2e-6 and equal >= 0.999, (what, equal, worst)
    return equal


def test_tangents_match_the_reference_library_on_smooth_seamed_and_mirrored_meshes():
    assert check(*soup(*sphere(12, 16)), "sphere") == 1.0                      # bit for bit
    assert check(*soup(*sphere(24, 40, bump=0.2)), "bumpy sphere") == 1.0
    check(*soup(*sphere(10, 14, bump=0.1, mirror=True)), "mirrored lower half")
    # flat-shaded: every triangle its own normals, so welding joins nothing across edges with different normals
    pos, nrm, uv, faces = sphere(8, 10, bump=0.3)
    p, n, t = soup(pos, nrm, uv, faces)
    flat = unit(np.cross(p[1::3] - p[0::3], p[2::3] - p[0::3]).astype(np.float64))
    check(p, np.repeat(flat, 3, axis=0), t, "flat shaded")


def test_tangents_match_on_degenerate_and_pathological_input():
    rng = np.random.default_rng(7)
    # unconnected random triangles, some with all three texture coordinates equal (no usable mapping), some with two equal positions
    p = rng.normal(size=(300, 3, 3)).astype(np.float32)
    t = rng.random(size=(300, 3, 2)).astype(np.float32)
    t[::7] = t[::7, :1]
    p[::11, 1] = p[::11, 0]
    n = unit(rng.normal(size=(300, 3, 3)))
    check(p.reshape(-1, 3), n.reshape(-1, 3), t.reshape(-1, 2), "random soup")
    # a fan of four triangles around one edge (more than two triangles per edge), shared normals and coordinates at the shared vertices
    a, b = np.array([0, 0, 0], np.float32), np.array([0, 1, 0], np.float32)
    wings = [np.array([np.cos(k), 0.3, np.sin(k)], np.float32) for k in (0.0, 1.3, 2.9, 4.4)]
    pos = np.array([[a, b, w] if i % 2 == 0 else [b, a, w] for i, w in enumerate(wings)], np.float32).reshape(-1, 3)
    uv = np.array([[[0, 0], [0, 1], [1, 0.5]] if i % 2 == 0 else [[0, 1], [0, 0], [1, 0.5]] for i in range(4)], np.float32).reshape(-1, 2)
    nrm = np.tile(np.array([[1, 0, 0]], np.float32), (12, 1))
    check(pos, nrm, uv, "butterfly")
    # a mesh whose texture coordinates collapse to a line on one side (zero-area mapping next to a healthy one)
    pos, nrm, uv, faces = sphere(6, 8)
    uv[: len(uv) // 2, 0] = 0.25
    check(*soup(pos, nrm, uv, faces), "collapsed mapping")


def _own_mesh(tmp_path, asset_path):
    scene = tmp_path / "one_mesh.scene"
    scene.write_text("material type=lambert albedo=0.5,0.5,0.5\nmesh path=%s material=0\n" % asset_path)
    host = pt.HostScene.load(str(scene), str(tmp_path))
    assert host.desc.meshCount == 1
    m = host.desc.meshes[0]
    v, i = int(m.vertexCount), int(m.indexCount)
    return (np.ctypeslib.as_array(m.positions, (v, 3)).copy(), np.ctypeslib.as_array(m.normals, (v, 3)).copy(), np.ctypeslib.as_array(m.indices, (i,)).copy(), host)


def _same_mesh(own, want, fallback_normals):
    positions, normals, indices, _ = own
    assert positions.shape == want["positions"].shape and indices.shape == want["indices"].shape
    assert np.array_equal(indices, want["indices"]) and np.array_equal(positions.view(np.uint32), want["positions"].view(np.uint32))
    if fallback_normals:   # vertices the file gives no normal get their triangle's (ApplyFallbackNormals, SceneManager.mm:69-94): only the given ones compare
        given = np.linalg.norm(want["normals"], axis=1) > 0
        assert np.array_equal(normals[given].view(np.uint32), want["normals"][given].view(np.uint32))
    else:
        assert np.array_equal(normals.view(np.uint32), want["normals"].view(np.uint32))


def test_obj_reader_matches_tinyobjloader(tmp_path):
    from scenes.gen_assets import ensure_assets
    ensure_assets()
    for name in ("blob_1152.obj", "blob_70688.obj"):          # generated assets: v / vn, triangles
        path = os.path.join(SCENES, "assets", name)
        _same_mesh(_own_mesh(tmp_path, path), ref.load_mesh(path), False)
    # hand-written: texture coordinates, negative (relative) indices, a quad and a pentagon to triangulate, two groups, a vertex used
    # with two different normals (two vertices), comments, exponents, trailing whitespace
    text = """# test
o first
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 1.5 1e-1
vn 0 0 1
vn 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
f 1/1/1 2/2/1 3/3/1 4/4/1
g second
f -5//2 -4//2 -3//2 -2//2 -1//2
f 1/1/2 3/3/1 2/2/1   
"""
    path = tmp_path / "hand.obj"
    path.write_text(text)
    _same_mesh(_own_mesh(tmp_path, str(path)), ref.load_mesh(str(path)), False)


def test_ply_reader_matches_tinyply(tmp_path):
    from scenes.gen_assets import ensure_assets, ensure_large_asset
    ensure_assets()
    ensure_large_asset("blob_125000.ply")   # (generated on demand: a fresh checkout does not have it)
    path = os.path.join(SCENES, "assets", "blob_125000.ply")      # generated asset: binary little endian, float positions, int faces
    _same_mesh(_own_mesh(tmp_path, path), ref.load_mesh(path), True)
    # ASCII, double precision, normals, uchar list counts
    ascii_ply = """ply
format ascii 1.0
comment hand-written
element vertex 5
property double x
property double y
property double z
property float nx
property float ny
property float nz
element face 3
property list uchar int vertex_indices
end_header
0 0 0 0 0 1
1 0 0 0 0 1
1 1 0.125 0 0 1
0 1 0 0 0 1
0.5 0.5 1 0 1 0
3 0 1 2
3 0 2 3
3 3 2 4
"""
    p1 = tmp_path / "hand_ascii.ply"
    p1.write_text(ascii_ply)
    _same_mesh(_own_mesh(tmp_path, str(p1)), ref.load_mesh(str(p1)), False)
    # (the reference asks its library for the face lists with a size hint of 3, SceneManager.mm:262-265: a file of quads does not parse
    # there at all; this reader fans them from the first corner)
    quads = tmp_path / "quads.ply"
    quads.write_text(ascii_ply.replace("element face 3", "element face 1").replace("3 0 1 2\n3 0 2 3\n3 3 2 4\n", "4 0 1 2 3\n"))
    with pytest.raises(RuntimeError):
        ref.load_mesh(str(quads))
    assert np.array_equal(_own_mesh(tmp_path, str(quads))[2], [0, 1, 2, 0, 2, 3])
    # binary little endian without normals, ushort indices under the alternative property name
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
              "element face 2\nproperty list uchar ushort vertex_index\nend_header\n").encode()
    body = struct.pack("<12f", 0, 0, 0, 2, 0, 0, 2, 2, 0, 0, 2, 1) + struct.pack("<BHHH", 3, 0, 1, 2) + struct.pack("<BHHH", 3, 0, 2, 3)
    p2 = tmp_path / "hand_binary.ply"
    p2.write_bytes(header + body)
    _same_mesh(_own_mesh(tmp_path, str(p2)), ref.load_mesh(str(p2)), True)
