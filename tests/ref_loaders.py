"""ctypes access to oracle/_ref/libref_mikktspace.so and libref_mesh_loaders.so: the reference's own tangent library and OBJ / PLY
parsers, compiled from /root/reference by oracle/Makefile behind drivers of ours (oracle/ref_mikktspace.c, ref_mesh_loaders.cpp).
TEST INFRASTRUCTURE: only tests import this."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
MIKK_PATH = os.path.join(REF_DIR, "libref_mikktspace.so")
MESH_PATH = os.path.join(REF_DIR, "libref_mesh_loaders.so")


def available():
    return os.path.exists(MIKK_PATH) and os.path.exists(MESH_PATH)


_fp = C.POINTER(C.c_float)


def mikktspace(positions, normals, uvs):
    """genTangSpaceDefault on a triangle soup ([3T, 3], [3T, 3], [3T, 2] float32) -> ([3T, 4] tangent + sign, ok)."""
    lib = C.CDLL(MIKK_PATH)
    lib.ref_mikktspace.argtypes = [_fp, _fp, _fp, C.c_int, _fp]
    lib.ref_mikktspace.restype = C.c_int
    p = np.ascontiguousarray(positions, np.float32)
    n = np.ascontiguousarray(normals, np.float32)
    t = np.ascontiguousarray(uvs, np.float32)
    out = np.zeros((p.shape[0], 4), np.float32)
    ok = lib.ref_mikktspace(p.ctypes.data_as(_fp), n.ctypes.data_as(_fp), t.ctypes.data_as(_fp), p.shape[0] // 3, out.ctypes.data_as(_fp))
    return out, bool(ok)


def load_mesh(path):
    """The reference's parser for the file's extension -> dict(positions [V, 3], normals [V, 3], uvs [V, 2], indices [I])."""
    lib = C.CDLL(MESH_PATH)
    for name in ("ref_mesh_open_obj", "ref_mesh_open_ply"):
        getattr(lib, name).argtypes = [C.c_char_p]
        getattr(lib, name).restype = C.c_void_p
    lib.ref_mesh_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_uint64]
    lib.ref_mesh_copy.argtypes = [C.c_void_p, _fp, _fp, _fp, C.POINTER(C.c_uint32)]
    lib.ref_mesh_close.argtypes = [C.c_void_p]
    opener = lib.ref_mesh_open_obj if path.lower().endswith(".obj") else lib.ref_mesh_open_ply
    handle = opener(path.encode())
    try:
        counts = (C.c_uint64 * 2)()
        err = C.create_string_buffer(512)
        if lib.ref_mesh_counts(handle, counts, err, len(err)) != 0:
            raise RuntimeError(err.value.decode("utf-8", "replace"))
        v, i = int(counts[0]), int(counts[1])
        out = {"positions": np.zeros((v, 3), np.float32), "normals": np.zeros((v, 3), np.float32), "uvs": np.zeros((v, 2), np.float32),
               "indices": np.zeros(i, np.uint32)}
        lib.ref_mesh_copy(handle, out["positions"].ctypes.data_as(_fp), out["normals"].ctypes.data_as(_fp), out["uvs"].ctypes.data_as(_fp),
                          out["indices"].ctypes.data_as(C.POINTER(C.c_uint32)))
        return out
    finally:
        lib.ref_mesh_close(handle)
