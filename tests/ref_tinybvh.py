"""ctypes access to oracle/_ref/libref_tinybvh.so — TEST INFRASTRUCTURE ONLY.

The library is the reference's own BVH dependency (tinybvh 1.6.7, header-only, compiled from /root/reference by
oracle/Makefile) behind the small driver oracle/ref_tinybvh.cpp.  Only tests/ may import this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "_ref", "libref_tinybvh.so")


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(LIB_PATH)
        fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        l.ref_bvh_build.argtypes = [fp, C.c_uint32]
        l.ref_bvh_build.restype = C.c_void_p
        l.ref_bvh_free.argtypes = [C.c_void_p]
        l.ref_bvh_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        l.ref_bvh_intersect.argtypes = [C.c_void_p, fp, C.c_uint64, fp, up, fp]
        l.ref_bvh_occluded.argtypes = [C.c_void_p, fp, C.c_uint64, C.POINTER(C.c_uint8)]
        _lib = l
    return _lib


def mesh_world_triangles(desc, mesh_index: int) -> np.ndarray:
    """[T, 3, 3] float32 world-space vertices of one mesh of a PtrSceneDesc (column-major localToWorld, as the bake applies it)."""
    m = desc.meshes[mesh_index]
    pos = np.ctypeslib.as_array(m.positions, shape=(m.vertexCount, 3)).astype(np.float32)
    idx = np.ctypeslib.as_array(m.indices, shape=(m.indexCount // 3, 3))
    M = np.array(list(m.localToWorld), dtype=np.float32).reshape(4, 4)
    world = ((pos[:, 0:1] * M[0, :3] + pos[:, 1:2] * M[1, :3]) + pos[:, 2:3] * M[2, :3]) + M[3, :3]
    return np.ascontiguousarray(world.astype(np.float32)[idx])


class RefBvh:
    def __init__(self, triangles: np.ndarray):
        tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
        self.count = tri.shape[0]
        self._h = lib().ref_bvh_build(tri.ctypes.data_as(C.POINTER(C.c_float)), self.count)

    def info(self) -> dict:
        out = (C.c_uint64 * 4)()
        lib().ref_bvh_info(self._h, out)
        return {"nodes": int(out[0]), "leaves": int(out[1]), "sah_cost": out[2] / 1000.0, "version": int(out[3])}

    def intersect(self, rays: np.ndarray):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        uv = np.zeros((n, 2), np.float32)
        lib().ref_bvh_intersect(self._h, rays.ctypes.data_as(C.POINTER(C.c_float)), n, t.ctypes.data_as(C.POINTER(C.c_float)),
                                prim.ctypes.data_as(C.POINTER(C.c_uint32)), uv.ctypes.data_as(C.POINTER(C.c_float)))
        return t, prim, uv

    def occluded(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], np.uint8)
        lib().ref_bvh_occluded(self._h, rays.ctypes.data_as(C.POINTER(C.c_float)), rays.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out.astype(bool)

    def close(self):
        if self._h:
            lib().ref_bvh_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
