"""ctypes access to oracle/liboracle.so — TEST INFRASTRUCTURE ONLY (see oracle/README.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import importlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
pt = importlib.import_module("metal-pathtracer-arm64_amd")

_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(ORACLE_LIB)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint32)
        l.oracle_scene_create.argtypes = [C.POINTER(pt.PtrSceneDesc)]
        l.oracle_scene_create.restype = vp
        l.oracle_scene_destroy.argtypes = [vp]
        l.oracle_scene_info.argtypes = [vp, C.POINTER(u64)]
        l.oracle_render.argtypes = [vp, C.POINTER(pt.PtrSettings), u32, u32, u32, u32, fp, C.POINTER(u64)]
        l.oracle_render.restype = C.c_double
        l.oracle_render_signatures.argtypes = [vp, C.POINTER(pt.PtrSettings), u32, u32, u32, u32, fp, up, C.POINTER(C.c_uint8)]
        l.oracle_render_signatures.restype = C.c_double
        l.oracle_texture_sample.argtypes = [C.POINTER(pt.PtrSceneDesc), u32, fp, u64, fp]
        l.oracle_trace_rays.argtypes = [vp, fp, u64, C.c_int, C.c_int, vp]
        l.oracle_surface_hits.argtypes = [vp, fp, u64, fp]
        l.oracle_rng_hash.argtypes = [u32]
        l.oracle_rng_hash.restype = u32
        l.oracle_rng_floats.argtypes = [u32, u32, fp, up]
        l.oracle_build_camera.argtypes = [C.POINTER(pt.PtrSettings), fp]
        l.oracle_camera_rays.argtypes = [C.POINTER(pt.PtrSettings), up, u64, fp, up]
        l.oracle_eval_bsdf.argtypes = [C.POINTER(pt.PtrMaterial), C.POINTER(pt.PtrSettings), fp, u64, fp]
        l.oracle_sample_bsdf.argtypes = [C.POINTER(pt.PtrMaterial), C.POINTER(pt.PtrSettings), fp, up, up, u64, fp, up]
        l.oracle_env_build.argtypes = [fp, u32, u32, fp, up, fp, up, fp, fp]
        l.oracle_env_build.restype = C.c_int
        l.oracle_env_sample.argtypes = [fp, u32, u32, C.c_float, C.c_float, fp, u64, fp, fp]
        l.oracle_env_sample.restype = C.c_int
        _lib = l
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class OracleScene:
    def __init__(self, host_scene):
        self._host = host_scene  # keeps the arrays alive
        self._h = lib().oracle_scene_create(C.byref(host_scene.desc))

    def info(self):
        out = (C.c_uint64 * 4)()
        lib().oracle_scene_info(self._h, out)
        return {"prims": int(out[0]), "nodes": int(out[1]), "geoms": int(out[2])}

    def render(self, settings, spp, threads=0, rows=None, count=False):
        h, w = settings.height, settings.width
        y0, y1 = rows if rows else (0, h)
        img = np.zeros((h, w, 3), dtype=np.float32)
        counters = (C.c_uint64 * 8)()
        secs = lib().oracle_render(self._h, C.byref(settings), spp, threads, y0, y1, _f(img), counters if count else None)
        keys = ("extendRays", "shadowRays", "nodes", "prims", "shadedHits", "triangleHits")
        return img, secs, dict(zip(keys, [int(c) for c in counters]))

    def render_signatures(self, settings, threads=0):
        """1 spp image plus, per pixel, the path signature (as the HIP counting build computes it) and whether one of the
        path's rectangle-light shadow tests flips within +-2e-6 (relative) of the shadow ray's length."""
        h, w = settings.height, settings.width
        img = np.zeros((h, w, 3), dtype=np.float32)
        sig = np.zeros((h, w), dtype=np.uint32)
        marginal = np.zeros((h, w), dtype=np.uint8)
        lib().oracle_render_signatures(self._h, C.byref(settings), 1, threads, 0, h, _f(img), _u(sig), marginal.ctypes.data_as(C.POINTER(C.c_uint8)))
        return img, sig, marginal.astype(bool)

    def trace_rays(self, rays, any_hit=False, brute_force=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=pt.HIT_DTYPE)
        lib().oracle_trace_rays(self._h, _f(rays), rays.shape[0], int(any_hit), int(brute_force), out.ctypes.data_as(C.c_void_p))
        return out

    def surface_hits(self, rays):
        """IntersectScene + OffsetRayOrigin: rays [n, 9] {origin, direction, next direction} -> [n, 16]
        {hit, t, position, normal, shading normal, front face, next origin, 0}."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 9)
        out = np.zeros((rays.shape[0], 16), dtype=np.float32)
        lib().oracle_surface_hits(self._h, _f(rays), rays.shape[0], _f(out))
        return out

    def close(self):
        if self._h:
            lib().oracle_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def texture_sample(host_scene, texture, uv_lod):
    """The oracle's restatement of the texture filtering rule: uv_lod [n, 3] -> [n, 4] RGBA."""
    uv_lod = np.ascontiguousarray(uv_lod, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((uv_lod.shape[0], 4), dtype=np.float32)
    lib().oracle_texture_sample(C.byref(host_scene.desc), texture, _f(uv_lod), uv_lod.shape[0], _f(out))
    return out


def rng_hash(x):
    return int(lib().oracle_rng_hash(x & 0xFFFFFFFF))


def rng_floats(state, n):
    out = np.zeros(n, dtype=np.float32)
    st = C.c_uint32()
    lib().oracle_rng_floats(state, n, _f(out), C.byref(st))
    return out, int(st.value)


def build_camera(settings):
    out = np.zeros(19, dtype=np.float32)
    lib().oracle_build_camera(C.byref(settings), _f(out))
    return out


def camera_rays(settings, xys):
    xys = np.ascontiguousarray(xys, dtype=np.uint32).reshape(-1, 3)
    out = np.zeros((xys.shape[0], 6), dtype=np.float32)
    states = np.zeros(xys.shape[0], dtype=np.uint32)
    lib().oracle_camera_rays(C.byref(settings), _u(xys), xys.shape[0], _f(out), _u(states))
    return out, states


def eval_bsdf(material, settings, inputs):
    inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 12)
    out = np.zeros((inputs.shape[0], 5), dtype=np.float32)
    lib().oracle_eval_bsdf(C.byref(material), C.byref(settings), _f(inputs), inputs.shape[0], _f(out))
    return out


def sample_bsdf(material, settings, inputs, front, states):
    inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 9)
    front = np.ascontiguousarray(front, dtype=np.uint32)
    states = np.ascontiguousarray(states, dtype=np.uint32)
    out = np.zeros((inputs.shape[0], 8), dtype=np.float32)
    out_states = np.zeros(inputs.shape[0], dtype=np.uint32)
    lib().oracle_sample_bsdf(C.byref(material), C.byref(settings), _f(inputs), _u(front), _u(states), inputs.shape[0], _f(out), _u(out_states))
    return out, out_states


def env_build(rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w = rgba.shape[0], rgba.shape[1]
    pdf = np.zeros((h, w), dtype=np.float32)
    ca = np.zeros((h, w), dtype=np.uint32)
    ct = np.zeros((h, w), dtype=np.float32)
    ma = np.zeros(h, dtype=np.uint32)
    mt = np.zeros(h, dtype=np.float32)
    total = C.c_float()
    rc = lib().oracle_env_build(_f(rgba), w, h, _f(pdf), _u(ca), _f(ct), _u(ma), _f(mt), C.byref(total))
    return rc, dict(pdf=pdf, cond_alias=ca, cond_threshold=ct, marg_alias=ma, marg_threshold=mt, total=total.value)


def env_sample(rgba, rotation, intensity, u):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((u.shape[0], 7), dtype=np.float32)
    look = np.zeros((u.shape[0], 4), dtype=np.float32)
    rc = lib().oracle_env_sample(_f(rgba), rgba.shape[1], rgba.shape[0], rotation, intensity, _f(u), u.shape[0], _f(out), _f(look))
    return rc, out, look
