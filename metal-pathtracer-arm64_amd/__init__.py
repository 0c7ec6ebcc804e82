"""MI355X wavefront path tracer — Python host bindings over the C-ABI (include/ptr_abi.h).

The package is only plumbing: it loads ``libptr_hip.so`` (hand-written HIP kernels + C++ host layer) with
ctypes and mirrors the POD structs of the ABI.  There is no CPU fallback: every render entry point needs the
HIP library and a GPU, and raises if either is missing.

Reference surface mirrored here (paths relative to the reference checkout):
  HostScene.load      <- SceneManager::loadSceneFromPath      src/renderer/SceneManager.mm:677-722
  DeviceScene         <- SceneResources::rebuildAccelerationStructures  src/renderer/SceneResources.mm:2055-2259
  DeviceScene.render  <- IHeadlessRenderer::render            include/headless/IHeadlessRenderer.h:42-52
  write_image         <- WriteImage                           src/renderer/ImageWriter.mm:609-627
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libptr_hip.so")
CLI_PATH = os.path.join(_PKG_DIR, "PathTracerHeadless")

# ----------------------------------------------------------------------------- ABI structs


class PtrSphere(C.Structure):
    _fields_ = [("centerRadius", C.c_float * 4), ("materialIndex", C.c_uint32 * 4)]


class PtrRect(C.Structure):
    _fields_ = [
        ("corner", C.c_float * 4),
        ("edgeU", C.c_float * 4),
        ("edgeV", C.c_float * 4),
        ("normalAndPlane", C.c_float * 4),
        ("materialTwoSided", C.c_uint32 * 4),
    ]


class PtrMaterial(C.Structure):
    _fields_ = [
        ("baseColorRoughness", C.c_float * 4),
        ("typeEta", C.c_float * 4),
        ("emission", C.c_float * 4),
        ("conductorEta", C.c_float * 4),
        ("conductorK", C.c_float * 4),
        ("coatParams", C.c_float * 4),
        ("coatTint", C.c_float * 4),
        ("coatAbsorption", C.c_float * 4),
        ("dielectricSigmaA", C.c_float * 4),
        ("sssSigmaA", C.c_float * 4),
        ("sssSigmaS", C.c_float * 4),
        ("sssParams", C.c_float * 4),
        ("carpaintBaseParams", C.c_float * 4),
        ("carpaintFlakeParams", C.c_float * 4),
        ("carpaintBaseEta", C.c_float * 4),
        ("carpaintBaseK", C.c_float * 4),
        ("carpaintBaseTint", C.c_float * 4),
        ("textureIndices0", C.c_uint32 * 4),
        ("textureIndices1", C.c_uint32 * 4),
        ("materialFlags", C.c_uint32),
        ("materialPad", C.c_uint32 * 3),
        ("pbrParams", C.c_float * 4),
        ("pbrExtras", C.c_float * 4),
        ("textureUvSet0", C.c_uint32 * 4),
        ("textureUvSet1", C.c_uint32 * 4),
        ("textureTransform", (C.c_float * 4) * 12),
    ]


class PtrMeshDesc(C.Structure):
    _fields_ = [
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("indices", C.POINTER(C.c_uint32)),
        ("vertexCount", C.c_uint32),
        ("indexCount", C.c_uint32),
        ("localToWorld", C.c_float * 16),
        ("materialIndex", C.c_uint32),
        ("pad", C.c_uint32),
        ("uv0", C.POINTER(C.c_float)),
        ("uv1", C.POINTER(C.c_float)),
        ("tangents", C.POINTER(C.c_float)),
    ]


class PtrTexture(C.Structure):
    _fields_ = [
        ("rgba", C.POINTER(C.c_float)),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("wrapS", C.c_uint32),
        ("wrapT", C.c_uint32),
        ("filter", C.c_uint32),
        ("pad", C.c_uint32),
    ]


class PtrSceneDesc(C.Structure):
    _fields_ = [
        ("spheres", C.POINTER(PtrSphere)),
        ("rects", C.POINTER(PtrRect)),
        ("materials", C.POINTER(PtrMaterial)),
        ("meshes", C.POINTER(PtrMeshDesc)),
        ("envRgba", C.POINTER(C.c_float)),
        ("sphereCount", C.c_uint32),
        ("rectCount", C.c_uint32),
        ("materialCount", C.c_uint32),
        ("meshCount", C.c_uint32),
        ("envWidth", C.c_uint32),
        ("envHeight", C.c_uint32),
        ("textures", C.POINTER(PtrTexture)),
        ("textureCount", C.c_uint32),
        ("pad", C.c_uint32),
    ]


class PtrSettings(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("maxDepth", C.c_uint32),
        ("seed", C.c_uint32),
        ("enableRussianRoulette", C.c_uint32),
        ("enableSpecularNee", C.c_uint32),
        ("enableMnee", C.c_uint32),
        ("enableMneeSecondary", C.c_uint32),
        ("cameraTarget", C.c_float * 3),
        ("cameraDistance", C.c_float),
        ("cameraYaw", C.c_float),
        ("cameraPitch", C.c_float),
        ("cameraVerticalFov", C.c_float),
        ("cameraDefocusAngle", C.c_float),
        ("cameraFocusDistance", C.c_float),
        ("backgroundMode", C.c_uint32),
        ("backgroundColor", C.c_float * 3),
        ("environmentRotation", C.c_float),
        ("environmentIntensity", C.c_float),
        ("fireflyClampEnabled", C.c_uint32),
        ("fireflyClampFactor", C.c_float),
        ("fireflyClampFloor", C.c_float),
        ("throughputClamp", C.c_float),
        ("specularTailClampBase", C.c_float),
        ("specularTailClampRoughnessScale", C.c_float),
        ("minSpecularPdf", C.c_float),
        ("fireflyClampMaxContribution", C.c_float),
        ("emissionScale", C.c_float),
        ("metalSemantics", C.c_uint32),
        ("sssMode", C.c_uint32),
        ("sssMaxSteps", C.c_uint32),
        ("debugShadowSlack", C.c_float),
    ]

    def copy(self) -> "PtrSettings":
        out = PtrSettings()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(PtrSettings))
        return out


class PtrRenderStats(C.Structure):
    _fields_ = [
        ("totalSeconds", C.c_double),
        ("avgMsPerSample", C.c_double),
        ("uploadSeconds", C.c_double),
        ("traceKernelMs", C.c_double),
        ("shadeKernelMs", C.c_double),
        ("shadowKernelMs", C.c_double),
        ("traceLaunches", C.c_uint64),
        ("samples", C.c_uint64),
        ("primaryRays", C.c_uint64),
        ("extendRays", C.c_uint64),
        ("shadowRays", C.c_uint64),
        ("nodesVisited", C.c_uint64),
        ("leafPrimTests", C.c_uint64),
        ("extendNodesVisited", C.c_uint64),
        ("extendLeafPrimTests", C.c_uint64),
        ("shadedHits", C.c_uint64),
        ("triangleHits", C.c_uint64),
        ("shadowEarlyExits", C.c_uint64),
        ("tailKernelMs", C.c_double),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class PtrHit(C.Structure):
    _fields_ = [
        ("t", C.c_float),
        ("u", C.c_float),
        ("v", C.c_float),
        ("primType", C.c_uint32),
        ("geomIndex", C.c_uint32),
        ("primIndex", C.c_uint32),
        ("ng", C.c_float * 3),
        ("pad", C.c_uint32),
    ]


HIT_DTYPE = np.dtype(
    [("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("primType", "<u4"), ("geomIndex", "<u4"), ("primIndex", "<u4"),
     ("ng", "<f4", (3,)), ("pad", "<u4")]
)

assert C.sizeof(PtrSphere) == 32 and C.sizeof(PtrRect) == 80 and C.sizeof(PtrMaterial) == 576
assert C.sizeof(PtrHit) == HIT_DTYPE.itemsize == 40

# Every symbol include/ptr_abi.h declares.
ABI_SYMBOLS = (
    "ptr_device_count", "ptr_scene_upload", "ptr_scene_release", "ptr_scene_info", "ptr_render",
    "ptr_render_bands_device", "ptr_part_band_count", "ptr_render_bands", "ptr_trace_rays", "ptr_render_aovs",
    "ptr_host_scene_load", "ptr_host_scene_free", "ptr_host_scene_desc", "ptr_host_write_image", "ptr_host_write_exr_multilayer",
    "ptr_host_read_pfm", "ptr_version", "ptr_render_multi", "ptr_host_write_exr_aovs", "ptr_host_decode_image",
    "ptr_scene_timings", "ptr_scene_prepare_geometry", "ptr_scene_upload_prepared",
)
# include/ptr_debug.h (test-only device-function probes)
DEBUG_SYMBOLS = ("ptr_debug_eval_bsdf", "ptr_debug_sample_bsdf", "ptr_debug_camera_rays", "ptr_debug_env_distribution",
                 "ptr_debug_scene_geometry", "ptr_debug_render_signatures", "ptr_debug_render_multi_on", "ptr_debug_texture_sample",
                 "ptr_debug_generate_tangents", "ptr_debug_surface_hits", "ptr_debug_shade_kernel_set", "ptr_debug_exact_division", "ptr_debug_walk_counts")

_lib: Optional[C.CDLL] = None


class PtrError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load libptr_hip.so (built in-tree by __graft_entry__.build()).  No fallback: raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PTR_HIP_LIBRARY", LIB_PATH)   # override: A/B builds of the same library (tools/)
    if not os.path.exists(path):
        raise PtrError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    lib = C.CDLL(path)
    vp, cp, u32, u64, sz = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_size_t
    lib.ptr_device_count.restype = C.c_int
    lib.ptr_scene_upload.argtypes = [C.POINTER(PtrSceneDesc), C.c_int, C.POINTER(vp), cp, sz]
    lib.ptr_scene_release.argtypes = [vp]
    lib.ptr_scene_timings.argtypes = [vp, C.POINTER(C.c_double)]
    lib.ptr_scene_prepare_geometry.argtypes = [C.POINTER(PtrSceneDesc), cp, C.POINTER(C.c_double), cp, sz]
    lib.ptr_scene_upload_prepared.argtypes = [C.POINTER(PtrSceneDesc), cp, C.c_int, C.POINTER(vp), cp, sz]
    lib.ptr_scene_release.restype = None
    lib.ptr_scene_info.argtypes = [vp, C.POINTER(u64)]
    lib.ptr_render.argtypes = [C.POINTER(PtrSceneDesc), C.POINTER(PtrSettings), u32, C.c_int, C.POINTER(C.c_float),
                               C.POINTER(PtrRenderStats), cp, sz]
    lib.ptr_render_bands_device.argtypes = [vp, C.POINTER(PtrSettings), u32, u32, u32, vp, vp, C.c_int,
                                            C.POINTER(PtrRenderStats), cp, sz]
    lib.ptr_part_band_count.argtypes = [u32, u32, u32]
    lib.ptr_part_band_count.restype = u32
    lib.ptr_render_bands.argtypes = [vp, C.POINTER(PtrSettings), u32, u32, u32, C.POINTER(C.c_float), C.c_int,
                                     C.POINTER(PtrRenderStats), cp, sz]
    lib.ptr_render_aovs.argtypes = [vp, C.POINTER(PtrSettings), u32, C.POINTER(C.c_float), C.POINTER(C.c_float), cp, sz]
    lib.ptr_trace_rays.argtypes = [vp, C.POINTER(C.c_float), u64, C.c_int, vp, C.POINTER(PtrRenderStats), cp, sz]
    lib.ptr_host_scene_load.argtypes = [cp, cp, C.POINTER(vp), cp, sz]
    lib.ptr_host_scene_free.argtypes = [vp]
    lib.ptr_host_scene_free.restype = None
    lib.ptr_host_scene_desc.argtypes = [vp, C.POINTER(PtrSceneDesc), C.POINTER(PtrSettings)]
    lib.ptr_host_write_image.argtypes = [cp, cp, C.POINTER(C.c_float), u32, u32, C.c_int, u32, u32, C.c_float,
                                         C.c_float, cp, sz]
    lib.ptr_host_write_exr_multilayer.argtypes = [cp, C.POINTER(C.c_float), u32, u32, C.POINTER(C.c_float), cp, cp, sz]
    lib.ptr_host_read_pfm.argtypes = [cp, C.POINTER(C.c_float), u32, C.POINTER(u32), C.POINTER(u32)]
    lib.ptr_version.restype = cp
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.ptr_debug_eval_bsdf.argtypes = [C.POINTER(PtrMaterial), C.POINTER(PtrSettings), fp, u64, fp, cp, sz]
    lib.ptr_debug_sample_bsdf.argtypes = [C.POINTER(PtrMaterial), C.POINTER(PtrSettings), fp, up, up, u64, fp, up, cp, sz]
    lib.ptr_debug_camera_rays.argtypes = [C.POINTER(PtrSettings), up, u64, fp, up, cp, sz]
    lib.ptr_debug_env_distribution.argtypes = [fp, u32, u32, fp, up, fp, up, fp, fp]
    lib.ptr_debug_scene_geometry.argtypes = [C.POINTER(PtrSceneDesc), u32, C.POINTER(u64), cp, sz]
    lib.ptr_debug_render_signatures.argtypes = [vp, C.POINTER(PtrSettings), fp, up, cp, sz]
    lib.ptr_debug_texture_sample.argtypes = [vp, u32, fp, u64, fp, cp, sz]
    lib.ptr_render_multi.argtypes = [C.POINTER(PtrSceneDesc), C.POINTER(PtrSettings), u32, C.c_int, C.c_int, fp, C.POINTER(PtrRenderStats), cp, sz]
    lib.ptr_debug_render_multi_on.argtypes = [C.POINTER(PtrSceneDesc), C.POINTER(PtrSettings), u32, C.POINTER(C.c_int), C.c_int, fp,
                                              C.POINTER(PtrRenderStats), cp, sz]
    lib.ptr_host_write_exr_aovs.argtypes = [cp, fp, fp, fp, u32, u32, cp, sz]
    lib.ptr_host_decode_image.argtypes = [cp, u64, C.POINTER(C.c_uint8), u64, up, up, cp, sz]
    _lib = lib
    return lib


def _err_buf():
    return C.create_string_buffer(1024)


def _check(rc: int, err) -> None:
    if rc != 0:
        raise PtrError(err.value.decode("utf-8", "replace") or f"libptr_hip call failed ({rc})")


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def device_count() -> int:
    return int(load_library().ptr_device_count())


BAND_ROWS = 8   # PTR_BAND_ROWS of include/ptr_abi.h


def band_count(height: int, part: int = 0, parts: int = 1) -> int:
    return int(load_library().ptr_part_band_count(height, part, parts))


# ----------------------------------------------------------------------------- host scene layer


class HostScene:
    """A parsed `.scene` file: SceneResources arrays + RenderSettings (host memory, no GPU needed)."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self.desc = PtrSceneDesc()
        self.settings = PtrSettings()
        load_library().ptr_host_scene_desc(self._h, C.byref(self.desc), C.byref(self.settings))

    @classmethod
    def load(cls, scene_path: str, asset_dir: Optional[str] = None) -> "HostScene":
        lib = load_library()
        h = C.c_void_p()
        err = _err_buf()
        rc = lib.ptr_host_scene_load(os.fsencode(scene_path), os.fsencode(asset_dir) if asset_dir else None,
                                     C.byref(h), err, len(err))
        _check(rc, err)
        return cls(h)

    def settings_for(self, width: Optional[int] = None, height: Optional[int] = None, max_depth: Optional[int] = None,
                     seed: Optional[int] = None, **overrides) -> PtrSettings:
        """CLI-style overrides applied after the scene file (main_headless.mm:418-449); size defaults 1280x720."""
        s = self.settings.copy()
        if width:
            s.width = width
        if height:
            s.height = height
        if max_depth is not None:
            s.maxDepth = max_depth
        if seed is not None:
            s.seed = seed
        for k, v in overrides.items():
            setattr(s, k, v)
        if s.width == 0:
            s.width = 1280
        if s.height == 0:
            s.height = 720
        return s

    def close(self) -> None:
        if self._h:
            load_library().ptr_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------- device path


def prepare_geometry(desc: PtrSceneDesc, path: str) -> float:
    """Host-only: bake the scene, build the BVH and write the device-independent arrays to `path` (a file under /dev/shm) for
    DeviceScene(..., prepared=path) in this or another process.  No GPU call.  Returns the seconds it took."""
    lib = load_library()
    err = _err_buf()
    seconds = C.c_double(0.0)
    _check(lib.ptr_scene_prepare_geometry(C.byref(desc), path.encode(), C.byref(seconds), err, len(err)), err)
    return float(seconds.value)


class DeviceScene:
    """A scene resident in HBM: SAH BVH + SoA primitive/material/light arrays (ptr_scene_upload)."""

    def __init__(self, desc: PtrSceneDesc, device: int = 0, keepalive=None, prepared: Optional[str] = None):
        """prepared: a geometry cache written by prepare_geometry() for this description - the BVH is read, not built
        (the processes of a multi-GPU render build it once)."""
        lib = load_library()
        if lib.ptr_device_count() <= 0:
            raise PtrError("no HIP device visible: the HIP render path has no CPU fallback")
        self._keepalive = keepalive
        self._h = C.c_void_p()
        err = _err_buf()
        if prepared:
            _check(lib.ptr_scene_upload_prepared(C.byref(desc), prepared.encode(), device, C.byref(self._h), err, len(err)), err)
        else:
            _check(lib.ptr_scene_upload(C.byref(desc), device, C.byref(self._h), err, len(err)), err)

    def timings(self) -> dict:
        """Seconds of the upload: geometry preparation (or cache read), shading tables, copies to the device."""
        out = (C.c_double * 4)()
        load_library().ptr_scene_timings(self._h, out)
        return {"geometry_s": out[0], "shading_tables_s": out[1], "copy_s": out[2], "geometry_from_cache": bool(out[3])}

    def info(self) -> dict:
        out = (C.c_uint64 * 8)()
        load_library().ptr_scene_info(self._h, out)
        keys = ("nodes", "leaves", "triangles", "spheres", "max_depth", "max_leaf", "sah_cost_x1000", "rect_lights")
        return dict(zip(keys, [int(v) for v in out]))

    def shade_kernel_set(self, settings: PtrSettings, count: bool = False) -> int:
        """ptr_debug_shade_kernel_set: the material / feature mask of the k_shade instantiation a render launches (0x3FF = full)."""
        lib = load_library()
        lib.ptr_debug_shade_kernel_set.argtypes = [C.c_void_p, C.POINTER(PtrSettings), C.c_int, C.POINTER(C.c_uint32)]
        out = C.c_uint32(0)
        if lib.ptr_debug_shade_kernel_set(self._h, C.byref(settings), int(count), C.byref(out)) != 0:
            raise PtrError("ptr_debug_shade_kernel_set failed")
        return int(out.value)

    def surface_hits(self, rays) -> np.ndarray:
        """ptr_debug_surface_hits: rays [n, 9] {origin, direction, next direction} -> [n, 16]."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 9)
        out = np.zeros((rays.shape[0], 16), dtype=np.float32)
        lib = load_library()
        lib.ptr_debug_surface_hits.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_float), C.c_char_p, C.c_size_t]
        err = _err_buf()
        _check(lib.ptr_debug_surface_hits(self._h, _fptr(rays), rays.shape[0], _fptr(out), err, len(err)), err)
        return out

    def render(self, settings: PtrSettings, spp: int, part: int = 0, parts: int = 1, count: bool = False
               ) -> Tuple[np.ndarray, PtrRenderStats]:
        """Render one partition to host memory.  Returns ([bands*BAND_ROWS, W, 3] float32, stats)."""
        lib = load_library()
        bands = band_count(settings.height, part, parts)
        out = np.zeros((bands * BAND_ROWS, settings.width, 3), dtype=np.float32)
        stats = PtrRenderStats()
        err = _err_buf()
        _check(lib.ptr_render_bands(self._h, C.byref(settings), spp, part, parts, _fptr(out), int(count),
                                    C.byref(stats), err, len(err)), err)
        return out, stats

    def render_image(self, settings: PtrSettings, spp: int, count: bool = False) -> Tuple[np.ndarray, PtrRenderStats]:
        out, stats = self.render(settings, spp, 0, 1, count)
        return out[: settings.height], stats

    def render_device(self, settings: PtrSettings, spp: int, d_out_ptr: int, stream: int = 0, part: int = 0,
                      parts: int = 1, count: bool = False, want_stats: bool = True, solo: bool = False
                      ) -> Optional[PtrRenderStats]:
        """Render into a caller-owned DEVICE buffer (e.g. a torch tensor's data_ptr()) on `stream`.
        solo=True runs the pool as one group (no concurrent kernels) so per-kernel timings are clean."""
        lib = load_library()
        stats = PtrRenderStats()
        err = _err_buf()
        _check(lib.ptr_render_bands_device(self._h, C.byref(settings), spp, part, parts, C.c_void_p(d_out_ptr),
                                           C.c_void_p(stream), int(count) | (2 if solo else 0), C.byref(stats) if want_stats else None,
                                           err, len(err)), err)
        return stats if want_stats else None

    def render_aovs(self, settings: PtrSettings, sample_index: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """First-hit feature buffers: ([H, W, 4] albedo rgb | hit flag, [H, W, 4] encoded normal | distance)."""
        albedo = np.zeros((settings.height, settings.width, 4), dtype=np.float32)
        normal = np.zeros((settings.height, settings.width, 4), dtype=np.float32)
        err = _err_buf()
        _check(load_library().ptr_render_aovs(self._h, C.byref(settings), sample_index, _fptr(albedo), _fptr(normal), err, len(err)), err)
        return albedo, normal

    def render_signatures(self, settings: PtrSettings) -> Tuple[np.ndarray, np.ndarray]:
        """One sample per pixel with the path signature of every pixel (include/ptr_debug.h): ([H, W, 3] image, [H, W] uint32)."""
        img = np.zeros((settings.height, settings.width, 3), dtype=np.float32)
        sig = np.zeros((settings.height, settings.width), dtype=np.uint32)
        err = _err_buf()
        _check(load_library().ptr_debug_render_signatures(self._h, C.byref(settings), _fptr(img), sig.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                          err, len(err)), err)
        return img, sig

    def texture_sample(self, texture: int, uv_lod: np.ndarray) -> np.ndarray:
        """Filtered texture lookups on the device (include/ptr_debug.h): uv_lod [n, 3] -> [n, 4] RGBA."""
        uv_lod = np.ascontiguousarray(uv_lod, dtype=np.float32).reshape(-1, 3)
        out = np.zeros((uv_lod.shape[0], 4), dtype=np.float32)
        err = _err_buf()
        _check(load_library().ptr_debug_texture_sample(self._h, texture, _fptr(uv_lod), uv_lod.shape[0], _fptr(out), err, len(err)), err)
        return out

    def trace_rays(self, rays: np.ndarray, any_hit: bool = False) -> Tuple[np.ndarray, PtrRenderStats]:
        """rays: [n, 8] float32 {ox,oy,oz,tmin,dx,dy,dz,tmax}; returns a structured array of PtrHit."""
        lib = load_library()
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        stats = PtrRenderStats()
        err = _err_buf()
        _check(lib.ptr_trace_rays(self._h, _fptr(rays), rays.shape[0], int(any_hit), out.ctypes.data_as(C.c_void_p),
                                  C.byref(stats), err, len(err)), err)
        return out, stats

    def close(self) -> None:
        if self._h:
            load_library().ptr_scene_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_multi(desc: PtrSceneDesc, settings: PtrSettings, spp: int, n_devices: int = 0, device_ids=None, verbose: bool = False
                 ) -> Tuple[np.ndarray, PtrRenderStats]:
    """The frame spread over several devices of this node (ptr_render_multi): [H, W, 3] float32 + stats.  `device_ids` (tests) names the
    devices explicitly; an id may repeat, so a one-GPU box can run the whole multi-device path."""
    lib = load_library()
    img = np.zeros((settings.height, settings.width, 3), dtype=np.float32)
    stats = PtrRenderStats()
    err = _err_buf()
    if device_ids is not None:
        ids = (C.c_int * len(device_ids))(*device_ids)
        _check(lib.ptr_debug_render_multi_on(C.byref(desc), C.byref(settings), spp, ids, len(device_ids), _fptr(img), C.byref(stats), err, len(err)), err)
    else:
        _check(lib.ptr_render_multi(C.byref(desc), C.byref(settings), spp, n_devices, int(verbose), _fptr(img), C.byref(stats), err, len(err)), err)
    return img, stats


def decode_image(data: bytes) -> np.ndarray:
    """PNG / baseline JPEG bytes -> [H, W, 4] uint8 through the library's own decoders (ptr_host_decode_image)."""
    lib = load_library()
    w, h = C.c_uint32(), C.c_uint32()
    err = _err_buf()
    _check(lib.ptr_host_decode_image(data, len(data), None, 0, C.byref(w), C.byref(h), err, len(err)), err)
    out = np.zeros((h.value, w.value, 4), dtype=np.uint8)
    _check(lib.ptr_host_decode_image(data, len(data), out.ctypes.data_as(C.POINTER(C.c_uint8)), out.nbytes, C.byref(w), C.byref(h), err, len(err)), err)
    return out


def write_exr_aovs(path: str, rgb: np.ndarray, albedo: np.ndarray, normal: np.ndarray) -> None:
    """Beauty + first-hit albedo / normal / depth layers in one EXR (ptr_host_write_exr_aovs)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    albedo = np.ascontiguousarray(albedo, dtype=np.float32)
    normal = np.ascontiguousarray(normal, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    err = _err_buf()
    _check(load_library().ptr_host_write_exr_aovs(os.fsencode(path), _fptr(rgb), _fptr(albedo), _fptr(normal), w, h, err, len(err)), err)


def assemble_bands(parts_out, width: int, height: int) -> np.ndarray:
    """Interleave per-partition band buffers ([bands*BAND_ROWS, W, 3] each, band b of part p = image band p + b*P)."""
    parts = len(parts_out)
    R = BAND_ROWS
    img = np.zeros((((height + R - 1) // R) * R, width, 3), dtype=np.float32)
    for p, buf in enumerate(parts_out):
        nb = buf.shape[0] // R
        for b in range(nb):
            g = p + b * parts
            img[g * R:(g + 1) * R] = buf[b * R:(b + 1) * R]
    return img[:height]


def write_image(path: str, rgb: np.ndarray, fmt: str = "pfm", rgba_exr: bool = False, tonemap: int = 1,
                aces_variant: int = 0, exposure: float = 0.0, reinhard_white: float = 1.5) -> None:
    lib = load_library()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    err = _err_buf()
    _check(lib.ptr_host_write_image(os.fsencode(path), fmt.encode(), _fptr(rgb), w, h, int(rgba_exr), tonemap,
                                    aces_variant, exposure, reinhard_white, err, len(err)), err)


def write_exr_multilayer(path: str, rgb: np.ndarray, sample_counts: Optional[np.ndarray] = None,
                         colorspace: str = "Linear sRGB") -> None:
    """RGBA EXR with a planar SAMPLES channel (per-pixel sample counts), or plain RGBA when sample_counts is None."""
    lib = load_library()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    counts = None
    if sample_counts is not None:
        counts = np.ascontiguousarray(sample_counts, dtype=np.float32).reshape(h, w)
    err = _err_buf()
    _check(lib.ptr_host_write_exr_multilayer(os.fsencode(path), _fptr(rgb), w, h, _fptr(counts) if counts is not None else None,
                                             colorspace.encode(), err, len(err)), err)


def read_pfm(path: str) -> np.ndarray:
    lib = load_library()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.ptr_host_read_pfm(os.fsencode(path), None, 0, C.byref(w), C.byref(h)) != 0:
        raise PtrError(f"cannot read PFM header: {path}")
    out = np.zeros((h.value, w.value, 3), dtype=np.float32)
    if lib.ptr_host_read_pfm(os.fsencode(path), _fptr(out), out.size, C.byref(w), C.byref(h)) != 0:
        raise PtrError(f"cannot read PFM data: {path}")
    return out


# ----------------------------------------------------------------------------- device-function probes (tests)


def _uptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def debug_eval_bsdf(material: PtrMaterial, settings: PtrSettings, inputs: np.ndarray) -> np.ndarray:
    inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 12)
    out = np.zeros((inputs.shape[0], 5), dtype=np.float32)
    err = _err_buf()
    _check(load_library().ptr_debug_eval_bsdf(C.byref(material), C.byref(settings), _fptr(inputs), inputs.shape[0],
                                             _fptr(out), err, len(err)), err)
    return out


def debug_sample_bsdf(material: PtrMaterial, settings: PtrSettings, inputs: np.ndarray, front: np.ndarray,
                      states: np.ndarray):
    inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 9)
    front = np.ascontiguousarray(front, dtype=np.uint32)
    states = np.ascontiguousarray(states, dtype=np.uint32)
    out = np.zeros((inputs.shape[0], 8), dtype=np.float32)
    out_states = np.zeros(inputs.shape[0], dtype=np.uint32)
    err = _err_buf()
    _check(load_library().ptr_debug_sample_bsdf(C.byref(material), C.byref(settings), _fptr(inputs), _uptr(front),
                                               _uptr(states), inputs.shape[0], _fptr(out), _uptr(out_states), err,
                                               len(err)), err)
    return out, out_states


def debug_camera_rays(settings: PtrSettings, xys: np.ndarray):
    xys = np.ascontiguousarray(xys, dtype=np.uint32).reshape(-1, 3)
    out = np.zeros((xys.shape[0], 6), dtype=np.float32)
    states = np.zeros(xys.shape[0], dtype=np.uint32)
    err = _err_buf()
    _check(load_library().ptr_debug_camera_rays(C.byref(settings), _uptr(xys), xys.shape[0], _fptr(out), _uptr(states),
                                               err, len(err)), err)
    return out, states


def debug_env_distribution(rgba: np.ndarray) -> dict:
    """Host-side environment importance tables (no GPU needed)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.float32)
    h, w = rgba.shape[0], rgba.shape[1]
    pdf = np.zeros((h, w), np.float32)
    ca = np.zeros((h, w), np.uint32)
    ct = np.zeros((h, w), np.float32)
    ma = np.zeros(h, np.uint32)
    mt = np.zeros(h, np.float32)
    total = C.c_float()
    rc = load_library().ptr_debug_env_distribution(_fptr(rgba), w, h, _fptr(pdf), _uptr(ca), _fptr(ct), _uptr(ma), _fptr(mt),
                                                   C.byref(total))
    if rc != 0:
        raise PtrError("environment map has no positive radiance")
    return dict(pdf=pdf, cond_alias=ca, cond_threshold=ct, marg_alias=ma, marg_threshold=mt, total=total.value)


GEOMETRY_FIELDS = ("nodes", "leaves", "triangles_referenced", "spheres_referenced", "max_depth", "max_leaf_size",
                   "unreferenced", "multiply_referenced", "box_violations", "quant_violations", "bad_refs", "triangles",
                   "spheres", "sah_cost_milli", "build_ms", "quantized_usable")


def debug_scene_geometry(desc: PtrSceneDesc, leaf_max: int = 0) -> dict:
    """Host-side (no GPU): build the BVH / leaf-order arrays exactly as ptr_scene_upload does and validate them."""
    out = (C.c_uint64 * 16)()
    err = _err_buf()
    _check(load_library().ptr_debug_scene_geometry(C.byref(desc), leaf_max, out, err, len(err)), err)
    g = dict(zip(GEOMETRY_FIELDS, [int(v) for v in out]))
    word = g["quantized_usable"]
    g["oversize"] = (word >> 8) & 0xFF                   # triangles kept out of the tree (tested first by every ray)
    g["wide_nodes"] = (word >> 16) & 0xFFFFFFFF          # four-wide nodes of the persistent traversal kernels
    g["wide_problems"] = word >> 63                      # bad references / primitives not reached exactly once through them
    g["quantized_usable"] = word & 0xFF
    return g
