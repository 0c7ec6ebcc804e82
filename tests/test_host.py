"""CPU tests of the host layer and of the C-ABI surface (no GPU, no compute calls on the device path)."""
import ctypes as C
import importlib
import os
import re
import struct
import subprocess

import numpy as np
import pytest

pt = importlib.import_module("metal-pathtracer-arm64_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _write(tmp_path, text, name="t.scene"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


# --------------------------------------------------------------------------- ABI surface
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ptr_abi.h")).read()
    declared = set(re.findall(r"\b(ptr_[a-z_]+)\s*\(", header))
    assert declared == set(pt.ABI_SYMBOLS)
    debug = set(re.findall(r"\b(ptr_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", "ptr_debug.h")).read()))
    assert debug == set(pt.DEBUG_SYMBOLS)
    declared |= debug
    lib = pt.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ptr_version().startswith(b"ptr-hip")


def test_struct_layouts_match_reference_contract():
    assert C.sizeof(pt.PtrSphere) == 32 and C.sizeof(pt.PtrRect) == 80 and C.sizeof(pt.PtrMaterial) == 576
    assert pt.PtrMaterial.pbrParams.offset == 19 * 16 + 16      # 17 float4 + 2 uint4 + 4 u32
    assert pt.PtrMaterial.textureTransform.offset == 576 - 12 * 16


def test_device_path_fails_loudly_without_gpu():
    if pt.device_count() > 0:
        pytest.skip("a GPU is present")
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    with pytest.raises(pt.PtrError, match="no HIP device|no such HIP device"):
        pt.DeviceScene(host.desc)
    out = np.zeros((64, 64, 3), np.float32)
    err = C.create_string_buffer(256)
    s = host.settings_for(64, 64)
    rc = pt.load_library().ptr_render(C.byref(host.desc), C.byref(s), 1, 0, out.ctypes.data_as(C.POINTER(C.c_float)), None, err, 256)
    assert rc != 0 and b"no CPU fallback" in err.value and not out.any()


def test_division_by_a_render_constant_is_exact():
    # csrc/kernels/device_types.h DivU32 (item -> sample / local pixel, pixel -> x / y in the kernels): same quotient as n // d for
    # every divisor class (1, powers of two, 2^k +- 1, image widths, 2^32 - 1) at the ends of the range and at multiples +- 1
    import ctypes as C
    lib = pt.load_library()
    lib.ptr_debug_exact_division.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(3)
    divisors = [1, 2, 3, 5, 7, 64, 641, 1920, 1080, 3840, 2073600, 8294400, 65535, 65536, 65537, 2**31 - 1, 2**31, 2**31 + 1, 2**32 - 1]
    divisors += [int(v) for v in rng.integers(1, 2**32, 40)] + [int(v) for v in rng.integers(1, 5000, 40)]
    for d in divisors:
        n = np.concatenate([np.array([0, 1, d - 1, d, min(d + 1, 2**32 - 1), 2**32 - 1, 2**32 - 2, 2**31, 2**31 - 1], dtype=np.uint64),
                            rng.integers(0, 2**32, 4000, dtype=np.uint64),
                            np.clip(rng.integers(0, max(2**32 // d, 1) + 1, 2000, dtype=np.uint64) * d + rng.integers(0, 3, 2000, dtype=np.uint64) - 1, 0, 2**32 - 1)])
        n32 = np.ascontiguousarray(n.astype(np.uint32))
        out = np.zeros(n32.shape[0], dtype=np.uint32)
        assert lib.ptr_debug_exact_division(d, n32.ctypes.data_as(C.POINTER(C.c_uint32)), n32.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint32))) == 0
        assert np.array_equal(out.astype(np.uint64), n32.astype(np.uint64) // d), d
    assert lib.ptr_debug_exact_division(0, None, 0, None) == 1


def test_walk_counts_of_collapsed_levels(tmp_path):
    # ptr_debug_walk_counts (host only): the same rays hit the same things whatever the node width; a wider node takes
    # fewer steps and never fewer box tests than the binary walk
    from scenes.gen_assets import ensure_assets
    ensure_assets()
    scenes = os.path.join(ROOT, "scenes")
    host = pt.HostScene.load(os.path.join(scenes, "cornell_mesh.scene"), scenes)
    lib = pt.load_library()
    lib.ptr_debug_walk_counts.argtypes = [C.POINTER(pt.PtrSceneDesc), C.POINTER(C.c_float), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(2)
    n = 4000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.ascontiguousarray(np.concatenate([rng.uniform(50, 500, (n, 3)), np.full((n, 1), 1e-3), d, np.full((n, 1), np.inf)], axis=1).astype(np.float32))
    res = []
    for levels in (1, 2, 3, 4, 5):   # 4: four-wide nodes whose children are chosen by box area; 5: the array BuildWideNodes ships
        out = (C.c_uint64 * 4)()
        err = C.create_string_buffer(256)
        assert lib.ptr_debug_walk_counts(C.byref(host.desc), rays.ctypes.data_as(C.POINTER(C.c_float)), n, levels, out, err, len(err)) == 0, err.value
        res.append([int(v) for v in out])
    assert res[0][3] == res[1][3] == res[2][3] == res[3][3] == res[4][3] > 0.4 * n          # the same rays hit
    assert abs(res[4][0] - res[3][0]) <= 0.01 * res[3][0]                      # the shipped array is that collapse (boxes quantised)
    assert res[3][0] < 0.95 * res[1][0] and res[3][1] <= 4 * res[3][0]         # by area: fewer four-wide steps than by level
    assert res[0][0] > res[1][0] > res[2][0]                      # fewer steps per level collapsed
    assert res[0][1] == 2 * res[0][0]                             # a binary step tests its two boxes
    assert res[2][1] > res[1][1] >= 0.95 * res[0][1]              # the third level costs box tests
    err = C.create_string_buffer(256)
    assert lib.ptr_debug_walk_counts(C.byref(host.desc), None, 0, 6, (C.c_uint64 * 4)(), err, len(err)) == 1


def test_band_partition_counts():
    src = open(os.path.join(ROOT, "include", "ptr_abi.h")).read()
    assert "#define PTR_BAND_ROWS %du" % pt.BAND_ROWS in src
    assert importlib.import_module("metal-pathtracer-arm64_amd.bands").BAND_ROWS == pt.BAND_ROWS
    for height in (8, 16, 17, 64, 1080, 2160):
        bands = (height + pt.BAND_ROWS - 1) // pt.BAND_ROWS
        for parts in (1, 2, 3, 8):
            counts = [pt.band_count(height, p, parts) for p in range(parts)]
            assert sum(counts) == bands and max(counts) - min(counts) <= 1
    assert pt.band_count(1080, 8, 8) == 0


# --------------------------------------------------------------------------- scene grammar
def test_smoke_scene_parses_like_the_reference_fixture():
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    d, s = host.desc, host.settings
    assert (d.sphereCount, d.materialCount, d.rectCount, d.meshCount) == (2, 2, 0, 0)
    assert (s.width, s.height, s.maxDepth) == (64, 64, 4)
    assert s.backgroundMode == 1 and np.allclose(list(s.backgroundColor), [0.7, 0.8, 1.0])
    assert np.allclose(list(s.cameraTarget), [0, 0, -1]) and s.cameraDistance == 3.5 and s.cameraVerticalFov == 45
    assert s.cameraFocusDistance == 3.5
    assert np.allclose(list(d.spheres[1].centerRadius), [0, -100.5, -1, 100]) and d.spheres[1].materialIndex[0] == 1
    m = d.materials[0]
    assert m.typeEta[0] == 0 and np.allclose(list(m.baseColorRoughness), [0.8, 0.3, 0.3, 0.0])
    # renderer defaults that a scene file does not touch
    assert s.enableRussianRoulette == 1 and s.enableSpecularNee == 1 and s.enableMnee == 0 and s.enableMneeSecondary == 1
    assert (s.fireflyClampEnabled, s.fireflyClampFactor, s.fireflyClampFloor, s.throughputClamp) == (1, 32.0, 4.0, 32.0)


def test_scene_defaults_size_and_gradient_background(tmp_path):
    host = pt.HostScene.load(_write(tmp_path, "material type=lambert\nsphere center=0,0,0 radius=1 material=0\n"))
    assert host.settings.width == 0 and host.settings.height == 0 and host.settings.backgroundMode == 0
    s = host.settings_for()
    assert (s.width, s.height, s.maxDepth) == (1280, 720, 50)   # CLI defaults (main_headless.mm:510-515), maxDepth 50


def test_grammar_comments_continuations_unknowns(tmp_path):
    text = (
        "# My Scene\n"
        "camera target=1,2,3 \\\n   distance=5 bogus=1 bareword\n"
        "unknowndirective a=b\n"
        "renderer maxDepth=7 envRotation=90 seed=42 russianRoulette=0 width=4 height=5\n"
        "material type=GLASS ior=1.33 sigmaA=1,2,3\n"
        "material type=light emit=1,2,3 emitEnv=1 name=L\n"
        "rect x=0,2 y=1 z=0,4 normal=-1 twoSided=1 material=1\n"
    )
    host = pt.HostScene.load(_write(tmp_path, text))
    s, d = host.settings, host.desc
    assert np.allclose(list(s.cameraTarget), [1, 2, 3]) and s.cameraDistance == 5
    assert s.maxDepth == 7 and s.seed == 42 and s.enableRussianRoulette == 0
    assert s.environmentRotation == pytest.approx(np.pi / 2, rel=1e-6)
    assert (s.width, s.height) == (8, 8)                         # renderer width/height are clamped to >= 8
    g = d.materials[0]
    assert g.typeEta[0] == 2 and g.typeEta[1] == pytest.approx(1.33) and np.allclose(list(g.dielectricSigmaA)[:3], [1, 2, 3])
    light = d.materials[1]
    assert light.typeEta[0] == 3 and light.typeEta[1] == 1.0 and light.emission[3] == 1.0
    r = d.rects[0]
    assert np.allclose(list(r.normalAndPlane), [0, -1, 0, -1]) and r.materialTwoSided[1] == 1
    assert np.allclose(list(r.corner)[:3], [0, 1, 4]) and np.allclose(list(r.edgeU)[:3], [2, 0, 0]) and np.allclose(list(r.edgeV)[:3], [0, 0, -4])
    assert r.edgeU[3] == pytest.approx(0.25) and r.edgeV[3] == pytest.approx(1 / 16)


@pytest.mark.parametrize(
    "line, message",
    [
        ("sphere center=0,0,0 radius=1 material=0", "line 1: sphere references material index that has not been defined yet"),
        ("material albedo=1,1,1", "line 1: material requires a type token"),
        ("material type=wood", "line 1: material type is not recognized"),
        ("material type=lambert\nrectangle x=0,1 y=0,1 z=0,1 material=0", "line 2: rectangle requires exactly one axis to be fixed to a single value"),
        ("material type=lambert\nrectangle x=0,1 y=0 material=0", "line 2: rectangle requires z token"),
        ("material type=lambert\nbox min=0,0,0 material=0", "line 2: box requires min, max, and material tokens"),
        ("background solid=1,1,1 env=a.hdr", "line 1: background cannot specify both solid and env"),
        ("background env=missing.hdr", "background env map not found"),
        ("material type=sss sigma_a=1,1,1", "line 1: material sigma_a and sigma_s must both be provided together"),
        ("material type=lambert\nmesh path=nope.obj material=0", "mesh file not found"),
        ("material type=lambert\nmesh type=plane", "line 2: mesh requires material token"),
        ("camera distance=abc", "line 1: camera distance expects a float"),
    ],
)
def test_parse_errors(tmp_path, line, message):
    with pytest.raises(pt.PtrError) as e:
        pt.HostScene.load(_write(tmp_path, line + "\n"))
    assert message in str(e.value) and "Failed parsing scene" in str(e.value)


def test_missing_scene_file():
    with pytest.raises(pt.PtrError, match="Failed to open scene file"):
        pt.HostScene.load("/nonexistent/x.scene")


# --------------------------------------------------------------------------- material derivation
def test_material_derived_fields():
    host = pt.HostScene.load(os.path.join(GOLDEN, "materials.scene"))
    mats = {i: host.desc.materials[i] for i in range(host.desc.materialCount)}
    plastic = mats[5]
    f0 = ((1.5 - 1) / (1.5 + 1)) ** 2
    avg = f0 + (1 - f0) / 21.0                                   # SceneResources.mm:823-832
    assert plastic.typeEta[0] == 4 and plastic.typeEta[1] == 1.5 and plastic.typeEta[2] == 1.5
    assert plastic.coatParams[3] == pytest.approx(avg, rel=1e-6)
    assert plastic.coatParams[2] == pytest.approx(max(avg * 2.5 + 0.1 * 0.5, 0.25), rel=1e-6)
    thick = mats[6]
    assert thick.typeEta[1] == pytest.approx(1.4) and thick.coatParams[1] == 0.5 and np.allclose(list(thick.coatAbsorption)[:3], [0.4, 0.1, 0.05])
    car = mats[8]
    assert car.typeEta[0] == 6 and car.coatParams[0] == pytest.approx(0.04)
    assert car.coatParams[2] == pytest.approx(max(avg * 2.5 + 0.04 * 0.5, 0.35), rel=1e-6)
    assert car.carpaintBaseParams[0] == pytest.approx(0.6) and car.carpaintBaseParams[1] == pytest.approx(0.3)
    assert car.carpaintFlakeParams[0] == pytest.approx(0.2)      # clamp(2e6 * 1e-7, 0, 0.6) * reflectance scale 1
    assert car.carpaintBaseEta[3] == 1.0 and np.allclose(list(car.carpaintBaseEta)[:3], [1.3456, 0.9652, 0.6172])
    car2 = mats[9]
    assert car2.carpaintBaseEta[3] == 0.0 and not any(list(car2.carpaintBaseEta)[:3])   # no conductor without metallic / eta / k
    assert car2.carpaintFlakeParams[0] == pytest.approx(0.5) and car2.carpaintBaseParams[2] == pytest.approx(40.0)
    assert car2.carpaintBaseParams[1] == pytest.approx(0.2)      # default base roughness
    gold = mats[3]
    assert gold.conductorEta[3] == 1.0 and gold.conductorK[3] == 1.0
    sss = mats[7]
    assert sss.typeEta[0] == 5 and sss.sssParams[0] == pytest.approx(0.5) and sss.coatParams[0] == pytest.approx(0.05)
    lam = mats[0]
    assert lam.coatParams[2] == 0.0 and lam.pbrParams[1] == lam.baseColorRoughness[3]
    assert list(lam.textureIndices0) == [0xFFFFFFFF] * 4 and list(lam.textureTransform[0]) == [1, 0, 0, 0] and list(lam.textureTransform[1]) == [0, 1, 0, 0]


def test_box_expands_to_oriented_rectangles(tmp_path):
    text = ("material type=lambert\n"
            "box min=0,0,0 max=1,2,3 material=0\n"
            "box min=0,0,0 max=1,1,1 material=0 includeBottom=0 translate=5,0,0 rotateY=90\n")
    host = pt.HostScene.load(_write(tmp_path, text))
    d = host.desc
    assert d.rectCount == 11
    normals = [tuple(np.round(list(d.rects[i].normalAndPlane)[:3], 5)) for i in range(6)]
    assert normals == [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    for i in range(d.rectCount):
        r = d.rects[i]
        n = np.array(list(r.normalAndPlane)[:3])
        eu, ev, c = np.array(list(r.edgeU)[:3]), np.array(list(r.edgeV)[:3]), np.array(list(r.corner)[:3])
        assert abs(np.dot(n, eu)) < 1e-5 and abs(np.dot(n, ev)) < 1e-5
        assert r.normalAndPlane[3] == pytest.approx(float(np.dot(n, c)), abs=1e-5)
    # rotated box: its +X face now points along -Z (rotation about Y by +90 degrees), translated by 5 in x
    n6 = np.array(list(d.rects[6].normalAndPlane)[:3])
    assert np.allclose(n6, [0, 0, -1], atol=1e-5)


def test_obj_and_ply_loaders(tmp_path):
    (tmp_path / "assets").mkdir()
    (tmp_path / "assets" / "tri.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\nf -4 -3 -2\n")
    (tmp_path / "assets" / "tri.ply").write_text(
        "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
        "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    binply = tmp_path / "assets" / "bin.ply"
    with open(binply, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty double x\nproperty double y\nproperty double z\n"
                b"property float nx\nproperty float ny\nproperty float nz\nelement face 1\nproperty list uchar uint vertex_index\nend_header\n")
        for p in ((0, 0, 0), (2, 0, 0), (0, 2, 0)):
            f.write(struct.pack("<3d3f", *p, 0, 0, 1))
        f.write(struct.pack("<B3I", 3, 0, 1, 2))
    text = ("material type=lambert name=m\n"
            "mesh path=assets/tri.obj material=m translate=0,0,1\n"
            "mesh file=assets/tri.ply material=0 scale=2\n"
            "mesh path=assets/bin.ply material=0 rotate=0,90,0\n"
            "mesh type=plane material=0 scale=2,1,4\n")
    host = pt.HostScene.load(_write(tmp_path, text))
    d = host.desc
    assert d.meshCount == 4
    obj = d.meshes[0]
    assert obj.indexCount == 9 and obj.vertexCount == 7     # quad fan (4 shared corners) + a face without normals (3 new)
    nrm = np.ctypeslib.as_array(obj.normals, shape=(obj.vertexCount * 3,)).reshape(-1, 3)
    assert np.allclose(nrm[:4], [0, 0, 1]) and np.allclose(nrm[4:], [0, 1, 0])   # quirk Q3: missing vn -> (0,1,0)
    assert obj.localToWorld[14] == 1.0
    ply = d.meshes[1]
    pn = np.ctypeslib.as_array(ply.normals, shape=(9,)).reshape(-1, 3)
    assert np.allclose(pn, [0, 0, 1]) and ply.localToWorld[0] == 2.0            # PLY without normals -> flat fallback
    rot = np.array(list(d.meshes[2].localToWorld)).reshape(4, 4).T
    assert np.allclose(rot[:3, :3] @ [1, 0, 0], [0, 0, -1], atol=1e-6)          # Ry(90): +x -> -z
    plane = d.meshes[3]
    assert plane.indexCount == 6 and plane.localToWorld[0] == 2.0 and plane.localToWorld[10] == 4.0


# --------------------------------------------------------------------------- glTF / GLB import
def _tiny_gltf(tmp_path, name="t.gltf", with_normals=False, with_camera=True, materials=True, index_type="u16"):
    """A two-triangle quad under root(scale 2) -> child(translation 1,0,0; rotation 90 deg about Y), data-URI buffer."""
    import base64
    import json
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], {"u16": np.uint16, "u8": np.uint8, "u32": np.uint32}[index_type])
    blob = pos.tobytes() + idx.tobytes()
    pad = (4 - len(blob) % 4) % 4
    blob += b"\0" * pad
    views = [{"buffer": 0, "byteOffset": 0, "byteLength": pos.nbytes}, {"buffer": 0, "byteOffset": pos.nbytes, "byteLength": idx.nbytes}]
    accessors = [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                 {"bufferView": 1, "componentType": {"u16": 5123, "u8": 5121, "u32": 5125}[index_type], "count": 6, "type": "SCALAR"}]
    attrs = {"POSITION": 0}
    if with_normals:
        nrm = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": nrm.nbytes})
        accessors.append({"bufferView": 2, "componentType": 5126, "count": 4, "type": "VEC3"})
        attrs["NORMAL"] = 2
        blob += nrm.tobytes()
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
           "nodes": [{"name": "root", "scale": [2, 2, 2], "children": [1] + ([2] if with_camera else [])},
                     {"name": "quad", "mesh": 0, "translation": [1, 0, 0], "rotation": [0, 0.70710678, 0, 0.70710678]}],
           "meshes": [{"name": "m", "primitives": [{"attributes": attrs, "indices": 1, "material": 0},
                                                   {"attributes": attrs, "indices": 1, "mode": 1}]}],
           "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
           "bufferViews": views, "accessors": accessors}
    if with_camera:
        doc["nodes"].append({"name": "cam", "camera": 0, "translation": [0, 1, 5]})
        doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": 0.5, "znear": 0.1}}]
    if materials:
        doc["materials"] = [{"name": "Gold \u00e9", "doubleSided": True, "alphaMode": "MASK", "alphaCutoff": 0.3,
                             "pbrMetallicRoughness": {"baseColorFactor": [1.0, 0.8, 0.3, 0.5], "metallicFactor": 1.5, "roughnessFactor": 0.25},
                             "emissiveFactor": [1, 0.5, 0.25],
                             "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0},
                                            "KHR_materials_ior": {"ior": 1.33},
                                            "KHR_materials_volume": {"thicknessFactor": 0.2, "attenuationColor": [0.5, 1.0, 0.25], "attenuationDistance": 2.0}}}]
    (tmp_path / "assets").mkdir(exist_ok=True)
    (tmp_path / "assets" / name).write_text(json.dumps(doc))
    return doc


def test_gltf_hierarchy_materials_generated_normals_and_camera(tmp_path):
    _tiny_gltf(tmp_path)
    host = pt.HostScene.load(_write(tmp_path, "renderer gltfEmissiveScale=0.5\nmesh path=assets/t.gltf translate=0,10,0\n"))
    d = host.desc
    assert d.meshCount == 1 and d.materialCount == 1        # the LINES primitive (mode 1) is skipped
    m = d.meshes[0]
    assert m.vertexCount == 4 and m.indexCount == 6
    # world = T(0,10,0) * S(2) * T(1,0,0) * Ry(90): local +x -> world -z (scaled by 2), origin -> (2,10,0)
    xf = np.array(list(m.localToWorld), np.float64).reshape(4, 4).T
    assert np.allclose(xf @ [0, 0, 0, 1], [2, 10, 0, 1], atol=1e-5)
    assert np.allclose(xf @ [1, 0, 0, 0], [0, 0, -2, 0], atol=1e-5)
    nrm = np.ctypeslib.as_array(m.normals, shape=(12,)).reshape(4, 3)
    assert np.allclose(nrm, [0, 0, 1], atol=1e-6)           # no NORMAL accessor: area-weighted face normals
    mat = d.materials[0]
    assert int(mat.typeEta[0]) == 7 and np.isclose(mat.typeEta[1], 1.33) and mat.typeEta[2] == 1.0 and np.isclose(mat.typeEta[3], 0.2)
    assert np.allclose(list(mat.baseColorRoughness), [1.0, 0.8, 0.3, 0.25])
    assert np.allclose(list(mat.pbrParams), [1.0, 0.25, 1.0, 1.0])                     # metallic clamped to 1
    assert np.allclose(list(mat.pbrExtras), [0.5, 0.3, 0.0, 1.0])                      # alpha, cutoff, transmission, MASK
    assert np.allclose(list(mat.emission)[:3], np.array([1, 0.5, 0.25]) * 4.0 * 0.5)   # strength * renderer gltfEmissiveScale
    assert np.allclose(list(mat.dielectricSigmaA)[:3], [np.log(2) / 2, 0.0, np.log(4) / 2], atol=1e-6)
    assert all(t == 0xFFFFFFFF for t in list(mat.textureIndices0) + list(mat.textureIndices1))
    # embedded perspective camera (no `camera` directive before the mesh): orbit parameters derived from its pose
    s = host.settings
    assert np.isclose(s.cameraVerticalFov, np.degrees(0.5), atol=1e-4) and s.cameraDefocusAngle == 0.0
    eye = np.array(list(s.cameraTarget)) + s.cameraDistance * np.array(
        [np.cos(s.cameraPitch) * np.cos(s.cameraYaw), np.sin(s.cameraPitch), np.cos(s.cameraPitch) * np.sin(s.cameraYaw)])
    assert np.allclose(eye, [0, 2, 10], atol=1e-3)          # camera node (0,1,5) under the scale-2 root; mesh transform not applied


def test_gltf_camera_directive_wins_and_variants(tmp_path):
    _tiny_gltf(tmp_path, with_normals=True, materials=False, index_type="u8")
    host = pt.HostScene.load(_write(tmp_path, "camera target=1,2,3 distance=4 vfov=33\nmesh path=assets/t.gltf\n"))
    assert np.allclose(list(host.settings.cameraTarget), [1, 2, 3]) and host.settings.cameraVerticalFov == 33.0
    d = host.desc
    assert d.materialCount == 1                              # default material: white, metallic 1, roughness 1
    assert np.allclose(list(d.materials[0].baseColorRoughness), [1, 1, 1, 1]) and d.materials[0].pbrParams[0] == 1.0
    nrm = np.ctypeslib.as_array(d.meshes[0].normals, shape=(12,)).reshape(4, 3)
    assert np.allclose(nrm, [0, 0, 1])
    _tiny_gltf(tmp_path, name="u32.gltf", index_type="u32", with_camera=False)
    host = pt.HostScene.load(_write(tmp_path, "mesh file=assets/u32.gltf\n"))
    assert host.desc.meshes[0].indexCount == 6


def test_glb_container_and_errors(tmp_path):
    from scenes.gen_assets import helmet_glb
    (tmp_path / "assets").mkdir()
    tris = helmet_glb(str(tmp_path / "assets" / "h.glb"), n=12)
    host = pt.HostScene.load(_write(tmp_path, "mesh path=assets/h.glb\n"))
    d = host.desc
    assert d.meshCount == 3 and d.materialCount == 3 and sum(d.meshes[i].indexCount // 3 for i in range(3)) == tris == 288
    assert [d.meshes[i].materialIndex for i in range(3)] == [0, 2, 1]
    assert np.allclose(list(d.materials[2].emission)[:3], [6.0, 2.7, 0.6], atol=1e-5)
    g = pt.debug_scene_geometry(d)
    assert g["triangles"] == 288 and g["unreferenced"] == 0 and g["box_violations"] == 0
    raw = (tmp_path / "assets" / "h.glb").read_bytes()
    cases = {"magic.glb": (b"XXXX" + raw[4:], "Invalid .glb magic"),
             "version.glb": (raw[:4] + struct.pack("<I", 1) + raw[8:], "Unsupported .glb version"),
             "short.glb": (raw[:8], "Invalid .glb header"),
             "chunk.glb": (raw[:12] + struct.pack("<I", 10 ** 9) + raw[16:], "Invalid .glb chunk length"),
             "nojson.glb": (raw[:12] + struct.pack("<II", 4, 0x004E4942) + b"abcd", "Missing JSON chunk"),
             "bad.gltf": (b"{\"asset\": [1, 2,,]}", "Failed to parse glTF JSON"),
             "nobuf.gltf": (b'{"buffers": [{"byteLength": 4}]}', "buffer missing uri"),
             "nopos.gltf": (b'{"scenes":[{"nodes":[0]}],"nodes":[{"mesh":0}],"meshes":[{"primitives":[{"attributes":{}}]}]}', "missing POSITION")}
    for name, (data, message) in cases.items():
        (tmp_path / "assets" / name).write_bytes(data)
        with pytest.raises(pt.PtrError, match=re.escape(message)):
            pt.HostScene.load(_write(tmp_path, "mesh path=assets/%s\n" % name))
    with pytest.raises(pt.PtrError, match="mesh file not found"):      # resolved (and rejected) before the loader runs
        pt.HostScene.load(_write(tmp_path, "mesh path=assets/none.glb\n"))


def test_config_scenes_parse():
    host = pt.HostScene.load(os.path.join(ROOT, "scenes", "helmet_env.scene"), os.path.join(ROOT, "scenes"))
    d = host.desc
    assert d.meshCount == 3 and sum(d.meshes[i].indexCount // 3 for i in range(3)) == 46208
    assert d.envWidth == 2048 and d.envHeight == 1024 and np.isclose(host.settings.environmentRotation, np.radians(30))
    assert [int(d.materials[i].typeEta[0]) for i in range(d.materialCount)] == [0, 7, 7, 7]


# --------------------------------------------------------------------------- BVH / leaf-order geometry (host side of ptr_scene_upload)
def _geometry_ok(g, tris, spheres):
    assert g["triangles"] == tris and g["spheres"] == spheres
    assert g["triangles_referenced"] == tris and g["spheres_referenced"] == spheres
    assert g["unreferenced"] == 0 and g["multiply_referenced"] == 0 and g["bad_refs"] == 0
    assert g["box_violations"] == 0 and g["quant_violations"] == 0
    assert g["max_depth"] < 48          # kMaxTreeDepth: the traversal stacks (76 entries) can never overflow
    # the four-wide nodes of the persistent kernels: every second level collapsed, every primitive reached exactly once
    assert g["wide_problems"] == 0
    assert g["wide_nodes"] <= g["nodes"] and (g["nodes"] == 0 or g["wide_nodes"] >= 1)


@pytest.mark.parametrize("leaf_max", [1, 2, 4, 8])
def test_bvh_invariants_config2_mesh(leaf_max):
    host = pt.HostScene.load(os.path.join(ROOT, "scenes", "cornell_mesh.scene"), os.path.join(ROOT, "scenes"))
    g = pt.debug_scene_geometry(host.desc, leaf_max)
    _geometry_ok(g, 70688 + 2 * 6, 0)
    assert g["max_leaf_size"] <= leaf_max
    assert g["nodes"] >= (70700 // leaf_max) - 1 and g["quantized_usable"] == 1
    # SAH cost of the flattened tree (c_trav = c_int = 1): a regression guard, loose enough for builder tweaks
    assert g["sah_cost_milli"] < 60_000 * (1 if leaf_max >= 2 else 2)


def test_bvh_of_a_large_mesh_does_not_depend_on_the_thread_count():
    # 871 k triangles: above the size where the builder splits its passes over the threads and hands subtrees to a task pool
    # (csrc/host/bvh_builder.cpp).  The tree must be the same tree however many threads build it.
    from scenes.gen_assets import ensure_assets, ensure_large_asset
    ensure_assets()
    ensure_large_asset("torus_knot_871200.ply")
    host = pt.HostScene.load(os.path.join(ROOT, "scenes", "knot_glass.scene"), os.path.join(ROOT, "scenes"))
    keys = ("nodes", "leaves", "max_depth", "max_leaf_size", "sah_cost_milli", "oversize", "quantized_usable")
    seen = []
    for threads in ("1", "3", None):
        if threads is None:
            os.environ.pop("PTR_BUILD_THREADS", None)
        else:
            os.environ["PTR_BUILD_THREADS"] = threads
        try:
            g = pt.debug_scene_geometry(host.desc)
        finally:
            os.environ.pop("PTR_BUILD_THREADS", None)
        _geometry_ok(g, 871200 + 2 * 6, 0)
        seen.append(tuple(g[k] for k in keys))
    assert seen[0] == seen[1] == seen[2]


def test_bvh_invariants_mixed_and_degenerate_scenes(tmp_path):
    for name, tris, spheres in (("materials.scene", None, None), ("env_materials.scene", None, None)):
        host = pt.HostScene.load(os.path.join(ROOT, "tests", "golden", name), os.path.join(ROOT, "scenes"))
        d = host.desc
        want_tris = 2 * d.rectCount + sum(d.meshes[i].indexCount // 3 for i in range(d.meshCount))
        _geometry_ok(pt.debug_scene_geometry(d), want_tris, d.sphereCount)
    # one primitive, an empty scene, coincident spheres (zero-extent centroids: the split must still terminate)
    one = pt.HostScene.load(_write(tmp_path, "material type=lambert\nsphere center=0,0,0 radius=1 material=0\n"))
    g = pt.debug_scene_geometry(one.desc)
    _geometry_ok(g, 0, 1)
    assert g["nodes"] == 1 and g["leaves"] == 1
    empty = pt.HostScene.load(_write(tmp_path, "material type=lambert\n"))
    g = pt.debug_scene_geometry(empty.desc)
    assert g["nodes"] == 0 and g["triangles"] == 0 and g["spheres"] == 0
    same = "material type=lambert\n" + "sphere center=1,2,3 radius=0.5 material=0\n" * 300
    host = pt.HostScene.load(_write(tmp_path, same))     # keep the host scene alive: desc points into it
    g = pt.debug_scene_geometry(host.desc)
    _geometry_ok(g, 0, 300)
    assert g["max_leaf_size"] <= 4


def test_bvh_keeps_oversize_triangles_out_of_the_tree(tmp_path):
    # a finely tessellated mesh in a room 600 times its size: inside the tree the floor would stretch the 16-bit grid of the
    # quantised nodes until a cell is coarser than the mesh triangles (-> 64 B float nodes); kept out of the tree (and tested
    # first by every ray) the grid covers the mesh and the light only
    text = ("material type=lambert\nrectangle x=-1500,1500 y=0 z=-1500,1500 normal=1 material=0\n"
            "rectangle x=-40,40 y=90 z=-40,40 normal=-1 material=0\n"
            "mesh path=assets/blob_70688.obj translate=0,10,0 scale=0.05 material=0\n")
    host = pt.HostScene.load(_write(tmp_path, text), os.path.join(ROOT, "scenes"))
    g = pt.debug_scene_geometry(host.desc)
    _geometry_ok(g, 70688 + 4, 0)                       # every triangle referenced exactly once, the two floor halves included
    assert g["oversize"] == 2 and g["quantized_usable"] == 1
    os.environ["PTR_NO_OVERSIZE"] = "1"
    try:
        g0 = pt.debug_scene_geometry(host.desc)
    finally:
        del os.environ["PTR_NO_OVERSIZE"]
    _geometry_ok(g0, 70688 + 4, 0)
    assert g0["oversize"] == 0 and g0["quantized_usable"] == 0
    # the scenes whose grid is fine anyway are left alone
    cfg2 = pt.HostScene.load(os.path.join(ROOT, "scenes", "cornell_mesh.scene"), os.path.join(ROOT, "scenes"))
    assert pt.debug_scene_geometry(cfg2.desc)["oversize"] == 0


def test_bvh_rejects_bad_mesh_indices(tmp_path):
    host = pt.HostScene.load(_write(tmp_path, "material type=lambert\nmesh type=plane material=0\n"))
    d = host.desc
    idx = np.ctypeslib.as_array(d.meshes[0].indices, shape=(d.meshes[0].indexCount,))
    keep = idx[0]
    idx[0] = 1000
    with pytest.raises(pt.PtrError, match="index out of range"):
        pt.debug_scene_geometry(d)
    idx[0] = keep


# --------------------------------------------------------------------------- image output
def test_pfm_roundtrip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.random((5, 7, 3)).astype(np.float32) * 10
    p = str(tmp_path / "a.pfm")
    pt.write_image(p, img, "pfm")
    raw = open(p, "rb").read()
    assert raw.startswith(b"PF\n7 5\n-1.0\n")
    body = np.frombuffer(raw[len(b"PF\n7 5\n-1.0\n"):], dtype="<f4").reshape(5, 7, 3)
    assert np.array_equal(body[::-1], img)                  # bottom row first
    assert np.array_equal(pt.read_pfm(p), img)


def test_ppm_tonemap_and_exr_layout(tmp_path):
    img = np.zeros((2, 3, 3), dtype=np.float32)
    img[0, 0] = [1, 0.5, 0.0]
    img[1, 2] = [4, 4, 4]
    p = str(tmp_path / "a.ppm")
    pt.write_image(p, img, "ppm", tonemap=1)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n3 2\n255\n")
    px = np.frombuffer(raw[len(b"P6\n3 2\n255\n"):], dtype=np.uint8).reshape(2, 3, 3)
    assert px[0, 0].tolist() == [255, round(0.5 ** (1 / 2.2) * 255), 0] and px[1, 2].tolist() == [255, 255, 255]
    for mode in (2, 3, 4):
        pt.write_image(p, img, "ppm", tonemap=mode)
        assert os.path.getsize(p) == len(b"P6\n3 2\n255\n") + 18
    e = str(tmp_path / "a.exr")
    pt.write_image(e, img, "exr")
    raw = open(e, "rb").read()
    assert struct.unpack("<II", raw[:8]) == (20000630, 2)
    assert raw[8:17] == b"channels\x00" and b"B\x00" in raw[:80] and b"colorspace" not in raw
    header_end = raw.index(b"lineOrder\x00lineOrder\x00") + len(b"lineOrder\x00lineOrder\x00") + 4 + 1 + 1
    assert len(raw) == header_end + 2 * 8 + 2 * (8 + 3 * 3 * 4)
    first_line = raw[header_end + 16:]
    y, size = struct.unpack("<iI", first_line[:8])
    assert (y, size) == (0, 36)
    planes = np.frombuffer(first_line[8:8 + 36], dtype="<f4").reshape(3, 3)
    assert np.array_equal(planes[0], img[0, :, 2]) and np.array_equal(planes[2], img[0, :, 0])    # B, G, R planar
    pt.write_image(e, img, "exr", rgba_exr=True)
    assert b"colorspace\x00string\x00" in open(e, "rb").read()
    # PNG: tonemapped RGBA8 + sRGB chunk; decode with zlib and undo the Sub filter
    import zlib
    pt.write_image(str(tmp_path / "a.png"), img, "png", tonemap=2, exposure=0.5)
    raw = (tmp_path / "a.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    chunks, at = [], 8
    while at < len(raw):
        n = struct.unpack(">I", raw[at:at + 4])[0]
        kind, data = raw[at + 4:at + 8], raw[at + 8:at + 8 + n]
        assert struct.unpack(">I", raw[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + data)
        chunks.append((kind, data))
        at += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    w_, h_, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (w_, h_, depth, ctype) == (img.shape[1], img.shape[0], 8, 6)
    rows = np.frombuffer(zlib.decompress(chunks[2][1]), np.uint8).reshape(h_, 1 + 4 * w_)
    assert (rows[:, 0] == 1).all()
    px = np.cumsum(rows[:, 1:].reshape(h_, w_, 4).astype(np.uint32), axis=1).astype(np.uint8)     # Sub filter: prefix sums mod 256
    pt.write_image(str(tmp_path / "b.ppm"), img, "ppm", tonemap=2, exposure=0.5)
    ppm = (tmp_path / "b.ppm").read_bytes()
    ldr = np.frombuffer(ppm[ppm.index(b"255\n") + 4:], np.uint8).reshape(h_, w_, 3)
    assert np.array_equal(px[..., :3], ldr) and (px[..., 3] == 255).all()
    # a large smooth image compresses (LZ77 + fixed Huffman), and still round-trips
    big = np.linspace(0, 1, 256 * 192 * 3, dtype=np.float32).reshape(192, 256, 3) ** 3
    pt.write_image(str(tmp_path / "big.png"), big, "png")
    rawb = (tmp_path / "big.png").read_bytes()
    assert len(rawb) < 0.5 * 192 * 256 * 4
    idat = rawb[rawb.index(b"IDAT") + 4:rawb.index(b"IEND") - 8]
    assert len(zlib.decompress(idat)) == 192 * (1 + 4 * 256)
    # multilayer EXR: B,G,R,A + planar SAMPLES
    counts = np.arange(img.shape[0] * img.shape[1], dtype=np.float32).reshape(img.shape[:2])
    pt.write_exr_multilayer(str(tmp_path / "m.exr"), img, counts)
    data = (tmp_path / "m.exr").read_bytes()
    names = [n for n in (b"A\x00", b"B\x00", b"G\x00", b"R\x00", b"SAMPLES\x00") if n in data[:400]]
    assert len(names) == 5
    h0, w0 = img.shape[:2]
    body = np.frombuffer(data[-(h0 * (8 + 5 * w0 * 4)):], np.uint8).reshape(h0, 8 + 5 * w0 * 4)
    planes = body[:, 8:].copy().view(np.float32).reshape(h0, 5, w0)
    assert np.array_equal(planes[:, 4], counts) and np.array_equal(planes[:, 2], img[..., 0]) and (planes[:, 3] == 1.0).all()


def test_aov_exr_layers(tmp_path):
    # beauty + first-hit feature layers (ptr_host_write_exr_aovs): channel list in alphabetical order, normals decoded to unit vectors
    h, w = 3, 4
    rng = np.random.default_rng(5)
    rgb = rng.random((h, w, 3), dtype=np.float32)
    albedo = np.concatenate([rng.random((h, w, 3), dtype=np.float32), np.ones((h, w, 1), np.float32)], axis=2)
    n = rng.normal(size=(h, w, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=2, keepdims=True)
    normal = np.concatenate([n * 0.5 + 0.5, rng.uniform(1, 9, size=(h, w, 1)).astype(np.float32)], axis=2).astype(np.float32)
    albedo[0, 0] = 0.0                      # a miss: no hit flag
    normal[0, 0] = (0.5, 0.5, 0.5, 0.0)
    path = tmp_path / "aov.exr"
    pt.write_exr_aovs(str(path), rgb, albedo, normal)
    data = path.read_bytes()
    assert struct.unpack("<II", data[:8]) == (20000630, 2)
    names = [b"B", b"G", b"R", b"albedo.B", b"albedo.G", b"albedo.R", b"depth.Z", b"normal.X", b"normal.Y", b"normal.Z"]
    at = data.index(b"chlist\x00") + 7 + 4
    found = []
    while data[at] != 0:
        end = data.index(b"\x00", at)
        found.append(data[at:end])
        at = end + 1 + 16
    assert found == names
    body = np.frombuffer(data[-(h * (8 + 10 * w * 4)):], np.uint8).reshape(h, 8 + 10 * w * 4)
    planes = body[:, 8:].copy().view(np.float32).reshape(h, 10, w)
    assert np.array_equal(planes[:, 2], rgb[..., 0]) and np.array_equal(planes[:, 0], rgb[..., 2])
    assert np.array_equal(planes[:, 5], albedo[..., 0]) and np.array_equal(planes[:, 6], normal[..., 3])
    dec = np.stack([planes[:, 7], planes[:, 8], planes[:, 9]], axis=2)
    assert np.allclose(dec[1:], n[1:], atol=1e-6) and np.allclose(np.linalg.norm(dec[1:], axis=2), 1.0, atol=1e-5)
    assert (dec[0, 0] == 0).all() and planes[0, 6, 0] == 0.0


# --------------------------------------------------------------------------- CLI surface
def test_cli_flag_surface():
    exe = pt.CLI_PATH
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--sppTotal" in r.stdout and "--enableSoftwareRayTracing" in r.stdout
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "--scene is required" in r.stderr
    r = subprocess.run([exe, "--scene=x.scene", "--denoiser=1"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unknown option: --denoiser" in r.stderr       # documented-but-unparsed flag (quirk Q5)
    r = subprocess.run([exe, "--scene=x.scene", "--width=4"], capture_output=True, text=True)
    assert r.returncode == 1 and "--width must be >= 8" in r.stderr
    r = subprocess.run([exe, "--scene=/nonexistent.scene"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to load scene" in r.stderr
    # backend selection of the reference (main_headless.mm:344-371) is accepted: the names pick the integrator semantics
    for flags in (["--backend=embree"], ["--backend", "metal"], ["--enableEmbree"], ["--enableEmbree=0"], ["--backend=hip", "--semantics=metal"],
                  ["--devices=2"], ["--aovExr=/tmp/unused.exr"]):
        r = subprocess.run([exe, "--scene=/nonexistent.scene"] + flags, capture_output=True, text=True)
        assert r.returncode == 1 and "Failed to load scene" in r.stderr and "Unknown option" not in r.stderr, flags
    r = subprocess.run([exe, "--scene=x.scene", "--backend=optix"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid value for --backend" in r.stderr
    if pt.device_count() == 0:
        r = subprocess.run([exe, "--scene", os.path.join(GOLDEN, "smoke.scene"), "--sppTotal=1"], capture_output=True, text=True)
        assert r.returncode == 1 and "Render failed" in r.stderr


def test_cli_scene_identifiers_and_catalogue(tmp_path):
    # `--scene=<identifier>` is looked up among the *.scene files of <cwd>/assets; failures list the catalogue
    # sorted by display name (first `#` comment line, else the file stem)
    exe = pt.CLI_PATH
    assets = tmp_path / "assets"
    assets.mkdir()
    (assets / "alpha.scene").write_text("# Zebra room\nmaterial type=lambert\nsphere center=0,0,0 radius=1 material=0\n")
    (assets / "beta.scene").write_text("material type=lambert\n")
    (assets / "notes.txt").write_text("ignored")
    r = subprocess.run([exe, "--scene=gamma"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "Unknown scene identifier: gamma" in r.stderr
    listing = r.stderr[r.stderr.index("Available scenes:"):].split()
    assert listing[2:4] == ["alpha", "beta"]          # by display name: "Zebra room" < "beta" (byte order)
    r = subprocess.run([exe, "--scene=alpha", "--sppTotal=1", "--output", str(tmp_path / "o.pfm")], capture_output=True, text=True, cwd=tmp_path)
    assert "Unknown scene identifier" not in r.stderr and "Failed to load scene" not in r.stderr
    if pt.device_count() == 0:
        assert r.returncode == 1 and "Render failed" in r.stderr


# --------------------------------------------------------------------------- environment maps
def test_env_tables_match_oracle_bitwise_and_hdr_loader(tmp_path):
    import oracle_lib as ol

    host = pt.HostScene.load(os.path.join(GOLDEN, "env_materials.scene"), os.path.join(ROOT, "scenes"))
    d = host.desc
    assert (d.envWidth, d.envHeight) == (96, 48) and host.settings.backgroundMode == 2
    assert host.settings.environmentRotation == pytest.approx(np.radians(30.0), rel=1e-6)
    rgba = np.ctypeslib.as_array(d.envRgba, shape=(48, 96, 4)).copy()
    assert np.isfinite(rgba).all() and rgba[..., 3].min() == 1.0 and rgba[..., :3].max() > 1.0e4   # the brightest sun survives RGBE
    ours = pt.debug_env_distribution(rgba)
    rc, theirs = ol.env_build(rgba)
    assert rc == 0
    for key in ("pdf", "cond_alias", "cond_threshold", "marg_alias", "marg_threshold"):
        assert np.array_equal(ours[key], theirs[key]), key
    assert ours["total"] == theirs["total"]
    with pytest.raises(pt.PtrError):
        pt.debug_env_distribution(np.zeros((4, 8, 4), np.float32))
    # PFM is accepted as an environment format too (rows stored bottom-up)
    small = np.random.default_rng(1).random((4, 8, 3)).astype(np.float32)
    (tmp_path / "assets").mkdir()
    pt.write_image(str(tmp_path / "assets" / "e.pfm"), small, "pfm")
    (tmp_path / "s.scene").write_text("background env=assets/e.pfm\n")
    h2 = pt.HostScene.load(str(tmp_path / "s.scene"))
    got = np.ctypeslib.as_array(h2.desc.envRgba, shape=(4, 8, 4))
    assert np.array_equal(got[..., :3], small)
