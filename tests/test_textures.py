"""Material textures of the Metal metallic-roughness model (PtrSettings.metalSemantics bit PTR_METAL_PBR): the loader, the
filtering rule, the oracle's restatement (CPU), and - marked gpu - the HIP path against the oracle.

Fixture: tests/golden/textured.glb (written by tests/golden/make_textured_glb.py): PNG base colour with an alpha mask and a
KHR_texture_transform, one ORM PNG for metallic-roughness + occlusion, a PNG normal map (with and without vertex tangents),
a baseline-JPEG emissive map on TEXCOORD_1, mirrored-repeat / clamp samplers, alphaMode MASK and BLEND."""
import importlib
import io
import os

import numpy as np
import pytest

import oracle_lib as ol

pt = importlib.import_module("metal-pathtracer-arm64_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def textured():
    return pt.HostScene.load(os.path.join(GOLDEN, "textured.scene"), GOLDEN)


def _texels(desc, index):
    t = desc.textures[index]
    return np.ctypeslib.as_array(t.rgba, shape=(t.height, t.width, 4)).copy(), t


def test_loader_decodes_and_binds_the_textures(textured):
    d = textured.desc
    assert d.textureCount == 6 and d.meshCount == 3
    sizes = [(d.textures[i].width, d.textures[i].height) for i in range(6)]
    assert sizes == [(256, 128), (64, 64), (96, 96), (128, 64), (32, 32), (96, 96)]
    assert (d.textures[0].wrapS, d.textures[0].wrapT) == (2, 0) and (d.textures[5].wrapS, d.textures[5].wrapT) == (1, 1)     # mirrored / clamp
    # texture coordinates and tangents travel with the meshes; the ground quad's file has no TANGENT attribute: the loader gives it
    # MikkTSpace tangents, one vertex per triangle corner, as the reference does (src/assets/TangentGen.mm:181-230)
    assert all(bool(d.meshes[i].uv0) and bool(d.meshes[i].uv1) for i in range(3))
    assert all(bool(d.meshes[i].tangents) for i in range(3))
    ground_mesh = d.meshes[2]
    assert ground_mesh.vertexCount == ground_mesh.indexCount == 6
    assert list(np.ctypeslib.as_array(ground_mesh.indices, (6,))) == [0, 1, 2, 3, 4, 5]
    tangents = np.ctypeslib.as_array(ground_mesh.tangents, (6, 4))
    assert np.allclose(np.linalg.norm(tangents[:, :3], axis=1), 1.0, atol=1e-6) and np.all(np.abs(tangents[:, 3]) == 1.0)
    assert np.allclose(tangents, tangents[0], atol=1e-6)                    # a flat quad with one linear mapping: one frame
    shell, glow, ground = d.materials[1], d.materials[2], d.materials[3]
    assert list(shell.textureIndices0) == [0, 1, 2, 1] and list(glow.textureIndices1)[:2] == [3, 0xFFFFFFFF]
    assert list(ground.textureIndices0) == [4, 0xFFFFFFFF, 5, 0xFFFFFFFF]
    assert list(shell.pbrExtras)[3] == 1.0 and list(glow.pbrExtras)[3] == 2.0 and abs(list(glow.pbrExtras)[0] - 0.6) < 1e-6     # MASK / BLEND
    assert list(glow.textureUvSet1)[0] == 1                                 # the emissive map reads TEXCOORD_1
    rows = np.array([list(shell.textureTransform[0])[:3], list(shell.textureTransform[1])[:3]])
    c, s = np.cos(0.2), np.sin(0.2)
    # rows as the reference builds them (src/assets/GltfLoader.mm:615-631): (c sx, -s sy, tx), (s sx, c sy, ty)
    assert np.allclose(rows, [[2 * c, -1 * s, 0.1], [2 * s, 1 * c, 0.0]], atol=1e-6)
    # sRGB-encoded base colour is decoded to linear, data textures are not; alpha is always linear
    Image = pytest.importorskip("PIL.Image")
    import struct
    raw = open(os.path.join(GOLDEN, "textured.glb"), "rb").read()
    jlen = struct.unpack("<I", raw[12:16])[0]
    import json
    doc = json.loads(raw[20:20 + jlen])
    binary = raw[20 + jlen + 8:]

    def image_bytes(i):
        v = doc["bufferViews"][doc["images"][i]["bufferView"]]
        return binary[v["byteOffset"]:v["byteOffset"] + v["byteLength"]]

    base8 = np.array(Image.open(io.BytesIO(image_bytes(0))).convert("RGBA")).astype(np.float64) / 255.0
    linear = np.where(base8 <= 0.04045, base8 / 12.92, ((base8 + 0.055) / 1.055) ** 2.4)
    tex0, _ = _texels(d, 0)
    assert np.allclose(tex0[..., :3], linear[..., :3], atol=2e-6) and np.allclose(tex0[..., 3], base8[..., 3], atol=1e-7)
    orm8 = np.array(Image.open(io.BytesIO(image_bytes(1))).convert("RGB")).astype(np.float64) / 255.0
    assert np.allclose(_texels(d, 1)[0][..., :3], orm8, atol=1e-7)
    jpg8 = np.array(Image.open(io.BytesIO(image_bytes(3))).convert("RGB")).astype(int)
    mine = pt.decode_image(image_bytes(3))[..., :3].astype(int)
    assert np.abs(mine - jpg8).mean() < 2.0                                     # the library's own JPEG decoder (4:2:0, chroma replicated)


def _reference_sample(tex, info, u, v, lod):
    """Independent numpy statement of the filtering rule (csrc/kernels/texture.h): box-filtered mips, bilinear per level with texel
    centres at (i + 0.5) / W, wrap per sampler, linear between the two nearest levels.  float32 arithmetic in the same order."""
    f = np.float32
    levels = [tex.astype(np.float32)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        src = levels[-1]
        h, w = src.shape[:2]
        nh, nw = max(h // 2, 1), max(w // 2, 1)
        ys0, ys1 = np.minimum(2 * np.arange(nh), h - 1), np.minimum(2 * np.arange(nh) + 1, h - 1)
        xs0, xs1 = np.minimum(2 * np.arange(nw), w - 1), np.minimum(2 * np.arange(nw) + 1, w - 1)
        a, b = src[ys0][:, xs0], src[ys0][:, xs1]
        c, d = src[ys1][:, xs0], src[ys1][:, xs1]
        levels.append(((a + b) + (c + d)) * f(0.25))

    def wrap(i, n, mode):
        if mode == 1:
            return min(max(i, 0), n - 1)
        if mode == 2:
            j = i % (2 * n)
            return j if j < n else 2 * n - 1 - j
        return i % n

    def bilinear(level):
        img = levels[level]
        h, w = img.shape[:2]
        fx, fy = f(u) * f(w) - f(0.5), f(v) * f(h) - f(0.5)
        x0f, y0f = np.floor(fx), np.floor(fy)
        tx, ty = f(fx - x0f), f(fy - y0f)
        x0, x1 = wrap(int(x0f), w, info.wrapS), wrap(int(x0f) + 1, w, info.wrapS)
        y0, y1 = wrap(int(y0f), h, info.wrapT), wrap(int(y0f) + 1, h, info.wrapT)
        ix, iy = f(1) - tx, f(1) - ty
        return (img[y0, x0] * ix + img[y0, x1] * tx) * iy + (img[y1, x0] * ix + img[y1, x1] * tx) * ty

    l = min(max(f(lod), f(0)), f(len(levels) - 1))
    l0 = int(np.floor(l))
    frac = f(l - f(l0))
    l1 = min(l0 + 1, len(levels) - 1)
    a = bilinear(l0)
    if not frac > 0 or l1 == l0:
        return a
    b = bilinear(l1)
    return a + (b - a) * frac


def test_oracle_filtering_rule_matches_an_independent_statement(textured):
    d = textured.desc
    rng = np.random.default_rng(3)
    for index in (0, 2, 5):                      # mirrored / repeat / clamp samplers; 256x128 and 96x96 (odd mip sizes on the way down)
        tex, info = _texels(d, index)
        q = np.stack([rng.uniform(-1.5, 2.5, 200), rng.uniform(-1.5, 2.5, 200), rng.uniform(-0.5, 8.5, 200)], axis=1).astype(np.float32)
        got = ol.texture_sample(textured, index, q)
        want = np.array([_reference_sample(tex, info, *row) for row in q], dtype=np.float32)
        assert np.allclose(got, want, rtol=2e-6, atol=2e-7), index
    # level 0 at a texel centre returns that texel; the coarsest level is the mean of the image; a missing texture reports -1
    tex, info = _texels(d, 1)
    centre = ol.texture_sample(textured, 1, np.array([[(10 + 0.5) / 64, (20 + 0.5) / 64, 0.0]], np.float32))[0]
    assert np.allclose(centre, tex[20, 10], atol=1e-6)
    coarse = ol.texture_sample(textured, 1, np.array([[0.3, 0.7, 99.0]], np.float32))[0]
    assert np.allclose(coarse, tex.reshape(-1, 4).mean(axis=0), atol=2e-4)
    assert (ol.texture_sample(textured, 77, np.zeros((1, 3), np.float32)) == -1).all()


def test_oracle_textured_model_changes_the_image_where_it_should(textured):
    osc = ol.OracleScene(textured)
    s0 = textured.settings_for(width=96, height=72, seed=1337)
    s1 = s0.copy()
    s1.metalSemantics = 32
    plain, _, _ = osc.render(s0, 16, threads=4)
    tex, _, _ = osc.render(s1, 16, threads=4)
    assert np.isfinite(tex).all() and tex.min() >= 0
    assert float(np.sqrt(((plain - tex) ** 2).mean())) > 0.05                   # Embree-parity mode reads the factors only
    # the emissive lower sphere glows (orange) only in the textured Metal mode; the Embree backend ignores PBR emission
    lower = (slice(44, 56), slice(38, 58))
    assert tex[lower][..., 0].mean() > 1.3 * tex[lower][..., 2].mean()
    # deterministic: the stochastic alpha test draws from the path's own stream
    again, _, _ = osc.render(s1, 16, threads=2)
    assert np.array_equal(again, tex)


# --------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_filtering_matches_the_oracle(textured):
    dev = pt.DeviceScene(textured.desc, 0, keepalive=textured)
    rng = np.random.default_rng(5)
    for index in range(6):
        q = np.stack([rng.uniform(-2, 3, 4000), rng.uniform(-2, 3, 4000), rng.uniform(-0.5, 9.0, 4000)], axis=1).astype(np.float32)
        g = dev.texture_sample(index, q)
        o = ol.texture_sample(textured, index, q)
        assert np.array_equal(g, o), index                                      # same operations in the same order: bit for bit
    assert (dev.texture_sample(99, np.zeros((4, 3), np.float32)) == -1).all()


@pytest.mark.gpu
def test_textured_scene_image_parity(textured):
    dev, osc = pt.DeviceScene(textured.desc, 0, keepalive=textured), ol.OracleScene(textured)
    lum = np.array([0.2126, 0.7152, 0.0722])
    for sem in (32, 127):                                                       # the textured model alone, and with every other Metal behaviour
        s = textured.settings_for(width=160, height=120, max_depth=6, seed=1337, metalSemantics=sem)
        g1, _ = dev.render_image(s, 1)
        o1, _, _ = osc.render(s, 1, threads=8)
        rel = np.abs(g1 - o1) / (np.abs(o1) + 1e-2)
        assert float((rel.max(axis=2) <= 1e-3).mean()) >= 0.95, sem
        g, _ = dev.render_image(s, 32)
        o, _, _ = osc.render(s, 32, threads=8)
        s2 = s.copy()
        s2.seed = 1338
        o2, _, _ = osc.render(s2, 32, threads=8)
        rmse = lambda a, b: float(np.sqrt(np.mean((a.astype(np.float64) - b) ** 2)))
        assert rmse(g, o) <= 1.25 * rmse(o, o2) and abs((g @ lum).mean() / (o @ lum).mean() - 1.0) <= 0.005, sem
    # the Embree-parity integrator is untouched by the textures
    s0 = textured.settings_for(width=96, height=72, max_depth=5, seed=1337)
    g0, _ = dev.render_image(s0, 8)
    o0, _, _ = osc.render(s0, 8, threads=8)
    assert rmse(g0, o0) < 0.02 * max(float(o0.mean()), 1e-3) + 1e-3
