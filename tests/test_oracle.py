"""CPU tests of the oracle (oracle/): known answers, self-consistency, and the reference's own fixtures.

Reference pins covered here:
  * tests/scenes/smoke.scene (copied as data to tests/golden/smoke.scene) rendered with the paper's command line
    (paper/paper.md:160-188): the EXR the reference's Embree backend writes is 66,925 bytes — reproduced exactly.
    Its sha256 comes from an Apple-Silicon build (Apple libm + real Embree) and is not reproducible here; the
    test records the mismatch instead of asserting it ("parity unpinned" at content level).
  * tests/public/headless_smoke_test.sh: output exists and is non-empty (covered via the CLI in test_host.py).
  * src/headless/EmbreeSmokeTest.cpp: one triangle, one ray, must hit.
"""
import ctypes as C
import hashlib
import importlib
import os

import numpy as np
import pytest

import oracle_lib as ol

pt = importlib.import_module("metal-pathtracer-arm64_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PAPER_SHA256 = "6a5e6c9b2d50bc81c67a5463937575aeb0a367aec547036eef0ab2905a16406c"  # paper/paper.md:187


def lowbias32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def test_rng_hash_known_answers():
    # independent Python statement of lowbias32 (EmbreeHeadlessRenderer.mm:55-62)
    for seed in (0, 1, 1337, 0x9E3779B9, 0xFFFFFFFF, 123456789):
        assert ol.rng_hash(seed) == lowbias32(seed)
    floats, state = ol.rng_floats(1337, 8)
    s = 1337
    for f in floats:
        s = lowbias32(s)
        assert f == np.float32((s & 0xFFFFFF) / 16777216.0)
    assert state == s
    assert np.all((floats >= 0) & (floats < 1))


def test_smoke_scene_exr_size_matches_paper(tmp_path):
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    s = host.settings_for(width=64, height=64, max_depth=4, seed=1337)
    img, _, _ = ol.OracleScene(host).render(s, 4, threads=1)
    assert img.shape == (64, 64, 3) and np.isfinite(img).all() and img.min() >= 0
    out = tmp_path / "smoke.ppm"  # the reference keeps the .ppm name but writes an RGBA EXR (main_headless.mm:549-583)
    pt.write_image(str(out), img, "exr", rgba_exr=True)
    data = out.read_bytes()
    assert len(data) == 66925
    assert data[:4] == (20000630).to_bytes(4, "little")
    # background pixels are exactly the solid colour of the scene file
    assert np.allclose(img[0, 0], [0.7, 0.8, 1.0])


@pytest.mark.xfail(strict=False, reason="parity unpinned: the paper's sha256 was taken on Apple Silicon (Apple libm, real Embree 4.4 SIMD "
                   "kernels); the oracle restates that code path with glibc libm and a scalar ray caster, so low-order bits differ")
def test_smoke_scene_exr_sha256_matches_paper(tmp_path):
    """The reference's one content-level pin (paper/paper.md:183-188).  Kept as a visible expected failure: it shows that
    the oracle is pinned by structure (byte count, header, exact background) but not by content."""
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    s = host.settings_for(width=64, height=64, max_depth=4, seed=1337)
    img, _, _ = ol.OracleScene(host).render(s, 4, threads=1)
    out = tmp_path / "smoke.ppm"
    pt.write_image(str(out), img, "exr", rgba_exr=True)
    assert hashlib.sha256(out.read_bytes()).hexdigest() == PAPER_SHA256


def test_render_is_independent_of_thread_count_and_rows():
    host = pt.HostScene.load(os.path.join(GOLDEN, "cornell_small_mesh.scene"), os.path.join(ROOT, "scenes"))
    s = host.settings_for(width=48, height=40, max_depth=4)
    osc = ol.OracleScene(host)
    a, _, _ = osc.render(s, 3, threads=1)
    b, _, _ = osc.render(s, 3, threads=4)
    assert np.array_equal(a, b)
    c, _, _ = osc.render(s, 3, threads=2, rows=(16, 32))
    assert np.array_equal(c[16:32], a[16:32]) and not c[:16].any() and not c[32:].any()


def _single_triangle_desc():
    desc = pt.PtrSceneDesc()
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    nrm = np.tile(np.array([0, 0, 1], dtype=np.float32), (3, 1))
    idx = np.array([0, 1, 2], dtype=np.uint32)
    mesh = pt.PtrMeshDesc()
    mesh.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
    mesh.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
    mesh.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
    mesh.vertexCount, mesh.indexCount = 3, 3
    for i, v in enumerate(np.eye(4, dtype=np.float32).T.reshape(-1)):
        mesh.localToWorld[i] = v
    meshes = (pt.PtrMeshDesc * 1)(mesh)
    desc.meshes = meshes
    desc.meshCount = 1
    return desc, (pos, nrm, idx, meshes)


def test_embree_smoke_test_triangle_hit():
    # src/headless/EmbreeSmokeTest.cpp:6-75: one triangle, one ray, exit 0 iff hit
    desc, keep = _single_triangle_desc()

    class H:
        pass

    h = H()
    h.desc = desc
    h.keep = keep
    osc = ol.OracleScene(h)
    rays = np.array([[0.25, 0.25, -1.0, 0.0, 0, 0, 1, np.inf], [2.0, 2.0, -1.0, 0.0, 0, 0, 1, np.inf]], dtype=np.float32)
    hits = osc.trace_rays(rays)
    assert hits["t"][0] == pytest.approx(1.0) and hits["u"][0] == pytest.approx(0.25) and hits["v"][0] == pytest.approx(0.25)
    assert hits["t"][1] < 0
    assert osc.trace_rays(rays, any_hit=True)["t"].tolist() == [0.0, -1.0]


def test_bvh_matches_brute_force():
    host = pt.HostScene.load(os.path.join(GOLDEN, "cornell_small_mesh.scene"), os.path.join(ROOT, "scenes"))
    osc = ol.OracleScene(host)
    rng = np.random.default_rng(7)
    n = 4000
    org = rng.uniform(10, 545, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(50, 400, n)).astype(np.float32)
    rays = np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, tmax[:, None]], axis=1)
    a = osc.trace_rays(rays)
    b = osc.trace_rays(rays, brute_force=True)
    assert np.array_equal(a["t"], b["t"])
    tie_free = a["t"] >= 0
    assert (a["primIndex"][tie_free] == b["primIndex"][tie_free]).mean() > 0.999
    assert np.array_equal(osc.trace_rays(rays, any_hit=True)["t"], osc.trace_rays(rays, any_hit=True, brute_force=True)["t"])


def test_sphere_semantics_front_then_back_root():
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    osc = ol.OracleScene(host)
    # from outside: front root; from inside the small sphere: back root
    rays = np.array([[0, 0, 2.5, 1e-4, 0, 0, -1, np.inf], [0, 0, -1, 1e-4, 0, 0, -1, np.inf]], dtype=np.float32)
    hits = osc.trace_rays(rays)
    assert hits["t"][0] == pytest.approx(3.0, abs=1e-5) and hits["primType"][0] == 1 and hits["primIndex"][0] == 0
    assert hits["t"][1] == pytest.approx(0.5, abs=1e-5)


def test_camera_basis_matches_closed_form():
    host = pt.HostScene.load(os.path.join(ROOT, "scenes", "cornell.scene"))
    s = host.settings_for()
    cam = ol.build_camera(s)
    origin, lower_left, horizontal, vertical = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    assert np.allclose(origin, [278, 278, -800], atol=2e-2)  # target + 1078*(cos p cos y, sin p, cos p sin y), yaw=-1.5708
    h = 2 * np.tan(np.radians(40) / 2) * 1078
    assert np.linalg.norm(vertical) == pytest.approx(h, rel=1e-5)
    assert np.linalg.norm(horizontal) == pytest.approx(h * s.width / s.height, rel=1e-5)
    centre = lower_left + 0.5 * horizontal + 0.5 * vertical
    assert np.allclose(centre, [278, 278, 278], atol=5e-2)
    assert cam[18] == 0.0
    rays, states = ol.camera_rays(s, np.array([[0, 0, 0], [s.width - 1, s.height - 1, 3]], dtype=np.uint32))
    assert np.allclose(np.linalg.norm(rays[:, 3:], axis=1), 1.0, atol=1e-6)
    assert rays[0, 3] > 0 and rays[0, 4] > 0      # pixel (0,0) is top-left: +u is -x here, so dir.x > 0
    assert rays[1, 3] < 0 and rays[1, 4] < 0


def test_env_distribution_properties():
    rng = np.random.default_rng(3)
    h, w = 8, 16
    rgba = np.ones((h, w, 4), dtype=np.float32)
    rgba[..., :3] = rng.uniform(0.01, 1.0, size=(h, w, 3)).astype(np.float32)
    rgba[2, 5, :3] = 500.0
    rc, d = ol.env_build(rgba)
    assert rc == 0
    theta = (np.arange(h) + 0.5) * np.pi / h
    cell = np.sin(theta) * (np.pi / h) * (2 * np.pi / w)
    assert np.sum(d["pdf"] * cell[:, None]) == pytest.approx(1.0, rel=1e-4)     # pdf integrates to 1 over the sphere
    assert ((d["cond_threshold"] >= 0) & (d["cond_threshold"] <= 1)).all() and (d["cond_alias"] < w).all()
    assert (d["marg_alias"] < h).all()
    # alias tables reproduce the distribution they encode
    lum = 0.2126 * rgba[..., 0] + 0.7152 * rgba[..., 1] + 0.0722 * rgba[..., 2]
    weights = lum * cell[:, None]
    row_p = weights.sum(1) / weights.sum()
    recon = np.zeros(h)
    for y in range(h):
        recon[y] += d["marg_threshold"][y] / h
        recon[d["marg_alias"][y]] += (1 - d["marg_threshold"][y]) / h
    assert np.allclose(recon, row_p, atol=1e-5)
    # sampling: directions are unit length, pdf is the sampled texel's pdf, bright texel dominates
    u = rng.random((4000, 3)).astype(np.float32)
    rc, out, look = ol.env_sample(rgba, 0.3, 1.0, u)
    assert rc == 0 and np.allclose(np.linalg.norm(out[:, :3], axis=1), 1.0, atol=1e-5)
    bright = np.isclose(out[:, 3], 500.0)
    expected = weights[2, 5] / weights.sum()
    assert abs(bright.mean() - expected) < 0.03
    # quirk Q2: the lookup of a sampled direction lands half a turn away from the sampled texel
    assert not np.allclose(look[bright][:, 0], 500.0)


def test_bsdf_reciprocity_and_energy_bounds():
    host = pt.HostScene.load(os.path.join(GOLDEN, "materials.scene"))
    s = host.settings_for(width=16, height=16)
    rng = np.random.default_rng(11)
    n = 512

    def hemi(k):
        v = rng.normal(size=(k, 3))
        v[:, 2] = np.abs(v[:, 2]) + 0.05
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    wo, wi = hemi(n), hemi(n)
    normal = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    pos = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    fwd = np.concatenate([pos, normal, wo, wi], axis=1)
    rev = np.concatenate([pos, normal, wi, wo], axis=1)
    for mi in range(host.desc.materialCount):
        mat = host.desc.materials[mi]
        mtype = int(mat.typeEta[0])
        a = ol.eval_bsdf(mat, s, fwd)
        b = ol.eval_bsdf(mat, s, rev)
        assert np.isfinite(a).all() and (a[:, :4] >= 0).all()
        if mtype in (0, 5):          # Lambert / subsurface: albedo/pi, cosine pdf
            assert np.allclose(a[:, :3], np.clip(np.array(mat.baseColorRoughness[:3]), 0, 1) / np.pi, rtol=1e-6)
            assert np.allclose(a[:, 3], wi[:, 2] / np.pi, rtol=1e-5)
        if mtype in (0, 1, 5):       # these lobes are reciprocal in value (metal: F uses wi.wh == wo.wh)
            assert np.allclose(a[:, :3], b[:, :3], rtol=2e-4, atol=1e-6)
        if mtype == 2:
            assert (a[:, 4] == 1).all() and (a[:, :4] == 0).all()


def test_sample_bsdf_matches_eval_pdf_and_weight():
    host = pt.HostScene.load(os.path.join(GOLDEN, "materials.scene"))
    s = host.settings_for(width=16, height=16)
    rng = np.random.default_rng(5)
    n = 2000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.2
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    pos = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    inp = np.concatenate([pos, normal, wo], axis=1)
    for mi in range(host.desc.materialCount):
        mat = host.desc.materials[mi]
        mtype = int(mat.typeEta[0])
        out, new_states = ol.sample_bsdf(mat, s, inp, front, states)
        ok = out[:, 6] > 0
        assert np.isfinite(out).all() and ok.mean() > 0.5
        assert (new_states != states).all() or mtype == 1
        d = out[ok, :3]
        assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-5)
        if mtype in (0, 4, 5, 6, 7) or (mtype == 1 and mat.baseColorRoughness[3] > 1e-3):
            nd = ~(out[ok, 7] > 0)
            ev = ol.eval_bsdf(mat, s, np.concatenate([pos[ok], normal[ok], wo[ok], d], axis=1))
            # weight = f * cos / pdf and pdf agrees with EvaluateBsdf's pdf for the sampled direction
            if mtype in (0, 5, 4, 7):
                assert np.allclose(out[ok, 6][nd], ev[nd, 3], rtol=2e-3, atol=1e-6)
                w = ev[nd, :3] * d[nd, 2:3] / ev[nd, 3:4]
                assert np.allclose(out[ok, 3:6][nd], w, rtol=5e-3, atol=1e-5)
        if mtype == 2:
            assert (out[:, 7] == 1).all() and (out[:, 6] == 1).all()


def test_lambert_furnace_converges_to_albedo_series():
    # closed white-ish box seen from inside with a solid background never reached: radiance must stay 0;
    # open sky version: a single Lambert sphere under a uniform white sky returns ~albedo at normal incidence
    host = pt.HostScene.load(os.path.join(GOLDEN, "furnace.scene"))
    s = host.settings_for(width=16, height=16, max_depth=64)
    img, _, _ = ol.OracleScene(host).render(s, 256, threads=0)
    centre = img[6:10, 6:10].mean(axis=(0, 1))
    # sum_k albedo^k over bounces that escape = albedo for a convex body under a unit sky
    assert np.allclose(centre, [0.5, 0.5, 0.5], atol=0.02)


# --------------------------------------------------------------------------- Metal-only media semantics (oracle extension)
SLAB_SCENE = ("camera target=0,0,0 distance=10 yaw=1.5708 pitch=0 vfov=2\n"
              "renderer width=16 height=16 maxDepth=12 seed=5 russianRoulette=0\n"
              "background solid=1,1,1\n"
              "material type=dielectric ior=1.5 sigmaA=0.30,0.10,0.02 name=tinted\n"
              "material type=dielectric ior=1.5 sigmaA=0.30,0.10,0.02 thin=1 name=thin_tinted\n"
              "box min=-4,-4,-1 max=4,4,1 material=%d\n")


def _slab_centre(host, semantics, spp=64):
    s = host.settings_for(seed=5, metalSemantics=semantics, fireflyClampEnabled=0)
    img, _, _ = ol.OracleScene(host).render(s, spp, threads=0)
    return img[6:10, 6:10].reshape(-1, 3).mean(axis=0).astype(np.float64)


def test_beer_lambert_through_a_glass_slab(tmp_path):
    """A camera looks straight through a 2-unit slab of tinted glass at a white background.  With the Metal media
    semantics the radiance is the clear-glass radiance times exp(-sigmaA * 2) (internal double reflections add a
    factor 1 + O(Fr^4) ~ 1e-5); on the Embree path sigmaA is ignored.  shaders/pathtrace.metal:5869-5876, 6694-6709."""
    p = tmp_path / "slab.scene"
    p.write_text(SLAB_SCENE % 0)
    host = pt.HostScene.load(str(p))
    fr = ((1.5 - 1.0) / (1.5 + 1.0)) ** 2
    # Embree-path quirk: the unflipped normal turns the exit refraction into a U-turn, so only the first-surface
    # reflection (picked with probability Fr, weighted Fr) reaches the background
    assert np.allclose(_slab_centre(host, 0), fr * fr, rtol=0.15)
    clear = _slab_centre(host, 4)            # ray-facing normals, no absorption
    tinted = _slab_centre(host, 4 | 1)
    # quirk shared by both backends: the refraction weight is not divided by its selection probability, so each
    # interface contributes (1-Fr)^2 * scale and the two scales cancel
    assert np.allclose(clear, (1.0 - fr) ** 4 + fr * fr, rtol=0.03)
    expect = np.exp(-np.array([0.30, 0.10, 0.02]) * 2.0)
    assert np.allclose((tinted - fr * fr) / (clear - fr * fr), expect, rtol=0.02), (tinted, clear, expect)
    # thin-walled glass never enters a medium: no absorption, and both faces refract air -> glass
    p.write_text(SLAB_SCENE % 1)
    thin_host = pt.HostScene.load(str(p))
    thin_metal = _slab_centre(thin_host, 7)
    assert np.allclose(_slab_centre(thin_host, 4), clear, rtol=0.03)        # without the THIN bit the flag is ignored
    # air->glass at both faces: weight (1-Fr)^2 * 2.25 per interface, not cancelled by a glass->air exit
    assert np.allclose(thin_metal, ((1.0 - fr) ** 2 * 2.25) ** 2 + fr * fr, rtol=0.03), thin_metal


def test_metal_specular_semantics_consistency_and_furnace(tmp_path):
    """PTR_METAL_SPECULAR (shaders/pathtrace.metal:3724-3739, 3770-3797, 4610-4630, 5000-5023, 5228-5283): rough
    metals sample visible normals, report the G1 pdf and carry the multiple-scattering compensation."""
    p = tmp_path / "furnace.scene"
    p.write_text("camera target=0,0,0 distance=6 yaw=1.0 pitch=0.3 vfov=25\nrenderer width=24 height=24 maxDepth=10 seed=3 russianRoulette=0\n"
                 "background solid=1,1,1\nmaterial type=metal albedo=1,1,1 roughness=0.7\nsphere center=0,0,0 radius=1 material=0\n")
    host = pt.HostScene.load(str(p))
    mat = host.desc.materials[0]
    rng = np.random.default_rng(11)
    n = 4000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.1
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    pos = np.zeros((n, 3), np.float32)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    inp = np.concatenate([pos, normal, wo], axis=1)
    s8 = host.settings_for(metalSemantics=8, fireflyClampEnabled=0)
    s0 = host.settings_for(metalSemantics=0, fireflyClampEnabled=0)
    out8, st8 = ol.sample_bsdf(mat, s8, inp, front, states)
    out0, st0 = ol.sample_bsdf(mat, s0, inp, front, states)
    assert np.array_equal(st8, st0)                       # both samplers draw two numbers
    ok = out8[:, 6] > 0
    assert ok.mean() > 0.8 and ok.mean() > (out0[:, 6] > 0).mean()      # visible normals waste fewer samples
    d = out8[ok, :3]
    ev = ol.eval_bsdf(mat, s8, np.concatenate([pos[ok], normal[ok], wo[ok], d], axis=1))
    assert np.allclose(out8[ok, 6], ev[:, 3], rtol=2e-3, atol=1e-6)     # sampled pdf == evaluated pdf (with G1)
    assert np.allclose(out8[ok, 3:6], ev[:, :3] * d[:, 2:3] / ev[:, 3:4], rtol=5e-3, atol=1e-5)
    # (the Metal pdf D*G1*cos(h)/(4 wo.wh) is not the density of the VNDF sampler - that would be D*G1/(4 cos(o)) -
    #  so weights are F * comp * G1(wi) * wo.wh / (cos(o) cos(h)): the reference's own formula, restated as is)
    # white furnace: a white rough metal sphere under a white sky loses energy to single scattering; the
    # compensation puts most of it back (scale is clamped to [1, 2], so never above the sky)
    osc = ol.OracleScene(host)
    img0, _, _ = osc.render(s0, 256, threads=0)
    img8, _, _ = osc.render(s8, 256, threads=0)
    c0, c8 = float(img0[8:16, 8:16].mean()), float(img8[8:16, 8:16].mean())
    assert c0 < 0.93 and c8 > c0 + 0.03, (c0, c8)


SSS_SCENE = ("camera target=0,0,0 distance=6 yaw=1.0 pitch=0.3 vfov=25\n"
             "renderer width=24 height=24 maxDepth=6 seed=3 russianRoulette=0 sss=%s\n"
             "background solid=1,1,1\n"
             "material type=sss albedo=0.8,0.5,0.3 mfp=0.25 name=skin\n"
             "material type=sss albedo=0.8,0.5,0.3 mfp=0.25 method=randomwalk name=walker\n"
             "sphere center=0,0,0 radius=1 material=0\n")


def _dipole_reflectance(albedo, mfp):
    """pi * integral of the reference's normalized_diffusion_profile over the tangent plane, truncated at 10 mfp like the
    sampler's radius (numerical; shaders/pathtrace.metal:3916-3971)."""
    sigma_t = 1.0 / mfp
    sigma_s = np.clip(albedo, 0.0, 0.999) * sigma_t            # g = 0
    sigma_a = np.maximum(sigma_t - sigma_s, 1e-6)
    stp = sigma_a + sigma_s
    alpha = sigma_s / stp
    D = 1.0 / (3.0 * stp)
    str_ = np.sqrt(sigma_a / D)
    r = np.linspace(1e-4, 10.0 * mfp, 400001)[:, None]
    zr = 1.0 / stp
    vr = zr + 4.0 * D
    dr = np.sqrt(r * r + zr * zr)
    dv = np.sqrt(r * r + vr * vr)
    prof = alpha / (4 * np.pi) * (zr * (1 + str_ * dr) * np.exp(-str_ * dr) / dr ** 3 + vr * (1 + str_ * dv) * np.exp(-str_ * dv) / dv ** 3)
    return np.pi * np.trapezoid(prof * 2 * np.pi * r, r[:, 0], axis=0)


def test_metal_separable_subsurface_sampler(tmp_path):
    """PTR_METAL_SSS with sssMode = 1 (shaders/pathtrace.metal:5398-5481): exit point on the tangent plane at a radius
    drawn from exp(-sigma_tr r), cosine direction, weight = profile * cos / (pdfArea * pdfDir).  The expected weight is
    pi x the integral of the profile (the reference's weight has no 1/pi), which pins the restated profile and pdfs;
    evaluation returns zero (no NEE); without the bit, or with sssMode 0 / a random-walk material, type 5 stays Lambert."""
    p = tmp_path / "sss.scene"
    p.write_text(SSS_SCENE % "separable")
    host = pt.HostScene.load(str(p))
    mat = host.desc.materials[0]
    n = 200000
    rng = np.random.default_rng(5)
    normal = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    wo = np.tile(np.array([0.3, 0.1, 0.9], np.float32) / np.linalg.norm([0.3, 0.1, 0.9]).astype(np.float32), (n, 1))
    pos = np.zeros((n, 3), np.float32)
    inp = np.concatenate([pos, normal, wo], axis=1).astype(np.float32)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    s = host.settings_for(metalSemantics=16, fireflyClampEnabled=0)
    assert s.sssMode == 1                                   # `renderer sss=separable` reaches the settings
    out, st = ol.sample_bsdf(mat, s, inp, front, states)
    ok = out[:, 6] > 0
    assert ok.mean() > 0.999
    assert np.allclose(out[ok, 6], out[ok, 2] / np.pi, rtol=1e-4)        # the pdf carried on is the cosine pdf of the direction
    mean_w = out[:, 3:6].astype(np.float64).mean(axis=0)
    # coatParams.w = the coat's average Fresnel reflectance; the weight carries (1 - that) even without a coat (:5437)
    expect = _dipole_reflectance(np.array([0.8, 0.5, 0.3]), 0.25) * (1.0 - mat.coatParams[3])
    assert np.allclose(mean_w, expect, rtol=0.03), (mean_w, expect)
    # four numbers per sample: radius, angle, two for the direction (Lambert draws two)
    s0 = host.settings_for(metalSemantics=0, fireflyClampEnabled=0)
    out0, st0 = ol.sample_bsdf(mat, s0, inp, front, states)
    assert not np.array_equal(st, st0)
    assert np.allclose(out0[:, 3:6], np.array([0.8, 0.5, 0.3]), atol=1e-5)          # Lambert weight = albedo
    # evaluation: zero with the bit, Lambert without
    wi = np.tile(np.array([0.0, 0.0, 1.0], np.float32), (8, 1))
    ein = np.concatenate([pos[:8], normal[:8], wo[:8], wi], axis=1)
    assert np.all(ol.eval_bsdf(mat, s, ein) == 0.0)
    assert np.allclose(ol.eval_bsdf(mat, s0, ein)[:, :3], np.array([0.8, 0.5, 0.3]) / np.pi, rtol=1e-5)
    # a material that asks for the random walk, or sssMode off, falls back to Lambert (two draws) even with the bit
    walker = host.desc.materials[1]
    outw, stw = ol.sample_bsdf(walker, s, inp, front, states)
    assert np.array_equal(stw, st0) and np.allclose(outw[:, 3:6], out0[:, 3:6])
    s_off = s.copy()
    s_off.sssMode = 0
    outo, sto = ol.sample_bsdf(mat, s_off, inp, front, states)
    assert np.array_equal(sto, st0)
    # image level: the separable sphere differs from the Lambert one and stays finite
    osc = ol.OracleScene(host)
    img_l, _, _ = osc.render(s0, 32, threads=0)
    img_s, _, _ = osc.render(s, 32, threads=0)
    assert np.isfinite(img_s).all() and img_s.min() >= 0
    assert np.sqrt(np.mean((img_l - img_s) ** 2)) > 0.01


def test_metal_random_walk_subsurface(tmp_path):
    """PTR_METAL_SSS with sssMode = 2 (shaders/pathtrace.metal:4060-4311, 6650-6676): coat lobe or a walk of closest-hit
    queries.  As written, a walk leaves its medium only where the geometric normal faces the ray, so inside a closed sphere
    it is reflected back until the throughput cutoff or the step limit and the bounce falls back to the Lambert sample."""
    p = tmp_path / "walk.scene"
    p.write_text(SSS_SCENE % "randomwalk" + "sphere center=3,0,0 radius=1 material=1\n")
    host = pt.HostScene.load(str(p))
    osc = ol.OracleScene(host)
    s = host.settings_for(metalSemantics=16, fireflyClampEnabled=0)
    assert s.sssMode == 2 and s.sssMaxSteps == 32
    img, _, c32 = osc.render(s, 16, threads=0, count=True)
    assert np.isfinite(img).all() and img.min() >= 0
    # render is a pure function of (seed, pixel, sample) with the walk in it, too
    img_b, _, _ = osc.render(s, 16, threads=3, count=True)
    assert np.array_equal(img, img_b)
    # the step limit bounds the queries of a walk: fewer closest-hit rays with sssMaxSteps = 2 than with 32
    s2 = s.copy()
    s2.sssMaxSteps = 2
    _, _, c2 = osc.render(s2, 16, threads=0, count=True)
    assert c32["extendRays"] > c2["extendRays"]
    # only materials that ask for the walk take it: with the walker material off screen the frame matches separable-off Lambert
    s_off = s.copy()
    s_off.metalSemantics = 0
    base, _, c0 = osc.render(s_off, 16, threads=0, count=True)
    assert c32["extendRays"] > c0["extendRays"]
    assert np.sqrt(np.mean((img - base) ** 2)) > 0.005


def _pbr_material(transmission=0.6, metallic=0.2, roughness=0.4):
    host = pt.HostScene.load(os.path.join(ROOT, "scenes", "helmet_env.scene"), os.path.join(ROOT, "scenes"))
    m = host.desc.materials[1]
    assert int(m.typeEta[0]) == 7
    m.baseColorRoughness[0], m.baseColorRoughness[1], m.baseColorRoughness[2], m.baseColorRoughness[3] = 0.8, 0.6, 0.3, roughness
    m.pbrParams[0] = metallic
    m.pbrExtras[2] = transmission
    m.typeEta[1] = 1.45          # ior
    m.typeEta[3] = 0.5           # thickness (KHR_materials_volume) -> transmission tint through sigmaA
    m.dielectricSigmaA[0], m.dielectricSigmaA[1], m.dielectricSigmaA[2] = 0.2, 0.5, 1.0
    return host, m


def test_metal_pbr_model_sample_and_eval_agree():
    """PTR_METAL_PBR (shaders/pathtrace.metal:4598-4948): specular / diffuse / transmission lobes picked by weight; the
    delta limits (roughness <= 1e-3) mirror and refract by Snell's law; transmitted weights carry the thickness tint;
    without the bit type 7 is the Embree-path two-lobe model that never crosses the surface."""
    host, m = _pbr_material()
    s = host.settings_for(metalSemantics=32, fireflyClampEnabled=0)
    s0 = host.settings_for(metalSemantics=0, fireflyClampEnabled=0)
    rng = np.random.default_rng(21)
    n = 20000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.1
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    normal = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    pos = np.zeros((n, 3), np.float32)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    out, _ = ol.sample_bsdf(m, s, np.concatenate([pos, normal, wo], axis=1), front, states)
    ok = out[:, 6] > 0
    assert ok.mean() > 0.85
    d = out[ok, :3]
    below = d[:, 2] < 0
    assert 0.25 < below.mean() < 0.65                       # the transmission lobe is taken about wTrans / sum of the time
    ev = ol.eval_bsdf(m, s, np.concatenate([pos[ok], normal[ok], wo[ok], d], axis=1))
    # a sample reports the pdf of the lobe it took (pLobe x pdfLobe), the evaluation the sum over the reflection lobes: the
    # sampled pdf never exceeds the evaluated one above the surface, and equals it where only one lobe has density
    above = ~below
    ratio = out[ok, 6][above] / np.maximum(ev[above, 3], 1e-30)
    assert (ratio <= 1.0 + 1e-3).all() and np.median(ratio) > 0.5
    # below the surface the reference evaluates with the half vector normalize(wo + wi * eta) while the sampler refracted about
    # the half vector it drew (eta = etaI / etaT in both, :4722 vs :4893): the two do not describe the same microfacet, so the
    # evaluated pdf is not the sampler's - restated as written, checked here only for being finite and non-negative
    assert np.isfinite(ev[below]).all() and (ev[below, :4] >= 0).all()
    assert np.isfinite(out[ok, 3:7]).all() and (out[ok, 3:6] >= 0).all()
    # transmitted weights carry the thickness tint: blue is absorbed most
    wt = out[ok][below][:, 3:6].astype(np.float64).mean(axis=0)
    assert wt[0] > wt[1] > wt[2] > 0
    # without the bit: the Embree-path model never goes below the surface
    out0, _ = ol.sample_bsdf(m, s0, np.concatenate([pos, normal, wo], axis=1), front, states)
    assert (out0[out0[:, 6] > 0, 2] > 0).all()
    # delta limits
    host2, m2 = _pbr_material(roughness=0.0)
    outd, _ = ol.sample_bsdf(m2, s, np.concatenate([pos, normal, wo], axis=1), front, states)
    okd = outd[:, 6] > 0
    delta = outd[:, 7] == 1.0                                # the diffuse lobe stays non-delta
    assert 0.5 < delta[okd].mean() < 1.0 and (outd[okd & ~delta, 2] > 0).all()
    up = okd & delta & (outd[:, 2] > 0)
    assert np.allclose(outd[up, :2], -wo[up, :2], atol=1e-5) and np.allclose(outd[up, 2], wo[up, 2], atol=1e-5)      # mirror
    dn = okd & (outd[:, 2] < 0)
    assert dn.sum() > 1000 and delta[dn].all()
    sin_o = np.linalg.norm(wo[dn, :2], axis=1)
    sin_t = np.linalg.norm(outd[dn, :2], axis=1)
    assert np.allclose(sin_t * 1.45, sin_o, atol=1e-4)                                                               # Snell


def test_metal_clamp_variants_restated():
    """PTR_METAL_CLAMPS (shaders/pathtrace.metal:3550-3633) on the oracle side: with the default
    settings (tail base = scale = 0, minSpecularPdf = 0, maxContribution = 1000) the Embree variants cap a smooth metal lobe's
    luminance at the clamp floor (4) while the Metal variants leave it alone."""
    host = pt.HostScene.load(os.path.join(GOLDEN, "materials.scene"))
    d = host.desc
    metal = next(i for i in range(d.materialCount) if int(d.materials[i].typeEta[0]) == 1 and 0.05 < d.materials[i].baseColorRoughness[3] < 0.3)
    s0 = host.settings_for(width=32, height=32)
    s1 = s0.copy()
    s1.metalSemantics = 64
    wo = np.array([0.3, 0.1, 0.95])
    wo /= np.linalg.norm(wo)
    wi = wo * np.array([-1, -1, 1])          # the mirror direction: the peak of the lobe
    inputs = np.concatenate([np.zeros(3), [0, 0, 1.0], wo, wi]).astype(np.float32)[None]
    capped = ol.eval_bsdf(d.materials[metal], s0, inputs)[0]
    free = ol.eval_bsdf(d.materials[metal], s1, inputs)[0]
    lum = lambda v: 0.2126 * v[0] + 0.7152 * v[1] + 0.0722 * v[2]
    assert lum(capped[:3]) <= 4.0 + 1e-4 < lum(free[:3])
    assert free[3] >= capped[3] > 0                 # pdf: passed through / floored at 1e-8
    # the firefly limit: a 500-luminance contribution survives under the Metal variant only
    img0, _, _ = ol.OracleScene(host).render(s0, 8, threads=2)
    img1, _, _ = ol.OracleScene(host).render(s1, 8, threads=2)
    assert img1.mean() >= img0.mean()
