"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same seeded inputs.

Tolerances (floating-point path; north star: "within the repo's own RMSE threshold on linear PFM"; the reference
publishes no number, so SURVEY.md section 8(d) defines them):
  * ray queries:   t within max(1e-3, 1e-4*t) and same primitive except ties — the reference's own HWRT/SWRT
                   parity thresholds (shaders/pathtrace.metal:6813-6837).  In practice t is bit-identical.
  * device functions (camera, BSDF eval/sample): relative 2e-4 (libm vs ocml sin/cos/exp differ in the last ulps),
                   RNG state after sampling identical (same number of draws).
  * images:        deterministic-stream check at low spp (fraction of pixels within 1e-3 relative), and
                   RMSE(build, oracle) <= 1.25 * N with N = RMSE(oracle seed 1337, oracle seed 1338), mean ratio 0.5 %.
"""
import importlib
import os

import numpy as np
import pytest

import oracle_lib as ol

pt = importlib.import_module("metal-pathtracer-arm64_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")

pytestmark = pytest.mark.gpu

# worker threads of the CPU oracle (test infrastructure): bounded, a GPU box gives one card's share of its host cores
ORACLE_THREADS = min(32, os.cpu_count() or 1)


def _rel(a, b):
    return np.abs(a - b) / (np.abs(b) + 1e-2)


def _rmse(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def _random_rays(n, lo, hi, seed, finite_fraction=0.3):
    rng = np.random.default_rng(seed)
    org = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.random(n) < finite_fraction, rng.uniform(0.1 * (hi - lo), hi - lo, n), np.inf).astype(np.float32)
    return np.concatenate([org, np.full((n, 1), 1e-4, np.float32), d, tmax[:, None]], axis=1).astype(np.float32)


@pytest.fixture(scope="module")
def cornell_small():
    host = pt.HostScene.load(os.path.join(GOLDEN, "cornell_small_mesh.scene"), SCENES)
    return host, pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)


@pytest.fixture(scope="module")
def materials_scene():
    host = pt.HostScene.load(os.path.join(GOLDEN, "materials.scene"))
    return host, pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)


# --------------------------------------------------------------------------- ray level
@pytest.mark.parametrize("any_hit", [False, True])
def test_ray_queries_match_oracle(cornell_small, any_hit):
    host, dev, osc = cornell_small
    rays = _random_rays(50000, 5.0, 550.0, 11)
    g, stats = dev.trace_rays(rays, any_hit=any_hit)
    o = osc.trace_rays(rays, any_hit=any_hit)
    assert np.array_equal(g["t"] >= 0, o["t"] >= 0)
    if not any_hit:
        hit = o["t"] >= 0
        assert hit.mean() > 0.5
        tol = np.maximum(1e-3, 1e-4 * o["t"][hit])
        assert (np.abs(g["t"][hit] - o["t"][hit]) <= tol).all()
        assert np.array_equal(g["t"], o["t"])                       # in fact bit-identical
        same = (g["primType"][hit] == o["primType"][hit]) & (g["primIndex"][hit] == o["primIndex"][hit]) & \
               (g["geomIndex"][hit] == o["geomIndex"][hit])
        assert same.mean() > 0.999                                  # ties on shared edges may pick the other triangle
        assert np.allclose(g["u"][hit][same], o["u"][hit][same], atol=1e-6)
        tri = same & (o["primType"][hit] != 1)                      # spheres: the integrator recomputes the normal
        assert np.allclose(g["ng"][hit][tri], o["ng"][hit][tri], rtol=1e-6, atol=1e-6)
    assert stats.nodesVisited > 0 and stats.leafPrimTests > 0


def test_ray_queries_spheres_and_empty_and_degenerate(materials_scene):
    host, dev, osc = materials_scene
    rays = _random_rays(20000, -6.0, 6.0, 5)
    rays[:, 1] = np.abs(rays[:, 1]) + 0.05
    g, _ = dev.trace_rays(rays)
    o = osc.trace_rays(rays)
    assert np.array_equal(g["t"], o["t"]) and np.array_equal(g["primType"], o["primType"]) and np.array_equal(g["primIndex"], o["primIndex"])
    assert (o["primType"][o["t"] >= 0] == 1).any() and (o["primType"][o["t"] >= 0] == 2).any()
    # empty batch and axis-aligned directions (zero components -> infinite reciprocals)
    e, _ = dev.trace_rays(np.zeros((0, 8), np.float32))
    assert e.shape == (0,)
    axis = np.array([[0, 5, 0, 1e-4, 0, -1, 0, np.inf], [-3.6, 0.6, 5, 1e-4, 0, 0, -1, np.inf], [50, 0.6, 0, 1e-4, -1, 0, 0, np.inf]], np.float32)
    ga, _ = dev.trace_rays(axis)
    oa = osc.trace_rays(axis)
    assert np.array_equal(ga["t"], oa["t"]) and (ga["t"] > 0).all()


def test_large_mesh_bvh_depth_and_parity():
    host = pt.HostScene.load(os.path.join(SCENES, "cornell_mesh.scene"), SCENES)
    dev = pt.DeviceScene(host.desc, 0, keepalive=host)
    info = dev.info()
    assert info["triangles"] == 70688 + 12 and info["max_depth"] < 48 and info["max_leaf"] <= 4 and info["rect_lights"] == 1
    osc = ol.OracleScene(host)
    rays = _random_rays(30000, 5.0, 550.0, 3, finite_fraction=0.0)
    g, stats = dev.trace_rays(rays)
    o = osc.trace_rays(rays)
    assert np.array_equal(g["t"], o["t"])
    assert 5 < stats.nodesVisited / len(rays) < 200


# --------------------------------------------------------------------------- device functions
def test_camera_rays_match_oracle():
    host = pt.HostScene.load(os.path.join(SCENES, "cornell.scene"))
    for defocus in (0.0, 2.5):
        s = host.settings_for(width=320, height=200, cameraDefocusAngle=defocus, cameraFocusDistance=900.0)
        rng = np.random.default_rng(0)
        xys = np.stack([rng.integers(0, 320, 4096), rng.integers(0, 200, 4096), rng.integers(0, 64, 4096)], axis=1).astype(np.uint32)
        g, gs = pt.debug_camera_rays(s, xys)
        o, os_ = ol.camera_rays(s, xys)
        assert np.array_equal(gs, os_)                              # same number of draws (lens rejection loop included)
        assert np.array_equal(g[:, :3], o[:, :3]) or np.allclose(g[:, :3], o[:, :3], rtol=1e-6, atol=1e-4)
        assert np.allclose(g[:, 3:], o[:, 3:], atol=2e-6)


def _bsdf_inputs(n, seed):
    rng = np.random.default_rng(seed)

    def hemi(k, zmin):
        v = rng.normal(size=(k, 3))
        v[:, 2] = np.abs(v[:, 2]) + zmin
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    normal = np.tile(np.array([0, 0, 1], np.float32), (n, 1))
    pos = rng.uniform(-2, 2, size=(n, 3)).astype(np.float32)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    return pos, normal, hemi(n, 0.05), hemi(n, 0.05), states


def _all_materials(host):
    mats = [(int(host.desc.materials[i].typeEta[0]), host.desc.materials[i]) for i in range(host.desc.materialCount)]
    pbr = []
    for metallic, rough in ((0.0, 0.6), (1.0, 0.3), (0.5, 0.0)):
        m = pt.PtrMaterial.from_buffer_copy(bytes(host.desc.materials[0]))
        m.typeEta[0] = 7.0
        m.typeEta[1] = 1.5
        m.baseColorRoughness[3] = rough
        m.pbrParams[0] = metallic
        pbr.append((7, m))
    return mats + pbr


def test_eval_bsdf_matches_oracle(materials_scene):
    host, _, _ = materials_scene
    s = host.settings_for(width=16, height=16)
    pos, normal, wo, wi, _ = _bsdf_inputs(4096, 21)
    inp = np.concatenate([pos, normal, wo, wi], axis=1)
    for mtype, mat in _all_materials(host):
        g = pt.debug_eval_bsdf(mat, s, inp)
        o = ol.eval_bsdf(mat, s, inp)
        assert np.array_equal(g[:, 4], o[:, 4]), mtype
        assert np.allclose(g[:, :4], o[:, :4], rtol=2e-4, atol=1e-6), mtype


def test_sample_bsdf_matches_oracle(materials_scene):
    host, _, _ = materials_scene
    s = host.settings_for(width=16, height=16)
    pos, normal, wo, _, states = _bsdf_inputs(4096, 22)
    inp = np.concatenate([pos, normal, wo], axis=1)
    for front_value in (1, 0):
        front = np.full(len(pos), front_value, np.uint32)
        for mtype, mat in _all_materials(host):
            g, gs = pt.debug_sample_bsdf(mat, s, inp, front, states)
            o, os_ = ol.sample_bsdf(mat, s, inp, front, states)
            assert np.array_equal(gs, os_), mtype                   # identical RNG consumption
            # a sin/cos ulp can flip a discrete choice (valid <-> rejected) in a handful of samples
            agree = (g[:, 6] > 0) == (o[:, 6] > 0)
            assert agree.mean() > 0.998, mtype
            both = agree & (o[:, 6] > 0)
            close = np.isclose(g[both], o[both], rtol=5e-4, atol=2e-5).all(axis=1)
            # narrow GGX lobes (car-paint flakes: alpha ~ 0.02) amplify last-ulp differences of the half vector
            assert close.mean() > (0.985 if mtype == 6 else 0.995), (mtype, close.mean())
            # (car-paint clear coat: roughness 0.04 -> alpha^2 = 2.6e-6, D is ill-conditioned in f32 on both sides)
            assert np.isclose(g[both], o[both], rtol=5e-2, atol=1e-3).all(axis=1).mean() > (0.99 if mtype == 6 else 0.999), mtype


# --------------------------------------------------------------------------- image level
def _image_parity(host, dev, osc, width, height, depth, low_spp, high_spp, min_fraction, **overrides):
    s = host.settings_for(width=width, height=height, max_depth=depth, seed=1337, **overrides)
    img1, st1 = dev.render_image(s, low_spp, count=True)
    ref1, _, c1 = osc.render(s, low_spp, threads=0, count=True)
    assert img1.shape == ref1.shape and np.isfinite(img1).all() and img1.min() >= 0
    frac = float((_rel(img1, ref1).max(axis=2) <= 1e-3).mean())
    # same stream => (almost) the same number of rays, hits and mesh hits
    assert abs(st1.extendRays - c1["extendRays"]) <= 0.002 * c1["extendRays"] + 2
    assert abs(st1.shadedHits - c1["shadedHits"]) <= 0.002 * c1["shadedHits"] + 2
    if os.environ.get("PTR_TEST_VERBOSE"):   # what the thresholds were set from: measured fraction per call site (minus a 0.01 margin)
        import inspect
        caller = inspect.stack()[1]
        with open(os.path.join(ROOT, "gpurun_out", "parity_fractions.txt"), "a") as f:
            f.write("%s:%d  %dx%d d%d  %s  fraction %.4f  (threshold %.3f)\n" % (caller.function, caller.lineno, width, height, depth, overrides, frac, min_fraction))
    assert frac >= min_fraction, frac
    imgN, _ = dev.render_image(s, high_spp)
    refN, _, _ = osc.render(s, high_spp, threads=0)
    s2 = s.copy()
    s2.seed = 1338
    refM, _, _ = osc.render(s2, high_spp, threads=0)
    noise = _rmse(refN, refM)
    err = _rmse(imgN, refN)
    lum = np.array([0.2126, 0.7152, 0.0722])
    ratio = float((imgN @ lum).mean() / (refN @ lum).mean())
    assert err <= 1.25 * noise, (err, noise)
    assert abs(ratio - 1.0) <= 0.005, ratio
    return frac, err, noise, ratio


def test_smoke_scene_image_parity():
    # the reference's smoke render (64x64, 4 spp, depth 4, seed 1337): no lights, solid background
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    frac, err, noise, _ = _image_parity(host, dev, osc, 64, 64, 4, 4, 64, 0.99)
    assert err < 0.25 * noise            # far below the noise floor: the streams are the same


def test_cornell_image_parity(cornell_small):
    host, dev, osc = cornell_small
    # (Cornell-type scenes: every differing pixel is a marginal rectangle-light shadow decision, quirk Q9 - see
    # test_deterministic_stream_criterion_and_what_the_rest_comes_from; thresholds = measured fraction - 0.01)
    _image_parity(host, dev, osc, 64, 64, 4, 1, 64, 0.966)
    _image_parity(host, dev, osc, 80, 48, 8, 1, 32, 0.959)


def test_materials_image_parity_with_and_without_specular_nee(materials_scene):
    host, dev, osc = materials_scene
    _image_parity(host, dev, osc, 96, 64, 6, 1, 32, 0.987)
    _image_parity(host, dev, osc, 96, 64, 6, 1, 16, 0.987, enableSpecularNee=0, enableRussianRoulette=0)
    _image_parity(host, dev, osc, 96, 64, 12, 1, 16, 0.988, fireflyClampEnabled=0)


def test_environment_lit_image_parity():
    # env NEE (alias tables), MIS against the env pdf, bilinear lookups, specular NEE along delta bounces, env portal
    host = pt.HostScene.load(os.path.join(GOLDEN, "env_materials.scene"), SCENES)
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    _image_parity(host, dev, osc, 96, 64, 6, 1, 32, 0.989)
    _image_parity(host, dev, osc, 64, 48, 4, 1, 16, 0.99, environmentRotation=2.1, environmentIntensity=0.5, enableSpecularNee=0)


def test_config3_standin_glb_under_hdr_environment():
    # BASELINE config 3 stand-in at reduced resolution: GLB import (3 PBR primitives incl. an emissive one),
    # 2048x1024 RGBE environment with three suns of very different radiance (alias tables), envRotation 30 deg
    host = pt.HostScene.load(os.path.join(SCENES, "helmet_env.scene"), SCENES)
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    _image_parity(host, dev, osc, 192, 108, 8, 1, 32, 0.99)


def test_config4_standin_glass_knot_depth16():
    # BASELINE config 4 stand-in: 871,200-triangle dielectric torus knot in the Cornell box, depth 16, RR on
    from scenes.gen_assets import ensure_large_asset
    ensure_large_asset("torus_knot_871200.ply")
    host = pt.HostScene.load(os.path.join(SCENES, "knot_glass.scene"), SCENES)
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    info = dev.info()
    assert info["triangles"] == 871200 + 12 and info["max_depth"] < 48
    if os.environ.get("PTR_TEST_VERBOSE"):
        print("knot scene", info)
    _image_parity(host, dev, osc, 160, 90, 16, 1, 32, 0.97)
    rays = _random_rays(200_000, -50.0, 600.0, 11)
    hits = dev.trace_rays(rays, any_hit=False)[0]
    ref = osc.trace_rays(rays, any_hit=False)
    assert np.array_equal(hits["t"], ref["t"]) and (ref["t"] >= 0).mean() > 0.3   # bit-identical distances, 0.87 M triangles


def test_config5_reduced_sss_and_carpaint_meshes():
    from scenes.gen_assets import ensure_large_asset
    ensure_large_asset("blob_125000.ply")
    host = pt.HostScene.load(os.path.join(GOLDEN, "lucy_small.scene"), SCENES)
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    _image_parity(host, dev, osc, 160, 90, 12, 1, 32, 0.989)


def test_material_set_instantiations_render_what_the_full_kernel_renders(tmp_path):
    # k_shade is compiled per material / feature set (csrc/kernels/wavefront.hip shadeKernelSet); the counting render always runs
    # the full kernel, so the two images of one scene must be the same bits whichever set the scene selects
    from scenes.gen_assets import ensure_large_asset
    ensure_large_asset("torus_knot_871200.ply")
    ensure_large_asset("blob_125000.ply")
    metals = tmp_path / "metals.scene"
    metals.write_text("""camera target=0,0.6,0 distance=9 yaw=1.2 pitch=0.25 vfov=35
renderer width=96 height=64 maxDepth=6 seed=1337
background solid=0.1,0.1,0.15
material type=lambert albedo=0.8,0.3,0.3 name=lambert
material type=metal albedo=0.9,0.8,0.5 roughness=0.35 name=rough_metal
material type=metal albedo=0.95,0.95,0.95 roughness=0.0 name=mirror
material type=metal eta=0.2,0.9,1.1 k=3.9,2.4,2.2 roughness=0.2 name=gold_like
material type=dielectric ior=1.5 name=glass
material type=lambert albedo=0.5,0.5,0.5 name=floor
material type=diffuse_light emit=9,9,9 name=light
sphere center=-2.4,0.6,0 radius=0.6 material=0
sphere center=-1.2,0.6,0 radius=0.6 material=1
sphere center=0,0.6,0 radius=0.6 material=2
sphere center=1.2,0.6,0 radius=0.6 material=3
sphere center=2.4,0.6,0 radius=0.6 material=4
rectangle x=-8,8 y=0 z=-8,8 normal=1 material=5
rectangle x=-2,2 y=5 z=-2,2 normal=-1 material=6
""")
    D, G, M, CP, P7 = 0b101001, 0b101101, 0b101111, 0b1101101, 0b10101101
    cases = [(os.path.join(GOLDEN, "cornell_small_mesh.scene"), D, 6), (os.path.join(SCENES, "knot_glass.scene"), G, 12),
             (str(metals), M, 6), (os.path.join(GOLDEN, "lucy_small.scene"), CP, 8), (os.path.join(SCENES, "helmet_env.scene"), P7 | 0x100, 6),
             (os.path.join(GOLDEN, "materials.scene"), 0x3FF, 6), (os.path.join(GOLDEN, "env_materials.scene"), 0x3FF, 6)]
    for path, expected, depth in cases:
        host = pt.HostScene.load(path, SCENES)
        dev = pt.DeviceScene(host.desc, 0, keepalive=host)
        s = host.settings_for(width=160, height=96, max_depth=depth, seed=77)
        assert dev.shade_kernel_set(s) == expected, (path, bin(dev.shade_kernel_set(s)))
        assert dev.shade_kernel_set(s, count=True) == 0x3FF
        lean, _ = dev.render_image(s, 8)
        full, st = dev.render_image(s, 8, count=True)
        assert np.array_equal(lean.view(np.uint32), full.view(np.uint32)), path
        assert st.shadedHits > 0
        # the Metal medium rules are a feature of the set: with them the scene falls back to a set that has them
        s2 = host.settings_for(width=64, height=48, max_depth=depth, seed=77, metalSemantics=5)   # PTR_METAL_MEDIA | PTR_METAL_FACE_NORMAL
        assert dev.shade_kernel_set(s2) == 0x3FF
        lean2, _ = dev.render_image(s2, 4)
        full2, _ = dev.render_image(s2, 4, count=True)
        assert np.array_equal(lean2.view(np.uint32), full2.view(np.uint32)), path
        dev.close()


def test_metal_media_semantics(tmp_path):
    # Metal-only integrator semantics (PtrSettings.metalSemantics): Beer-Lambert media with the 8-deep stack, thin-walled
    # glass, ray-facing glass normals.  Checked against the oracle's restatement of shaders/pathtrace.metal:1187-1191,
    # 5649-5683, 5869-5876, 6694-6709 and against the closed form for a slab.
    text = ("camera target=0,1,0 distance=9 yaw=1.1 pitch=0.3 vfov=38\nrenderer maxDepth=12 seed=9\nbackground solid=0.6,0.7,0.9\n"
            "material type=lambert albedo=0.7,0.7,0.7\n"
            "material type=dielectric ior=1.5 sigmaA=0.9,0.25,0.05 name=amber\n"
            "material type=dielectric ior=1.33 sigmaA=0.05,0.3,0.6 name=water\n"
            "material type=dielectric ior=1.5 sigmaA=2,2,2 thin=1 name=pane\n"
            "material type=light emit=12,11,10\n"
            "rectangle x=-8,8 y=0 z=-8,8 normal=1 material=0\n"
            "sphere center=-1.6,1,0 radius=1 material=1\n"
            "sphere center=-1.6,1,0 radius=0.55 material=2\n"          # nested media: water inside amber glass
            "box min=0.4,0,-0.8 max=2.0,1.6,0.8 material=1\n"
            "rectangle x=-3,3 y=0,3 z=2.2 normal=-1 material=3 twoSided=1\n"     # thin pane between the camera... and
            "rectangle x=-1,1 y=6 z=-1,1 normal=-1 material=4\n")
    p = tmp_path / "media.scene"
    p.write_text(text)
    host = pt.HostScene.load(str(p))
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    for sem in (7, 5, 4):
        _image_parity(host, dev, osc, 96, 64, 12, 1, 32, 0.989, metalSemantics=sem, enableRussianRoulette=0)
    # the modes really differ from the Embree-parity image and from each other
    s0 = host.settings_for(width=96, height=64, max_depth=12, seed=9)
    base, _ = dev.render_image(s0, 16)
    imgs = []
    for sem in (4, 5, 7):
        s = s0.copy()
        s.metalSemantics = sem
        imgs.append(dev.render_image(s, 16)[0])
    assert _rmse(base, imgs[0]) > 0.01 and _rmse(imgs[0], imgs[1]) > 0.005 and _rmse(imgs[1], imgs[2]) > 0.002
    # closed form: straight through a 2-unit slab, radiance = (1-Fr)^4 exp(-2 sigma) + Fr^2
    slab = tmp_path / "slab.scene"
    slab.write_text("camera target=0,0,0 distance=10 yaw=1.5708 pitch=0 vfov=2\nrenderer width=16 height=16 maxDepth=12 seed=5 russianRoulette=0\n"
                    "background solid=1,1,1\nmaterial type=dielectric ior=1.5 sigmaA=0.30,0.10,0.02\nbox min=-4,-4,-1 max=4,4,1 material=0\n")
    sh = pt.HostScene.load(str(slab))
    sdev = pt.DeviceScene(sh.desc, 0, keepalive=sh)
    ss = sh.settings_for(seed=5, metalSemantics=5, fireflyClampEnabled=0)
    centre = sdev.render_image(ss, 256)[0][6:10, 6:10].reshape(-1, 3).mean(axis=0)
    fr = 0.04
    assert np.allclose(centre, (1 - fr) ** 4 * np.exp(-2.0 * np.array([0.30, 0.10, 0.02])) + fr * fr, rtol=0.02), centre


def test_metal_specular_semantics(materials_scene):
    # PTR_METAL_SPECULAR: rough metals with VNDF sampling, the G1 pdf and energy compensation (Metal formulas, restated in
    # the oracle too): device functions and images agree with the oracle, and differ from the Embree-parity mode
    host, dev, osc = materials_scene
    _image_parity(host, dev, osc, 96, 64, 6, 1, 32, 0.987, metalSemantics=8)
    s0 = host.settings_for(width=96, height=64, max_depth=6, seed=1337)
    s8 = s0.copy()
    s8.metalSemantics = 8
    assert _rmse(dev.render_image(s0, 32)[0], dev.render_image(s8, 32)[0]) > 0.005
    d = host.desc
    rough = [i for i in range(d.materialCount) if int(d.materials[i].typeEta[0]) == 1 and d.materials[i].baseColorRoughness[3] > 0.05]
    assert rough
    rng = np.random.default_rng(3)
    n = 3000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.1
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    inp = np.concatenate([np.zeros((n, 3), np.float32), np.tile(np.array([0, 0, 1], np.float32), (n, 1)), wo], axis=1)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    g, gs = pt.debug_sample_bsdf(d.materials[rough[0]], s8, inp, front, states)
    o, os_ = ol.sample_bsdf(d.materials[rough[0]], s8, inp, front, states)
    assert np.array_equal(gs, os_)
    both = (g[:, 6] > 0) & (o[:, 6] > 0)
    assert both.mean() > 0.75 and np.array_equal(g[:, 6] > 0, o[:, 6] > 0)
    assert np.allclose(g[both, :3], o[both, :3], atol=2e-4) and np.allclose(g[both, 3:7], o[both, 3:7], rtol=2e-3, atol=1e-4)


def test_metal_subsurface_semantics(tmp_path):
    # PTR_METAL_SSS: type 5 evaluates to zero (no NEE) and, with `renderer sss=separable`, is sampled with the separable
    # diffusion profile - exit point on the tangent plane, biased next-ray origin (shaders/pathtrace.metal:3916-3994,
    # 5398-5507, 6740-6766; restated in the oracle too).  Device functions and images agree with the oracle and differ
    # from the Embree-parity mode; so does the random-walk mode (shaders/pathtrace.metal:4060-4311, 6650-6676).
    scene = ("camera target=0,0.2,0 distance=7 yaw=0.8 pitch=0.35 vfov=30\n"
             "renderer width=96 height=64 maxDepth=6 seed=1337 sss=%s\n"
             "background solid=0.5,0.6,0.8\n"
             "material type=lambert albedo=0.7,0.7,0.7 name=floor\n"
             "material type=sss albedo=0.8,0.5,0.3 mfp=0.3%s name=skin\n"
             "material type=sss albedo=0.3,0.6,0.8 mfp=0.15 g=0.3 sigma_a=0.4,0.2,0.1 sigma_s=3.0,4.0,5.0 coat=on name=jade\n"
             "material type=diffuse_light emit=12,11,10 name=lamp\n"
             "rectangle x=-6,6 y=-1 z=-6,6 normal=1 material=0\n"
             "rectangle x=-1,1 y=4 z=-1,1 normal=-1 material=3\n"
             "sphere center=-1.1,0,0 radius=1 material=1\n"
             "sphere center=1.1,0,0 radius=1 material=2\n")
    p = tmp_path / "sss.scene"
    p.write_text(scene % ("separable", ""))
    host = pt.HostScene.load(str(p))
    assert host.desc.materials[2].sssParams[2] == 1.0 and host.desc.materials[2].sssSigmaA[3] == 1.0     # coat on, sigma override
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    s0 = host.settings_for(seed=1337)
    assert s0.sssMode == 1
    _image_parity(host, dev, osc, 96, 64, 6, 1, 32, 0.99, metalSemantics=16)
    s16 = s0.copy()
    s16.metalSemantics = 16
    assert _rmse(dev.render_image(s0, 32)[0], dev.render_image(s16, 32)[0]) > 0.005
    d = host.desc
    rng = np.random.default_rng(9)
    n = 4000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.1
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    inp = np.concatenate([rng.normal(size=(n, 3)).astype(np.float32), np.tile(np.array([0, 0, 1], np.float32), (n, 1)), wo], axis=1)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    for mi in (1, 2):
        g, gs = pt.debug_sample_bsdf(d.materials[mi], s16, inp, front, states)
        o, os_ = ol.sample_bsdf(d.materials[mi], s16, inp, front, states)
        assert np.array_equal(gs, os_)                              # four draws on both sides
        assert np.array_equal(g[:, 6] > 0, o[:, 6] > 0) and (o[:, 6] > 0).mean() > 0.99
        assert np.allclose(g[:, :3], o[:, :3], atol=2e-4) and np.allclose(g[:, 3:7], o[:, 3:7], rtol=3e-3, atol=1e-5)
        wi = g[:, :3]
        ev = pt.debug_eval_bsdf(d.materials[mi], s16, np.concatenate([inp, wi], axis=1))
        assert np.all(ev[:, :4] == 0.0)
    # random walk (sssMode 2 on a material that asks for it): every boundary query of the walk is one extend/shade iteration of
    # the wavefront, the walk state rides in the slot.  The small sphere inside the walker is what lets a walk leave its
    # medium at all: the reference takes the exit only where the geometric normal faces the ray (see bsdf.h: sssWalkStep).
    p.write_text((scene % ("randomwalk", " method=randomwalk")) + "sphere center=-1.1,0,0 radius=0.35 material=0\n")
    wh = pt.HostScene.load(str(p))
    wdev, wosc = pt.DeviceScene(wh.desc, 0, keepalive=wh), ol.OracleScene(wh)
    sw = wh.settings_for(seed=1337, metalSemantics=16)
    assert sw.sssMode == 2 and sw.sssMaxSteps == 32
    _image_parity(wh, wdev, wosc, 96, 64, 6, 1, 32, 0.988, metalSemantics=16)
    _image_parity(wh, wdev, wosc, 64, 48, 6, 1, 16, 0.988, metalSemantics=16, sssMaxSteps=3, enableRussianRoulette=0)
    # the walk costs closest-hit queries: more rays than the same frame without it, and the image differs from separable mode
    _, st_walk = wdev.render_image(sw, 4, count=True)
    s_sep = sw.copy()
    s_sep.sssMode = 1
    img_sep, st_sep = wdev.render_image(s_sep, 4, count=True)
    assert st_walk.extendRays > 1.2 * st_sep.extendRays
    sw0 = sw.copy()
    sw0.metalSemantics = 0                                           # Embree-parity mode ignores the setting
    assert np.isfinite(wdev.render_image(sw0, 1)[0]).all()


def test_metal_pbr_semantics():
    # PTR_METAL_PBR: the three-lobe metallic-roughness model of the Metal integrator (shaders/pathtrace.metal:4598-4948;
    # restated in the oracle too) on the config-3 stand-in, with one of its primitives made transmissive and another a
    # perfect mirror-and-refractor (roughness 0: delta lobes, which then also make the surface delta for NEE)
    host = pt.HostScene.load(os.path.join(SCENES, "helmet_env.scene"), SCENES)
    d = host.desc
    glassy, sharp = d.materials[1], d.materials[2]
    glassy.pbrExtras[2], glassy.pbrParams[0], glassy.baseColorRoughness[3] = 0.7, 0.1, 0.35
    glassy.typeEta[1], glassy.typeEta[3] = 1.45, 0.5
    glassy.dielectricSigmaA[0], glassy.dielectricSigmaA[1], glassy.dielectricSigmaA[2] = 0.2, 0.5, 1.0
    sharp.pbrExtras[2], sharp.pbrParams[0], sharp.baseColorRoughness[3] = 0.5, 0.0, 0.0
    dev, osc = pt.DeviceScene(d, 0, keepalive=host), ol.OracleScene(host)
    _image_parity(host, dev, osc, 160, 90, 8, 1, 32, 0.99, metalSemantics=32)
    s0 = host.settings_for(width=160, height=90, max_depth=8, seed=1337)
    s32 = s0.copy()
    s32.metalSemantics = 32
    assert _rmse(dev.render_image(s0, 32)[0], dev.render_image(s32, 32)[0]) > 0.005
    rng = np.random.default_rng(4)
    n = 4000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.1
    wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
    inp = np.concatenate([np.zeros((n, 3), np.float32), np.tile(np.array([0, 0, 1], np.float32), (n, 1)), wo], axis=1)
    states = rng.integers(1, 2**32 - 1, size=n, dtype=np.uint64).astype(np.uint32)
    front = np.ones(n, dtype=np.uint32)
    for m in (glassy, sharp):
        g, gs = pt.debug_sample_bsdf(m, s32, inp, front, states)
        o, os_ = ol.sample_bsdf(m, s32, inp, front, states)
        assert np.array_equal(gs, os_)
        agree = (g[:, 6] > 0) == (o[:, 6] > 0)
        assert agree.mean() > 0.995
        both = agree & (o[:, 6] > 0)
        assert (o[both, 2] < 0).mean() > 0.2                                        # the transmission lobe is exercised
        assert np.allclose(g[both, :3], o[both, :3], atol=3e-4)
        close = np.isclose(g[both, 3:7], o[both, 3:7], rtol=5e-3, atol=1e-4).all(axis=1)
        assert close.mean() > 0.99, close.mean()                                    # rough refraction: ill-conditioned Jacobian in f32
        wi = o[both, :3]
        ge = pt.debug_eval_bsdf(m, s32, np.concatenate([inp[both], wi], axis=1))
        oe = ol.eval_bsdf(m, s32, np.concatenate([inp[both], wi], axis=1))
        ok = np.isclose(ge, oe, rtol=5e-3, atol=1e-5).all(axis=1)
        assert ok.mean() > 0.99, ok.mean()


def test_first_hit_aovs(materials_scene):
    # denoiser inputs: albedo = base colour of the first hit, normal = shading normal * 0.5 + 0.5, distance in normal.w
    host, dev, osc = materials_scene
    s = host.settings_for(width=96, height=64, max_depth=4, seed=1337, cameraDefocusAngle=0.0)
    albedo, normal = dev.render_aovs(s, 0)
    xs, ys = np.meshgrid(np.arange(96, dtype=np.uint32), np.arange(64, dtype=np.uint32))
    xys = np.stack([xs.ravel(), ys.ravel(), np.zeros(96 * 64, np.uint32)], axis=1)
    rays, _ = pt.debug_camera_rays(s, xys)                       # the same jittered camera rays, sample 0
    batch = np.concatenate([rays[:, :3], np.full((len(rays), 1), 1e-4, np.float32), rays[:, 3:], np.full((len(rays), 1), np.inf, np.float32)], axis=1)
    hits = osc.trace_rays(batch.astype(np.float32), any_hit=False)
    hit = (hits["t"] >= 0).reshape(64, 96)
    assert np.array_equal(albedo[..., 3] > 0.5, hit) and hit.mean() > 0.5
    assert np.array_equal(normal[..., 3][hit], hits["t"].reshape(64, 96)[hit])        # same closest hit, bit for bit
    assert np.allclose(normal[..., :3][~hit], 0.5) and (albedo[..., :3][~hit] == 0).all()
    n = normal[..., :3][hit] * 2.0 - 1.0
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-4)
    # every albedo is the base colour of one of the scene's materials
    d = host.desc
    table = np.array([list(d.materials[i].baseColorRoughness)[:3] for i in range(d.materialCount)], np.float32).clip(0, 1)
    dist = np.abs(albedo[..., :3][hit][:, None, :] - table[None]).max(axis=2).min(axis=1)
    assert dist.max() < 1e-6


def test_gradient_sky_and_thin_lens(materials_scene):
    host, dev, osc = materials_scene
    _image_parity(host, dev, osc, 64, 48, 5, 1, 16, 0.987, backgroundMode=0, cameraDefocusAngle=1.5, cameraFocusDistance=8.0)


def test_mnee_modes(tmp_path):
    text = ("camera target=0,1,0 distance=7 yaw=1.0 pitch=0.4 vfov=40\nrenderer maxDepth=6 seed=3 enableMnee=1\nbackground solid=0.05,0.05,0.08\n"
            "material type=lambert albedo=0.7,0.7,0.7\nmaterial type=dielectric ior=1.5\nmaterial type=light emit=20,18,15\n"
            "rectangle x=-6,6 y=0 z=-6,6 normal=1 material=0\nsphere center=0,1,0 radius=1 material=1\nsphere center=2.2,0.7,0.5 radius=0.7 material=1\n"
            "rectangle x=-1,1 y=5 z=-1,1 normal=-1 material=2\n")
    p = tmp_path / "mnee.scene"
    p.write_text(text)
    host = pt.HostScene.load(str(p))
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    _image_parity(host, dev, osc, 64, 48, 6, 1, 32, 0.99, enableMnee=1, enableMneeSecondary=1)
    _image_parity(host, dev, osc, 64, 48, 6, 1, 16, 0.99, enableMnee=1, enableMneeSecondary=0)


def test_specular_connections_settled_in_shade_and_traced(tmp_path):
    # A delta bounce connects to rectangle lights straight along its direction.  With up to 8 lights k_shade works out which light the
    # direction meets (the lights' own triangles) and queues an any-hit ray up to it, or none; with more it queues the reference's
    # closest-hit ray.  Both against the oracle, on the same glass-and-mirror scene lit by one light / by nine.
    head = ("camera target=0,1,0 distance=7 yaw=1.0 pitch=0.4 vfov=40\nrenderer maxDepth=6 seed=5\nbackground solid=0.02,0.02,0.03\n"
            "material type=lambert albedo=0.7,0.7,0.7\nmaterial type=dielectric ior=1.5\nmaterial type=light emit=20,18,15\nmaterial type=metal albedo=0.9,0.9,0.9 fuzz=0\n"
            "rectangle x=-6,6 y=0 z=-6,6 normal=1 material=0\nsphere center=0,1,0 radius=1 material=1\nsphere center=2.2,0.7,0.5 radius=0.7 material=3\n"
            "rectangle x=-2.7 y=0,2 z=-1,1 normal=1 twoSided=1 material=0\n")   # an occluder between some bounce directions and the lights
    one = "rectangle x=-1.5,1.5 y=5 z=-1.5,1.5 normal=-1 material=2\n"
    nine = "".join("rectangle x=%g,%g y=5 z=%g,%g normal=-1 material=2\n" % (-1.5 + i, -0.6 + i, -1.5 + j, -0.6 + j) for i in range(3) for j in range(3))
    for name, lights, count in (("one", one, 1), ("nine", nine, 9)):
        p = tmp_path / ("spec_%s.scene" % name)
        p.write_text(head + lights)
        host = pt.HostScene.load(str(p))
        dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
        assert dev.info()["rect_lights"] == count
        _image_parity(host, dev, osc, 64, 48, 6, 1, 16, 0.99)
        _image_parity(host, dev, osc, 64, 48, 6, 1, 16, 0.99, enableMnee=1, enableMneeSecondary=1)
        dev.close()


def test_scene_from_a_prepared_geometry_file(tmp_path, cornell_small):
    # one BVH build for the processes of a multi-GPU render: the geometry written by prepare_geometry (host only) gives the same
    # device scene as building it in place - same tree, same image bit for bit; a file made for another scene is refused
    host, dev, _ = cornell_small
    path = str(tmp_path / "geometry.bin")
    assert pt.prepare_geometry(host.desc, path) > 0.0
    shared = pt.DeviceScene(host.desc, 0, keepalive=host, prepared=path)
    assert shared.info() == dev.info() and shared.timings()["geometry_from_cache"] and not dev.timings()["geometry_from_cache"]
    s = host.settings_for(width=72, height=56, max_depth=5)
    assert np.array_equal(shared.render_image(s, 4)[0], dev.render_image(s, 4)[0])
    shared.close()
    other = pt.HostScene.load(os.path.join(SCENES, "cornell.scene"), SCENES)
    with pytest.raises(pt.PtrError, match="another scene"):
        pt.DeviceScene(other.desc, 0, keepalive=other, prepared=path)
    with pytest.raises(pt.PtrError, match="cannot open"):
        pt.DeviceScene(host.desc, 0, keepalive=host, prepared=str(tmp_path / "missing.bin"))


def test_rccl_gather_and_reduce_on_the_device(tmp_path):
    # bench.py --gpus N: one process per GPU, RCCL gather of the band buffers to rank 0 and an all_reduce(MAX) of the timing.  A
    # one-GPU box can only run world size 1, but that still loads RCCL, creates the communicator on the device and drives both
    # collectives on device buffers (in a child process: a process group is per process)
    import subprocess
    import sys
    code = """
import importlib, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, HSA_ENABLE_IPC_MODE_LEGACY="0")
bands = importlib.import_module("metal-pathtracer-arm64_amd.bands")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rows = bands.max_band_count(40, 1) * bands.BAND_ROWS
local = torch.arange(rows * 24 * 3, dtype=torch.float32, device="cuda").reshape(rows, 24, 3)
recv = [torch.empty_like(local)]
dist.gather(local, gather_list=recv, dst=0)          # the collective gather_bands issues when there is more than one rank
img = bands.assemble(recv, 40)
assert img.is_cuda and torch.equal(img, local[:40]) and torch.equal(bands.gather_bands(local, 40, 0, 1), img)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 1.25
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
""" % (ROOT, str(29500 + os.getpid() % 2000))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stderr[-2000:]


def test_surface_records_and_next_ray_origins_match_the_oracle():
    # a13 / a14 at function level: what a bounce is built from - the hit point, the geometric normal, the interpolated shading normal
    # flipped to the geometric side (E:2348-2367), the facing flag, and the origin OffsetRayOrigin (E:917-931) gives the next ray -
    # for rays at a smooth-shaded mesh, rectangles and spheres, from outside and from inside, next directions on either side
    host = pt.HostScene.load(os.path.join(GOLDEN, "cornell_small_mesh.scene"), SCENES)
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    rng = np.random.default_rng(11)
    n = 40000
    org = rng.uniform(30.0, 525.0, (n, 3)).astype(np.float32)
    target = rng.uniform(0.0, 555.0, (n, 3)).astype(np.float32)
    target[: n // 2] = np.array([278.0, 200.0, 278.0], np.float32) + rng.normal(0, 60.0, (n // 2, 3)).astype(np.float32)   # at the mesh
    d = target - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    nxt = rng.normal(size=(n, 3))
    nxt = (nxt / np.linalg.norm(nxt, axis=1, keepdims=True)).astype(np.float32)
    rays = np.concatenate([org, d, nxt], axis=1)
    got, want = dev.surface_hits(rays), osc.surface_hits(rays)
    hit = want[:, 0] > 0
    assert np.array_equal(got[:, 0], want[:, 0]) and hit.mean() > 0.9   # (the box is open at the front)
    assert np.array_equal(got[hit, 1], want[hit, 1])                                  # same distance, bit for bit (a7)
    assert np.array_equal(got[hit, 2:5], want[hit, 2:5])                              # same hit point
    assert np.array_equal(got[hit, 11], want[hit, 11])                                # same facing
    # normals go through a normalisation (sqrt + division: IEEE on both sides) of the same operands: equal, up to primitives hit
    # exactly on a shared edge where the two sides may pick different triangles
    same = np.all(got[hit, 5:11] == want[hit, 5:11], axis=1)
    assert same.mean() > 0.999 and np.abs(got[hit, 5:11] - want[hit, 5:11])[same].max() == 0.0
    assert np.allclose(np.linalg.norm(got[hit, 8:11], axis=1), 1.0, atol=1e-5)
    facing = np.where(got[hit, 11:12] > 0, 1.0, -1.0) * got[hit, 5:8]
    on_side = np.einsum("ij,ij->i", got[hit, 8:11], facing) >= -1e-6                  # meshes and rectangles: flipped to the geometric side;
    assert on_side.mean() > 0.9                                                       # a sphere seen from inside keeps its outward normal (E:2389-2404)
    assert np.array_equal(got[hit][same][:, 12:15], want[hit][same][:, 12:15])        # a14: the next origin, bit for bit
    moved = np.linalg.norm(got[hit, 12:15] - got[hit, 2:5], axis=1)
    # |offset| <= max(t 1e-4, 1e-4) along the normal + 0.5e-4 along the direction (+ the rounding of coordinates of a few hundred units)
    assert (moved > 0.0).all() and (moved <= np.maximum(got[hit, 1] * 1e-4, 1e-4) + 0.5e-4 + 4e-4).all()
    dev.close()


def test_partition_and_pool_size_invariance(cornell_small):
    host, dev, _ = cornell_small
    s = host.settings_for(width=72, height=56, max_depth=5)
    full, _ = dev.render_image(s, 6)
    for parts in (2, 3):
        pieces = [dev.render(s, 6, p, parts)[0] for p in range(parts)]
        assert np.array_equal(pt.assemble_bands(pieces, 72, 56), full)      # bit-identical under any partition
    again, _ = dev.render_image(s, 6)
    assert np.array_equal(again, full)                                       # and run to run


def test_scheduling_knobs_do_not_change_the_image():
    # How the work is scheduled - pool groups, the end-of-frame kernels, the pool size, the node format, the order k_shade visits
    # the slots of a block in - must not show in the result: same image bit for bit, same ray and hit counters.
    # 1080p x 12 spp = 25 M work items: a 12 Mi-slot pool in 4 groups, large enough for every mechanism to engage.
    host = pt.HostScene.load(os.path.join(SCENES, "cornell_mesh.scene"), SCENES)
    s = host.settings_for(width=1920, height=1080, max_depth=8, seed=1337)

    def render(env):
        os.environ.update(env)       # the knobs are read when the scene is uploaded
        try:
            dev = pt.DeviceScene(host.desc, 0, keepalive=host)
            image, _ = dev.render_image(s, 12)
            _, stats = dev.render_image(s, 12, count=True)
            dev.close()
        finally:
            for k in env:
                del os.environ[k]
        return image, (stats.extendRays, stats.shadowRays, stats.shadedHits, stats.extendNodesVisited, stats.samples)

    base, base_counts = render({})
    assert np.isfinite(base).all() and base.mean() > 0.01 and base_counts[4] == 1920 * 1080 * 12
    # every environment variable the library reads (csrc/host/knobs.h) appears here
    for env in ({"PTR_POOL_GROUPS": "1"}, {"PTR_POOL_GROUPS": "4"},   # (default: two groups, each with its k_connect beside the next k_extend)
                {"PTR_CONNECT_OVERLAP": "0"},   # k_connect on its group's own stream
                {"PTR_TAIL_BELOW": "0"}, {"PTR_POOL_SLOTS": str(3 << 20), "PTR_REFILL_BELOW": "24"},
                {"PTR_WIDE_NODES": "0"},        # the binary walk instead of the four-wide nodes (same tree, one level at a time)
                {"PTR_WIDE_NODES": "2"},        # four-wide nodes collapsed by level instead of by box area (same tree, other groupings)
                {"PTR_QUANTIZED_NODES": "0"},   # 64 B float nodes (box tests only prune: the hits are the same)
                {"PTR_MAX_ITEMS": str(1920 * 1080 * 12)},   # the whole frame still fits one pass
                {"PTR_BUILD_THREADS": "3", "PTR_VERBOSE": "build"},   # same tree from any number of builder threads
                {"PTR_NO_OVERSIZE": "1"}):      # (this scene keeps every triangle in the tree anyway)
        image, counts = render(env)
        assert np.array_equal(image, base), env
        if "PTR_QUANTIZED_NODES" in env:   # float boxes are tighter than their 16-bit roundings: fewer node visits, the same rays and hits
            assert counts[:3] == base_counts[:3] and counts[4] == base_counts[4] and counts[3] <= base_counts[3], env
        else:
            assert counts == base_counts, env


def test_frames_rendered_in_several_passes(cornell_small):
    # a frame whose per-sample accumulators do not fit the memory budget is rendered in passes of equal sample counts that add up
    # in the output buffer; the sample streams do not depend on the split, so only the order of the float additions differs
    host, dev, osc = cornell_small
    s = host.settings_for(width=64, height=64, max_depth=4, seed=1337)
    whole, st_whole = dev.render_image(s, 24, count=True)
    os.environ["PTR_MAX_ITEMS"] = str(64 * 64 * 5)            # 5 samples per pixel and pass -> 5 passes of 5, 5, 5, 5, 4
    try:
        split, st_split = dev.render_image(s, 24, count=True)
        parts = [pt.assemble_bands([dev.render(s, 24, p, 3)[0] for p in range(3)], 64, 64)]   # passes x partitions
    finally:
        del os.environ["PTR_MAX_ITEMS"]
    assert np.allclose(split, whole, rtol=2e-6, atol=1e-7) and not np.array_equal(split, np.zeros_like(split))
    assert np.array_equal(parts[0], split)                    # still independent of the partition
    assert st_split.extendRays == st_whole.extendRays and st_split.shadedHits == st_whole.shadedHits
    assert st_split.samples == st_whole.samples == 64 * 64 * 24
    ref, _, _ = osc.render(s, 24, threads=0)
    assert _rmse(split, ref) < 2.0 * _rmse(whole, ref) + 1e-6


def test_edge_cases(cornell_small, tmp_path):
    host, dev, osc = cornell_small
    # depth 1: only directly visible emission / background
    s = host.settings_for(width=32, height=32, max_depth=1)
    img, _ = dev.render_image(s, 2)
    ref, _, _ = osc.render(s, 2, threads=0)
    assert np.allclose(img, ref, rtol=1e-5, atol=1e-6)
    # ragged size (not a multiple of 8 or 16) and spp smaller than the slots-per-pixel target
    s = host.settings_for(width=37, height=21, max_depth=3)
    img, st = dev.render_image(s, 1)
    ref, _, _ = osc.render(s, 1, threads=0)
    assert img.shape == (21, 37, 3) and st.samples == 37 * 21
    assert (_rel(img, ref).max(axis=2) <= 1e-3).mean() > 0.9
    # empty scene: every ray escapes to the sky gradient
    p = tmp_path / "empty.scene"
    p.write_text("camera target=0,0,0 distance=3 yaw=0.3 pitch=0.1 vfov=60\n")
    eh = pt.HostScene.load(str(p))
    ed = pt.DeviceScene(eh.desc, 0, keepalive=eh)
    es = eh.settings_for(width=24, height=16, max_depth=3)
    img, _ = ed.render_image(es, 2)
    ref, _, _ = ol.OracleScene(eh).render(es, 2, threads=1)
    assert np.allclose(img, ref, rtol=1e-5, atol=1e-6) and img.min() > 0.4


def test_cli_smoke_script_equivalent(tmp_path):
    # tests/public/headless_smoke_test.sh:33-62 — same flags, passes iff the output file is non-empty
    import subprocess

    out = tmp_path / "smoke.ppm"
    r = subprocess.run([pt.CLI_PATH, "--scene=" + os.path.join(GOLDEN, "smoke.scene"), "--width=64", "--height=64", "--sppTotal=4",
                        "--maxDepth=4", "--seed=1337", "--enableSoftwareRayTracing=1", "--format=ppm", "--output=" + str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.stat().st_size == len(b"P6\n64 64\n255\n") + 64 * 64 * 3
    assert "Rendered 4 spp at 64x64" in r.stdout
    exr = tmp_path / "smoke.exr"
    r = subprocess.run([pt.CLI_PATH, "--scene", os.path.join(GOLDEN, "smoke.scene"), "--sppTotal=4", "--seed=1337", "--rgbaExr=1", "--output", str(exr)],
                       capture_output=True, text=True)
    assert r.returncode == 0 and exr.stat().st_size == 66925       # the byte count the paper publishes


# --------------------------------------------------------------------------- BASELINE configurations at their defining sizes
def test_config1_cornell_box_at_full_size():
    # BASELINE configs[0]: scenes/cornell.scene (analytic spheres + rectangles, Lambertian + emissive), 512x512, depth 4,
    # 64 spp, seed 1337: SURVEY.md section 8(d) protocol on the whole frame (the noise floor on a 128-row strip)
    host = pt.HostScene.load(os.path.join(SCENES, "cornell.scene"))
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    s = host.settings_for(seed=1337)
    assert (s.width, s.height, s.maxDepth) == (512, 512, 4)
    info = dev.info()
    assert info["spheres"] == 2 and info["triangles"] == 12 and info["rect_lights"] == 1
    img, st = dev.render_image(s, 64, count=True)
    ref, _, c = osc.render(s, 64, threads=ORACLE_THREADS, count=True)
    assert st.samples == 512 * 512 * 64 and np.isfinite(img).all() and img.min() >= 0
    assert abs(st.extendRays - c["extendRays"]) <= 0.002 * c["extendRays"]
    s2 = s.copy()
    s2.seed = 1338
    rows = (192, 320)
    other, _, _ = osc.render(s2, 64, threads=ORACLE_THREADS, rows=rows)
    noise = _rmse(ref[rows[0]:rows[1]], other[rows[0]:rows[1]])
    err = _rmse(img, ref)
    lum = np.array([0.2126, 0.7152, 0.0722])
    ratio = float((img @ lum).mean() / (ref @ lum).mean())
    assert err <= 1.25 * noise and abs(ratio - 1.0) <= 0.005, (err, noise, ratio)
    assert err < 0.25 * noise                # same sample streams: far below the seed-to-seed noise


def test_config5_lucy_standin_at_full_size():
    # BASELINE configs[4]: scenes/lucy_standin.scene, 28,005,128 + 1,002,528 mesh triangles (BVH > 1 GB: nodes, triangles and
    # normals do not fit any cache), 3840x2160, depth 12.  Scene facts, 200 k ray queries bit-equal to the oracle's, and the
    # image protocol on a strip of the full-resolution frame.
    from scenes.gen_assets import ensure_large_asset
    for a in ("lucy_standin_28005128.ply", "blob_1002528.ply"):
        ensure_large_asset(a)
    host = pt.HostScene.load(os.path.join(SCENES, "lucy_standin.scene"), SCENES)
    dev = pt.DeviceScene(host.desc, 0, keepalive=host)
    info = dev.info()
    assert info["triangles"] == 28005128 + 1002528 + 4 and info["max_depth"] < 48 and info["max_leaf"] <= 4 and info["rect_lights"] == 1
    assert info["nodes"] * 32 > (1 << 28)                                  # node array alone beyond 256 MiB
    osc = ol.OracleScene(host)
    rays = _random_rays(200_000, -400.0, 400.0, 17)
    rays[:, 1] = np.abs(rays[:, 1]) + 1.0
    g, stats = dev.trace_rays(rays)
    o = osc.trace_rays(rays)
    assert np.array_equal(g["t"], o["t"]) and (o["t"] >= 0).mean() > 0.5
    hit = o["t"] >= 0
    same = (g["primType"][hit] == o["primType"][hit]) & (g["primIndex"][hit] == o["primIndex"][hit]) & (g["geomIndex"][hit] == o["geomIndex"][hit])
    assert same.mean() > 0.999
    ga, _ = dev.trace_rays(rays, any_hit=True)
    oa = osc.trace_rays(rays, any_hit=True)
    assert np.array_equal(ga["t"] >= 0, oa["t"] >= 0)
    s = host.settings_for(seed=1337)
    assert (s.width, s.height, s.maxDepth) == (3840, 2160, 12)
    rows = (1040, 1104)
    img1, _ = dev.render_image(s, 1)
    ref1, _, _ = osc.render(s, 1, threads=ORACLE_THREADS, rows=rows)
    frac = float((_rel(img1[rows[0]:rows[1]], ref1[rows[0]:rows[1]]).max(axis=2) <= 1e-3).mean())
    assert frac >= 0.99, frac
    img, st = dev.render_image(s, 8)
    assert st.samples == 3840 * 2160 * 8 and np.isfinite(img).all()
    ref, _, _ = osc.render(s, 8, threads=ORACLE_THREADS, rows=rows)
    s2 = s.copy()
    s2.seed = 1338
    other, _, _ = osc.render(s2, 8, threads=ORACLE_THREADS, rows=rows)
    sl = slice(rows[0], rows[1])
    noise, err = _rmse(ref[sl], other[sl]), _rmse(img[sl], ref[sl])
    lum = np.array([0.2126, 0.7152, 0.0722])
    ratio = float((img[sl] @ lum).mean() / (ref[sl] @ lum).mean())
    assert err <= 1.25 * noise and abs(ratio - 1.0) <= 0.005, (err, noise, ratio)
    dev.close()
    osc.close()


def test_ray_queries_against_the_reference_bvh_library():
    # the HIP traversal against tinybvh 1.6.7 itself (oracle/_ref, built from /root/reference/external/tinybvh by oracle/Makefile):
    # the library the reference's software path builds its trees with, traversed by the library's own intersector
    import ref_tinybvh as rt
    if not rt.available():
        pytest.skip("oracle/_ref/libref_tinybvh.so not built")
    host = pt.HostScene.load(os.path.join(SCENES, "cornell_mesh.scene"), SCENES)
    dev = pt.DeviceScene(host.desc, 0, keepalive=host)
    tris = rt.mesh_world_triangles(host.desc, 0)
    ref = rt.RefBvh(tris)
    rng = np.random.default_rng(21)
    centre = tris.reshape(-1, 3).mean(axis=0)
    radius = float(np.linalg.norm(tris.reshape(-1, 3) - centre, axis=1).max())
    n = 100_000
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = centre + radius * d * rng.uniform(1.25, 1.7, size=(n, 1))
    dirs = (centre + radius * 0.7 * rng.uniform(-1, 1, size=(n, 3))) - org
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True).astype(np.float32)
    rays = np.concatenate([org.astype(np.float32), np.full((n, 1), 1e-4, np.float32), dirs, np.full((n, 1), np.inf, np.float32)], axis=1)
    g, _ = dev.trace_rays(rays)
    rt_t, rt_prim, _ = ref.intersect(rays)
    mesh_hit = (g["t"] >= 0) & (g["primType"] == 0)
    assert mesh_hit.mean() > 0.3 and (rt_t[mesh_hit] >= 0).all()
    rel = np.abs(rt_t[mesh_hit] - g["t"][mesh_hit]) / np.maximum(g["t"][mesh_hit], 1.0)
    assert rel.max() < 2e-5, rel.max()
    assert (rt_prim[mesh_hit] == g["primIndex"][mesh_hit]).mean() > 0.999
    other = ~mesh_hit
    assert not ((rt_t[other] >= 0) & ((g["t"][other] < 0) | (rt_t[other] < g["t"][other] * (1 - 1e-5)))).any()


def test_multi_device_render_matches_single_device(cornell_small):
    # ptr_render_multi: one preparation of the scene, one upload + one host thread per device, interleaved bands handed to the first
    # device and interleaved there.  A one-GPU box runs the same path with the device named several times.
    host, dev, _ = cornell_small
    s = host.settings_for(width=88, height=72, max_depth=5, seed=1337)
    single, _ = dev.render_image(s, 6)
    # ids given as -(id + 1): that partition's bands take the pinned-host staging path (devices that cannot address each other)
    for ids in ([0], [0, 0], [0, 0, 0], [0] * 9, [0, -1, -1]):   # 9 bands of 8 rows: [0] * 9 gives every partition one band
        multi, st = pt.render_multi(host.desc, s, 6, device_ids=ids)
        assert np.array_equal(multi, single), ids             # bit-identical, whatever the number of partitions
        assert st.samples == 88 * 72 * 6 and st.totalSeconds > 0
    whole, st = pt.render_multi(host.desc, s, 6, n_devices=1)
    assert np.array_equal(whole, single)
    with pytest.raises(pt.PtrError):
        pt.render_multi(host.desc, s, 6, n_devices=pt.device_count() + 1)


def test_cli_devices_and_aov_export(tmp_path):
    import subprocess

    out, aov = tmp_path / "frame.pfm", tmp_path / "features.exr"
    r = subprocess.run([pt.CLI_PATH, "--scene=" + os.path.join(GOLDEN, "smoke.scene"), "--width=64", "--height=48", "--sppTotal=4", "--seed=1337",
                        "--format=pfm", "--output=" + str(out), "--aovExr=" + str(aov), "--devices=1", "--backend=embree"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Feature layers written to" in r.stdout
    data = aov.read_bytes()
    assert b"albedo.R\x00" in data[:600] and b"normal.Z\x00" in data[:600] and b"depth.Z\x00" in data[:600]
    assert len(data) > 64 * 48 * 10 * 4
    body = np.frombuffer(data[-(48 * (8 + 10 * 64 * 4)):], np.uint8).reshape(48, 8 + 10 * 64 * 4)
    planes = body[:, 8:].copy().view(np.float32).reshape(48, 10, 64)
    host = pt.HostScene.load(os.path.join(GOLDEN, "smoke.scene"))
    dev = pt.DeviceScene(host.desc, 0, keepalive=host)
    s = host.settings_for(width=64, height=48, seed=1337)
    img, _ = dev.render_image(s, 4)
    albedo, normal = dev.render_aovs(s, 0)
    assert np.array_equal(planes[:, 2], img[..., 0]) and np.array_equal(planes[:, 5], albedo[..., 0]) and np.array_equal(planes[:, 6], normal[..., 3])
    # all visible devices through the CLI: the same image as one device
    out2 = tmp_path / "frame_all.pfm"
    r = subprocess.run([pt.CLI_PATH, "--scene=" + os.path.join(GOLDEN, "smoke.scene"), "--width=64", "--height=48", "--sppTotal=4", "--seed=1337",
                        "--format=pfm", "--output=" + str(out2), "--devices=0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out2.read_bytes() == out.read_bytes()


# --------------------------------------------------------------------------- deterministic-stream criterion
_STREAM_CASES = [
    ("golden/smoke.scene", 64, 64), ("golden/cornell_small_mesh.scene", 128, 128), ("golden/materials.scene", 192, 128),
    ("golden/env_materials.scene", 192, 128), ("golden/lucy_small.scene", 160, 90), ("scenes/cornell.scene", 256, 256),
    ("scenes/cornell_mesh.scene", 320, 180), ("scenes/helmet_env.scene", 320, 180), ("scenes/knot_glass.scene", 320, 180),
]


@pytest.mark.parametrize("case", _STREAM_CASES, ids=[c[0].split("/")[-1] for c in _STREAM_CASES])
def test_deterministic_stream_criterion_and_what_the_rest_comes_from(case):
    """>= 99 % of the pixels within 1e-3 (relative) of the oracle at 1 spp, depth 2, on every parity scene - the criterion as
    SURVEY.md section 8(d) states it.  At the scenes' own depths the Cornell-box scenes fall below it; the path signatures
    (which vertices received a rectangle-light sample that contributed + a hash of the primitives hit, computed by the HIP
    counting build and by the oracle) say why, pixel by pixel: the paths hit the same primitives and differ in ONE thing, the
    outcome of a rectangle-light shadow test - and the oracle flags exactly those tests as decided within +-2e-6 of the
    shadow ray's length (quirk Q9: for a surface perpendicular to the light the ray ends 0.5e-4 before the light's own
    plane, half an ulp of the distance).  Taking that one decision out of the rounding noise (debugShadowSlack, on both
    sides) restores the criterion at every depth."""
    from scenes.gen_assets import ensure_large_asset
    rel_path, w, h = case
    if "knot" in rel_path:
        ensure_large_asset("torus_knot_871200.ply")
    if "lucy_small" in rel_path:
        ensure_large_asset("blob_125000.ply")
    path = os.path.join(GOLDEN, rel_path.split("/", 1)[1]) if rel_path.startswith("golden/") else os.path.join(ROOT, rel_path)
    host = pt.HostScene.load(path, SCENES)
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    own_depth = int(host.settings_for(width=w, height=h).maxDepth)
    for depth in sorted({2, own_depth}):
        s = host.settings_for(width=w, height=h, max_depth=depth, seed=1337)
        g, gsig = dev.render_signatures(s)
        plain, _ = dev.render_image(s, 1)
        assert np.array_equal(plain, g)                      # the counting build renders the same image as the timed build
        o, osig, marginal = osc.render_signatures(s, threads=ORACLE_THREADS)
        bad = _rel(g, o).max(axis=2) > 1e-3
        frac = 1.0 - float(bad.mean())
        nee = ((gsig ^ osig) & 0xFFFF) != 0
        prim = ((gsig ^ osig) >> 16) != 0
        if depth == 2:
            assert frac >= 0.99, (rel_path, depth, frac)
        if frac < 0.99:
            shadow_flip = bad & nee & ~prim & marginal       # same primitives, another outcome of a marginal light-sample shadow test
            assert shadow_flip.sum() >= 0.95 * bad.sum(), (rel_path, depth, int(bad.sum()), int(shadow_flip.sum()))
        # signatures agree wherever the pixels agree (up to ties that change nothing visible)
        assert (~bad & (nee | prim)).mean() < 1e-3
        s2 = s.copy()
        s2.debugShadowSlack = 1e-3
        g2, _ = dev.render_image(s2, 1)
        o2, _, _ = osc.render(s2, 1, threads=ORACLE_THREADS)
        frac2 = float((_rel(g2, o2).max(axis=2) <= 1e-3).mean())
        assert frac2 >= 0.99, (rel_path, depth, frac2)
        if os.environ.get("PTR_TEST_VERBOSE"):
            print("%s depth %d: %.4f within 1e-3 (%d differ, %d marginal shadow flips); with the shadow test out of the noise: %.4f"
                  % (rel_path, depth, frac, int(bad.sum()), int((bad & nee & ~prim & marginal).sum()), frac2))
    dev.close()
    osc.close()


def test_oversize_triangles_outside_the_tree(tmp_path):
    # the floor of a room 600 times the size of a finely tessellated mesh is kept out of the BVH (so that the 16-bit grid of the
    # 32 B nodes covers the mesh, not the room) and tested first by every ray: same hits as the oracle, same images
    text = ("camera target=0,10,0 distance=40 yaw=1.0 pitch=0.3 vfov=40\nrenderer maxDepth=5 seed=1337\nbackground solid=0.1,0.1,0.12\n"
            "material type=lambert albedo=0.6,0.6,0.6\nmaterial type=diffuse_light emit=14,14,14\nmaterial type=lambert albedo=0.8,0.5,0.3\n"
            "rectangle x=-1500,1500 y=0 z=-1500,1500 normal=1 material=0\n"
            "rectangle x=-40,40 y=90 z=-40,40 normal=-1 material=1\n"
            "mesh path=assets/blob_70688.obj translate=0,10,0 scale=0.05 material=2\n")
    p = tmp_path / "room.scene"
    p.write_text(text)
    host = pt.HostScene.load(str(p), SCENES)
    g = pt.debug_scene_geometry(host.desc)
    assert g["oversize"] == 2 and g["quantized_usable"] == 1
    dev, osc = pt.DeviceScene(host.desc, 0, keepalive=host), ol.OracleScene(host)
    rays = _random_rays(60_000, -30.0, 30.0, 4)
    rays[:, 1] = np.abs(rays[:, 1]) + 0.5
    for any_hit in (False, True):
        gh, _ = dev.trace_rays(rays, any_hit=any_hit)
        oh = osc.trace_rays(rays, any_hit=any_hit)
        assert np.array_equal(gh["t"] >= 0, oh["t"] >= 0)
        if not any_hit:
            assert np.array_equal(gh["t"], oh["t"])
            hit = oh["t"] >= 0
            assert ((gh["primType"][hit] == oh["primType"][hit]) & (gh["primIndex"][hit] == oh["primIndex"][hit])).mean() > 0.999
            assert (oh["primType"][hit] == 2).any() and (oh["primType"][hit] == 0).any()     # floor halves and mesh triangles
    _image_parity(host, dev, osc, 96, 64, 5, 1, 16, 0.95)


def test_metal_clamp_variants(materials_scene):
    # PTR_METAL_CLAMPS (shaders/pathtrace.metal:3550-3633): firefly limit raised to
    # fireflyClampMaxContribution, clamp_specular_tail skipped while base = scale = 0, clamp_specular_pdf passing the pdf through.
    # Parity against the oracle's restatement; and the variant really differs from the Embree-parity clamps.
    host, dev, osc = materials_scene
    assert host.settings.fireflyClampMaxContribution == 1000.0 and host.settings.specularTailClampBase == 0.0
    _image_parity(host, dev, osc, 96, 64, 6, 1, 32, 0.988, metalSemantics=64)
    _image_parity(host, dev, osc, 96, 64, 6, 1, 16, 0.987, metalSemantics=64 | 8, minSpecularPdf=1e-3, specularTailClampBase=2.0)
    s0 = host.settings_for(width=96, height=64, max_depth=6, seed=1337)
    s1 = s0.copy()
    s1.metalSemantics = 64
    a, b = dev.render_image(s0, 64)[0], dev.render_image(s1, 64)[0]
    assert _rmse(a, b) > 1e-3 and b.mean() > a.mean()        # looser clamps keep more energy
    # device functions: the BSDF values of a smooth metal are no longer capped at the clamp floor
    d = host.desc
    metals = [i for i in range(d.materialCount) if int(d.materials[i].typeEta[0]) == 1]
    assert metals
    rng = np.random.default_rng(2)
    n = 2000
    wo = rng.normal(size=(n, 3))
    wo[:, 2] = np.abs(wo[:, 2]) + 0.2
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = wo * np.array([-1, -1, 1]) + rng.normal(scale=0.02, size=(n, 3))
    wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    inputs = np.concatenate([np.zeros((n, 3)), np.tile([0, 0, 1.0], (n, 1)), wo, wi], axis=1).astype(np.float32)
    for mi in metals[:2]:
        g = pt.debug_eval_bsdf(d.materials[mi], s1, inputs)
        o = ol.eval_bsdf(d.materials[mi], s1, inputs)
        both = np.isfinite(g).all(axis=1) & np.isfinite(o).all(axis=1)
        assert np.isclose(g[both], o[both], rtol=5e-3, atol=1e-4).all(axis=1).mean() > 0.99
