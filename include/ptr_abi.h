/*
 * ptr_abi.h — C-ABI of the MI355X wavefront path tracer (libptr_hip.so).
 *
 * This is the drop-in boundary for the SWRT render path of
 * dariopagliaricci/Metal-PathTracer-arm64.  Everything a host in any language
 * needs is plain C: POD structs, raw pointers, sizes.  No C++/torch types.
 *
 * Reference interface each entry point replaces (paths relative to the
 * reference checkout):
 *   ptr_render / ptr_render_tiles   <- IHeadlessRenderer::render(...)
 *                                      include/headless/IHeadlessRenderer.h:42-52
 *                                      (caller: src/main_headless.mm:524-545)
 *   PtrSceneDesc                    <- HeadlessScene::resources, i.e. the CPU arrays
 *                                      SceneResources exposes to the Embree backend
 *                                      (include/renderer/SceneResources.h:165,248-267;
 *                                      consumed at src/headless/EmbreeHeadlessRenderer.mm:2077-2300,2478-2481)
 *   PtrSphere/PtrRect/PtrMaterial   <- SphereData/RectData/MaterialData
 *                                      include/MetalShaderTypes.h:44-97 (32 B / 80 B / 576 B)
 *   PtrSettings                     <- the RenderSettings fields the integrator reads
 *                                      include/renderer/RenderSettings.h:16-145
 *   PtrRenderStats                  <- HeadlessRenderOutput timing fields
 *                                      (IHeadlessRenderer.h:33-40) + PathtraceStats counters
 *                                      (include/MetalShaderTypes.h:215-226)
 *   ptr_scene_upload/release        <- SceneResources::rebuildAccelerationStructures +
 *                                      SoftwareBvhAccel::rebuild (src/renderer/SceneResources.mm:2055-2259,
 *                                      src/renderer/SceneAccel.mm:23-325)
 *   ptr_trace_rays                  <- trace_scene_software closest/any hit
 *   ptr_render_aovs                 <- first-hit albedo / normal outputs of pathtraceIntegrateKernel (denoiser inputs)
 *                                      (shaders/pathtrace.metal:2266-2382) == rtcIntersect1/rtcOccluded1
 *                                      call sites (EmbreeHeadlessRenderer.mm:2302-2433)
 *   ptr_host_*                      <- SceneManager::loadSceneFromPath (src/renderer/SceneManager.mm:677-722),
 *                                      WriteImage (src/renderer/ImageWriter.mm:609-627)
 */
#ifndef PTR_ABI_H
#define PTR_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene PODs (layouts match the reference's host/device contract) ---- */

typedef struct PtrSphere {          /* 32 B, MetalShaderTypes.h:44-47 */
    float centerRadius[4];          /* xyz = centre, w = radius */
    uint32_t materialIndex[4];      /* x = material index */
} PtrSphere;

typedef struct PtrRect {            /* 80 B, MetalShaderTypes.h:49-55 */
    float corner[4];
    float edgeU[4];                 /* w = 1/|U|^2 */
    float edgeV[4];                 /* w = 1/|V|^2 */
    float normalAndPlane[4];        /* xyz = unit normal, w = dot(normal, corner) */
    uint32_t materialTwoSided[4];   /* x = material index, y = two-sided flag */
} PtrRect;

enum PtrMaterialType {
    PTR_MAT_LAMBERTIAN = 0,
    PTR_MAT_METAL = 1,
    PTR_MAT_DIELECTRIC = 2,
    PTR_MAT_DIFFUSE_LIGHT = 3,
    PTR_MAT_PLASTIC = 4,
    PTR_MAT_SUBSURFACE = 5,
    PTR_MAT_CARPAINT = 6,
    PTR_MAT_PBR = 7
};

typedef struct PtrMaterial {        /* 576 B, MetalShaderTypes.h:57-97 */
    float baseColorRoughness[4];
    float typeEta[4];               /* x = type, y = IOR, z = coat IOR, w = thin flag */
    float emission[4];              /* w = env-portal flag */
    float conductorEta[4];
    float conductorK[4];
    float coatParams[4];            /* roughness, thickness, sample weight, Fresnel average */
    float coatTint[4];
    float coatAbsorption[4];
    float dielectricSigmaA[4];
    float sssSigmaA[4];
    float sssSigmaS[4];
    float sssParams[4];
    float carpaintBaseParams[4];    /* metallic, roughness, flake scale, flake reflectance scale */
    float carpaintFlakeParams[4];   /* sample weight, roughness, anisotropy, normal strength */
    float carpaintBaseEta[4];
    float carpaintBaseK[4];
    float carpaintBaseTint[4];
    uint32_t textureIndices0[4];
    uint32_t textureIndices1[4];
    uint32_t materialFlags;
    uint32_t materialPad[3];
    float pbrParams[4];             /* metallic, roughness, occlusion strength, normal scale */
    float pbrExtras[4];
    uint32_t textureUvSet0[4];
    uint32_t textureUvSet1[4];
    float textureTransform[12][4];
} PtrMaterial;

typedef struct PtrMeshDesc {
    const float* positions;         /* vertexCount * 3, object space */
    const float* normals;           /* vertexCount * 3, object space (may be zero vectors) */
    const uint32_t* indices;        /* indexCount (multiple of 3) */
    uint32_t vertexCount;
    uint32_t indexCount;
    float localToWorld[16];         /* column-major 4x4 */
    uint32_t materialIndex;
    uint32_t pad;
    /* texture coordinates and tangents (SceneVertex, include/MetalShaderTypes.h:351-356); each may be NULL.  Read only by the
     * textured metallic-roughness model (PTR_METAL_PBR), as in the reference, whose Embree backend never samples textures. */
    const float* uv0;               /* vertexCount * 2 */
    const float* uv1;               /* vertexCount * 2 */
    const float* tangents;          /* vertexCount * 4: object-space tangent, w = handedness (0: no tangent) */
} PtrMeshDesc;

/* A material texture: level 0 as linear RGBA floats (sRGB-encoded images are decoded to linear when loaded), row 0 = top.
 * The renderer builds the mip chain itself (2x2 box filter).  wrap: 0 repeat, 1 clamp to edge, 2 mirrored repeat (glTF sampler
 * wrapS / wrapT 10497 / 33071 / 33648); filter: 0 nearest, 1 linear (glTF magFilter 9728 / 9729). */
typedef struct PtrTexture {
    const float* rgba;
    uint32_t width, height;
    uint32_t wrapS, wrapT;
    uint32_t filter;
    uint32_t pad;
} PtrTexture;

typedef struct PtrSceneDesc {
    const PtrSphere* spheres;
    const PtrRect* rects;
    const PtrMaterial* materials;
    const PtrMeshDesc* meshes;
    const float* envRgba;           /* envWidth*envHeight*4 linear floats, row 0 = top; NULL if none */
    uint32_t sphereCount;
    uint32_t rectCount;
    uint32_t materialCount;
    uint32_t meshCount;
    uint32_t envWidth;
    uint32_t envHeight;
    const PtrTexture* textures;     /* what PtrMaterial.textureIndices0/1 index (0xFFFFFFFF = no texture); NULL if none */
    uint32_t textureCount;
    uint32_t pad;
} PtrSceneDesc;

enum PtrBackgroundMode { PTR_BG_GRADIENT = 0, PTR_BG_SOLID = 1, PTR_BG_ENVIRONMENT = 2 };

typedef struct PtrSettings {
    uint32_t width;
    uint32_t height;
    uint32_t maxDepth;
    uint32_t seed;                      /* fixedRngSeed; 0 -> 0x9e3779b9 */
    uint32_t enableRussianRoulette;
    uint32_t enableSpecularNee;
    uint32_t enableMnee;
    uint32_t enableMneeSecondary;
    float cameraTarget[3];
    float cameraDistance;
    float cameraYaw;
    float cameraPitch;
    float cameraVerticalFov;
    float cameraDefocusAngle;
    float cameraFocusDistance;
    uint32_t backgroundMode;
    float backgroundColor[3];
    float environmentRotation;          /* radians */
    float environmentIntensity;
    uint32_t fireflyClampEnabled;
    float fireflyClampFactor;
    float fireflyClampFloor;
    float throughputClamp;
    float specularTailClampBase;
    float specularTailClampRoughnessScale;
    float minSpecularPdf;
    float fireflyClampMaxContribution;
    float emissionScale;                /* PATH_TRACER_EMBREE_EMISSION_SCALE analogue; 1 = off */
    /* Metal-only integrator semantics the Embree oracle path does not have; 0 = the
     * Embree-parity integrator.  Bit 0 (PTR_METAL_MEDIA): Beer-Lambert absorption inside refractive dielectrics with
     * the 8-deep medium stack of shaders/pathtrace.metal:5768-5773, 5869-5876, 6694-6709.  Bit 1 (PTR_METAL_THIN):
     * thin-walled dielectrics (typeEta.w > 0.5) keep etaI = 1 on both faces and never enter a medium (:4589-4592,
     * 5649-5659, 5683).  Bit 2 (PTR_METAL_FACE_NORMAL): dielectrics are shaded with the geometric normal turned
     * towards the incoming ray (set_face_normal, :1187-1191); the Embree backend passes the unflipped normal
     * (EmbreeHeadlessRenderer.mm:2333-2339, 2641-2644), which sends a ray that tries to leave a glass body back inside.
     * Bit 3 (PTR_METAL_SPECULAR): rough metals (type 1) sample the distribution of visible normals, use the pdf
     * D*G1*cos(h)/(4 wo.wh) and scale the lobe by the multiple-scattering energy compensation (:3724-3739, 3770-3797,
     * 4610-4630, 5000-5023, 5228-5283); the Embree backend samples half vectors with pdf D*cos(h)/(4 wo.wh) and has no
     * compensation.  Bit 4 (PTR_METAL_SSS): subsurface materials (type 5) follow the Metal integrator: their BSDF
     * evaluates to zero (is_bssrdf: no next-event estimation, :5078-5084), and with sssMode = 1 they are sampled with
     * the separable diffusion profile - an exit point on the tangent plane at a radius drawn from exp(-sigma_tr r),
     * a cosine-distributed direction from it, weight = profile * cos / (area pdf * directional pdf), and the next
     * ray starts at the biased exit point (:3916-3994, 5398-5507, 6740-6766); otherwise the Lambert fallback.  The
     * Embree backend treats type 5 as Lambertian with NEE.  With sssMode = 2, front-face hits of materials that ask for
     * it (sssParams.y >= 0.5) run the random walk of sample_sss_random_walk_software (:4060-4311): coat lobe or refraction
     * into the medium, free flights against the closest boundary with Henyey-Greenstein scattering, up to sssMaxSteps
     * queries, Lambert fallback when the walk is abandoned - each query one extend/shade iteration of the wavefront.
     * Bit 5 (PTR_METAL_PBR): the metallic-roughness model (type 7) follows evaluate_ / sample_pbr_metallic_roughness
     * (:4598-4948): specular, diffuse and transmission lobes picked by weight (KHR_materials_transmission factor in
     * pbrExtras.z, thickness tint :3295-3306), visible-normal sampling with the G1 pdf, energy compensation on the
     * specular lobe, rough refraction with the Walter et al. Jacobian, delta mirror / delta refraction at roughness
     * <= 1e-3 (which then also counts as a delta surface, :4570-4586), and the material textures of the glTF path
     * (:5919-6400; PtrSceneDesc.textures, filtering rule in csrc/kernels/texture.h).
     * Bit 6 (PTR_METAL_CLAMPS): the Metal kernel's variants of the luminance clamps (shaders/pathtrace.metal:3550-3633; SURVEY.md
     * Appendix A rows 4-6): the firefly limit is raised to fireflyClampMaxContribution when that is positive (default 1000: far
     * looser than the Embree backend's max(32 lum(throughput), 4)); clamp_specular_tail is skipped while base and roughness
     * scale are both zero (the default - the Embree backend then still caps the lobe's luminance at the clamp floor);
     * clamp_specular_pdf returns 0 for a non-finite or non-positive pdf and the pdf itself while minSpecularPdf <= 0.  With
     * them go the next-event-estimation weights of the Metal kernel (:6532-6552, 6624-6645; Appendix A row 13): a light or
     * environment sample counts when the BSDF value is positive (not: when it has a density), its balance weight is clamped to
     * [1e-4, 0.9999], and the weight is 1 where the BSDF reports no density. */
    uint32_t metalSemantics;
    uint32_t sssMode;   /* RenderSettings::SssMode: 0 off, 1 separable, 2 random walk; read only with PTR_METAL_SSS */
    uint32_t sssMaxSteps;   /* RenderSettings::sssMaxSteps (32): closest-hit queries per random walk, at least 1 */
    /* Test knob, 0 in every product render.  The reference measures the length of a rectangle-light shadow ray from the
     * un-offset hit point (EmbreeHeadlessRenderer.mm:2745-2750, quirk Q9), so for a surface perpendicular to the light the
     * ray ends 0.5e-4 before the light's own plane: the occlusion test is decided inside float rounding noise.  A value s
     * in (0, 1) shortens those rays to distance * (1 - s) - 1e-4, which takes the decision out of the noise; the
     * deterministic-stream tests use it (on both sides) to show what the pixels that differ at s = 0 come from. */
    float debugShadowSlack;
} PtrSettings;

enum { PTR_METAL_MEDIA = 1u, PTR_METAL_THIN = 2u, PTR_METAL_FACE_NORMAL = 4u, PTR_METAL_SPECULAR = 8u, PTR_METAL_SSS = 16u, PTR_METAL_PBR = 32u,
       PTR_METAL_CLAMPS = 64u };

typedef struct PtrRenderStats {
    double totalSeconds;                /* integrate phase only (reference: out.totalSeconds) */
    double avgMsPerSample;
    double uploadSeconds;               /* BVH build + H2D, reported separately */
    double traceKernelMs;               /* sum of closest-hit kernel durations (HIP events) */
    double shadeKernelMs;
    double shadowKernelMs;
    uint64_t traceLaunches;
    uint64_t samples;                   /* pixel-samples integrated */
    /* PathtraceStats-style counters (filled only when countTraversal != 0) */
    uint64_t primaryRays;
    uint64_t extendRays;                /* closest-hit rays (primary + bounce + specular-NEE) */
    uint64_t shadowRays;
    uint64_t nodesVisited;              /* nodes popped-and-tested, all rays */
    uint64_t leafPrimTests;
    uint64_t extendNodesVisited;        /* subset of the above for the closest-hit kernel */
    uint64_t extendLeafPrimTests;
    uint64_t shadedHits;
    uint64_t triangleHits;
    uint64_t shadowEarlyExits;
    double tailKernelMs;                /* end-of-frame kernels (k_tail_collect + k_tail_run): the last paths, one lane each */
} PtrRenderStats;

typedef struct PtrHit {                 /* result of ptr_trace_rays */
    float t;                            /* < 0 on miss */
    float u, v;
    uint32_t primType;                  /* 0 mesh triangle, 1 sphere, 2 rectangle */
    uint32_t geomIndex;                 /* mesh index (primType 0) else 0 */
    uint32_t primIndex;                 /* triangle / sphere / rectangle index */
    float ng[3];                        /* unnormalised geometric normal (triangles) */
    uint32_t pad;
} PtrHit;

typedef struct PtrDeviceScene PtrDeviceScene;

/* ---- device path (HIP) ---- */

/* Number of visible HIP devices (0 when none / runtime unavailable). */
int ptr_device_count(void);

/* Build the SAH BVH on the host, flatten to device SoA arrays and upload to `device`. */
int ptr_scene_upload(const PtrSceneDesc* scene, int device, PtrDeviceScene** out_scene,
                     char* err, size_t err_cap);
void ptr_scene_release(PtrDeviceScene* scene);

/* BVH facts for tests/diagnostics: [0]=nodes,[1]=leaves,[2]=triangles,[3]=spheres,[4]=max depth,[5]=max leaf size */
int ptr_scene_info(const PtrDeviceScene* scene, uint64_t out[8]);
/* Where the upload's time went: [0] geometry preparation on the host (bake, BVH build, leaf order, wide nodes) or reading it from a
 * geometry cache, [1] shading tables (materials, lights, environment alias tables, texture mips), [2] copies to the device,
 * [3] 1.0 when the geometry came from a cache file. */
int ptr_scene_timings(const PtrDeviceScene* scene, double out[4]);

/* One BVH build for the processes of a multi-GPU render (one process per device).  ptr_scene_prepare_geometry runs the
 * device-independent geometry preparation on the host - no GPU call - and writes it to cache_path (a file under /dev/shm);
 * ptr_scene_upload_prepared is ptr_scene_upload with the geometry read from that file instead of built.  The file carries a
 * fingerprint of the scene description; a process holding another description is refused.
 * ("BVH build once"; the reference builds per process, SceneAccel.mm:23-325.) */
int ptr_scene_prepare_geometry(const PtrSceneDesc* scene, const char* cache_path, double* seconds, char* err, size_t err_cap);
int ptr_scene_upload_prepared(const PtrSceneDesc* scene, const char* cache_path, int device, PtrDeviceScene** out_scene,
                              char* err, size_t err_cap);

/* Render the whole image, result copied to host `out_rgb` (width*height*3, row 0 = top). */
int ptr_render(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, int verbose,
               float* out_rgb, PtrRenderStats* stats, char* err, size_t err_cap);

/* The same frame on `n_devices` devices of this node (0 = all visible): the scene is prepared once, uploaded to every device,
 * device k renders the bands b = k (mod n) (see below), the band buffers are handed to device 0 over the fabric and
 * interleaved there.  Pixel for pixel the image is the one ptr_render produces (every pixel's sample streams and sums are
 * independent of the partition).  Extends the one call site of the reference, src/main_headless.mm:524-545, which drives one
 * device.  stats->totalSeconds = the slowest device's integrate + hand-over time. */
int ptr_render_multi(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, int n_devices, int verbose,
                     float* out_rgb, PtrRenderStats* stats, char* err, size_t err_cap);

/* Rows per image band, the unit the frame is partitioned in.  8 keeps the largest partition of a 1080-row frame
 * within 0.7 % of the mean for 2/4/8 partitions (16-row bands left 6.7 % at 8) and is one row of the 8x8 pixel
 * blocks the local pixel order walks. */
#define PTR_BAND_ROWS 8u

/*
 * Render the subset of PTR_BAND_ROWS-row image bands owned by `part_index` of `part_count`
 * (band b belongs to part b % part_count) into a DEVICE buffer laid out
 * [localBand][PTR_BAND_ROWS][width][3] floats; `d_out_rgb` must hold
 * ptr_part_band_count(height, part_index, part_count)*PTR_BAND_ROWS*width*3 floats.
 * `stream` is a hipStream_t (NULL = default stream).  Asynchronous unless
 * stats != NULL (stats need a stream sync to read event timers / counters).
 * count_traversal: bit 0 selects the counting build of the same kernels; bit 1 runs the path-slot pool as a single
 * group on the caller's stream (no concurrent kernels: clean per-kernel timings for profiling).
 */
int ptr_render_bands_device(PtrDeviceScene* scene, const PtrSettings* settings, uint32_t spp,
                            uint32_t part_index, uint32_t part_count, void* d_out_rgb, void* stream,
                            int count_traversal, PtrRenderStats* stats, char* err, size_t err_cap);
uint32_t ptr_part_band_count(uint32_t height, uint32_t part_index, uint32_t part_count);

/* Same partition, host output buffer of the same layout (used by ptr_render and tests). */
int ptr_render_bands(PtrDeviceScene* scene, const PtrSettings* settings, uint32_t spp,
                     uint32_t part_index, uint32_t part_count, float* out_rgb_bands,
                     int count_traversal, PtrRenderStats* stats, char* err, size_t err_cap);

/* First-hit feature buffers ("AOVs") of the whole frame, the inputs the reference gives its denoiser
 * (shaders/pathtrace.metal:6424-6435, 9813-9815; src/renderer/Accumulation.mm:130): for every pixel the camera ray of
 * sample `sample_index` (same jitter stream as the path tracer) is traced to its closest hit.
 * out_albedo: width*height*4 floats {base colour rgb, 1 if hit else 0}; out_normal: width*height*4 floats
 * {shading normal * 0.5 + 0.5, hit distance (0 on a miss)}; either may be NULL.  Row 0 = top. */
int ptr_render_aovs(PtrDeviceScene* scene, const PtrSettings* settings, uint32_t sample_index, float* out_albedo,
                    float* out_normal, char* err, size_t err_cap);

/* Closest-hit (any_hit = 0) or occlusion (any_hit = 1) queries for a ray batch.
 * rays: n * 8 floats {ox,oy,oz,tmin,dx,dy,dz,tmax}; out: n PtrHit (any-hit: t>=0 means occluded). */
int ptr_trace_rays(PtrDeviceScene* scene, const float* rays, uint64_t n, int any_hit, PtrHit* out,
                   PtrRenderStats* stats, char* err, size_t err_cap);

/* ---- host scene layer (no GPU needed) ---- */

typedef struct PtrHostScene PtrHostScene;

/* Parse a .scene file (reference grammar, SceneManager.mm:791-2633). asset_dir resolves relative
 * mesh/env paths (the reference's "scene directory"); NULL -> directory of the .scene file. */
int ptr_host_scene_load(const char* scene_path, const char* asset_dir, PtrHostScene** out,
                        char* err, size_t err_cap);
void ptr_host_scene_free(PtrHostScene* scene);
/* Pointers stay valid until ptr_host_scene_free. */
int ptr_host_scene_desc(const PtrHostScene* scene, PtrSceneDesc* out_desc, PtrSettings* out_settings);

/* format: "pfm" | "exr" | "ppm" | "png" (src/renderer/ImageWriter.mm:164-214, 239-464, 480-565); EXR = RGB
 * (channels B,G,R) unless rgba != 0 (B,G,R,A + colorspace attr, what the reference's Embree backend writes,
 * main_headless.mm:568-583).  PNG/PPM are tonemapped 8-bit (PNG: RGBA, alpha 255, sRGB chunk). */
int ptr_host_write_image(const char* path, const char* format, const float* linear_rgb,
                         uint32_t width, uint32_t height, int rgba_exr,
                         uint32_t tonemap_mode, uint32_t aces_variant, float exposure, float reinhard_white,
                         char* err, size_t err_cap);
/* RGBA EXR plus a planar SAMPLES channel holding per-pixel sample counts (ImageWriter::WriteEXR_Multilayer,
 * src/renderer/ImageWriter.mm:657-684).  sample_counts: width*height floats, or NULL for plain RGBA. */
int ptr_host_write_exr_multilayer(const char* path, const float* linear_rgb, uint32_t width, uint32_t height,
                                  const float* sample_counts, const char* colorspace, char* err, size_t err_cap);
/* Beauty image + the first-hit feature buffers of ptr_render_aovs as layers of one scanline EXR (channels R G B, albedo.R/G/B,
 * normal.X/Y/Z decoded to unit vectors, depth.Z): the inputs the reference gives its denoiser (src/renderer/Accumulation.mm:130,
 * shaders/pathtrace.metal:9813-9815), written with the layer convention of ImageWriter::WriteEXR_Multilayer (ImageWriter.mm:657-684). */
int ptr_host_write_exr_aovs(const char* path, const float* linear_rgb, const float* albedo_rgba, const float* normal_rgba,
                            uint32_t width, uint32_t height, char* err, size_t err_cap);
/* PNG / baseline JPEG bytes -> 8-bit RGBA (row 0 = top), the decoders the glTF loader uses for material textures
 * (csrc/host/image_decoders.h; the reference leaves this to MTKTextureLoader, src/renderer/SceneResources.mm:213-420).
 * out_rgba may be NULL to query the size. */
int ptr_host_decode_image(const uint8_t* data, uint64_t size, uint8_t* out_rgba, uint64_t cap_bytes, uint32_t* width, uint32_t* height,
                          char* err, size_t err_cap);
int ptr_host_read_pfm(const char* path, float* out_rgb, uint32_t cap_floats, uint32_t* width, uint32_t* height);

const char* ptr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTR_ABI_H */
