/*
 * ptr_debug.h — test-only entry points of libptr_hip.so: run single device functions of the integrator on a
 * batch of inputs so the parity tests can compare them with the oracle function by function
 * (SURVEY.md section 8(c) "fixtures": Rng::Hash, BuildCamera/GenerateCameraRay, EvaluateBsdf, SampleBsdf).
 * Reference twins: src/headless/EmbreeHeadlessRenderer.mm 52-68, 150-232, 1315-1491, 1493-1918.
 */
#ifndef PTR_DEBUG_H
#define PTR_DEBUG_H

#include "ptr_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* in: n*12 floats {position, normal, wo, wi}; out: n*5 floats {value rgb, pdf, isDelta} */
int ptr_debug_eval_bsdf(const PtrMaterial* material, const PtrSettings* settings, const float* in, uint64_t n,
                        float* out, char* err, size_t err_cap);
/* in: n*9 floats {position, normal, wo} (incident = -wo); out: n*8 floats {direction, weight, pdf, isDelta} */
int ptr_debug_sample_bsdf(const PtrMaterial* material, const PtrSettings* settings, const float* in,
                          const uint32_t* front_face, const uint32_t* rng_states, uint64_t n, float* out,
                          uint32_t* out_states, char* err, size_t err_cap);
/* xys: n*3 {x, y, sample}; out: n*6 floats {origin, direction}; out_states: rng state after ray generation */
int ptr_debug_camera_rays(const PtrSettings* settings, const uint32_t* xys, uint64_t n, float* out,
                          uint32_t* out_states, char* err, size_t err_cap);

/* One sample per pixel through the counting build of the render kernels, with every pixel's path signature
 * (csrc/kernels/device_types.h kSig*: bits 0..15 = which path vertices received a rectangle-light sample that contributed,
 * bits 16..31 = hash chain over the primitives hit).  The deterministic-stream tests compare it with the oracle's to say
 * why a pixel differs.  out_rgb (width*height*3, may be NULL), out_signature (width*height). */
int ptr_debug_render_signatures(PtrDeviceScene* scene, const PtrSettings* settings, float* out_rgb, uint32_t* out_signature,
                                char* err, size_t err_cap);

/* The texture filtering rule of the textured metallic-roughness model (csrc/kernels/texture.h) on a batch: in n * 3 floats
 * {u, v, lod}, out n * 4 floats RGBA (-1 in every channel when the scene has no such texture). */
int ptr_debug_texture_sample(PtrDeviceScene* scene, uint32_t texture, const float* in, uint64_t n, float* out, char* err, size_t err_cap);

/* Closest hit, surface record and next-ray origin for a batch of rays: what k_shade reconstructs at a hit (interpolated shading normal
 * flipped to the geometric side, shaders/pathtrace.metal:597-638 / EmbreeHeadlessRenderer.mm:2348-2367) and where it starts the next ray
 * (offset_ray_origin, pathtrace.metal:1196-1208 / OffsetRayOrigin, EmbreeHeadlessRenderer.mm:917-931).
 * in: n x 9 floats {origin, direction, next direction}; out: n x 16 floats {hit (1/0), t, position xyz, geometric normal xyz,
 * shading normal xyz, front face (1/0), next origin xyz, 0}. */
int ptr_debug_surface_hits(PtrDeviceScene* scene, const float* in, uint64_t n, float* out, char* err, size_t err_cap);

/* Host-side count of what a closest-hit walk of the scene's BVH costs with 1, 2 or 3 binary levels collapsed per step (2-, 4-, 8-wide nodes,
 * children in order of entry distance): rays n x 8 {origin, tmin, direction, tmax}; out = {node steps, box tests, primitive tests, rays that
 * hit}.  No GPU involved. */
int ptr_debug_walk_counts(const PtrSceneDesc* scene, const float* rays, uint64_t n, uint32_t levels, uint64_t out[4], char* err, size_t err_cap);

/* The kernels' exact division by a per-render divisor (csrc/kernels/device_types.h DivU32), evaluated on the host: out[i] = n[i] / d. */
int ptr_debug_exact_division(uint32_t d, const uint32_t* n, uint64_t count, uint32_t* out);

/* Which k_shade instantiation a render of `scene` with `settings` launches: bit t (0..7) = material type t compiled in, bit 8 = the
 * environment map, bit 9 = the Metal medium stack / face-normal rule; 0x3FF = the full kernel.  (The product picks the smallest
 * compiled set that covers the scene: the tests check that every set renders what the full kernel renders.) */
int ptr_debug_shade_kernel_set(const PtrDeviceScene* scene, const PtrSettings* settings, int count, uint32_t* out);

/* ptr_render_multi on an explicit list of devices; an id may appear more than once, which lets a one-GPU box run the whole
 * multi-device path (threads, partitions, hand-over, interleave).  An id given as -(id + 1) routes that partition's bands through the
 * pinned-host staging copy ptr_render_multi falls back to when two devices cannot address each other (hipDeviceCanAccessPeer). */
int ptr_debug_render_multi_on(const PtrSceneDesc* scene, const PtrSettings* settings, uint32_t spp, const int* device_ids, int n,
                              float* out_rgb, PtrRenderStats* stats, char* err, size_t err_cap);

/* Host-side (no GPU): the environment importance tables the device sampler is fed
 * (src/renderer/EnvImportanceSampler.mm:70-171).  Outputs sized by the caller: texel_pdf, cond_alias,
 * cond_threshold: w*h; marg_alias, marg_threshold: h.  Returns non-zero if the map has no positive radiance. */
int ptr_debug_env_distribution(const float* rgba, uint32_t w, uint32_t h, float* texel_pdf, uint32_t* cond_alias,
                               float* cond_threshold, uint32_t* marg_alias, float* marg_threshold, float* total_weight);

/* Host-side (no GPU): run the geometry preparation ptr_scene_upload performs (world-space bake, BVH build,
 * leaf-order flattening; src/renderer/SceneAccel.mm:23-325 is the reference's counterpart) and walk the result.
 * out[0..15]: nodes, leaves, triangles referenced, spheres referenced, max depth, max leaf size, unreferenced
 * primitives, multiply referenced primitives, box containment violations, quantised-box violations, bad child
 * references, triangle count, sphere count, SAH cost * 1000, build milliseconds (gather+build+flatten), quantised
 * nodes usable (grid fine enough; bits 8..15: triangles kept out of the tree so that it is; bits 16..47: four-wide nodes of the
 * persistent kernels; bit 63: those nodes have a bad reference or do not reach every primitive exactly once).  leaf_max = 0 uses the default.  Returns non-zero with a message on bad input. */
int ptr_debug_scene_geometry(const PtrSceneDesc* scene, uint32_t leaf_max, uint64_t out[16], char* err, size_t err_cap);

/* Host-side (no GPU): the tangent generator behind glTF primitives without TANGENT (csrc/host/tangent_space.cpp, the MikkTSpace method;
 * reference: src/assets/TangentGen.mm:181-230 over external/MikkTSpace/mikktspace.c).  A triangle soup - corner c of triangle f is
 * element 3 f + c of positions (xyz), unit normals (xyz) and uvs (st); out_tangents: xyz + sign per corner.  0 on success. */
int ptr_debug_generate_tangents(const float* positions, const float* normals, const float* uvs, uint64_t triangle_count, float* out_tangents);

#ifdef __cplusplus
}
#endif
#endif /* PTR_DEBUG_H */
