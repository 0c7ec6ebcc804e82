// Material evaluation / sampling on the device: the eight material models of the reference with the
// Embree path's semantics (half-vector GGX sampling with pdf D*cos/(4 wo.wh), specular-tail clamp, no energy
// compensation, dielectric weights not divided by the pick probability, subsurface == Lambert, factor-only PBR).
// Oracle twins: src/headless/EmbreeHeadlessRenderer.mm EvaluateBsdf 1315-1491, SampleBsdf 1493-1918, helpers
// 312-885, car paint 1170-1313 (Metal counterparts: shaders/pathtrace.metal evaluate_bsdf 4950, sample_bsdf 5136).
// Random numbers are consumed in the oracle's order so a path follows the same stream.
#pragma once

#include "device_types.h"
#include "vec.h"

namespace ptrk {

struct ClampCfg {
    float factor, floorLum, throughput, tailBase, tailRoughScale, minSpecPdf;
    bool enabled;
    bool thinDielectrics;   // PTR_METAL_THIN: honour the thin-walled flag of dielectrics (Metal semantics)
    bool metalSpecular;     // PTR_METAL_SPECULAR: VNDF sampling, G1 pdf and energy compensation for rough metals
    bool metalSss;          // PTR_METAL_SSS: type 5 evaluates to zero (no NEE); separable diffusion sampling when sssMode == 1
    uint32_t sssMode;
    bool metalPbr;          // PTR_METAL_PBR: three-lobe metallic-roughness model of the Metal integrator (compiled with SSS = true)
    bool metalClamps;       // PTR_METAL_CLAMPS: the Metal kernel's variants of the three clamps below (pathtrace.metal:3550-3633)
    float maxContribution;  // fireflyClampMaxContribution (>= 0), read only with metalClamps
    float minSpecPdfRaw;    // minSpecularPdf as given (the Embree variant floors it at 1e-8), read only with metalClamps
};

__device__ __forceinline__ uint32_t rngHash(uint32_t x) {  // lowbias32
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float rngNext(uint32_t& state) {
    state = rngHash(state);
    return static_cast<float>(state & 0x00FFFFFFu) / 16777216.0f;
}

__device__ __forceinline__ float luminance(f3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }

// Material record view: 16 float4, fetched on demand (most branches need 2-4 of them).
struct Mat {
    const float4* p;
    // per-hit values of the textured metallic-roughness model (shaders/pathtrace.metal:6391-6394 writes them back into its copy
    // of the material): [0] base colour | roughness, [1] emission, [2] = (metallic, transmission, diffuse occlusion, 0).
    // Null everywhere else - and always in the kernels compiled without the Metal-only models.
    const float4* o = nullptr;
    __device__ __forceinline__ float4 v(uint32_t slot) const {
        if (o) {
            if (slot == kMatBaseColorRoughness) return o[0];
            if (slot == kMatEmission) return o[1];
            if (slot == kMatCoatTint) {
                float4 r = p[slot];
                r.w = o[2].x;
                return r;
            }
            if (slot == kMatDielectricSigmaA) {
                float4 r = p[slot];
                r.w = o[2].y;
                return r;
            }
        }
        return p[slot];
    }
    __device__ __forceinline__ float occlusion() const { return o ? o[2].z : 1.0f; }
    __device__ __forceinline__ uint32_t type() const { return static_cast<uint32_t>(p[kMatTypeEta].x); }
    __device__ __forceinline__ f3 baseColor() const { return vclamp(mk3(v(kMatBaseColorRoughness)), 0.0f, 1.0f); }
    __device__ __forceinline__ float roughness01() const { return clampf(v(kMatBaseColorRoughness).w, 0.0f, 1.0f); }
    __device__ __forceinline__ float ior() const { return p[kMatTypeEta].y; }
    __device__ __forceinline__ bool thinFlag() const { return p[kMatTypeEta].w > 0.5f; }   // material_is_thin_dielectric
    __device__ __forceinline__ f3 sigmaA() const { return vmax0(mk3(p[kMatDielectricSigmaA])); }
};

// ------------------------------------------------------------------ clamps
__device__ __forceinline__ f3 clampFirefly(f3 throughput, f3 contribution, const ClampCfg& c) {
    f3 combined = throughput * contribution;
    if (!finite3(combined)) return mk3(0.0f);
    f3 positive = vmax0(combined);
    if (!c.enabled) return positive;
    const float lum = luminance(positive);
    float maxLum = smax(luminance(vmax0(throughput)) * c.factor, c.floorLum);
    if (c.metalClamps && c.maxContribution > 0.0f) maxLum = smax(maxLum, c.maxContribution);   // pathtrace.metal:3563-3568
    if (lum > maxLum && lum > 0.0f) {
        combined *= maxLum / smax(lum, 1.0e-6f);
        positive = vmax0(combined);
    }
    return positive;
}

__device__ __forceinline__ float clampSpecPdf(float pdf, const ClampCfg& c) {
    if (c.metalClamps) {   // pathtrace.metal:3579-3590
        if (!isfinite(pdf) || pdf <= 0.0f) return 0.0f;
        return c.minSpecPdfRaw <= 0.0f ? pdf : smax(pdf, c.minSpecPdfRaw);
    }
    const float minPdf = smax(c.minSpecPdf, 1.0e-8f);
    return isfinite(pdf) ? smax(pdf, minPdf) : minPdf;
}

__device__ __forceinline__ f3 clampThroughput(f3 t, const ClampCfg& c) {
    if (!finite3(t)) return mk3(0.0f);
    if (!c.enabled || c.throughput <= 0.0f) return t;
    const float lum = luminance(vmax0(t));
    if (lum > c.throughput && lum > 0.0f) return t * (c.throughput / smax(lum, 1.0e-6f));
    return t;
}

__device__ __forceinline__ f3 clampSpecTail(f3 value, float roughness, f3 f0, const ClampCfg& c) {
    if (!finite3(value)) return mk3(0.0f);
    f3 positive = vmax0(value);
    if (!c.enabled) return positive;
    if (c.metalClamps && c.tailBase <= 0.0f && c.tailRoughScale <= 0.0f) return positive;   // pathtrace.metal:3619-3621
    const float strength = smax(smax(f0.x, f0.y), smax(f0.z, 1.0e-3f));
    const float limit = smax((c.tailBase + c.tailRoughScale * roughness) * strength, c.floorLum);
    const float lum = luminance(positive);
    if (lum > limit && lum > 0.0f) positive *= limit / smax(lum, 1.0e-6f);
    return positive;
}

// ------------------------------------------------------------------ frames, Fresnel, GGX
struct Frame {
    f3 t, b, n;
};

__device__ __forceinline__ Frame makeFrame(f3 n) {
    Frame f;
    f.n = normalize(n);
    const f3 up = (fabsf(f.n.z) < 0.999f) ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    f.t = normalize(cross(up, f.n));
    f.b = cross(f.n, f.t);
    return f;
}

__device__ __forceinline__ f3 frameToWorld(f3 l, const Frame& f) { return (l.x * f.t + l.y * f.b) + l.z * f.n; }

__device__ __forceinline__ f3 reflectDir(f3 v, f3 n) { return v - 2.0f * dot(v, n) * n; }

__device__ __forceinline__ f3 cosineHemisphere(uint32_t& rng, f3 n, float& pdf) {
    const float r1 = rngNext(rng);
    const float r2 = rngNext(rng);
    const float r = sqrtf(smax(r1, 0.0f));
    const float phi = 2.0f * kPi * r2;
    const float x = cosf(phi) * r;
    const float y = sinf(phi) * r;
    const float z = sqrtf(smax(1.0f - r1, 0.0f));
    pdf = z / kPi;
    return normalize(frameToWorld(mk3(x, y, z), makeFrame(n)));
}

__device__ __forceinline__ float schlickW(float cosTheta) {
    const float m = clampf(1.0f - cosTheta, 0.0f, 1.0f);
    const float m2 = m * m;
    return m2 * m2 * m;
}

__device__ __forceinline__ f3 schlick(f3 f0, float cosTheta) { return f0 + (mk3(1.0f) - f0) * schlickW(cosTheta); }

__device__ __forceinline__ float fresnelDielectric(float cosI, float etaI, float etaT, float& cosTOut) {
    cosI = clampf(cosI, -1.0f, 1.0f);
    const float ac = fabsf(cosI);
    const float sinI2 = smax(0.0f, 1.0f - ac * ac);
    const float eta = etaI / etaT;
    const float sinT2 = eta * eta * sinI2;
    if (sinT2 >= 1.0f) {
        cosTOut = 0.0f;
        return 1.0f;
    }
    const float cosT = sqrtf(smax(0.0f, 1.0f - sinT2));
    cosTOut = cosT;
    const float a = etaI * ac, b = etaT * cosT;
    const float rs = (a - b) / (a + b);
    const float rp = (etaT * ac - etaI * cosT) / (etaT * ac + etaI * cosT);
    return 0.5f * (rs * rs + rp * rp);
}

__device__ __forceinline__ f3 fresnelConductor(float cosI, f3 eta, f3 k) {
    cosI = clampf(cosI, -1.0f, 1.0f);
    const float cos2 = cosI * cosI;
    const float sin2 = smax(0.0f, 1.0f - cos2);
    const f3 eta2 = eta * eta, k2 = k * k;
    const f3 t0 = (eta2 - k2) - mk3(sin2);
    const f3 a2b2 = vsqrt(vmax0(t0 * t0 + (4.0f * eta2) * k2));
    const f3 a = vsqrt(vmax0(0.5f * (a2b2 + t0)));
    const f3 term1 = a2b2 + mk3(cos2);
    const f3 term2 = (2.0f * mk3(cosI)) * a;
    const f3 rs = (term1 - term2) / (term1 + term2);
    const f3 term3 = mk3(cos2) * a2b2 + mk3(sin2 * sin2);
    const f3 term4 = term2 * mk3(sin2);
    const f3 rp = (term3 - term4) / (term3 + term4);
    return vclamp(0.5f * (rs * rs + rp * rp), 0.0f, 1.0f);
}

__device__ __forceinline__ float ggxLambda(float alpha, float cosTheta) {
    const float ac = fabsf(cosTheta);
    if (ac <= 0.0f) return 0.0f;
    const float sinTheta = sqrtf(smax(0.0f, 1.0f - ac * ac));
    if (sinTheta == 0.0f) return 0.0f;
    const float a = alpha * (sinTheta / ac);
    return (-1.0f + sqrtf(1.0f + a * a)) * 0.5f;
}

__device__ __forceinline__ float ggxG1(float alpha, float cosTheta) { return 1.0f / (1.0f + ggxLambda(alpha, cosTheta)); }

__device__ __forceinline__ float ggxD(float alpha, float cosH) {
    const float c = fabsf(cosH);
    const float a2 = alpha * alpha;
    const float denom = c * c * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * denom * denom);
}

__device__ __forceinline__ float ggxPdf(float alpha, f3 n, f3 wo, f3 wi) {
    const f3 wh = normalize(wo + wi);
    const float cosH = dot(n, wh);
    const float denom = 4.0f * smax(dot(wo, wh), 1.0e-6f);
    return ggxD(alpha, smax(cosH, 0.0f)) * smax(cosH, 0.0f) / denom;
}

// Metal-only (PTR_METAL_SPECULAR): pdf of reflecting a VNDF-sampled half vector, shaders/pathtrace.metal:3724-3739
__device__ __forceinline__ float ggxPdfVisible(float alpha, f3 n, f3 wo, f3 wi) {
    const f3 wh = normalize(wo + wi);
    const float cosH = dot(n, wh), woh = dot(wo, wh), cosO = dot(n, wo);
    if (cosO <= 0.0f || cosH <= 0.0f || woh <= 0.0f) return 0.0f;
    return ggxD(alpha, cosH) * ggxG1(alpha, cosO) * cosH / (4.0f * smax(woh, 1.0e-6f));
}

// Metal-only: multiple-scattering compensation of a specular lobe, shaders/pathtrace.metal:4610-4630
__device__ __forceinline__ f3 specularEnergyCompensation(f3 f0, float roughness, float nov) {
    const float n = clampf(nov, 0.0f, 1.0f);
    const float rx = roughness * -1.0f + 1.0f, ry = roughness * -0.0275f + 0.0425f;
    const float rz = roughness * -0.572f + 1.04f, rw = roughness * 0.022f + -0.04f;
    const float a004 = smin(rx * rx, exp2f(-9.28f * n)) * rx + ry;
    const float dfgX = -1.04f * a004 + rz, dfgY = 1.04f * a004 + rw;
    float out[3];
    const float f[3] = {f0.x, f0.y, f0.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float fss = clampf(f[c] * dfgX + dfgY, 0.0f, 0.99f);
        const float favg = f[c] + (1.0f - f[c]) * (1.0f / 21.0f);
        const float oneMinus = clampf(1.0f - fss, 0.0f, 1.0f);
        const float fms = (favg * oneMinus) / smax(1.0f - favg * oneMinus, 1.0e-3f);
        out[c] = clampf((fss + fms) / smax(fss, 1.0e-4f), 1.0f, 2.0f);
    }
    return mk3(out[0], out[1], out[2]);
}

__device__ __forceinline__ f3 sampleGgxHalf(uint32_t& rng, float alpha, f3 n) {
    const float u1 = rngNext(rng);
    const float u2 = rngNext(rng);
    const float phi = 2.0f * kPi * u1;
    const float denom = 1.0f + (alpha * alpha - 1.0f) * u2;
    const float cosT = sqrtf(smax((1.0f - u2) / smax(denom, 1.0e-6f), 0.0f));
    const float sinT = sqrtf(smax(0.0f, 1.0f - cosT * cosT));
    return normalize(frameToWorld(mk3(cosf(phi) * sinT, sinf(phi) * sinT, cosT), makeFrame(n)));
}

__device__ __forceinline__ f3 sampleGgxVndf(uint32_t& rng, float roughness, f3 n, f3 wo) {
    const float alpha = smax(roughness * roughness, 1.0e-4f);
    const Frame fr = makeFrame(n);
    const f3 won = normalize(wo);
    f3 wl = mk3(dot(won, fr.t), dot(won, fr.b), dot(won, fr.n));
    wl.z = smax(wl.z, 1.0e-6f);
    const f3 vh = normalize(mk3(alpha * wl.x, alpha * wl.y, wl.z));
    const float lensq = vh.x * vh.x + vh.y * vh.y;
    const f3 t1 = lensq > 0.0f ? mk3(-vh.y, vh.x, 0.0f) / sqrtf(lensq) : mk3(1.0f, 0.0f, 0.0f);
    const f3 t2 = cross(vh, t1);
    const float u1 = rngNext(rng);
    const float u2 = rngNext(rng);
    const float r = sqrtf(u1);
    const float phi = 2.0f * kPi * u2;
    const float t1r = r * cosf(phi);
    const float t2r = r * sinf(phi);
    const float s = 0.5f * (1.0f + vh.z);
    const float t2a = (1.0f - s) * sqrtf(smax(0.0f, 1.0f - t1r * t1r)) + s * t2r;
    const float t3 = sqrtf(smax(0.0f, (1.0f - t1r * t1r) - t2a * t2a));
    const f3 nh = (t1r * t1 + t2a * t2) + t3 * vh;
    const f3 ne = normalize(mk3(alpha * nh.x, alpha * nh.y, smax(nh.z, 0.0f)));
    return normalize(frameToWorld(ne, fr));
}

__device__ __forceinline__ float lambertPdf(f3 n, f3 dir) {
    const float c = smax(dot(n, normalize(dir)), 0.0f);
    return c > 0.0f ? (c / kPi) : 0.0f;
}

__device__ __forceinline__ f3 microfacet(f3 F, float alpha, f3 n, f3 wh, float cosO, float cosI) {
    const float D = ggxD(alpha, dot(n, wh));
    const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
    return F * (D * G / smax(4.0f * cosO * cosI, 1.0e-6f));
}

__device__ __forceinline__ bool halfOk(f3 wh, f3 n, f3 wo, f3 wi) {
    return dot(wh, n) > 0.0f && dot(wo, wh) > 0.0f && dot(wi, wh) > 0.0f;
}

__device__ __forceinline__ float dielectricF0(float ior) {
    const float eta = smax(ior, 1.0f);
    const float r = (eta - 1.0f) / (eta + 1.0f);
    return r * r;
}

__device__ __forceinline__ float dielectricF0Clamped(float ior) {
    const float eta = smax(ior, 1.0f);
    const float q = (eta - 1.0f) / smax(eta + 1.0f, 1.0e-6f);
    return clampf(q * q, 0.0f, 0.99f);
}

__device__ __forceinline__ f3 exp3(f3 v) { return mk3(expf(v.x), expf(v.y), expf(v.z)); }
__device__ __forceinline__ f3 fract3(f3 v) { return mk3(v.x - floorf(v.x), v.y - floorf(v.y), v.z - floorf(v.z)); }

// ------------------------------------------------------------------ coat helpers (plastic / car paint)
struct Coat {
    float roughness, thickness, sampleWeight, fresnelAvg, ior;
    f3 tint, absorption;
};

__device__ __forceinline__ Coat loadCoat(const Mat& m) {
    Coat c;
    const float4 cp = m.v(kMatCoatParams);
    c.roughness = smax(clampf(cp.x, 0.0f, 1.0f), 1.0e-3f);
    c.thickness = smax(cp.y, 0.0f);
    c.sampleWeight = cp.z;
    c.fresnelAvg = clampf(cp.w, 0.0f, 1.0f);
    c.ior = smax(m.ior(), 1.0f);
    c.tint = vclamp(mk3(m.v(kMatCoatTint)), 0.0f, 1.0f);
    c.absorption = vmax0(mk3(m.v(kMatCoatAbsorption)));
    return c;
}

__device__ __forceinline__ f3 coatSpecTint(const Coat& c) {
    if (c.thickness <= 0.0f) return c.tint;
    if (c.absorption.x <= 1.0e-6f && c.absorption.y <= 1.0e-6f && c.absorption.z <= 1.0e-6f) return c.tint;
    return vclamp(c.tint * exp3(-c.absorption * c.thickness), 0.0f, 1.0f);
}

__device__ __forceinline__ f3 coatDiffuseTransmission(const Coat& c, float cosI, float cosO) {
    if (c.thickness <= 0.0f) return c.tint;
    const float ci = smax(cosI, 1.0e-3f), co = smax(cosO, 1.0e-3f);
    const f3 ai = exp3(-c.absorption * (c.thickness / ci));
    const f3 ao = exp3(-c.absorption * (c.thickness / co));
    return vclamp((c.tint * ai) * ao, 0.0f, 1.0f);
}

// Plastic = GGX coat + Fresnel/absorption attenuated Lambert base.
__device__ __forceinline__ void plasticLobes(const Mat& m, const Coat& c, f3 n, f3 wo, f3 wi, float cosO, float cosI,
                                             const ClampCfg& cc, f3& spec, float& specPdf, f3& diffuse) {
    const float alpha = c.roughness * c.roughness;
    const f3 f0 = mk3(dielectricF0(c.ior));
    spec = mk3(0.0f);
    specPdf = 0.0f;
    const f3 wh = normalize(wo + wi);
    if (halfOk(wh, n, wo, wi)) {
        spec = microfacet(schlick(f0, dot(wi, wh)), alpha, n, wh, cosO, cosI);
        spec = clampSpecTail(spec, c.roughness, f0, cc);
        spec *= coatSpecTint(c);
        const float pdf = ggxPdf(alpha, n, wo, wi);
        if (pdf > 0.0f) specPdf = clampSpecPdf(pdf, cc);
        spec = vmax0(spec);
    }
    diffuse = m.baseColor() / kPi;
    diffuse *= coatDiffuseTransmission(c, cosI, cosO);
    diffuse *= (mk3(1.0f) - schlick(f0, cosI)) * (mk3(1.0f) - schlick(f0, cosO));
    diffuse *= smax(1.0f - c.fresnelAvg, 0.0f);
    diffuse = vmax0(diffuse);
}

// ------------------------------------------------------------------ conductors
__device__ __forceinline__ bool hasConductorIor(float4 eta, float4 k) {
    return eta.w > 0.0f || k.w > 0.0f || eta.x > 0.0f || eta.y > 0.0f || eta.z > 0.0f || k.x > 0.0f || k.y > 0.0f || k.z > 0.0f;
}

struct Conductor {
    bool spectral;
    f3 eta, k, f0;
};

__device__ __forceinline__ Conductor loadMetal(const Mat& m) {
    Conductor c;
    const float4 e = m.v(kMatConductorEta), k = m.v(kMatConductorK);
    c.spectral = hasConductorIor(e, k);
    c.eta = mk3(e);
    c.k = mk3(k);
    c.f0 = c.spectral ? fresnelConductor(1.0f, c.eta, c.k) : m.baseColor();
    return c;
}

__device__ __forceinline__ f3 conductorF(const Conductor& c, float cosTheta) {
    return c.spectral ? fresnelConductor(cosTheta, c.eta, c.k) : schlick(c.f0, cosTheta);
}

// ------------------------------------------------------------------ car paint
struct CarPaint {
    Coat coat;
    float baseMetallic, baseRoughness, flakeScale;
    float flakeWeight, flakeRoughness, flakeAniso, flakeStrength;
    bool baseConductor;
    f3 baseEta, baseK, base;
    float pCoat, pFlake, pBase;
};

__device__ __forceinline__ CarPaint loadCarPaint(const Mat& m) {
    CarPaint c;
    c.coat = loadCoat(m);
    const float4 bp = m.v(kMatCarpaintBase), fp = m.v(kMatCarpaintFlake);
    const float4 be = m.v(kMatCarpaintBaseEta), bk = m.v(kMatCarpaintBaseK);
    c.baseMetallic = clampf(bp.x, 0.0f, 1.0f);
    c.baseRoughness = clampf(bp.y, 0.0f, 1.0f);
    c.flakeScale = smax(bp.z, 1.0e-4f);
    c.flakeWeight = clampf(fp.x, 0.0f, 0.95f);
    c.flakeRoughness = clampf(fp.y, 0.0f, 1.0f);
    c.flakeAniso = clampf(fp.z, -0.99f, 0.99f);
    c.flakeStrength = clampf(fp.w, 0.0f, 1.0f);
    c.baseConductor = be.w > 0.0f || bk.w > 0.0f;
    c.baseEta = vmax0(mk3(be));
    c.baseK = vmax0(mk3(bk));
    c.base = m.baseColor();
    // lobe pick probabilities (coat weight is clamped to 0.95 here, 1.0 for plastic)
    float pc = clampf(c.coat.sampleWeight, 0.0f, 0.95f);
    float pf = c.flakeWeight;
    float pb = smax(1.0f - (pc + pf), 0.0f);
    float norm = (pc + pf) + pb;
    if (norm <= 1.0e-6f) {
        pb = 1.0f;
        pc = 0.0f;
        pf = 0.0f;
        norm = 1.0f;
    }
    c.pCoat = pc / norm;
    c.pFlake = pf / norm;
    c.pBase = pb / norm;
    return c;
}

__device__ __forceinline__ f3 carpaintBaseF0(const CarPaint& c) {
    return c.baseConductor ? fresnelConductor(1.0f, c.baseEta, c.baseK) : c.base;
}

__device__ __forceinline__ f3 carpaintHash3(f3 p) {
    f3 v = fract3(p * 0.3183099f + mk3(0.1f, 0.3f, 0.7f));
    const float d = dot(v, mk3(v.y + 33.33f, v.z + 55.55f, v.x + 77.77f));
    v += mk3(d);
    return fract3(mk3(v.x + v.y, v.x + v.z, v.y + v.z) * 13.5453123f);
}

__device__ __forceinline__ f3 carpaintFlakeNormal(const CarPaint& c, f3 position, f3 n) {
    const f3 rnd = carpaintHash3(position * c.flakeScale);
    const float ax = smax(1.0f - c.flakeAniso, 1.0e-3f);
    const float ay = smax(1.0f + c.flakeAniso, 1.0e-3f);
    const float phi = 2.0f * kPi * rnd.x;
    const float r = sqrtf(smax(rnd.y, 1.0e-4f));
    const float x = r * cosf(phi) * ax;
    const float y = r * sinf(phi) * ay;
    const float m2 = clampf(x * x + y * y, 0.0f, 0.99f);
    const float z = sqrtf(smax(1.0f - m2, 0.0f));
    const Frame fr = makeFrame(n);
    const f3 perturbed = normalize((x * fr.t + y * fr.b) + z * fr.n);
    return normalize(n * (1.0f - c.flakeStrength) + perturbed * c.flakeStrength);
}

struct Lobe {
    f3 value;
    float pdf;
};

__device__ __forceinline__ Lobe carpaintCoat(const CarPaint& c, f3 n, f3 wo, f3 wi, const ClampCfg& cc) {
    Lobe r{mk3(0.0f), 0.0f};
    const float cosO = smax(dot(n, wo), 0.0f), cosI = smax(dot(n, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const float alpha = c.coat.roughness * c.coat.roughness;
    const f3 wh = normalize(wo + wi);
    if (!halfOk(wh, n, wo, wi)) return r;
    const f3 f0 = mk3(dielectricF0(c.coat.ior));
    f3 spec = microfacet(schlick(f0, dot(wi, wh)), alpha, n, wh, cosO, cosI);
    spec = clampSpecTail(spec, c.coat.roughness, f0, cc);
    spec *= coatSpecTint(c.coat);
    spec = vmax0(spec);
    const float pdf = ggxPdf(alpha, n, wo, wi);
    if (pdf > 0.0f) {
        r.pdf = clampSpecPdf(pdf, cc);
        r.value = spec;
    }
    return r;
}

__device__ __forceinline__ Lobe carpaintFlake(const CarPaint& c, f3 position, f3 n, f3 wo, f3 wi, const ClampCfg& cc) {
    Lobe r{mk3(0.0f), 0.0f};
    const f3 fn = carpaintFlakeNormal(c, position, n);
    const float cosO = smax(dot(fn, wo), 0.0f), cosI = smax(dot(fn, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const float rough = smax(c.flakeRoughness, 1.0e-3f);
    const float alpha = rough * rough;
    const f3 wh = normalize(wo + wi);
    if (!halfOk(wh, fn, wo, wi)) return r;
    const f3 f0 = carpaintBaseF0(c);
    f3 spec = microfacet(schlick(f0, dot(wi, wh)), alpha, fn, wh, cosO, cosI);
    spec = clampSpecTail(spec * coatSpecTint(c.coat), rough, f0, cc);
    spec *= smax(1.0f - c.coat.fresnelAvg, 0.0f);
    spec = vmax0(spec);
    const float pdf = ggxPdf(alpha, fn, wo, wi);
    if (pdf > 0.0f) {
        r.pdf = clampSpecPdf(pdf, cc);
        r.value = spec;
    }
    return r;
}

__device__ __forceinline__ Lobe carpaintBase(const CarPaint& c, f3 n, f3 wo, f3 wi, const ClampCfg& cc) {
    Lobe r{mk3(0.0f), 0.0f};
    const float cosO = smax(dot(n, wo), 0.0f), cosI = smax(dot(n, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const float dw = smax(1.0f - c.baseMetallic, 0.0f);
    const float sw = smax(c.baseMetallic, 0.0f);
    if (dw <= 1.0e-4f && sw <= 1.0e-4f) return r;
    f3 combined = mk3(0.0f);
    float pdfD = 0.0f, pdfS = 0.0f;
    if (dw > 1.0e-4f) {
        f3 diffuse = c.base / kPi;
        diffuse *= coatDiffuseTransmission(c.coat, cosI, cosO) * smax(1.0f - c.coat.fresnelAvg, 0.0f);
        diffuse = vmax0(diffuse);
        combined += dw * diffuse;
        pdfD = lambertPdf(n, wi);
    }
    if (sw > 1.0e-4f) {
        const float rough = smax(c.baseRoughness, 1.0e-3f);
        const float alpha = rough * rough;
        const f3 wh = normalize(wo + wi);
        if (halfOk(wh, n, wo, wi)) {
            const f3 f0 = c.baseConductor ? fresnelConductor(1.0f, c.baseEta, c.baseK) : c.base;
            const f3 F = c.baseConductor ? fresnelConductor(dot(wi, wh), c.baseEta, c.baseK) : schlick(c.base, dot(wi, wh));
            f3 spec = microfacet(F, alpha, n, wh, cosO, cosI);
            spec = clampSpecTail((spec * coatSpecTint(c.coat)) * smax(1.0f - c.coat.fresnelAvg, 0.0f), rough, f0, cc);
            spec = vmax0(spec);
            combined += sw * spec;
            const float pdf = ggxPdf(alpha, n, wo, wi);
            if (pdf > 0.0f) pdfS = clampSpecPdf(pdf, cc);
        }
    }
    r.value = vmax0(combined);
    r.pdf = dw * pdfD + sw * pdfS;
    return r;
}

// ------------------------------------------------------------------ PBR (factor only)
struct Pbr {
    f3 f0, diffuseColor;
    float roughness, specWeight;
};

__device__ __forceinline__ Pbr loadPbr(const Mat& m) {
    Pbr p;
    const f3 base = m.baseColor();
    const float metallic = clampf(m.v(kMatCoatTint).w, 0.0f, 1.0f);
    p.roughness = m.roughness01();
    const float df0 = dielectricF0Clamped(m.ior());
    p.f0 = base * metallic + mk3(df0) * (1.0f - metallic);
    p.diffuseColor = base * (1.0f - metallic);
    p.specWeight = clampf(smax(p.f0.x, smax(p.f0.y, p.f0.z)), 0.05f, 0.95f);
    return p;
}

// ------------------------------------------------------------------ public: delta test, eval, sample
__device__ __forceinline__ bool materialIsDelta(const Mat& m) {
    const uint32_t type = m.type();
    if (type == 2u) return true;
    if (type == 1u) return smax(m.roughness01(), 1.0e-3f) <= 1.0e-3f;
    return false;
}

struct BsdfEvalResult {
    f3 value;
    float pdf;
    bool isDelta;
};

// ---- metallic-roughness model of the Metal integrator (PTR_METAL_PBR): evaluate_ / sample_pbr_metallic_roughness,
// shaders/pathtrace.metal:4598-4948.  Factors only (no textures); restated line by line in the oracle as well.
struct PbrMetal {
    f3 f0, diffuseColor;
    float roughness, transmission, reflectScale, pSpec, pDiff, pTrans;
    bool valid;
};

__device__ __forceinline__ PbrMetal loadPbrMetal(const Mat& m) {   // :4656-4678 / 4789-4811
    PbrMetal p;
    const f3 base = m.baseColor();
    const float metallic = clampf(m.v(kMatCoatTint).w, 0.0f, 1.0f);
    p.roughness = m.roughness01();
    const f3 d0 = mk3(dielectricF0Clamped(m.ior()));
    p.f0 = d0 + (base - d0) * metallic;   // mix(dielectricF0, baseColor, metallic)
    p.diffuseColor = base * (1.0f - metallic);
    p.diffuseColor = p.diffuseColor * clampf(m.occlusion(), 0.0f, 1.0f);   // diffuseOcclusion (:4661; 1 without an occlusion texture)
    p.transmission = clampf(m.v(kMatDielectricSigmaA).w, 0.0f, 1.0f) * (1.0f - metallic);   // pbrExtras.z rides in this lane
    p.reflectScale = 1.0f - p.transmission;
    const float specWeightBase = clampf(smax(p.f0.x, smax(p.f0.y, p.f0.z)), 0.05f, 0.95f);
    const float wSpec = specWeightBase * p.reflectScale, wDiff = (1.0f - specWeightBase) * p.reflectScale, wTrans = p.transmission;
    const float sum = wSpec + wDiff + wTrans;
    p.valid = sum > 0.0f;
    p.pSpec = p.valid ? wSpec / sum : 0.0f;
    p.pDiff = p.valid ? wDiff / sum : 0.0f;
    p.pTrans = p.valid ? wTrans / sum : 0.0f;
    return p;
}

__device__ __forceinline__ f3 transmissionTint(const Mat& m, float cosTheta) {   // :3295-3306
    const float thickness = smax(m.v(kMatTypeEta).w, 0.0f);
    if (thickness <= 0.0f) return mk3(1.0f);
    const f3 sigmaA = m.sigmaA();
    if (sigmaA.x <= 0.0f && sigmaA.y <= 0.0f && sigmaA.z <= 0.0f) return mk3(1.0f);
    const float distance = thickness / smax(fabsf(cosTheta), 1.0e-3f);
    return vclamp(vexp(-(sigmaA * distance)), 0.0f, 1.0f);
}

__device__ __forceinline__ float ggxVndfPdf(float alpha, f3 n, f3 wo, f3 wh) {   // :3741-3754
    const float cosO = dot(n, wo), cosH = dot(n, wh);
    if (cosO <= 0.0f || cosH <= 0.0f) return 0.0f;
    return ggxD(alpha, cosH) * ggxG1(alpha, cosO) * cosH / smax(dot(wo, wh), 1.0e-6f);
}

// the rough transmission term shared by evaluation and sampling (:4729-4757 / 4906-4930); false = rejected
__device__ bool roughTransmission(const Mat& m, const PbrMetal& p, f3 n, f3 wo, f3 wi, f3 wh, float eta, float etaI, float etaT, f3& ft,
                                  float& pdfTrans) {
    const float absCosO = fabsf(dot(n, wo)), absCosI = fabsf(dot(n, wi));
    const float cosOWh = dot(wo, wh), cosIWh = dot(wi, wh);
    if (cosOWh * cosIWh > 0.0f) return false;
    const float alpha = smax(p.roughness * p.roughness, 1.0e-4f);
    const float D = ggxD(alpha, smax(dot(n, wh), 0.0f));
    const float G = ggxG1(alpha, absCosO) * ggxG1(alpha, absCosI);
    float cosT = 0.0f;
    const float F = fresnelDielectric(cosOWh, etaI, etaT, cosT);
    const float denom = cosOWh + eta * cosIWh;
    const float denomSq = denom * denom;
    if (fabsf(denomSq) <= 1.0e-8f) return false;
    float factor = (eta * eta) * fabsf(cosIWh) * fabsf(cosOWh);
    factor /= smax(absCosO * absCosI * denomSq, 1.0e-6f);
    ft = mk3((1.0f - F) * D * G * factor) * transmissionTint(m, absCosI);
    ft = ft * p.transmission;
    const float pdfWh = ggxVndfPdf(alpha, n, wo, wh);
    const float dwhDwi = fabsf((eta * eta * cosIWh) / smax(denomSq, 1.0e-8f));
    pdfTrans = pdfWh * dwhDwi;
    return true;
}

__device__ BsdfEvalResult evalPbrMetal(const Mat& m, f3 n, f3 wo, f3 wi, const ClampCfg& cc) {   // :4632-4762
    BsdfEvalResult r{mk3(0.0f), 0.0f, false};
    const float cosO = dot(n, wo), cosI = dot(n, wi);
    const float absCosO = fabsf(cosO), absCosI = fabsf(cosI);
    if (absCosO <= 0.0f || absCosI <= 0.0f) return r;
    const PbrMetal p = loadPbrMetal(m);
    if (!p.valid) return r;
    if (cosO * cosI > 0.0f) {
        if (cosO <= 0.0f || cosI <= 0.0f) return r;
        const float alpha = smax(p.roughness * p.roughness, 1.0e-4f);
        const f3 wh = normalize(wo + wi);
        if (dot(wh, n) <= 0.0f || dot(wo, wh) <= 0.0f || dot(wi, wh) <= 0.0f) return r;
        const float D = ggxD(alpha, dot(n, wh));
        const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
        f3 spec = schlick(p.f0, dot(wi, wh)) * (D * G / smax(4.0f * cosO * cosI, 1.0e-6f));
        spec = spec * specularEnergyCompensation(p.f0, p.roughness, absCosO);
        spec = clampSpecTail(spec, p.roughness, p.f0, cc);
        spec = spec * p.reflectScale;
        const float pdfSpec = ggxPdfVisible(alpha, n, wo, wi);
        const f3 diffuse = (p.diffuseColor / kPi) * p.reflectScale;
        const float pdf = p.pSpec * pdfSpec + p.pDiff * lambertPdf(n, wi);
        if (pdf > 0.0f) {
            r.value = vmax0(spec + diffuse);
            r.pdf = clampSpecPdf(pdf, cc);
        }
        return r;
    }
    if (p.transmission <= 0.0f) return r;
    float etaI = 1.0f, etaT = smax(m.ior(), 1.0f);
    if (cosO < 0.0f) {
        const float tmp = etaI;
        etaI = etaT;
        etaT = tmp;
    }
    const float eta = etaI / etaT;
    f3 wh = wo + wi * eta;
    if (!finite3(wh) || dot(wh, wh) <= 0.0f) return r;
    wh = normalize(wh);
    if (dot(wh, n) <= 0.0f) wh = -wh;
    f3 ft;
    float pdfTrans = 0.0f;
    if (!roughTransmission(m, p, n, wo, wi, wh, eta, etaI, etaT, ft, pdfTrans)) return r;
    const float pdf = p.pTrans * pdfTrans;
    if (pdf > 0.0f) {
        r.value = vmax0(ft);
        r.pdf = clampSpecPdf(pdf, cc);
    }
    return r;
}

// SSS: compiled with the Metal subsurface semantics (PTR_METAL_SSS).  A template parameter, not a run-time flag: the extra
// branch and the exit point it carries cost the default k_shade 3 % when they were merely predicated off.
// MATS: bit t set = material type t can occur (the host knows a scene's materials: an instantiation compiled for the types of a
// simple scene carries no registers or code for car paint, plastic or the metallic-roughness model).  A type outside the set cannot
// reach these functions; it would fall through to the Lambert branch.
constexpr uint32_t kFeatureEnvironment = 1u << 8;   // ... and two scene features k_shade compiles out with them: an environment map,
constexpr uint32_t kFeatureMedia = 1u << 9;         // the Metal integrator's medium stack / face-normal rule
constexpr uint32_t kAllMaterials = 0x3FFu;
template <bool SSS = false, uint32_t MATS = kAllMaterials>
__device__ BsdfEvalResult evalBsdf(const Mat& m, f3 position, f3 n, f3 wo, f3 wi, const ClampCfg& cc) {
    if (SSS && cc.metalPbr && m.type() == 7u) return evalPbrMetal(m, n, wo, wi, cc);   // before the same-side test below
    BsdfEvalResult r{mk3(0.0f), 0.0f, false};
    const float cosO = smax(dot(n, wo), 0.0f), cosI = smax(dot(n, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const uint32_t type = m.type();
    if (SSS && type == 5u && cc.metalSss) return r;   // is_bssrdf: value 0, pdf 0 -> no next-event estimation (pathtrace.metal:5078-5084)
    switch (type) {
        case 7u: {  // PBR metallic-roughness
            if (!(MATS & (1u << 7))) break;
            const Pbr p = loadPbr(m);
            const float alpha = smax(p.roughness * p.roughness, 1.0e-4f);
            const f3 wh = normalize(wo + wi);
            if (!halfOk(wh, n, wo, wi)) return r;
            f3 spec = microfacet(schlick(p.f0, dot(wi, wh)), alpha, n, wh, cosO, cosI);
            spec = vmax0(clampSpecTail(spec, p.roughness, p.f0, cc));
            const float pdfS = ggxPdf(alpha, n, wo, wi);
            const float pdfSc = (pdfS > 0.0f) ? clampSpecPdf(pdfS, cc) : 0.0f;
            const float pdf = p.specWeight * pdfSc + (1.0f - p.specWeight) * lambertPdf(n, wi);
            if (pdf > 0.0f) {
                r.value = vmax0(spec + p.diffuseColor / kPi);
                r.pdf = pdf;
            }
            return r;
        }
        case 4u: {  // plastic
            if (!(MATS & (1u << 4))) break;
            const Coat c = loadCoat(m);
            f3 spec, diffuse;
            float specPdf;
            plasticLobes(m, c, n, wo, wi, cosO, cosI, cc, spec, specPdf, diffuse);
            const float pCoat = clampf(c.sampleWeight, 0.0f, 1.0f);
            const float pdf = pCoat * specPdf + (1.0f - pCoat) * lambertPdf(n, wi);
            if (pdf > 0.0f) {
                r.value = spec + diffuse;
                r.pdf = pdf;
            }
            return r;
        }
        case 1u: {  // metal
            if (!(MATS & (1u << 1))) break;
            const float rough = m.roughness01();
            if (rough <= 1.0e-3f) {
                r.isDelta = true;
                return r;
            }
            const float alpha = rough * rough;
            const f3 wh = normalize(wo + wi);
            if (!halfOk(wh, n, wo, wi)) return r;
            const Conductor c = loadMetal(m);
            f3 spec = microfacet(conductorF(c, dot(wi, wh)), alpha, n, wh, cosO, cosI);
            if (cc.metalSpecular) spec = spec * specularEnergyCompensation(c.f0, rough, cosO);
            spec = clampSpecTail(spec, rough, c.f0, cc);
            const float pdf = cc.metalSpecular ? ggxPdfVisible(alpha, n, wo, wi) : ggxPdf(alpha, n, wo, wi);
            if (pdf > 0.0f) {
                r.value = vmax0(spec);
                r.pdf = clampSpecPdf(pdf, cc);
            }
            return r;
        }
        case 6u: {  // car paint: probability-weighted mixture of base, flake and coat lobes
            if (!(MATS & (1u << 6))) break;
            const CarPaint c = loadCarPaint(m);
            const Lobe coat = carpaintCoat(c, n, wo, wi, cc);
            const Lobe flake = carpaintFlake(c, position, n, wo, wi, cc);
            const Lobe base = carpaintBase(c, n, wo, wi, cc);
            r.value = (c.pBase * base.value + c.pFlake * flake.value) + c.pCoat * coat.value;
            r.pdf = (c.pBase * base.pdf + c.pFlake * flake.pdf) + c.pCoat * coat.pdf;
            return r;
        }
        case 2u:  // dielectric
            r.isDelta = true;
            return r;
        default:
            break;
    }
    // Lambert, subsurface (Lambert on this path), anything unknown
    r.value = m.baseColor() / kPi;
    r.pdf = lambertPdf(n, wi);
    return r;
}

struct BsdfSampleResult {
    f3 dir, weight;
    float pdf;
    bool isDelta;
    int mediumEvent;   // +1: refracted into a dielectric through its front face, -1: out through a back face, else 0
    bool hasExit;      // separable subsurface sample: the path leaves the surface at exitPoint (normal = the shading normal)
    f3 exitPoint;
    // which lobe the textured metallic-roughness model sampled (0 diffuse, 1 specular, 2 transmission) and its roughness: what the
    // ray cone of the path widens by (bsdf_cone_spread_increment, shaders/pathtrace.metal:5703-5715); read only in textured scenes
    int lobe = 0;
    float lobeRoughness = 0.0f;
};

// ---- separable subsurface scattering of the Metal integrator (shaders/pathtrace.metal:3916-3994) ----
struct SssCoefficients {
    f3 sigmaA, sigmaSPrime;
};

// sss_sigma_a / sss_sigma_s_prime (:3916-3950): the material's override pair, or derived from base colour and mean free path
__device__ __forceinline__ SssCoefficients sssCoefficients(const Mat& m, float meanFreePath, float anisotropy) {
    SssCoefficients c;
    const float4 a = m.v(kMatSssSigmaA);
    const float reduce = smax(1.0f - anisotropy, 0.01f);
    if (a.w > 0.5f) {
        c.sigmaA = vmaxs(mk3(a), 1.0e-6f);
        c.sigmaSPrime = vmax0(mk3(m.v(kMatSssSigmaS))) * reduce;
        return c;
    }
    const float sigmaT = 1.0f / smax(meanFreePath, 1.0e-4f);
    f3 sigmaS = vmax0(vclamp(m.baseColor(), 0.0f, 0.999f) * sigmaT) * reduce;
    c.sigmaA = vmaxs(mk3(sigmaT) - sigmaS, 1.0e-6f);
    c.sigmaSPrime = sigmaS;
    return c;
}

// normalized_diffusion_profile (:3952-3971): dipole reflectance at `radius`, per colour channel
__device__ __forceinline__ f3 diffusionProfile(float radius, f3 sigmaA, f3 sigmaSPrime) {
    const f3 sigmaTPrime = vmaxs(sigmaA + sigmaSPrime, 1.0e-6f);
    const f3 alphaPrime = vclamp(sigmaSPrime / sigmaTPrime, 0.0f, 1.0f);
    const f3 D = mk3(1.0f) / vmaxs(3.0f * sigmaTPrime, 1.0e-6f);
    const f3 sigmaTr = vsqrt(vmaxs(sigmaA / D, 1.0e-6f));
    const float r = smax(radius, 1.0e-4f);
    const f3 zr = mk3(1.0f) / sigmaTPrime;
    const f3 dr = vsqrt(mk3(r * r) + zr * zr);
    const f3 vr = zr + 4.0f * D;
    const f3 dv = vsqrt(mk3(r * r) + vr * vr);
    const f3 expDr = vexp(-(sigmaTr * dr));
    const f3 expDv = vexp(-(sigmaTr * dv));
    const f3 termDr = (zr * (mk3(1.0f) + sigmaTr * dr)) / vmaxs(dr * dr * dr, 1.0e-6f);
    const f3 termDv = (vr * (mk3(1.0f) + sigmaTr * dv)) / vmaxs(dv * dv * dv, 1.0e-6f);
    return vmax0((alphaPrime / (4.0f * kPi)) * (termDr * expDr + termDv * expDv));
}

// sss_sigma_tr_scalar (:3973-3980)
__device__ __forceinline__ float sssSigmaTrScalar(f3 sigmaA, f3 sigmaSPrime) {
    const f3 sigmaTPrime = vmaxs(sigmaA + sigmaSPrime, 1.0e-6f);
    const f3 D = mk3(1.0f) / vmaxs(3.0f * sigmaTPrime, 1.0e-6f);
    return smax(luminance(vsqrt(vmaxs(sigmaA / D, 1.0e-6f))), 1.0e-4f);
}

// The separable branch of sample_bsdf case 5 (:5398-5481).  Returns false when it does not apply or gives up on the
// way (the caller then takes the Lambert fallback with the generator wherever this left it, like the reference).
__device__ bool sampleSeparableSss(const Mat& m, f3 position, f3 n, f3 wo, uint32_t& rng, const ClampCfg& cc, BsdfSampleResult& r) {
    const float4 params = m.v(kMatSssParams);
    const float meanFreePath = smax(params.x, 1.0e-4f);
    if (!(cc.sssMode == 1u && params.y < 0.5f && meanFreePath > 1.0e-4f)) return false;
    const float anisotropy = clampf(m.v(kMatSssSigmaS).w, -0.99f, 0.99f);
    const SssCoefficients c = sssCoefficients(m, meanFreePath, anisotropy);
    const float sigmaTr = sssSigmaTrScalar(c.sigmaA, c.sigmaSPrime);
    if (sigmaTr <= 0.0f) return false;
    const float u = clampf(rngNext(rng), 1.0e-6f, 1.0f - 1.0e-6f);
    float radius = -logf(1.0f - u) / smax(sigmaTr, 1.0e-4f);
    radius = smin(radius, meanFreePath * 10.0f);
    const float sigma = smax(sigmaTr, 1.0e-4f);
    const float pdfRadius = radius <= 0.0f ? 0.0f : sigma * expf(-sigma * radius);
    if (pdfRadius <= 0.0f || !isfinite(pdfRadius)) return false;
    const float phi = 2.0f * kPi * rngNext(rng);
    const Frame fr = makeFrame(n);
    const f3 exitPoint = position + fr.t * (radius * cosf(phi)) + fr.b * (radius * sinf(phi));
    float pdfDir = 0.0f;
    const f3 wi = cosineHemisphere(rng, n, pdfDir);
    const float cosExit = dot(n, wi);
    pdfDir = cosExit > 0.0f ? cosExit / kPi : 0.0f;   // lambert_pdf of the normalised direction
    const float pdfArea = pdfRadius / (2.0f * kPi * smax(radius, 1.0e-4f));
    if (cosExit <= 0.0f || pdfDir <= 0.0f || pdfArea <= 0.0f) return false;
    f3 profile = diffusionProfile(radius, c.sigmaA, c.sigmaSPrime);
    const float4 coat = m.v(kMatCoatParams);
    const float coatAverage = 1.0f - clampf(coat.w, 0.0f, 1.0f);
    float coatTransmission = 1.0f;
    if (params.z > 0.5f) {
        const float coatIor = smax(m.v(kMatTypeEta).z, 1.0f);
        float f0 = (coatIor - 1.0f) / (coatIor + 1.0f);
        f0 *= f0;
        const float transIn = 1.0f - (f0 + (1.0f - f0) * schlickW(smax(dot(n, wo), 0.0f)));
        const float transOut = 1.0f - (f0 + (1.0f - f0) * schlickW(cosExit));
        coatTransmission = clampf(transIn * transOut, 0.0f, 1.0f);
        profile = profile * vclamp(mk3(m.v(kMatCoatTint)), 0.0f, 1.0f);
    }
    const float denom = smax(pdfArea * pdfDir, 1.0e-6f);
    const f3 weight = vmax0((profile * (cosExit * coatAverage * coatTransmission)) / denom);
    if (!finite3(weight)) return false;
    r.dir = wi;
    r.weight = weight;
    // The reference reports pdf = area pdf x directional pdf and keeps the directional pdf for the MIS weight of the next
    // emitter hit (lastBsdfPdf = directionalPdf, :7269); both are positive here, so the one pdf this struct carries is
    // the directional one.
    r.pdf = pdfDir;
    r.isDelta = false;
    r.hasExit = true;
    r.exitPoint = exitPoint;
    return true;
}

// ---- random-walk subsurface scattering of the Metal integrator (sample_sss_random_walk_software,
// shaders/pathtrace.metal:4060-4311), cut into "begin" and "step": the reference runs the walk as a loop of closest-hit
// queries inside the sampling step; in the wavefront every query is one extend/shade iteration, and the slot carries
// the walk state in between (k_shade).  oracle/oracle_integrator.cpp holds the same two functions.
struct SssWalk {
    f3 position, direction, throughput;
    uint32_t step;
};
constexpr int kWalkFallback = 0, kWalkSample = 1, kWalkWalking = 2;

__device__ __forceinline__ f3 refractMetal(f3 i, f3 n, float eta) {   // MSL refract(): zero vector on total internal reflection
    const float ndi = dot(n, i);
    const float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return mk3(0.0f);
    return eta * i - (eta * ndi + sqrtf(k)) * n;
}

__device__ __forceinline__ f3 offsetSurfacePoint(f3 point, f3 normal, f3 direction) {   // :1210-1220
    const f3 n = (finite3(normal) && dot(normal, normal) > 0.0f) ? normalize(normal) : mk3(0.0f, 1.0f, 0.0f);
    const float sign = dot(direction, n) >= 0.0f ? 1.0f : -1.0f;
    f3 origin = point + n * (sign * 1.0e-4f * 4.0f);
    origin += (direction * 1.0e-4f) * 0.5f;
    return origin;
}

// lobe pick, then the coat reflection (a finished sample) or the refraction into the medium (walk state)
__device__ int sssWalkBegin(const Mat& m, f3 point, f3 entryNormal, f3 wo, f3 incident, uint32_t& rng, const ClampCfg& cc,
                            BsdfSampleResult& sample, SssWalk& walk) {
    const Coat c = loadCoat(m);
    const float pCoat = clampf(c.sampleWeight, 0.0f, 1.0f);
    const float randLobe = rngNext(rng);
    const float alpha = c.roughness * c.roughness;
    const float ratio = (c.ior - 1.0f) / smax(c.ior + 1.0f, 1.0e-6f);
    const f3 f0 = mk3(clampf(ratio * ratio, 0.0f, 0.999f));   // plastic_coat_f0, :3861-3866
    const f3 specTint = coatSpecTint(c);
    if (pCoat > 0.0f && randLobe < pCoat) {   // :4104-4155
        const f3 wh = sampleGgxVndf(rng, c.roughness, entryNormal, wo);
        if (dot(wh, entryNormal) <= 0.0f) return kWalkFallback;
        f3 wi = reflectDir(-wo, wh);
        if (!(dot(wi, wi) > 0.0f)) return kWalkFallback;
        wi = normalize(wi);
        if (!finite3(wi)) return kWalkFallback;
        const float cosI = dot(entryNormal, wi), cosO = dot(entryNormal, wo);
        if (cosI <= 0.0f || cosO <= 0.0f) return kWalkFallback;
        const float dotWiWh = dot(wi, wh);
        if (dotWiWh <= 0.0f) return kWalkFallback;
        const float D = ggxD(alpha, dot(entryNormal, wh));
        const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
        const f3 F = schlick(f0, dotWiWh);
        f3 spec = F * (D * G / smax(4.0f * cosO * cosI, 1.0e-6f));
        spec = clampSpecTail(spec * specTint, c.roughness, f0, cc);
        const float specPdfRaw = ggxPdfVisible(alpha, entryNormal, wo, wi);
        if (specPdfRaw <= 0.0f) return kWalkFallback;
        const float specPdf = clampSpecPdf(specPdfRaw, cc);
        const float combinedPdf = smax(pCoat * specPdf, 1.0e-6f);
        const f3 weight = vmax0(spec * cosI / combinedPdf);
        if (!finite3(weight)) return kWalkFallback;
        sample.dir = wi;
        sample.weight = weight;
        sample.pdf = specPdf;   // the directional pdf (see sampleSeparableSss)
        sample.isDelta = false;
        sample.mediumEvent = 0;
        sample.hasExit = false;
        return kWalkSample;
    }
    const float pDiffuse = smax(1.0f - pCoat, 1.0e-3f);
    f3 throughput = mk3(1.0f / pDiffuse);
    const float etaInside = smax(m.ior(), 1.0f);
    const float cosThetaI = dot(-incident, entryNormal);
    if (cosThetaI <= 0.0f) return kWalkFallback;
    float cosThetaT = 0.0f;
    const float frEntry = fresnelDielectric(cosThetaI, 1.0f, etaInside, cosThetaT);
    f3 enterDir = refractMetal(incident, entryNormal, 1.0f / etaInside);
    if (!finite3(enterDir) || dot(enterDir, enterDir) <= 0.0f) return kWalkFallback;
    enterDir = normalize(enterDir);
    const float scaleEntry = (etaInside * etaInside) * (cosThetaT / smax(cosThetaI, 1.0e-6f));
    throughput *= smax(1.0f - frEntry, 0.0f) * scaleEntry;
    if (m.v(kMatSssParams).z > 0.5f) throughput = throughput * specTint;
    walk.position = offsetSurfacePoint(point, -entryNormal, enterDir);
    walk.direction = enterDir;
    walk.throughput = throughput;
    walk.step = 0u;
    return kWalkWalking;
}

__device__ __forceinline__ f3 sampleHenyeyGreenstein(f3 reference, float g, uint32_t& rng) {   // :4011-4036
    const float u1 = rngNext(rng);
    const float u2 = rngNext(rng);
    float cosTheta;
    if (fabsf(g) < 1.0e-3f) {
        cosTheta = 1.0f - 2.0f * u1;
    } else {
        const float s = (1.0f - g * g) / (1.0f - g + 2.0f * g * u1);
        cosTheta = clampf((1.0f + g * g - s * s) / (2.0f * g), -1.0f, 1.0f);
    }
    const float sinTheta = sqrtf(smax(0.0f, 1.0f - cosTheta * cosTheta));
    const float phi = 2.0f * kPi * u2;
    const f3 local = mk3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
    if (!(dot(reference, reference) > 0.0f)) return mk3(0.0f);
    const Frame fr = makeFrame(reference);
    const f3 world = (local.x * fr.t + local.y * fr.b) + local.z * fr.n;
    return dot(world, world) > 0.0f ? normalize(world) : mk3(0.0f);
}

// one pass of the reference's loop body, given the closest hit of the ray (walk.position, walk.direction)
__device__ int sssWalkStep(const Mat& m, uint32_t maxSteps, SssWalk& walk, bool hitBoundary, float hitT, f3 hitPoint, f3 outwardNormal,
                           uint32_t& rng, BsdfSampleResult& sample) {
    const float anisotropy = clampf(m.v(kMatSssSigmaS).w, -0.99f, 0.99f);
    const SssCoefficients c = sssCoefficients(m, smax(m.v(kMatSssParams).x, 1.0e-4f), anisotropy);
    const f3 sigmaT = vmaxs(c.sigmaA + c.sigmaSPrime, 1.0e-6f);
    const float sigmaTScalar = smax(smax(sigmaT.x, smax(sigmaT.y, sigmaT.z)), 1.0e-4f);
    const float xi = clampf(rngNext(rng), 1.0e-6f, 1.0f - 1.0e-6f);
    const float distance = -logf(1.0f - xi) / sigmaTScalar;
    if (!hitBoundary) return kWalkFallback;
    const float boundaryDistance = smax(hitT, 1.0e-4f);
    const uint32_t limit = max(maxSteps, 1u);
    if (distance < boundaryDistance) {   // scattering event inside, :4227-4245
        walk.throughput = walk.throughput * vexp(-(sigmaT * distance));
        walk.throughput = walk.throughput * vclamp(c.sigmaSPrime / vmaxs(sigmaT, 1.0e-6f), 0.0f, 1.0f);
        if (smax(walk.throughput.x, smax(walk.throughput.y, walk.throughput.z)) < 1.0e-3f) return kWalkFallback;
        walk.position += walk.direction * distance;
        const f3 scattered = sampleHenyeyGreenstein(-walk.direction, anisotropy, rng);
        if (!finite3(scattered) || dot(scattered, scattered) <= 0.0f) return kWalkFallback;
        walk.direction = normalize(scattered);
        ++walk.step;
        return walk.step < limit ? kWalkWalking : kWalkFallback;
    }
    walk.throughput = walk.throughput * vexp(-(sigmaT * boundaryDistance));
    if (smax(walk.throughput.x, smax(walk.throughput.y, walk.throughput.z)) < 1.0e-3f) return kWalkFallback;
    if (!finite3(outwardNormal) || dot(outwardNormal, outwardNormal) <= 0.0f) return kWalkFallback;
    outwardNormal = normalize(outwardNormal);
    const float etaI = smax(m.ior(), 1.0f);
    // :4262-4268 as written: the exit is taken only where the geometric normal faces the ray; a ray that reaches the
    // boundary of a closed mesh from inside is reflected back in
    const float cosExitI = dot(-walk.direction, outwardNormal);
    f3 refracted = mk3(0.0f);
    float cosExitT = 0.0f, frExit = 1.0f;
    bool leaves = cosExitI > 0.0f;
    if (leaves) {
        frExit = fresnelDielectric(cosExitI, etaI, 1.0f, cosExitT);
        refracted = refractMetal(walk.direction, outwardNormal, etaI);
        leaves = finite3(refracted) && dot(refracted, refracted) > 0.0f;
    }
    if (!leaves) {
        walk.position = hitPoint;
        walk.direction = normalize(reflectDir(walk.direction, outwardNormal));
        ++walk.step;
        return walk.step < limit ? kWalkWalking : kWalkFallback;
    }
    refracted = normalize(refracted);
    const float scaleExit = (1.0f / (etaI * etaI)) * (cosExitT / smax(cosExitI, 1.0e-6f));
    f3 throughput = walk.throughput * (smax(1.0f - frExit, 0.0f) * scaleExit);
    if (m.v(kMatSssParams).z > 0.5f) throughput = throughput * coatSpecTint(loadCoat(m));
    throughput = vmax0(throughput);
    if (!finite3(throughput)) return kWalkFallback;
    sample.dir = refracted;
    sample.weight = throughput;
    sample.pdf = 1.0f;   // directionalPdf (:4297)
    sample.isDelta = false;
    sample.mediumEvent = 0;
    sample.hasExit = true;
    sample.exitPoint = hitPoint;
    return kWalkSample;
}

__device__ BsdfSampleResult samplePbrMetal(const Mat& m, f3 n, f3 wo, f3 incident, uint32_t& rng, const ClampCfg& cc) {   // :4764-4948
    BsdfSampleResult r{mk3(0.0f), mk3(0.0f), 0.0f, false, 0, false, mk3(0.0f)};
    const PbrMetal p = loadPbrMetal(m);
    if (!p.valid) return r;
    const float choose = rngNext(rng);
    f3 wi = mk3(0.0f), f = mk3(0.0f);
    float pdfSpec = 0.0f, pdfDiffuse = 0.0f, pdfTrans = 0.0f;
    bool isDelta = false;
    r.lobe = choose < p.pSpec ? 1 : (choose < p.pSpec + p.pDiff ? 0 : 2);
    r.lobeRoughness = p.roughness;
    if (choose < p.pSpec) {
        if (p.roughness <= 1.0e-3f) {
            wi = reflectDir(incident, n);
            if (dot(n, wi) <= 0.0f) return r;
            pdfSpec = 1.0f;
            f = schlick(p.f0, smax(dot(n, wo), 0.0f)) * p.reflectScale;
            isDelta = true;
        } else {
            const f3 wh = sampleGgxVndf(rng, p.roughness, n, wo);
            wi = reflectDir(-wo, wh);
            const float cosI = dot(n, wi);
            if (cosI <= 0.0f) return r;
            const float alpha = smax(p.roughness * p.roughness, 1.0e-4f);
            const float cosO = smax(dot(n, wo), 0.0f);
            const float D = ggxD(alpha, dot(n, wh));
            const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
            f = schlick(p.f0, dot(wi, wh)) * (D * G / smax(4.0f * cosO * cosI, 1.0e-6f));
            f = f * specularEnergyCompensation(p.f0, p.roughness, cosO);
            f = clampSpecTail(f, p.roughness, p.f0, cc);
            f = f * p.reflectScale;
            pdfSpec = ggxPdfVisible(alpha, n, wo, wi);
        }
    } else if (choose < p.pSpec + p.pDiff) {
        float unused = 0.0f;
        wi = cosineHemisphere(rng, n, unused);
        if (dot(n, wi) <= 0.0f) return r;
        f = (p.diffuseColor / kPi) * p.reflectScale;
        pdfDiffuse = lambertPdf(n, wi);
    } else {
        const float cosO = dot(n, wo), absCosO = fabsf(cosO);
        float etaI = 1.0f, etaT = smax(m.ior(), 1.0f);
        if (cosO < 0.0f) {
            const float tmp = etaI;
            etaI = etaT;
            etaT = tmp;
        }
        const float eta = etaI / etaT;
        if (p.roughness <= 1.0e-3f) {
            wi = refractMetal(-wo, n, eta);
            if (dot(wi, wi) <= 0.0f) return r;
            wi = normalize(wi);
            float cosT = 0.0f;
            const float Fr = fresnelDielectric(cosO, etaI, etaT, cosT);
            const float directionScale = ((etaT * etaT) / (etaI * etaI)) * (fabsf(cosT) / smax(absCosO, 1.0e-6f));
            f = (mk3(smax(1.0f - Fr, 0.0f) * directionScale) * transmissionTint(m, fabsf(dot(n, wi)))) * p.transmission;
            pdfTrans = 1.0f;
            isDelta = true;
        } else {
            const f3 wh = sampleGgxVndf(rng, p.roughness, n, wo);
            wi = refractMetal(-wo, wh, eta);
            if (dot(wi, wi) <= 0.0f) return r;
            wi = normalize(wi);
            if (dot(wi, n) * cosO >= 0.0f) return r;
            if (!roughTransmission(m, p, n, wo, wi, wh, eta, etaI, etaT, f, pdfTrans)) return r;
        }
    }
    const float absCosI = fabsf(dot(n, wi));
    if (absCosI <= 0.0f) return r;
    const float pdf = p.pSpec * pdfSpec + p.pDiff * pdfDiffuse + p.pTrans * pdfTrans;
    if (pdf <= 0.0f) return r;
    r.dir = wi;
    r.pdf = pdf;
    r.isDelta = isDelta;
    r.weight = vmax0(f * absCosI / pdf);
    return r;
}

template <bool SSS = false, uint32_t MATS = kAllMaterials>
__device__ BsdfSampleResult sampleBsdf(const Mat& m, f3 position, f3 n, f3 wo, f3 incident, bool frontFace,
                                       uint32_t& rng, const ClampCfg& cc) {
    BsdfSampleResult r{mk3(0.0f), mk3(0.0f), 0.0f, false, 0, false, mk3(0.0f)};
    const uint32_t type = m.type();
    if (SSS && type == 5u && cc.metalSss && sampleSeparableSss(m, position, n, wo, rng, cc, r)) return r;
    if (SSS && type == 7u && cc.metalPbr) return samplePbrMetal(m, n, wo, incident, rng, cc);
    switch (type) {
        case 7u: {
            if (!(MATS & (1u << 7))) break;
            const Pbr p = loadPbr(m);
            f3 wi, f;
            float pdfS = 0.0f, pdfD = 0.0f;
            if (rngNext(rng) < p.specWeight) {
                if (p.roughness <= 1.0e-3f) {
                    wi = normalize(reflectDir(incident, n));
                    if (dot(n, wi) <= 0.0f) return r;
                    f = schlick(p.f0, smax(dot(n, wo), 0.0f));
                    pdfS = 1.0f;
                    r.isDelta = true;
                } else {
                    const float alpha = p.roughness * p.roughness;
                    const f3 wh = sampleGgxHalf(rng, alpha, n);
                    if (dot(wh, n) <= 0.0f) return r;
                    wi = normalize(reflectDir(-wo, wh));
                    const float cosI = smax(dot(n, wi), 0.0f), cosO = smax(dot(n, wo), 0.0f);
                    if (cosI <= 0.0f || cosO <= 0.0f) return r;
                    f = microfacet(schlick(p.f0, dot(wi, wh)), alpha, n, wh, cosO, cosI);
                    f = clampSpecTail(f, p.roughness, p.f0, cc);
                    pdfS = ggxPdf(alpha, n, wo, wi);
                }
            } else {
                wi = cosineHemisphere(rng, n, pdfD);
                if (pdfD <= 0.0f || smax(dot(n, wi), 0.0f) <= 0.0f) return r;
                f = p.diffuseColor / kPi;
            }
            const float cosI = smax(dot(n, wi), 0.0f);
            const float pdfSc = (pdfS > 0.0f) ? clampSpecPdf(pdfS, cc) : 0.0f;
            const float pdf = p.specWeight * pdfSc + (1.0f - p.specWeight) * pdfD;
            if (pdf <= 0.0f || cosI <= 0.0f) return r;
            const f3 w = f * cosI / pdf;
            if (!finite3(w)) return r;
            r.dir = wi;
            r.weight = vmax0(w);
            r.pdf = pdf;
            return r;
        }
        case 4u: {
            if (!(MATS & (1u << 4))) break;
            const Coat c = loadCoat(m);
            const float alpha = c.roughness * c.roughness;
            const float pCoat = clampf(c.sampleWeight, 0.0f, 1.0f);
            f3 wi;
            if (rngNext(rng) < pCoat) {
                const f3 wh = sampleGgxHalf(rng, alpha, n);
                if (dot(wh, n) <= 0.0f) return r;
                wi = normalize(reflectDir(-wo, wh));
            } else {
                float unused;
                wi = cosineHemisphere(rng, n, unused);
            }
            const float cosI = smax(dot(n, wi), 0.0f), cosO = smax(dot(n, wo), 0.0f);
            if (cosI <= 0.0f || cosO <= 0.0f) return r;
            f3 spec, diffuse;
            float specPdf;
            plasticLobes(m, c, n, wo, wi, cosO, cosI, cc, spec, specPdf, diffuse);
            const float pdf = pCoat * specPdf + (1.0f - pCoat) * lambertPdf(n, wi);
            if (pdf <= 0.0f) return r;
            const f3 w = (spec + diffuse) * cosI / pdf;
            if (!finite3(w)) return r;
            r.dir = wi;
            r.weight = vmax0(w);
            r.pdf = pdf;
            return r;
        }
        case 1u: {
            if (!(MATS & (1u << 1))) break;
            const float rough = m.roughness01();
            const Conductor c = loadMetal(m);
            if (rough <= 1.0e-3f) {
                const f3 wi = normalize(reflectDir(incident, n));
                if (dot(n, wi) <= 0.0f) return r;
                r.dir = wi;
                r.weight = conductorF(c, smax(dot(n, wo), 0.0f));
                r.pdf = 1.0f;
                r.isDelta = true;
                return r;
            }
            const float alpha = rough * rough;
            const f3 wh = cc.metalSpecular ? sampleGgxVndf(rng, rough, n, wo) : sampleGgxHalf(rng, alpha, n);
            if (dot(wh, n) <= 0.0f) return r;
            const f3 wi = normalize(reflectDir(-wo, wh));
            const float cosI = dot(n, wi), cosO = dot(n, wo);
            if (cosI <= 0.0f || cosO <= 0.0f) return r;
            const float woh = dot(wo, wh);
            if (woh <= 0.0f) return r;
            const f3 F = conductorF(c, dot(wi, wh));
            const float D = ggxD(alpha, dot(n, wh));
            const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
            f3 f = F * (D * G / smax(4.0f * cosO * cosI, 1.0e-6f));
            if (cc.metalSpecular) f = f * specularEnergyCompensation(c.f0, rough, cosO);
            f = clampSpecTail(f, rough, c.f0, cc);
            const float pdf = cc.metalSpecular ? ggxPdfVisible(alpha, n, wo, wi) : D * smax(dot(n, wh), 0.0f) / smax(4.0f * woh, 1.0e-6f);
            if (pdf <= 0.0f) return r;
            const float pdfC = clampSpecPdf(pdf, cc);
            const f3 w = f * cosI / pdfC;
            if (!finite3(w)) return r;
            r.dir = wi;
            r.weight = vmax0(w);
            r.pdf = pdfC;
            return r;
        }
        case 6u: {
            if (!(MATS & (1u << 6))) break;
            const CarPaint c = loadCarPaint(m);
            const float pick = rngNext(rng);
            uint32_t lobe = 0u;  // 0 base, 1 flake, 2 coat
            if (c.pCoat > 0.0f && pick < c.pCoat) {
                lobe = 2u;
            } else if (c.pFlake > 0.0f && pick < c.pCoat + c.pFlake) {
                lobe = 1u;
            } else if (c.pBase <= 1.0e-6f) {
                if (c.pFlake > c.pCoat && c.pFlake > 0.0f) lobe = 1u;
                else if (c.pCoat > 0.0f) lobe = 2u;
            }
            f3 wi;
            if (lobe == 2u) {
                const f3 wh = sampleGgxVndf(rng, c.coat.roughness, n, wo);
                if (dot(wh, n) <= 0.0f) return r;
                wi = normalize(reflectDir(-wo, wh));
            } else if (lobe == 1u) {
                const float fr = smax(c.flakeRoughness, 1.0e-3f);
                const f3 fn = carpaintFlakeNormal(c, position, n);
                const f3 wh = sampleGgxHalf(rng, fr * fr, fn);
                if (dot(wh, fn) <= 0.0f) return r;
                wi = normalize(reflectDir(-wo, wh));
            } else {
                const float dw = smax(1.0f - c.baseMetallic, 0.0f), sw = smax(c.baseMetallic, 0.0f);
                const float sum = dw + sw;
                const float choose = rngNext(rng);
                if ((sw > 0.0f) && (sum > 0.0f) && (choose < sw / smax(sum, 1.0e-6f))) {
                    const float br = smax(c.baseRoughness, 1.0e-3f);
                    const f3 wh = sampleGgxHalf(rng, br * br, n);
                    if (dot(wh, n) <= 0.0f) return r;
                    wi = normalize(reflectDir(-wo, wh));
                } else {
                    float unused;
                    wi = cosineHemisphere(rng, n, unused);
                }
            }
            if (!finite3(wi) || dot(n, wi) <= 0.0f) return r;
            const Lobe coat = carpaintCoat(c, n, wo, wi, cc);
            const Lobe flake = carpaintFlake(c, position, n, wo, wi, cc);
            const Lobe base = carpaintBase(c, n, wo, wi, cc);
            const float pdf = (c.pBase * base.pdf + c.pFlake * flake.pdf) + c.pCoat * coat.pdf;
            if (pdf <= 0.0f) return r;
            const Lobe sel = (lobe == 1u) ? flake : (lobe == 2u ? coat : base);
            if (sel.pdf <= 0.0f || !(sel.value.x > 0.0f || sel.value.y > 0.0f || sel.value.z > 0.0f)) return r;
            const float cosI = smax(dot(n, wi), 0.0f);
            if (cosI <= 0.0f) return r;
            const f3 w = sel.value * cosI / pdf;
            if (!finite3(w)) return r;
            r.dir = wi;
            r.weight = vmax0(w);
            r.pdf = pdf;
            return r;
        }
        case 2u: {  // smooth dielectric: Fresnel-weighted pick between mirror reflection and refraction
            if (!(MATS & (1u << 2))) break;
            r.isDelta = true;
            const float refIdx = smax(m.ior(), 1.0f);
            // thin-walled glass (Metal semantics only, pathtrace.metal:5649-5659): both faces see air -> glass
            const bool thin = cc.thinDielectrics && m.thinFlag();
            const bool entering = frontFace || thin;
            const float etaI = entering ? 1.0f : refIdx;
            const float etaT = entering ? refIdx : 1.0f;
            const float cosO = clampf(dot(-incident, n), -1.0f, 1.0f);
            float cosT = 0.0f;
            const float Fr = fresnelDielectric(cosO, etaI, etaT, cosT);
            f3 dir;
            f3 weight = mk3(Fr);
            bool refracted = false;
            if (!(rngNext(rng) < Fr)) {
                const float eta = etaI / etaT;
                const float ct = smin(-dot(incident, n), 1.0f);
                const f3 perp = eta * (incident + ct * n);
                const float k = 1.0f - dot(perp, perp);
                if (!(k < 0.0f)) {
                    const f3 out = perp + (-sqrtf(k)) * n;
                    if (dot(out, out) > 0.0f) {
                        refracted = true;
                        dir = normalize(out);
                        const float etaScale = (etaT * etaT) / (etaI * etaI);
                        const float dirScale = etaScale * (fabsf(cosT) / smax(fabsf(cosO), 1.0e-6f));
                        weight = mk3(smax(1.0f - Fr, 0.0f) * dirScale);
                        if (!thin) r.mediumEvent = frontFace ? 1 : -1;   // pathtrace.metal:5683
                    }
                }
            }
            if (!refracted) dir = reflectDir(incident, n);
            r.dir = normalize(dir);
            r.weight = weight;
            r.pdf = 1.0f;
            return r;
        }
        case 0u:
        case 5u: {  // Lambert / subsurface
            float pdf = 0.0f;
            const f3 wi = cosineHemisphere(rng, n, pdf);
            const float cosI = dot(n, wi);
            if (pdf <= 0.0f || cosI <= 0.0f) return r;
            const f3 w = (m.baseColor() / kPi) * cosI / pdf;
            if (!finite3(w)) return r;
            r.dir = wi;
            r.weight = vmax0(w);
            r.pdf = pdf;
            return r;
        }
        default:
            break;
    }
    // unknown type: cosine sampling with weight = albedo
    float pdf = 0.0f;
    const f3 wi = cosineHemisphere(rng, n, pdf);
    if (pdf <= 0.0f) return r;
    const f3 w = m.baseColor();
    if (!finite3(w)) return r;
    r.dir = wi;
    r.weight = vmax0(w);
    r.pdf = pdf;
    return r;
}

}  // namespace ptrk
