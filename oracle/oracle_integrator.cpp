// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md and oracle_integrator.h.
// "E:" citations = /root/reference/src/headless/EmbreeHeadlessRenderer.mm line numbers.
#include "oracle_integrator.h"

#include <atomic>
#include <cstring>
#include <limits>
#include <thread>

namespace oracle {
namespace {

constexpr float kEpsilon = 1.0e-4f;                 // E:31
constexpr float kSpecularNeePdfFloor = 1.0e-4f;     // E:32
constexpr float kSpecularNeeInvPdfClamp = 1.0e4f;   // E:33
constexpr float kMisWeightClampMin = 1.0e-4f;       // E:34
constexpr float kMisWeightClampMax = 0.9999f;       // E:35
constexpr float kInf = std::numeric_limits<float>::infinity();

inline uint32_t matType(const PtrMaterial& m) { return static_cast<uint32_t>(m.typeEta[0]); }
inline float degToRad(float d) { return d * (kPi / 180.0f); }
inline float luminance(V3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }  // E:377-379

// ---------------------------------------------------------------------------------------------
// background / environment                                                           E:234-310
// ---------------------------------------------------------------------------------------------
V3 skyColor(V3 direction) {
    const V3 unit = normalize(direction);
    const float t = 0.5f * (unit.y + 1.0f);
    return V3(1.0f, 1.0f, 1.0f) * (1.0f - t) + V3(0.5f, 0.7f, 1.0f) * t;
}

V3 rotateIntoMap(V3 unit, float rotation) {
    const float c = std::cos(rotation), s = std::sin(rotation);
    return {unit.x * c - unit.z * s, unit.y, unit.x * s + unit.z * c};
}

V3 evaluateBackground(const PtrSettings& s, const EnvMap* env, V3 direction) {  // E:293-310
    if (s.backgroundMode == PTR_BG_SOLID) return V3(s.backgroundColor);
    if (s.backgroundMode == PTR_BG_ENVIRONMENT && env) {
        return sampleEnvironment(*env, direction, s.environmentRotation, s.environmentIntensity);
    }
    return skyColor(direction);
}

// ---------------------------------------------------------------------------------------------
// small BSDF helpers                                                                 E:312-375
// ---------------------------------------------------------------------------------------------
V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; }

bool refract(V3 v, V3 n, float eta, V3& out) {
    const float cosTheta = std::min(-dot(v, n), 1.0f);
    const V3 perp = eta * (v + cosTheta * n);
    const float k = 1.0f - dot(perp, perp);
    if (k < 0.0f) return false;
    out = perp + (-std::sqrt(k)) * n;
    return true;
}

struct Onb {
    V3 tangent, bitangent, normal;
};

Onb buildOnb(V3 n) {  // E:341-350
    Onb o;
    o.normal = normalize(n);
    const V3 up = (std::fabs(o.normal.z) < 0.999f) ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
    o.tangent = normalize(cross(up, o.normal));
    o.bitangent = cross(o.normal, o.tangent);
    return o;
}

V3 onbToWorld(V3 l, const Onb& o) { return (l.x * o.tangent + l.y * o.bitangent) + l.z * o.normal; }
V3 onbToLocal(V3 v, const Onb& o) { return {dot(v, o.tangent), dot(v, o.bitangent), dot(v, o.normal)}; }

V3 sampleCosineHemisphere(Rng& rng, V3 n, float& outPdf) {  // E:352-365
    const float r1 = rng.nextFloat();
    const float r2 = rng.nextFloat();
    const float r = std::sqrt(std::max(r1, 0.0f));
    const float phi = 2.0f * kPi * r2;
    const float x = std::cos(phi) * r;
    const float y = std::sin(phi) * r;
    const float z = std::sqrt(std::max(1.0f - r1, 0.0f));
    const V3 dir = onbToWorld(V3(x, y, z), buildOnb(n));
    outPdf = z / kPi;
    return normalize(dir);
}

float schlickWeight(float cosTheta) {
    const float m = clampf(1.0f - cosTheta, 0.0f, 1.0f);
    const float m2 = m * m;
    return m2 * m2 * m;
}

V3 schlickFresnel(V3 f0, float cosTheta) { return f0 + (V3(1.0f, 1.0f, 1.0f) - f0) * schlickWeight(cosTheta); }

// ---------------------------------------------------------------------------------------------
// clamps                                                                              E:406-479
// ---------------------------------------------------------------------------------------------
// Next-event estimation, Embree path (E:2753-2756, 2800-2802): a light sample counts when the BSDF has a density for its direction,
// with the unclamped balance weight.  Metal kernel (M:6532-6552, 6624-6645; PTR_METAL_CLAMPS, Appendix A row 13): it counts when the BSDF
// value is positive; the weight is clamped to [1e-4, 0.9999], and 1 where the BSDF reports no density.
static bool neeContributes(const BsdfEval& be, const ClampParams& p) {
    if (be.isDelta) return false;
    if (p.metalClamps) return std::max(std::max(be.value.x, be.value.y), be.value.z) > 0.0f;
    return be.pdf > 0.0f;
}
static float neeWeight(float lightPdf, float bsdfPdf, const ClampParams& p) {
    if (!p.metalClamps) return lightPdf / (lightPdf + bsdfPdf);
    float weight = 1.0f;
    if (bsdfPdf > 0.0f) {
        const float denom = lightPdf + bsdfPdf;
        if (denom > 0.0f) weight = clampf(lightPdf / denom, kMisWeightClampMin, kMisWeightClampMax);
    }
    return weight;
}

V3 clampFireflyContribution(V3 throughput, V3 contribution, const ClampParams& p) {
    V3 combined = throughput * contribution;
    if (!finite3(combined)) return V3();
    V3 positive = vmax(combined, V3());
    if (p.enabled < 0.5f) return positive;
    const float lum = luminance(positive);
    const float throughputLum = luminance(vmax(throughput, V3()));
    float maxLum = std::max(throughputLum * p.clampFactor, p.clampFloor);
    if (p.metalClamps && p.maxContribution > 0.0f) maxLum = std::max(maxLum, p.maxContribution);   // M:3563-3568
    if (lum > maxLum && lum > 0.0f) {
        const float scale = maxLum / std::max(lum, 1.0e-6f);
        combined *= scale;
        positive = vmax(combined, V3());
    }
    return positive;
}

float clampSpecularPdf(float pdf, const ClampParams& p) {
    if (p.metalClamps) {   // M:3579-3590
        if (!std::isfinite(pdf) || pdf <= 0.0f) return 0.0f;
        return p.minSpecularPdfRaw <= 0.0f ? pdf : std::max(pdf, p.minSpecularPdfRaw);
    }
    const float minPdf = std::max(p.minSpecularPdf, 1.0e-8f);
    if (!std::isfinite(pdf)) return minPdf;
    return std::max(pdf, minPdf);
}

V3 clampPathThroughput(V3 throughput, const ClampParams& p) {
    if (!finite3(throughput)) return V3();
    if (p.enabled < 0.5f || p.throughputClamp <= 0.0f) return throughput;
    const float lum = luminance(vmax(throughput, V3()));
    if (lum > p.throughputClamp && lum > 0.0f) {
        return throughput * (p.throughputClamp / std::max(lum, 1.0e-6f));
    }
    return throughput;
}

V3 clampSpecularTail(V3 value, float roughness, V3 f0, const ClampParams& p) {
    if (!finite3(value)) return V3();
    V3 positive = vmax(value, V3());
    if (p.enabled < 0.5f) return positive;
    if (p.metalClamps && p.specularTailClampBase <= 0.0f && p.specularTailClampRoughnessScale <= 0.0f) return positive;   // M:3619-3621
    const float strength = std::max(std::max(f0.x, f0.y), std::max(f0.z, 1.0e-3f));
    float limit = (p.specularTailClampBase + p.specularTailClampRoughnessScale * roughness) * strength;
    limit = std::max(limit, p.clampFloor);
    const float lum = luminance(positive);
    if (lum > limit && lum > 0.0f) positive *= limit / std::max(lum, 1.0e-6f);
    return positive;
}

// ---------------------------------------------------------------------------------------------
// Fresnel / GGX                                                                       E:481-632
// ---------------------------------------------------------------------------------------------
float fresnelDielectricExact(float cosThetaI, float etaI, float etaT, float& outCosThetaT) {
    cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
    const float absCosI = std::fabs(cosThetaI);
    const float sinI2 = std::max(0.0f, 1.0f - absCosI * absCosI);
    const float eta = etaI / etaT;
    const float sinT2 = eta * eta * sinI2;
    if (sinT2 >= 1.0f) {
        outCosThetaT = 0.0f;
        return 1.0f;
    }
    const float cosT = std::sqrt(std::max(0.0f, 1.0f - sinT2));
    outCosThetaT = cosT;
    const float etaICosI = etaI * absCosI;
    const float etaTCosT = etaT * cosT;
    const float rs = (etaICosI - etaTCosT) / (etaICosI + etaTCosT);
    const float rp = (etaT * absCosI - etaI * cosT) / (etaT * absCosI + etaI * cosT);
    return 0.5f * (rs * rs + rp * rp);
}

float dielectricF0FromIor(float ior) {  // E:509-515 (clamped variant, PBR)
    const float eta = std::max(ior, 1.0f);
    const float num = eta - 1.0f;
    const float den = std::max(eta + 1.0f, 1.0e-6f);
    const float f0 = (num / den) * (num / den);
    return clampf(f0, 0.0f, 0.99f);
}

float dielectricF0(float ior) {  // E:667-671 (plastic / coat)
    const float eta = std::max(ior, 1.0f);
    const float r = (eta - 1.0f) / (eta + 1.0f);
    return r * r;
}

V3 fresnelConductor(float cosThetaI, V3 eta, V3 k) {
    cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
    const float cos2 = cosThetaI * cosThetaI;
    const float sin2 = std::max(0.0f, 1.0f - cos2);
    const V3 eta2 = eta * eta;
    const V3 k2 = k * k;
    const V3 t0 = (eta2 - k2) - splat(sin2);
    const V3 a2plusb2 = vsqrt(vmax(t0 * t0 + (4.0f * eta2) * k2, V3()));
    const V3 a = vsqrt(vmax(0.5f * (a2plusb2 + t0), V3()));
    const V3 term1 = a2plusb2 + splat(cos2);
    const V3 term2 = (2.0f * splat(cosThetaI)) * a;
    const V3 rs = (term1 - term2) / (term1 + term2);
    const V3 term3 = splat(cos2) * a2plusb2 + splat(sin2 * sin2);
    const V3 term4 = term2 * splat(sin2);
    const V3 rp = (term3 - term4) / (term3 + term4);
    return vclamp(0.5f * (rs * rs + rp * rp), 0.0f, 1.0f);
}

float ggxLambda(float alpha, float cosTheta) {
    const float absCos = std::fabs(cosTheta);
    if (absCos <= 0.0f) return 0.0f;
    const float sinTheta = std::sqrt(std::max(0.0f, 1.0f - absCos * absCos));
    if (sinTheta == 0.0f) return 0.0f;
    const float a = alpha * (sinTheta / absCos);
    return (-1.0f + std::sqrt(1.0f + a * a)) * 0.5f;
}

float ggxG1(float alpha, float cosTheta) { return 1.0f / (1.0f + ggxLambda(alpha, cosTheta)); }

float ggxDistribution(float alpha, float cosThetaH) {
    const float c = std::fabs(cosThetaH);
    const float a2 = alpha * alpha;
    const float denom = c * c * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * denom * denom);
}

float ggxPdf(float alpha, V3 normal, V3 wo, V3 wi) {  // half-vector pdf, no G1 (E:570-577)
    const V3 wh = normalize(wo + wi);
    const float cosThetaH = dot(normal, wh);
    const float denom = 4.0f * std::max(dot(wo, wh), 1.0e-6f);
    const float d = ggxDistribution(alpha, std::max(cosThetaH, 0.0f));
    return d * std::max(cosThetaH, 0.0f) / denom;
}

// Metal-only (PTR_METAL_SPECULAR): pdf of reflecting a VNDF-sampled half vector, shaders/pathtrace.metal:3724-3739
float ggxPdfVisible(float alpha, V3 normal, V3 wo, V3 wi) {
    const V3 wh = normalize(wo + wi);
    const float cosThetaH = dot(normal, wh), dotWoWh = dot(wo, wh), cosThetaO = dot(normal, wo);
    if (cosThetaO <= 0.0f || cosThetaH <= 0.0f || dotWoWh <= 0.0f) return 0.0f;
    return ggxDistribution(alpha, cosThetaH) * ggxG1(alpha, cosThetaO) * cosThetaH / (4.0f * std::max(dotWoWh, 1.0e-6f));
}

// Metal-only: multiple-scattering compensation of a specular lobe, shaders/pathtrace.metal:4610-4630
V3 specularEnergyCompensation(V3 f0, float roughness, float nov) {
    const float n = clampf(nov, 0.0f, 1.0f);
    const float rx = roughness * -1.0f + 1.0f, ry = roughness * -0.0275f + 0.0425f;
    const float rz = roughness * -0.572f + 1.04f, rw = roughness * 0.022f + -0.04f;
    const float a004 = std::min(rx * rx, std::exp2(-9.28f * n)) * rx + ry;
    const float dfgX = -1.04f * a004 + rz, dfgY = 1.04f * a004 + rw;
    auto channel = [&](float f) {
        const float fss = clampf(f * dfgX + dfgY, 0.0f, 0.99f);
        const float favg = f + (1.0f - f) * (1.0f / 21.0f);
        const float oneMinus = clampf(1.0f - fss, 0.0f, 1.0f);
        const float fms = (favg * oneMinus) / std::max(1.0f - favg * oneMinus, 1.0e-3f);
        return clampf((fss + fms) / std::max(fss, 1.0e-4f), 1.0f, 2.0f);
    };
    return V3(channel(f0.x), channel(f0.y), channel(f0.z));
}

V3 sampleGgxHalfVector(Rng& rng, float alpha, V3 n) {  // E:579-590
    const float u1 = rng.nextFloat();
    const float u2 = rng.nextFloat();
    const float phi = 2.0f * kPi * u1;
    const float denom = 1.0f + (alpha * alpha - 1.0f) * u2;
    const float cosTheta = std::sqrt(std::max((1.0f - u2) / std::max(denom, 1.0e-6f), 0.0f));
    const float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
    const V3 local(std::cos(phi) * sinTheta, std::sin(phi) * sinTheta, cosTheta);
    return normalize(onbToWorld(local, buildOnb(n)));
}

V3 sampleGgxVndf(Rng& rng, float roughness, V3 normal, V3 wo) {  // E:602-632
    const float alpha = std::max(roughness * roughness, 1.0e-4f);
    const Onb onb = buildOnb(normal);
    V3 woLocal = onbToLocal(normalize(wo), onb);
    woLocal.z = std::max(woLocal.z, 1.0e-6f);
    const V3 vh = normalize(V3(alpha * woLocal.x, alpha * woLocal.y, woLocal.z));
    const float lensq = vh.x * vh.x + vh.y * vh.y;
    const V3 t1 = lensq > 0.0f ? V3(-vh.y, vh.x, 0.0f) / std::sqrt(lensq) : V3(1.0f, 0.0f, 0.0f);
    const V3 t2 = cross(vh, t1);
    const float u1 = rng.nextFloat();
    const float u2 = rng.nextFloat();
    const float r = std::sqrt(u1);
    const float phi = 2.0f * kPi * u2;
    const float t1r = r * std::cos(phi);
    const float t2r = r * std::sin(phi);
    const float s = 0.5f * (1.0f + vh.z);
    const float t2Adj = (1.0f - s) * std::sqrt(std::max(0.0f, 1.0f - t1r * t1r)) + s * t2r;
    const float t3 = std::sqrt(std::max(0.0f, (1.0f - t1r * t1r) - t2Adj * t2Adj));
    const V3 nh = (t1r * t1 + t2Adj * t2) + t3 * vh;
    const V3 ne = normalize(V3(alpha * nh.x, alpha * nh.y, std::max(nh.z, 0.0f)));
    return normalize(onbToWorld(ne, onb));
}

// ---------------------------------------------------------------------------------------------
// material accessors                                                                  E:656-872
// ---------------------------------------------------------------------------------------------
V3 baseColor(const PtrMaterial& m) { return vclamp(V3(m.baseColorRoughness), 0.0f, 1.0f); }
float materialRoughness(const PtrMaterial& m) { return std::max(clampf(m.baseColorRoughness[3], 0.0f, 1.0f), 1.0e-3f); }
float coatIor(const PtrMaterial& m) { return std::max(m.typeEta[1], 1.0f); }  // PlasticCoatIor reads typeEta.y
float coatRoughness(const PtrMaterial& m) { return std::max(clampf(m.coatParams[0], 0.0f, 1.0f), 1.0e-3f); }
float coatThickness(const PtrMaterial& m) { return std::max(m.coatParams[1], 0.0f); }
float coatSampleWeight(const PtrMaterial& m) { return clampf(m.coatParams[2], 0.0f, 1.0f); }
float coatFresnelAverage(const PtrMaterial& m) { return clampf(m.coatParams[3], 0.0f, 1.0f); }
V3 coatTint(const PtrMaterial& m) { return vclamp(V3(m.coatTint), 0.0f, 1.0f); }
V3 coatAbsorption(const PtrMaterial& m) { return vmax(V3(m.coatAbsorption), V3()); }
V3 exp3(V3 v) { return {std::exp(v.x), std::exp(v.y), std::exp(v.z)}; }
V3 fract3(V3 v) { return {v.x - std::floor(v.x), v.y - std::floor(v.y), v.z - std::floor(v.z)}; }

V3 plasticSpecularTint(const PtrMaterial& m) {  // E:707-719
    const V3 tint = coatTint(m);
    const float thickness = coatThickness(m);
    if (thickness <= 0.0f) return tint;
    const V3 a = coatAbsorption(m);
    if (a.x <= 1.0e-6f && a.y <= 1.0e-6f && a.z <= 1.0e-6f) return tint;
    return vclamp(tint * exp3(-a * thickness), 0.0f, 1.0f);
}

V3 plasticDiffuseTransmission(const PtrMaterial& m, float cosThetaI, float cosThetaO) {  // E:721-735
    const V3 tint = coatTint(m);
    const float thickness = coatThickness(m);
    if (thickness <= 0.0f) return tint;
    const V3 a = coatAbsorption(m);
    const float ci = std::max(cosThetaI, 1.0e-3f), co = std::max(cosThetaO, 1.0e-3f);
    const V3 attenI = exp3(-a * (thickness / ci));
    const V3 attenO = exp3(-a * (thickness / co));
    return vclamp((tint * attenI) * attenO, 0.0f, 1.0f);
}

float cpBaseMetallic(const PtrMaterial& m) { return clampf(m.carpaintBaseParams[0], 0.0f, 1.0f); }
float cpBaseRoughness(const PtrMaterial& m) { return clampf(m.carpaintBaseParams[1], 0.0f, 1.0f); }
float cpFlakeScale(const PtrMaterial& m) { return std::max(m.carpaintBaseParams[2], 1.0e-4f); }
float cpFlakeSampleWeight(const PtrMaterial& m) { return clampf(m.carpaintFlakeParams[0], 0.0f, 0.95f); }
float cpFlakeRoughness(const PtrMaterial& m) { return clampf(m.carpaintFlakeParams[1], 0.0f, 1.0f); }
float cpFlakeAnisotropy(const PtrMaterial& m) { return clampf(m.carpaintFlakeParams[2], -0.99f, 0.99f); }
float cpFlakeNormalStrength(const PtrMaterial& m) { return clampf(m.carpaintFlakeParams[3], 0.0f, 1.0f); }
float cpCoatSampleWeight(const PtrMaterial& m) { return clampf(m.coatParams[2], 0.0f, 0.95f); }
bool cpHasBaseConductor(const PtrMaterial& m) { return m.carpaintBaseEta[3] > 0.0f || m.carpaintBaseK[3] > 0.0f; }
V3 cpBaseEta(const PtrMaterial& m) { return vmax(V3(m.carpaintBaseEta), V3()); }
V3 cpBaseK(const PtrMaterial& m) { return vmax(V3(m.carpaintBaseK), V3()); }
V3 cpBaseF0(const PtrMaterial& m) {
    return cpHasBaseConductor(m) ? fresnelConductor(1.0f, cpBaseEta(m), cpBaseK(m)) : baseColor(m);
}

V3 carpaintHash3(V3 p) {  // E:794-805
    V3 value = fract3(p * 0.3183099f + V3(0.1f, 0.3f, 0.7f));
    const float d = dot(value, V3(value.y + 33.33f, value.z + 55.55f, value.x + 77.77f));
    value += splat(d);
    const V3 mixed(value.x + value.y, value.x + value.z, value.y + value.z);
    return fract3(mixed * 13.5453123f);
}

V3 carpaintFlakeNormal(const PtrMaterial& m, V3 position, V3 normal) {  // E:807-827
    const V3 rnd = carpaintHash3(position * cpFlakeScale(m));
    const float anis = cpFlakeAnisotropy(m);
    const float ax = std::max(1.0f - anis, 1.0e-3f);
    const float ay = std::max(1.0f + anis, 1.0e-3f);
    const float phi = 2.0f * kPi * rnd.x;
    const float r = std::sqrt(std::max(rnd.y, 1.0e-4f));
    const float x = r * std::cos(phi) * ax;
    const float y = r * std::sin(phi) * ay;
    const float m2 = clampf(x * x + y * y, 0.0f, 0.99f);
    const float z = std::sqrt(std::max(1.0f - m2, 0.0f));
    const Onb onb = buildOnb(normal);
    const V3 perturbed = normalize((x * onb.tangent + y * onb.bitangent) + z * onb.normal);
    const float strength = cpFlakeNormalStrength(m);
    return normalize(normal * (1.0f - strength) + perturbed * strength);
}

bool hasConductorIor(const PtrMaterial& m) {  // E:829-834
    return m.conductorEta[3] > 0.0f || m.conductorK[3] > 0.0f || m.conductorEta[0] > 0.0f || m.conductorEta[1] > 0.0f ||
           m.conductorEta[2] > 0.0f || m.conductorK[0] > 0.0f || m.conductorK[1] > 0.0f || m.conductorK[2] > 0.0f;
}

V3 conductorF0(const PtrMaterial& m) {
    return hasConductorIor(m) ? fresnelConductor(1.0f, V3(m.conductorEta), V3(m.conductorK)) : baseColor(m);
}

bool materialIsDelta(const PtrMaterial& m) {  // E:849-861
    const uint32_t type = matType(m);
    if (type == PTR_MAT_DIELECTRIC) return true;
    if (type == PTR_MAT_METAL) return materialRoughness(m) <= 1.0e-3f;
    return false;
}

float pbrSpecularWeight(V3 f0) { return clampf(std::max(f0.x, std::max(f0.y, f0.z)), 0.05f, 0.95f); }

float lambertPdf(V3 normal, V3 direction) {
    const float c = std::max(dot(normal, normalize(direction)), 0.0f);
    return c > 0.0f ? (c / kPi) : 0.0f;
}

// Microfacet specular term F*D*G/(4 cosO cosI) shared by every glossy branch.
V3 specTerm(V3 F, float alpha, V3 normal, V3 wh, float cosThetaO, float cosThetaI) {
    const float D = ggxDistribution(alpha, dot(normal, wh));
    const float G = ggxG1(alpha, cosThetaO) * ggxG1(alpha, cosThetaI);
    const float denom = 4.0f * cosThetaO * cosThetaI;
    return F * (D * G / std::max(denom, 1.0e-6f));
}

bool halfVectorUsable(V3 wh, V3 n, V3 wo, V3 wi) { return dot(wh, n) > 0.0f && dot(wo, wh) > 0.0f && dot(wi, wh) > 0.0f; }

// ---------------------------------------------------------------------------------------------
// car paint lobes                                                                     E:1170-1313
// ---------------------------------------------------------------------------------------------
struct LobeResult {
    V3 value;
    float pdf = 0.0f;
};

LobeResult carpaintEvalCoat(const PtrMaterial& m, V3 normal, V3 wo, V3 wi, const ClampParams& cp) {
    LobeResult r;
    const float cosO = std::max(dot(normal, wo), 0.0f), cosI = std::max(dot(normal, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const float roughness = coatRoughness(m);
    const float alpha = roughness * roughness;
    const V3 wh = normalize(wo + wi);
    if (!halfVectorUsable(wh, normal, wo, wi)) return r;
    const V3 f0 = splat(dielectricF0(coatIor(m)));
    V3 spec = specTerm(schlickFresnel(f0, dot(wi, wh)), alpha, normal, wh, cosO, cosI);
    spec = clampSpecularTail(spec, roughness, f0, cp);
    spec *= plasticSpecularTint(m);
    spec = vmax(spec, V3());
    const float pdf = ggxPdf(alpha, normal, wo, wi);
    if (pdf > 0.0f) {
        r.pdf = clampSpecularPdf(pdf, cp);
        r.value = spec;
    }
    return r;
}

LobeResult carpaintEvalFlake(const PtrMaterial& m, V3 position, V3 normal, V3 wo, V3 wi, const ClampParams& cp) {
    LobeResult r;
    const V3 fn = carpaintFlakeNormal(m, position, normal);
    const float cosO = std::max(dot(fn, wo), 0.0f), cosI = std::max(dot(fn, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const float roughness = std::max(cpFlakeRoughness(m), 1.0e-3f);
    const float alpha = roughness * roughness;
    const V3 wh = normalize(wo + wi);
    if (!halfVectorUsable(wh, fn, wo, wi)) return r;
    const V3 f0 = cpBaseF0(m);
    V3 spec = specTerm(schlickFresnel(f0, dot(wi, wh)), alpha, fn, wh, cosO, cosI);
    spec = clampSpecularTail(spec * plasticSpecularTint(m), roughness, f0, cp);
    spec *= std::max(1.0f - coatFresnelAverage(m), 0.0f);
    spec = vmax(spec, V3());
    const float pdf = ggxPdf(alpha, fn, wo, wi);
    if (pdf > 0.0f) {
        r.pdf = clampSpecularPdf(pdf, cp);
        r.value = spec;
    }
    return r;
}

LobeResult carpaintEvalBase(const PtrMaterial& m, V3 normal, V3 wo, V3 wi, const ClampParams& cp) {
    LobeResult r;
    const float cosO = std::max(dot(normal, wo), 0.0f), cosI = std::max(dot(normal, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const float metallic = cpBaseMetallic(m);
    const float diffuseWeight = std::max(1.0f - metallic, 0.0f);
    const float specWeight = std::max(metallic, 0.0f);
    if (diffuseWeight <= 1.0e-4f && specWeight <= 1.0e-4f) return r;
    const float coatAvg = coatFresnelAverage(m);
    const V3 base = baseColor(m);
    V3 combined;
    float pdfDiffuse = 0.0f, pdfSpec = 0.0f;
    if (diffuseWeight > 1.0e-4f) {
        V3 diffuse = base / kPi;
        diffuse *= plasticDiffuseTransmission(m, cosI, cosO) * std::max(1.0f - coatAvg, 0.0f);
        diffuse = vmax(diffuse, V3());
        combined += diffuseWeight * diffuse;
        pdfDiffuse = lambertPdf(normal, wi);
    }
    if (specWeight > 1.0e-4f) {
        const float roughness = std::max(cpBaseRoughness(m), 1.0e-3f);
        const float alpha = roughness * roughness;
        const V3 wh = normalize(wo + wi);
        if (halfVectorUsable(wh, normal, wo, wi)) {
            const bool conductor = cpHasBaseConductor(m);
            const V3 eta = cpBaseEta(m), k = cpBaseK(m);
            const V3 f0 = conductor ? fresnelConductor(1.0f, eta, k) : base;
            const V3 F = conductor ? fresnelConductor(dot(wi, wh), eta, k) : schlickFresnel(base, dot(wi, wh));
            V3 spec = specTerm(F, alpha, normal, wh, cosO, cosI);
            spec = clampSpecularTail((spec * plasticSpecularTint(m)) * std::max(1.0f - coatAvg, 0.0f), roughness, f0, cp);
            spec = vmax(spec, V3());
            combined += specWeight * spec;
            const float pdf = ggxPdf(alpha, normal, wo, wi);
            if (pdf > 0.0f) pdfSpec = clampSpecularPdf(pdf, cp);
        }
    }
    r.value = vmax(combined, V3());
    r.pdf = diffuseWeight * pdfDiffuse + specWeight * pdfSpec;
    return r;
}

void carpaintLobeWeights(const PtrMaterial& m, float& pCoat, float& pFlake, float& pBase) {  // E:1459-1471 / 1747-1759
    pCoat = cpCoatSampleWeight(m);
    pFlake = cpFlakeSampleWeight(m);
    pBase = std::max(1.0f - (pCoat + pFlake), 0.0f);
    float norm = (pCoat + pFlake) + pBase;
    if (norm <= 1.0e-6f) {
        pBase = 1.0f;
        pCoat = 0.0f;
        pFlake = 0.0f;
        norm = 1.0f;
    }
    pCoat /= norm;
    pFlake /= norm;
    pBase /= norm;
}

// Plastic: coat specular + attenuated diffuse, shared by eval and sample (E:1374-1421, 1626-1657).
void plasticLobes(const PtrMaterial& m, V3 normal, V3 wo, V3 wi, float cosO, float cosI, const ClampParams& cp,
                  V3& specular, float& specularPdf, V3& diffuse) {
    const V3 albedo = baseColor(m);
    const float cr = coatRoughness(m);
    const float alpha = cr * cr;
    const V3 f0 = splat(dielectricF0(coatIor(m)));
    const V3 specTint = plasticSpecularTint(m);
    specular = V3();
    specularPdf = 0.0f;
    const V3 wh = normalize(wo + wi);
    if (halfVectorUsable(wh, normal, wo, wi)) {
        specular = specTerm(schlickFresnel(f0, dot(wi, wh)), alpha, normal, wh, cosO, cosI);
        specular = clampSpecularTail(specular, cr, f0, cp);
        specular *= specTint;
        const float pdf = ggxPdf(alpha, normal, wo, wi);
        if (pdf > 0.0f) specularPdf = clampSpecularPdf(pdf, cp);
        specular = vmax(specular, V3());
    }
    diffuse = albedo / kPi;
    const V3 tint = plasticDiffuseTransmission(m, cosI, cosO);
    const V3 Fi = schlickFresnel(f0, cosI);
    const V3 Fo = schlickFresnel(f0, cosO);
    diffuse *= tint;
    diffuse *= (V3(1.0f, 1.0f, 1.0f) - Fi) * (V3(1.0f, 1.0f, 1.0f) - Fo);
    diffuse *= std::max(1.0f - coatFresnelAverage(m), 0.0f);
    diffuse = vmax(diffuse, V3());
}

V3 metalFresnel(const PtrMaterial& m, V3 f0, float cosTheta) {
    return hasConductorIor(m) ? fresnelConductor(cosTheta, V3(m.conductorEta), V3(m.conductorK)) : schlickFresnel(f0, cosTheta);
}

// ---------------------------------------------------------------------------------------------
// lights                                                                              E:917-1168
// ---------------------------------------------------------------------------------------------
struct RectLight {  // RectLightInfo E:109-119
    uint32_t rectIndex = 0;
    V3 corner, edgeU, edgeV, normal, baseEmission;
    bool twoSided = false, emissionUsesEnv = false;
    float area = 0.0f;
};

struct RectLightSample {
    V3 direction, emission;
    float distance = 0.0f, pdf = 0.0f;
};

V3 offsetRayOrigin(const HitInfo& hit, V3 direction) {  // E:917-931
    V3 normal = hit.shadingNormal;
    if (dot(normal, normal) <= 0.0f) normal = hit.normal;
    if (dot(normal, normal) <= 0.0f) normal = V3(0.0f, 1.0f, 0.0f);
    normal = normalize(normal);
    const float sign = dot(direction, normal) >= 0.0f ? 1.0f : -1.0f;
    const float distance = std::max(std::fabs(hit.t) * 1.0e-4f, kEpsilon);
    V3 origin = hit.position + normal * (sign * distance);
    origin += (direction * kEpsilon) * 0.5f;
    return origin;
}

bool sampleRectLight(const std::vector<RectLight>& lights, const EnvMap* env, const PtrSettings& s, const HitInfo& hit,
                     Rng& rng, RectLightSample& out) {  // E:987-1055
    out = RectLightSample{};
    if (lights.empty()) return false;
    const uint32_t selected = std::min(static_cast<uint32_t>(rng.nextFloat() * lights.size()),
                                       static_cast<uint32_t>(lights.size() - 1u));
    const RectLight& light = lights[selected];
    const float u = rng.nextFloat();
    const float v = rng.nextFloat();
    const V3 samplePoint = (light.corner + u * light.edgeU) + v * light.edgeV;
    const V3 toLight = samplePoint - hit.position;
    const float distSq = dot(toLight, toLight);
    if (distSq <= 0.0f) return false;
    const float distance = std::sqrt(distSq);
    const V3 direction = toLight / distance;
    if (light.area <= 0.0f) return false;
    float cosLight = dot(-direction, light.normal);
    if (light.twoSided) {
        cosLight = std::fabs(cosLight);
    } else if (cosLight <= 0.0f) {
        return false;
    }
    if (cosLight <= 0.0f) return false;
    const float pdfArea = 1.0f / light.area;
    const float pdfDir = pdfArea * distSq / std::max(cosLight, 1.0e-6f);
    const float selectionPdf = 1.0f / static_cast<float>(lights.size());
    const float pdf = pdfDir * selectionPdf;
    if (!(pdf > 0.0f) || !std::isfinite(pdf)) return false;
    V3 emission = light.baseEmission;
    if (light.emissionUsesEnv && env && env->width > 0 && env->height > 0 && env->rgba) {
        emission *= sampleEnvironment(*env, -light.normal, s.environmentRotation, s.environmentIntensity);
    }
    if (!(dot(emission, emission) > 0.0f)) return false;
    out.direction = direction;
    out.distance = distance;
    out.pdf = pdf;
    out.emission = emission;
    return true;
}

float rectLightPdfForHit(const std::vector<int32_t>& lightIndexByRect, const PtrRect* rects, uint32_t rectCount,
                         uint32_t lightCount, const HitInfo& lightHit, V3 origin) {  // E:1057-1111
    if (lightCount == 0 || !rects || rectCount == 0) return 0.0f;
    if (lightHit.primitiveType != GeomType::Rectangles) return 0.0f;
    const uint32_t rectIndex = lightHit.primitiveIndex;
    if (rectIndex >= rectCount) return 0.0f;
    if (lightIndexByRect.empty() || rectIndex >= lightIndexByRect.size() || lightIndexByRect[rectIndex] < 0) return 0.0f;
    const PtrRect& rect = rects[rectIndex];
    const float area = length(cross(V3(rect.edgeU), V3(rect.edgeV)));
    if (area <= 0.0f) return 0.0f;
    const V3 toLight = lightHit.position - origin;
    const float distSq = dot(toLight, toLight);
    if (distSq <= 0.0f) return 0.0f;
    const float distance = std::sqrt(distSq);
    const V3 direction = toLight / distance;
    float cosLight = dot(-direction, V3(rect.normalAndPlane));
    if (rect.materialTwoSided[1] != 0u) {
        cosLight = std::fabs(cosLight);
    } else if (cosLight <= 0.0f) {
        return 0.0f;
    }
    if (cosLight <= 0.0f) return 0.0f;
    const float pdfArea = 1.0f / area;
    const float pdfDir = pdfArea * distSq / std::max(cosLight, 1.0e-6f);
    return pdfDir * (1.0f / static_cast<float>(lightCount));
}

struct RectLightHit {
    V3 emission;
    float pdf = 0.0f;
};

bool rectLightHitInfo(const std::vector<int32_t>& lightIndexByRect, const std::vector<RectLight>& lights,
                      const PtrRect* rects, uint32_t rectCount, const EnvMap* env, const PtrSettings& s,
                      const HitInfo& hit, V3 origin, RectLightHit& out) {  // E:1113-1168
    out = RectLightHit{};
    if (lights.empty() || !rects || rectCount == 0) return false;
    if (hit.primitiveType != GeomType::Rectangles) return false;
    const uint32_t rectIndex = hit.primitiveIndex;
    if (rectIndex >= rectCount || rectIndex >= lightIndexByRect.size()) return false;
    const int32_t li = lightIndexByRect[rectIndex];
    if (li < 0) return false;
    const RectLight& light = lights[static_cast<size_t>(li)];
    if (!hit.frontFace && !light.twoSided) return false;
    V3 emission = light.baseEmission;
    if (light.emissionUsesEnv && env && env->width > 0 && env->height > 0 && env->rgba) {
        emission *= sampleEnvironment(*env, -hit.shadingNormal, s.environmentRotation, s.environmentIntensity);
    }
    if (!(dot(emission, emission) > 0.0f)) return false;
    const float pdf = rectLightPdfForHit(lightIndexByRect, rects, rectCount, static_cast<uint32_t>(lights.size()), hit, origin);
    if (!(pdf > 0.0f) || !std::isfinite(pdf)) return false;
    out.emission = emission;
    out.pdf = pdf;
    return true;
}

}  // namespace

// =============================================================================================
// public pieces
// =============================================================================================

Camera buildCamera(const PtrSettings& s) {  // E:150-198
    Camera c;
    const uint32_t width = s.width, height = s.height;
    const float aspect = width > 0 ? static_cast<float>(width) / static_cast<float>(height) : 1.0f;
    const float vfov = clampf(s.cameraVerticalFov, 1.0f, 179.0f);
    const float defocusAngle = std::max(s.cameraDefocusAngle, 0.0f);
    const float theta = degToRad(vfov);
    const float h = std::tan(theta * 0.5f);
    const float viewportHeight = 2.0f * h;
    const float viewportWidth = aspect * viewportHeight;
    const float distance = std::max(s.cameraDistance, 0.1f);
    const float cosPitch = std::cos(s.cameraPitch), sinPitch = std::sin(s.cameraPitch);
    const float cosYaw = std::cos(s.cameraYaw), sinYaw = std::sin(s.cameraYaw);
    const V3 offset(distance * cosPitch * cosYaw, distance * sinPitch, distance * cosPitch * sinYaw);
    const V3 lookAt(s.cameraTarget);
    const V3 lookFrom = lookAt + offset;
    const V3 w = normalize(lookFrom - lookAt);
    const V3 u = normalize(cross(V3(0.0f, 1.0f, 0.0f), w));
    const V3 v = cross(w, u);
    float focusDist = s.cameraFocusDistance;
    if (focusDist <= 0.0f) focusDist = distance;
    c.horizontal = (focusDist * viewportWidth) * u;
    c.vertical = (focusDist * viewportHeight) * v;
    c.lowerLeft = ((lookFrom - 0.5f * c.horizontal) - 0.5f * c.vertical) - focusDist * w;
    c.origin = lookFrom;
    c.u = u;
    c.v = v;
    c.lensRadius = focusDist * std::tan(degToRad(defocusAngle * 0.5f));
    return c;
}

Ray generateCameraRay(const Camera& cam, uint32_t width, uint32_t height, uint32_t x, uint32_t y, Rng& rng) {  // E:200-232
    const float u = (static_cast<float>(x) + rng.nextFloat()) / static_cast<float>(width);
    float v = (static_cast<float>(y) + rng.nextFloat()) / static_cast<float>(height);
    v = 1.0f - v;
    V3 direction = ((cam.lowerLeft + u * cam.horizontal) + v * cam.vertical) - cam.origin;
    V3 origin = cam.origin;
    if (cam.lensRadius > 0.0f) {
        V3 disk;  // SampleInUnitDisk: up to 8 rejection pairs, else (0,0,0)
        for (int i = 0; i < 8; ++i) {
            const float dx = rng.nextFloat() * 2.0f - 1.0f;
            const float dy = rng.nextFloat() * 2.0f - 1.0f;
            if (dx * dx + dy * dy <= 1.0f) {
                disk = V3(dx, dy, 0.0f);
                break;
            }
        }
        disk = disk * cam.lensRadius;
        const V3 offset = cam.u * disk.x + cam.v * disk.y;
        origin += offset;
        direction -= offset;
    }
    return Ray{origin, normalize(direction)};
}

ClampParams makeClampParams(const PtrSettings& s) {  // E:381-391
    ClampParams p;
    p.clampFactor = std::max(s.fireflyClampFactor, 0.0f);
    p.clampFloor = std::max(s.fireflyClampFloor, 0.0f);
    p.throughputClamp = std::max(s.throughputClamp, 0.0f);
    p.specularTailClampBase = std::max(s.specularTailClampBase, 0.0f);
    p.specularTailClampRoughnessScale = std::max(s.specularTailClampRoughnessScale, 0.0f);
    p.minSpecularPdf = std::max(s.minSpecularPdf, 1.0e-8f);
    p.enabled = s.fireflyClampEnabled ? 1.0f : 0.0f;
    p.thinDielectrics = (s.metalSemantics & PTR_METAL_THIN) != 0u;   // Metal-only semantics, off on the Embree path
    p.metalSpecular = (s.metalSemantics & PTR_METAL_SPECULAR) != 0u;
    p.metalSss = (s.metalSemantics & PTR_METAL_SSS) != 0u;
    p.sssMode = s.sssMode;
    p.sssMaxSteps = std::max(s.sssMaxSteps, 1u);
    p.metalPbr = (s.metalSemantics & PTR_METAL_PBR) != 0u;
    p.metalClamps = (s.metalSemantics & PTR_METAL_CLAMPS) != 0u;
    p.maxContribution = std::max(s.fireflyClampMaxContribution, 0.0f);
    p.minSpecularPdfRaw = s.minSpecularPdf;
    return p;
}

V3 sampleEnvironment(const EnvMap& env, V3 direction, float rotation, float intensity) {  // E:241-291
    if (env.width == 0 || env.height == 0 || !env.rgba) return V3();
    const V3 rotated = rotateIntoMap(normalize(direction), rotation);
    const float u = (std::atan2(rotated.z, rotated.x) + kPi) / (2.0f * kPi);
    const float v = 0.5f - std::asin(clampf(rotated.y, -1.0f, 1.0f)) / kPi;
    const float fx = u * static_cast<float>(env.width) - 0.5f;
    const float fy = v * static_cast<float>(env.height) - 0.5f;
    int x0 = static_cast<int>(std::floor(fx));
    int y0 = static_cast<int>(std::floor(fy));
    int x1 = x0 + 1, y1 = y0 + 1;
    const float tx = fx - static_cast<float>(x0);
    const float ty = fy - static_cast<float>(y0);
    auto wrap = [](int value, int max) {
        const int m = value % max;
        return m < 0 ? m + max : m;
    };
    const int W = static_cast<int>(env.width), H = static_cast<int>(env.height);
    x0 = wrap(x0, W);
    x1 = wrap(x1, W);
    y0 = std::min(std::max(y0, 0), H - 1);
    y1 = std::min(std::max(y1, 0), H - 1);
    auto fetch = [&](int px, int py) { return V3(env.rgba + (static_cast<size_t>(py) * env.width + static_cast<size_t>(px)) * 4u); };
    const V3 c00 = fetch(x0, y0), c10 = fetch(x1, y0), c01 = fetch(x0, y1), c11 = fetch(x1, y1);
    const V3 c0 = c00 * (1.0f - tx) + c10 * tx;
    const V3 c1 = c01 * (1.0f - tx) + c11 * tx;
    const V3 color = c0 * (1.0f - ty) + c1 * ty;
    return color * std::max(intensity, 0.0f);
}

float environmentPdf(const EnvMap& env, float rotation, V3 direction) {  // E:887-915
    if (!env.hasDistribution || env.width == 0 || env.height == 0 || env.dist.texelPdf.empty()) return 0.0f;
    const V3 rotated = rotateIntoMap(normalize(direction), rotation);
    float u = (std::atan2(rotated.z, rotated.x) + kPi) / (2.0f * kPi);
    float v = 0.5f - std::asin(clampf(rotated.y, -1.0f, 1.0f)) / kPi;
    u = clampf(u, 0.0f, 0.99999994f);
    v = clampf(v, 0.0f, 0.99999994f);
    const uint32_t width = std::max(env.dist.width, 1u), height = std::max(env.dist.height, 1u);
    const uint32_t x = std::min(static_cast<uint32_t>(u * static_cast<float>(width)), width - 1u);
    const uint32_t y = std::min(static_cast<uint32_t>(v * static_cast<float>(height)), height - 1u);
    const size_t index = static_cast<size_t>(y) * width + x;
    if (index >= env.dist.texelPdf.size()) return 0.0f;
    const float value = env.dist.texelPdf[index];
    return (std::isfinite(value) && value > 0.0f) ? value : 0.0f;
}

namespace {
BsdfEval evaluatePbrMetal(const PtrMaterial& m, V3 normal, V3 wo, V3 wi, const ClampParams& cp);
}

BsdfEval evaluateBsdf(const PtrMaterial& m, V3 position, V3 normal, V3 wo, V3 wi, const ClampParams& cp) {  // E:1315-1491
    BsdfEval r;
    if (matType(m) == PTR_MAT_PBR && cp.metalPbr) return evaluatePbrMetal(m, normal, wo, wi, cp);   // before the same-side test below
    const float cosO = std::max(dot(normal, wo), 0.0f);
    const float cosI = std::max(dot(normal, wi), 0.0f);
    if (cosI <= 0.0f || cosO <= 0.0f) return r;
    const uint32_t type = matType(m);
    if (type == PTR_MAT_SUBSURFACE && cp.metalSss) return r;   // M:5078-5084: isBssrdf, value 0, pdf 0 -> no NEE

    if (type == PTR_MAT_LAMBERTIAN || type == PTR_MAT_SUBSURFACE) {
        r.value = baseColor(m) / kPi;
        r.pdf = lambertPdf(normal, wi);
        return r;
    }
    if (type == PTR_MAT_PBR) {
        const V3 base = baseColor(m);
        const float metallic = clampf(m.pbrParams[0], 0.0f, 1.0f);
        const float roughness = clampf(m.baseColorRoughness[3], 0.0f, 1.0f);
        const float df0 = dielectricF0FromIor(m.typeEta[1]);
        const V3 f0 = base * metallic + splat(df0) * (1.0f - metallic);
        const V3 diffuseColor = base * (1.0f - metallic);
        const float alpha = std::max(roughness * roughness, 1.0e-4f);
        const V3 wh = normalize(wo + wi);
        if (!halfVectorUsable(wh, normal, wo, wi)) return r;
        V3 spec = specTerm(schlickFresnel(f0, dot(wi, wh)), alpha, normal, wh, cosO, cosI);
        spec = clampSpecularTail(spec, roughness, f0, cp);
        spec = vmax(spec, V3());
        const V3 diffuse = diffuseColor / kPi;
        const float pdfDiffuse = lambertPdf(normal, wi);
        const float pdfSpec = ggxPdf(alpha, normal, wo, wi);
        const float specPdfClamped = (pdfSpec > 0.0f) ? clampSpecularPdf(pdfSpec, cp) : 0.0f;
        const float specWeight = pbrSpecularWeight(f0);
        const float pdf = specWeight * specPdfClamped + (1.0f - specWeight) * pdfDiffuse;
        if (pdf > 0.0f) {
            r.value = vmax(spec + diffuse, V3());
            r.pdf = pdf;
        }
        return r;
    }
    if (type == PTR_MAT_PLASTIC) {
        V3 specular, diffuse;
        float specularPdf;
        plasticLobes(m, normal, wo, wi, cosO, cosI, cp, specular, specularPdf, diffuse);
        const float pdfDiffuse = lambertPdf(normal, wi);
        const float pCoat = coatSampleWeight(m);
        const float pdf = pCoat * specularPdf + (1.0f - pCoat) * pdfDiffuse;
        if (pdf > 0.0f) {
            r.value = specular + diffuse;
            r.pdf = pdf;
        }
        return r;
    }
    if (type == PTR_MAT_METAL) {
        const float roughness = clampf(m.baseColorRoughness[3], 0.0f, 1.0f);
        if (roughness <= 1.0e-3f) {
            r.isDelta = true;
            return r;
        }
        const float alpha = roughness * roughness;
        const V3 wh = normalize(wo + wi);
        if (!halfVectorUsable(wh, normal, wo, wi)) return r;
        const V3 f0 = conductorF0(m);
        V3 spec = specTerm(metalFresnel(m, f0, dot(wi, wh)), alpha, normal, wh, cosO, cosI);
        if (cp.metalSpecular) spec = spec * specularEnergyCompensation(f0, roughness, cosO);   // M:5011
        spec = clampSpecularTail(spec, roughness, f0, cp);
        const float pdf = cp.metalSpecular ? ggxPdfVisible(alpha, normal, wo, wi) : ggxPdf(alpha, normal, wo, wi);
        if (pdf > 0.0f) {
            r.value = vmax(spec, V3());
            r.pdf = clampSpecularPdf(pdf, cp);
        }
        return r;
    }
    if (type == PTR_MAT_CARPAINT) {
        float pCoat, pFlake, pBase;
        carpaintLobeWeights(m, pCoat, pFlake, pBase);
        const LobeResult coat = carpaintEvalCoat(m, normal, wo, wi, cp);
        const LobeResult flake = carpaintEvalFlake(m, position, normal, wo, wi, cp);
        const LobeResult base = carpaintEvalBase(m, normal, wo, wi, cp);
        r.value = (pBase * base.value + pFlake * flake.value) + pCoat * coat.value;
        r.pdf = (pBase * base.pdf + pFlake * flake.pdf) + pCoat * coat.pdf;
        return r;
    }
    if (type == PTR_MAT_DIELECTRIC) {
        r.isDelta = true;
        return r;
    }
    r.value = baseColor(m) / kPi;
    r.pdf = lambertPdf(normal, wi);
    return r;
}

// ---- separable subsurface scattering of the Metal integrator (Metal-only semantics, PTR_METAL_SSS) ----
namespace {

V3 vmaxs(V3 a, float s) { return {std::max(a.x, s), std::max(a.y, s), std::max(a.z, s)}; }
V3 vsqrt3(V3 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)}; }
V3 vexp3(V3 a) { return {std::exp(a.x), std::exp(a.y), std::exp(a.z)}; }
V3 vclamp01(V3 a) { return {clampf(a.x, 0.0f, 1.0f), clampf(a.y, 0.0f, 1.0f), clampf(a.z, 0.0f, 1.0f)}; }

// sss_sigma_a / sss_sigma_s_prime, M:3916-3950
void sssCoefficients(const PtrMaterial& m, float meanFreePath, float anisotropy, V3& sigmaA, V3& sigmaSPrime) {
    const float reduce = std::max(1.0f - anisotropy, 0.01f);
    if (m.sssSigmaA[3] > 0.5f) {
        sigmaA = vmaxs(V3(m.sssSigmaA[0], m.sssSigmaA[1], m.sssSigmaA[2]), 1.0e-6f);
        sigmaSPrime = vmax(V3(m.sssSigmaS[0], m.sssSigmaS[1], m.sssSigmaS[2]), V3()) * reduce;
        return;
    }
    const float sigmaT = 1.0f / std::max(meanFreePath, 1.0e-4f);
    const V3 bc = baseColor(m);
    V3 sigmaS = V3(clampf(bc.x, 0.0f, 0.999f), clampf(bc.y, 0.0f, 0.999f), clampf(bc.z, 0.0f, 0.999f)) * sigmaT;
    sigmaS = vmax(sigmaS, V3()) * reduce;
    sigmaA = vmaxs(V3(sigmaT, sigmaT, sigmaT) - sigmaS, 1.0e-6f);
    sigmaSPrime = sigmaS;
}

// normalized_diffusion_profile, M:3952-3971
V3 diffusionProfile(float radius, V3 sigmaA, V3 sigmaSPrime) {
    const V3 one(1.0f, 1.0f, 1.0f);
    const V3 sigmaTPrime = vmaxs(sigmaA + sigmaSPrime, 1.0e-6f);
    const V3 alphaPrime = vclamp01(sigmaSPrime / sigmaTPrime);
    const V3 D = one / vmaxs(3.0f * sigmaTPrime, 1.0e-6f);
    const V3 sigmaTr = vsqrt3(vmaxs(sigmaA / D, 1.0e-6f));
    const float r = std::max(radius, 1.0e-4f);
    const V3 rr(r * r, r * r, r * r);
    const V3 zr = one / sigmaTPrime;
    const V3 dr = vsqrt3(rr + zr * zr);
    const V3 vr = zr + 4.0f * D;
    const V3 dv = vsqrt3(rr + vr * vr);
    const V3 expDr = vexp3(-(sigmaTr * dr));
    const V3 expDv = vexp3(-(sigmaTr * dv));
    const V3 termDr = (zr * (one + sigmaTr * dr)) / vmaxs(dr * dr * dr, 1.0e-6f);
    const V3 termDv = (vr * (one + sigmaTr * dv)) / vmaxs(dv * dv * dv, 1.0e-6f);
    return vmax((alphaPrime / (4.0f * kPi)) * (termDr * expDr + termDv * expDv), V3());
}

// sss_sigma_tr_scalar, M:3973-3980
float sssSigmaTrScalar(V3 sigmaA, V3 sigmaSPrime) {
    const V3 sigmaTPrime = vmaxs(sigmaA + sigmaSPrime, 1.0e-6f);
    const V3 D = V3(1.0f, 1.0f, 1.0f) / vmaxs(3.0f * sigmaTPrime, 1.0e-6f);
    return std::max(luminance(vsqrt3(vmaxs(sigmaA / D, 1.0e-6f))), 1.0e-4f);
}

// The separable branch of sample_bsdf case 5, M:5398-5481.  false: not applicable / gave up (Lambert fallback follows,
// with the generator wherever this left it).
bool sampleSeparableSss(const PtrMaterial& m, V3 position, V3 normal, V3 wo, Rng& rng, const ClampParams& cp, BsdfSample& r) {
    const float meanFreePath = std::max(m.sssParams[0], 1.0e-4f);
    if (!(cp.sssMode == 1u && m.sssParams[1] < 0.5f && meanFreePath > 1.0e-4f)) return false;
    const float anisotropy = clampf(m.sssSigmaS[3], -0.99f, 0.99f);
    V3 sigmaA, sigmaSPrime;
    sssCoefficients(m, meanFreePath, anisotropy, sigmaA, sigmaSPrime);
    const float sigmaTr = sssSigmaTrScalar(sigmaA, sigmaSPrime);
    if (sigmaTr <= 0.0f) return false;
    const float u = clampf(rng.nextFloat(), 1.0e-6f, 1.0f - 1.0e-6f);   // sample_sss_radius, M:3982-3986
    float radius = -std::log(1.0f - u) / std::max(sigmaTr, 1.0e-4f);
    radius = std::min(radius, meanFreePath * 10.0f);
    const float sigma = std::max(sigmaTr, 1.0e-4f);                      // pdf_sss_radius, M:3988-3994
    const float pdfRadius = radius <= 0.0f ? 0.0f : sigma * std::exp(-sigma * radius);
    if (pdfRadius <= 0.0f || !std::isfinite(pdfRadius)) return false;
    const float phi = 2.0f * kPi * rng.nextFloat();
    const Onb onb = buildOnb(normal);
    const V3 exitPoint = position + onb.tangent * (radius * std::cos(phi)) + onb.bitangent * (radius * std::sin(phi));
    float pdfDir = 0.0f;
    const V3 wi = sampleCosineHemisphere(rng, normal, pdfDir);
    const float cosExit = dot(normal, wi);
    pdfDir = cosExit > 0.0f ? cosExit / kPi : 0.0f;   // lambert_pdf, M:967-971
    const float pdfArea = pdfRadius / (2.0f * kPi * std::max(radius, 1.0e-4f));
    if (cosExit <= 0.0f || pdfDir <= 0.0f || pdfArea <= 0.0f) return false;
    V3 profile = diffusionProfile(radius, sigmaA, sigmaSPrime);
    const float coatAverage = 1.0f - clampf(m.coatParams[3], 0.0f, 1.0f);
    float coatTransmission = 1.0f;
    if (m.sssParams[2] > 0.5f) {
        const float coatIor = std::max(m.typeEta[2], 1.0f);
        float f0 = (coatIor - 1.0f) / (coatIor + 1.0f);
        f0 *= f0;
        const float transIn = 1.0f - (f0 + (1.0f - f0) * schlickWeight(std::max(dot(normal, wo), 0.0f)));
        const float transOut = 1.0f - (f0 + (1.0f - f0) * schlickWeight(cosExit));
        coatTransmission = clampf(transIn * transOut, 0.0f, 1.0f);
        profile = profile * vclamp01(V3(m.coatTint[0], m.coatTint[1], m.coatTint[2]));
    }
    const float denom = std::max(pdfArea * pdfDir, 1.0e-6f);
    const V3 weight = vmax((profile * (cosExit * coatAverage * coatTransmission)) / denom, V3());
    if (!finite3(weight)) return false;
    r.direction = wi;
    r.weight = weight;
    r.pdf = pdfDir;   // the reference's pdf is area x directional; the directional one is what the next emitter hit's MIS uses (M:7269)
    r.isDelta = false;
    r.hasExitPoint = true;
    r.exitPoint = exitPoint;
    return true;
}

// Origin of the ray leaving a subsurface exit point, M:6740-6766 (offset_surface_point M:1210-1220 + the two biases)
V3 sssExitOrigin(V3 exitPoint, V3 exitNormal, V3 direction) {
    const V3 n = (finite3(exitNormal) && dot(exitNormal, exitNormal) > 0.0f) ? normalize(exitNormal) : V3(0.0f, 1.0f, 0.0f);
    const float sign = dot(direction, n) >= 0.0f ? 1.0f : -1.0f;
    V3 o = exitPoint + n * (sign * kEpsilon * 4.0f);
    o += (direction * kEpsilon) * 0.5f;
    o += n * std::max(5.0e-3f * 4.0f, kEpsilon * 32.0f);
    const V3 d = (finite3(direction) && dot(direction, direction) > 0.0f) ? normalize(direction) : n;
    o += d * std::max(5.0e-3f * 8.0f, kEpsilon * 32.0f);
    return o;
}


// ---- random-walk subsurface scattering of the Metal integrator, sample_sss_random_walk_software M:4060-4311 ----
// (Metal-only semantics, PTR_METAL_SSS with sssMode == 2 on front-face hits of materials whose sssParams.y >= 0.5.)
// The reference runs the walk as a loop of closest-hit queries inside the sampling step; here it is cut into "begin" and
// "step" so the HIP wavefront, where every query is one extend/shade iteration, runs the very same two functions' twins.
struct SssWalk {
    V3 position, direction, throughput;
    uint32_t step = 0;
};
enum class WalkOutcome { Fallback, Sample, Walking };

V3 refractMetal(V3 i, V3 n, float eta) {   // MSL refract(): zero vector on total internal reflection
    const float ndi = dot(n, i);
    const float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return V3();
    return eta * i - (eta * ndi + std::sqrt(k)) * n;
}

V3 offsetSurfacePoint(V3 point, V3 normal, V3 direction) {  // M:1210-1220
    const V3 n = (finite3(normal) && dot(normal, normal) > 0.0f) ? normalize(normal) : V3(0.0f, 1.0f, 0.0f);
    const float sign = dot(direction, n) >= 0.0f ? 1.0f : -1.0f;
    V3 origin = point + n * (sign * kEpsilon * 4.0f);
    origin += (direction * kEpsilon) * 0.5f;
    return origin;
}

void sssWalkMedium(const PtrMaterial& m, V3& sigmaT, V3& sigmaSPrime, float& sigmaTScalar, float& anisotropy) {  // M:4158-4166
    anisotropy = clampf(m.sssSigmaS[3], -0.99f, 0.99f);
    const float meanFreePath = std::max(m.sssParams[0], 1.0e-4f);
    V3 sigmaA;
    sssCoefficients(m, meanFreePath, anisotropy, sigmaA, sigmaSPrime);
    sigmaT = vmaxs(sigmaA + sigmaSPrime, 1.0e-6f);
    sigmaTScalar = std::max(std::max(sigmaT.x, std::max(sigmaT.y, sigmaT.z)), 1.0e-4f);
}

// Lobe pick, then either the coat reflection (a finished sample) or the refraction into the medium (walk state).
WalkOutcome sssWalkBegin(const PtrMaterial& m, V3 point, V3 entryNormal, V3 wo, V3 incidentDir, Rng& rng, const ClampParams& cp,
                         BsdfSample& sample, SssWalk& walk) {
    const float pCoat = clampf(m.coatParams[2], 0.0f, 1.0f);
    const float randLobe = rng.nextFloat();
    const float roughness = coatRoughness(m);
    const float alpha = roughness * roughness;
    const float eta = coatIor(m);
    const float ratio = (eta - 1.0f) / std::max(eta + 1.0f, 1.0e-6f);
    const V3 f0 = splat(clampf(ratio * ratio, 0.0f, 0.999f));   // plastic_coat_f0, M:3861-3866
    const V3 specTint = plasticSpecularTint(m);
    if (pCoat > 0.0f && randLobe < pCoat) {   // M:4104-4155
        const V3 wh = sampleGgxVndf(rng, roughness, entryNormal, wo);
        if (dot(wh, entryNormal) <= 0.0f) return WalkOutcome::Fallback;
        V3 wi = reflect(-wo, wh);
        if (!(dot(wi, wi) > 0.0f)) return WalkOutcome::Fallback;
        wi = normalize(wi);
        if (!finite3(wi)) return WalkOutcome::Fallback;
        const float cosI = dot(entryNormal, wi), cosO = dot(entryNormal, wo);
        if (cosI <= 0.0f || cosO <= 0.0f) return WalkOutcome::Fallback;
        const float dotWiWh = dot(wi, wh);
        if (dotWiWh <= 0.0f) return WalkOutcome::Fallback;
        const float D = ggxDistribution(alpha, dot(entryNormal, wh));
        const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
        const V3 F = schlickFresnel(f0, dotWiWh);
        V3 spec = F * (D * G / std::max(4.0f * cosO * cosI, 1.0e-6f));
        spec = clampSpecularTail(spec * specTint, roughness, f0, cp);
        const float specPdfRaw = ggxPdfVisible(alpha, entryNormal, wo, wi);
        if (specPdfRaw <= 0.0f) return WalkOutcome::Fallback;
        const float specPdf = clampSpecularPdf(specPdfRaw, cp);
        const float combinedPdf = std::max(pCoat * specPdf, 1.0e-6f);
        const V3 weight = vmax(spec * cosI / combinedPdf, V3());
        if (!finite3(weight)) return WalkOutcome::Fallback;
        sample = BsdfSample();
        sample.direction = wi;
        sample.weight = weight;
        sample.pdf = specPdf;   // directionalPdf: what the next emitter hit's MIS uses (M:7269); pdf = pCoat x this is > 0 with it
        return WalkOutcome::Sample;
    }
    const float pDiffuse = std::max(1.0f - pCoat, 1.0e-3f);
    V3 throughput = splat(1.0f / pDiffuse);
    const float etaInside = std::max(m.typeEta[1], 1.0f);
    const float cosThetaI = dot(-incidentDir, entryNormal);
    if (cosThetaI <= 0.0f) return WalkOutcome::Fallback;
    float cosThetaT = 0.0f;
    const float frEntry = fresnelDielectricExact(cosThetaI, 1.0f, etaInside, cosThetaT);
    V3 enterDir = refractMetal(incidentDir, entryNormal, 1.0f / etaInside);
    if (!finite3(enterDir) || dot(enterDir, enterDir) <= 0.0f) return WalkOutcome::Fallback;
    enterDir = normalize(enterDir);
    const float scaleEntry = (etaInside * etaInside) * (cosThetaT / std::max(cosThetaI, 1.0e-6f));
    throughput *= std::max(1.0f - frEntry, 0.0f) * scaleEntry;
    if (m.sssParams[2] > 0.5f) throughput = throughput * specTint;
    walk.position = offsetSurfacePoint(point, -entryNormal, enterDir);
    walk.direction = enterDir;
    walk.throughput = throughput;
    walk.step = 0;
    return WalkOutcome::Walking;
}

V3 sampleHenyeyGreenstein(V3 reference, float g, Rng& rng) {  // M:4011-4036
    const float u1 = rng.nextFloat();
    const float u2 = rng.nextFloat();
    float cosTheta;
    if (std::fabs(g) < 1.0e-3f) {
        cosTheta = 1.0f - 2.0f * u1;
    } else {
        const float s = (1.0f - g * g) / (1.0f - g + 2.0f * g * u1);
        cosTheta = clampf((1.0f + g * g - s * s) / (2.0f * g), -1.0f, 1.0f);
    }
    const float sinTheta = std::sqrt(std::max(0.0f, 1.0f - cosTheta * cosTheta));
    const float phi = 2.0f * kPi * u2;
    const V3 local(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta);
    if (!(dot(reference, reference) > 0.0f)) return V3();
    const Onb onb = buildOnb(reference);
    const V3 world = (local.x * onb.tangent + local.y * onb.bitangent) + local.z * onb.normal;
    return dot(world, world) > 0.0f ? normalize(world) : V3();
}

// One pass of the reference's loop body, given the closest hit of the ray (walk.position, walk.direction): Walking = go on
// with the updated ray, Sample = the path leaves the medium, Fallback = the walk is abandoned (pdf 0 in the reference).
WalkOutcome sssWalkStep(const PtrMaterial& m, uint32_t maxSteps, SssWalk& walk, bool hitBoundary, float hitT, V3 hitPoint, V3 outwardNormal,
                        Rng& rng, BsdfSample& sample) {
    V3 sigmaT, sigmaSPrime;
    float sigmaTScalar, anisotropy;
    sssWalkMedium(m, sigmaT, sigmaSPrime, sigmaTScalar, anisotropy);
    const float xi = clampf(rng.nextFloat(), 1.0e-6f, 1.0f - 1.0e-6f);
    const float distance = -std::log(1.0f - xi) / sigmaTScalar;
    if (!hitBoundary) return WalkOutcome::Fallback;
    const float boundaryDistance = std::max(hitT, 1.0e-4f);
    const uint32_t limit = std::max(maxSteps, 1u);
    auto goOn = [&]() {
        ++walk.step;
        return walk.step < limit ? WalkOutcome::Walking : WalkOutcome::Fallback;
    };
    if (distance < boundaryDistance) {   // scattering event inside, M:4227-4245
        walk.throughput = walk.throughput * vexp3(-(sigmaT * distance));
        walk.throughput = walk.throughput * vclamp01(sigmaSPrime / vmaxs(sigmaT, 1.0e-6f));
        if (std::max(walk.throughput.x, std::max(walk.throughput.y, walk.throughput.z)) < 1.0e-3f) return WalkOutcome::Fallback;
        walk.position += walk.direction * distance;
        const V3 scattered = sampleHenyeyGreenstein(-walk.direction, anisotropy, rng);
        if (!finite3(scattered) || dot(scattered, scattered) <= 0.0f) return WalkOutcome::Fallback;
        walk.direction = normalize(scattered);
        return goOn();
    }
    walk.throughput = walk.throughput * vexp3(-(sigmaT * boundaryDistance));
    if (std::max(walk.throughput.x, std::max(walk.throughput.y, walk.throughput.z)) < 1.0e-3f) return WalkOutcome::Fallback;
    if (!finite3(outwardNormal) || dot(outwardNormal, outwardNormal) <= 0.0f) return WalkOutcome::Fallback;
    outwardNormal = normalize(outwardNormal);
    const float etaI = std::max(m.typeEta[1], 1.0f);
    // M:4262-4268 as written: the exit is taken only where the geometric normal faces the ray; a ray that reaches the
    // boundary of a closed mesh from inside is reflected back in
    const float cosExitI = dot(-walk.direction, outwardNormal);
    if (cosExitI <= 0.0f) {
        walk.position = hitPoint;
        walk.direction = normalize(reflect(walk.direction, outwardNormal));
        return goOn();
    }
    float cosExitT = 0.0f;
    const float frExit = fresnelDielectricExact(cosExitI, etaI, 1.0f, cosExitT);
    V3 refracted = refractMetal(walk.direction, outwardNormal, etaI);
    if (!finite3(refracted) || dot(refracted, refracted) <= 0.0f) {
        walk.position = hitPoint;
        walk.direction = normalize(reflect(walk.direction, outwardNormal));
        return goOn();
    }
    refracted = normalize(refracted);
    const float scaleExit = (1.0f / (etaI * etaI)) * (cosExitT / std::max(cosExitI, 1.0e-6f));
    V3 throughput = walk.throughput * (std::max(1.0f - frExit, 0.0f) * scaleExit);
    if (m.sssParams[2] > 0.5f) throughput = throughput * plasticSpecularTint(m);
    throughput = vmax(throughput, V3());
    if (!finite3(throughput)) return WalkOutcome::Fallback;
    sample = BsdfSample();
    sample.direction = refracted;
    sample.weight = throughput;
    sample.pdf = 1.0f;   // directionalPdf (M:4297); the reference's pdf = max(pDiffuse, 1e-4) is positive with it
    sample.hasExitPoint = true;
    sample.exitPoint = hitPoint;
    sample.exitNormal = outwardNormal;
    return WalkOutcome::Sample;
}

}  // namespace


// ---- metallic-roughness model of the Metal integrator (PTR_METAL_PBR), evaluate_/sample_pbr_metallic_roughness M:4598-4948 ----
namespace {

struct PbrMetal {
    V3 f0, diffuseColor;
    float roughness = 0.0f, transmission = 0.0f, reflectScale = 1.0f, pSpec = 0.0f, pDiff = 0.0f, pTrans = 0.0f;
    bool valid = false;
};

V3 mix3(V3 x, V3 y, float a) { return x + (y - x) * a; }

PbrMetal loadPbrMetal(const PtrMaterial& m) {  // M:4656-4678 / 4789-4811
    PbrMetal p;
    const V3 base = baseColor(m);
    const float metallic = clampf(m.pbrParams[0], 0.0f, 1.0f);
    p.roughness = clampf(m.baseColorRoughness[3], 0.0f, 1.0f);
    p.f0 = mix3(splat(dielectricF0FromIor(m.typeEta[1])), base, metallic);
    p.diffuseColor = base * (1.0f - metallic);
    // diffuseOcclusion (M:4661): the textured path parks it in the per-hit copy of the material (materialPad[0], flagged by [1])
    if (m.materialPad[1] == 0x4F43434Cu) {
        float occlusion;
        std::memcpy(&occlusion, &m.materialPad[0], 4);
        p.diffuseColor = p.diffuseColor * clampf(occlusion, 0.0f, 1.0f);
    }
    p.transmission = clampf(m.pbrExtras[2], 0.0f, 1.0f) * (1.0f - metallic);
    p.reflectScale = 1.0f - p.transmission;
    const float specWeightBase = pbrSpecularWeight(p.f0);
    const float wSpec = specWeightBase * p.reflectScale, wDiff = (1.0f - specWeightBase) * p.reflectScale, wTrans = p.transmission;
    const float sum = wSpec + wDiff + wTrans;
    if (sum <= 0.0f) return p;
    p.pSpec = wSpec / sum;
    p.pDiff = wDiff / sum;
    p.pTrans = wTrans / sum;
    p.valid = true;
    return p;
}

V3 transmissionTint(const PtrMaterial& m, float cosTheta) {  // M:3295-3306
    const float thickness = std::max(m.typeEta[3], 0.0f);
    if (thickness <= 0.0f) return splat(1.0f);
    const V3 sigmaA = vmax(V3(m.dielectricSigmaA), V3());
    if (sigmaA.x <= 0.0f && sigmaA.y <= 0.0f && sigmaA.z <= 0.0f) return splat(1.0f);
    const float distance = thickness / std::max(std::fabs(cosTheta), 1.0e-3f);
    return vclamp01(vexp3(-(sigmaA * distance)));
}

float ggxVndfPdf(float alpha, V3 normal, V3 wo, V3 wh) {  // M:3741-3754
    const float cosO = dot(normal, wo), cosH = dot(normal, wh);
    if (cosO <= 0.0f || cosH <= 0.0f) return 0.0f;
    return ggxDistribution(alpha, cosH) * ggxG1(alpha, cosO) * cosH / std::max(dot(wo, wh), 1.0e-6f);
}

// the rough transmission term shared by evaluation and sampling (M:4729-4757 / 4906-4930); false = rejected
bool roughTransmission(const PtrMaterial& m, const PbrMetal& p, V3 normal, V3 wo, V3 wi, V3 wh, float eta, float etaI, float etaT, V3& ft,
                       float& pdfTrans) {
    const float absCosO = std::fabs(dot(normal, wo)), absCosI = std::fabs(dot(normal, wi));
    const float cosOWh = dot(wo, wh), cosIWh = dot(wi, wh);
    if (cosOWh * cosIWh > 0.0f) return false;
    const float alpha = std::max(p.roughness * p.roughness, 1.0e-4f);
    const float D = ggxDistribution(alpha, std::max(dot(normal, wh), 0.0f));
    const float G = ggxG1(alpha, absCosO) * ggxG1(alpha, absCosI);
    float cosT = 0.0f;
    const float F = fresnelDielectricExact(cosOWh, etaI, etaT, cosT);
    const float denom = cosOWh + eta * cosIWh;
    const float denomSq = denom * denom;
    if (std::fabs(denomSq) <= 1.0e-8f) return false;
    float factor = (eta * eta) * std::fabs(cosIWh) * std::fabs(cosOWh);
    factor /= std::max(absCosO * absCosI * denomSq, 1.0e-6f);
    ft = splat((1.0f - F) * D * G * factor) * transmissionTint(m, absCosI);
    ft = ft * p.transmission;
    const float pdfWh = ggxVndfPdf(alpha, normal, wo, wh);
    const float dwhDwi = std::fabs((eta * eta * cosIWh) / std::max(denomSq, 1.0e-8f));
    pdfTrans = pdfWh * dwhDwi;
    return true;
}

BsdfEval evaluatePbrMetal(const PtrMaterial& m, V3 normal, V3 wo, V3 wi, const ClampParams& cp) {  // M:4632-4762
    BsdfEval r;
    const float cosO = dot(normal, wo), cosI = dot(normal, wi);
    const float absCosO = std::fabs(cosO), absCosI = std::fabs(cosI);
    if (absCosO <= 0.0f || absCosI <= 0.0f) return r;
    const PbrMetal p = loadPbrMetal(m);
    if (!p.valid) return r;
    if (cosO * cosI > 0.0f) {
        if (cosO <= 0.0f || cosI <= 0.0f) return r;
        const float alpha = std::max(p.roughness * p.roughness, 1.0e-4f);
        const V3 wh = normalize(wo + wi);
        if (dot(wh, normal) <= 0.0f || dot(wo, wh) <= 0.0f || dot(wi, wh) <= 0.0f) return r;
        const float D = ggxDistribution(alpha, dot(normal, wh));
        const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
        V3 spec = schlickFresnel(p.f0, dot(wi, wh)) * (D * G / std::max(4.0f * cosO * cosI, 1.0e-6f));
        spec = spec * specularEnergyCompensation(p.f0, p.roughness, absCosO);
        spec = clampSpecularTail(spec, p.roughness, p.f0, cp);
        spec = spec * p.reflectScale;
        const float pdfSpec = ggxPdfVisible(alpha, normal, wo, wi);
        const V3 diffuse = (p.diffuseColor / kPi) * p.reflectScale;
        const float pdf = p.pSpec * pdfSpec + p.pDiff * lambertPdf(normal, wi);
        if (pdf > 0.0f) {
            r.value = vmax(spec + diffuse, V3());
            r.pdf = clampSpecularPdf(pdf, cp);
        }
        return r;
    }
    if (p.transmission <= 0.0f) return r;
    float etaI = 1.0f, etaT = std::max(m.typeEta[1], 1.0f);
    if (cosO < 0.0f) std::swap(etaI, etaT);
    const float eta = etaI / etaT;
    V3 wh = wo + wi * eta;
    if (!finite3(wh) || dot(wh, wh) <= 0.0f) return r;
    wh = normalize(wh);
    if (dot(wh, normal) <= 0.0f) wh = -wh;
    V3 ft;
    float pdfTrans = 0.0f;
    if (!roughTransmission(m, p, normal, wo, wi, wh, eta, etaI, etaT, ft, pdfTrans)) return r;
    const float pdf = p.pTrans * pdfTrans;
    if (pdf > 0.0f) {
        r.value = vmax(ft, V3());
        r.pdf = clampSpecularPdf(pdf, cp);
    }
    return r;
}

BsdfSample samplePbrMetal(const PtrMaterial& m, V3 normal, V3 wo, V3 incidentDir, Rng& rng, const ClampParams& cp) {  // M:4764-4948
    BsdfSample r;
    const PbrMetal p = loadPbrMetal(m);
    if (!p.valid) return r;
    const float choose = rng.nextFloat();
    V3 wi, f;
    float pdfSpec = 0.0f, pdfDiffuse = 0.0f, pdfTrans = 0.0f;
    bool isDelta = false;
    r.lobe = choose < p.pSpec ? 1 : (choose < p.pSpec + p.pDiff ? 0 : 2);
    r.lobeRoughness = p.roughness;
    if (choose < p.pSpec) {
        if (p.roughness <= 1.0e-3f) {
            wi = reflect(incidentDir, normal);
            if (dot(normal, wi) <= 0.0f) return r;
            pdfSpec = 1.0f;
            f = schlickFresnel(p.f0, std::max(dot(normal, wo), 0.0f)) * p.reflectScale;
            isDelta = true;
        } else {
            const V3 wh = sampleGgxVndf(rng, p.roughness, normal, wo);
            wi = reflect(-wo, wh);
            const float cosI = dot(normal, wi);
            if (cosI <= 0.0f) return r;
            const float alpha = std::max(p.roughness * p.roughness, 1.0e-4f);
            const float cosO = std::max(dot(normal, wo), 0.0f);
            const float D = ggxDistribution(alpha, dot(normal, wh));
            const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
            f = schlickFresnel(p.f0, dot(wi, wh)) * (D * G / std::max(4.0f * cosO * cosI, 1.0e-6f));
            f = f * specularEnergyCompensation(p.f0, p.roughness, cosO);
            f = clampSpecularTail(f, p.roughness, p.f0, cp);
            f = f * p.reflectScale;
            pdfSpec = ggxPdfVisible(alpha, normal, wo, wi);
        }
    } else if (choose < p.pSpec + p.pDiff) {
        float unused = 0.0f;
        wi = sampleCosineHemisphere(rng, normal, unused);
        if (dot(normal, wi) <= 0.0f) return r;
        f = (p.diffuseColor / kPi) * p.reflectScale;
        pdfDiffuse = lambertPdf(normal, wi);
    } else {
        const float cosO = dot(normal, wo), absCosO = std::fabs(cosO);
        float etaI = 1.0f, etaT = std::max(m.typeEta[1], 1.0f);
        if (cosO < 0.0f) std::swap(etaI, etaT);
        const float eta = etaI / etaT;
        if (p.roughness <= 1.0e-3f) {
            wi = refractMetal(-wo, normal, eta);
            if (dot(wi, wi) <= 0.0f) return r;
            wi = normalize(wi);
            float cosT = 0.0f;
            const float Fr = fresnelDielectricExact(cosO, etaI, etaT, cosT);
            const float directionScale = ((etaT * etaT) / (etaI * etaI)) * (std::fabs(cosT) / std::max(absCosO, 1.0e-6f));
            f = splat(std::max(1.0f - Fr, 0.0f) * directionScale) * transmissionTint(m, std::fabs(dot(normal, wi))) * p.transmission;
            pdfTrans = 1.0f;
            isDelta = true;
        } else {
            const V3 wh = sampleGgxVndf(rng, p.roughness, normal, wo);
            wi = refractMetal(-wo, wh, eta);
            if (dot(wi, wi) <= 0.0f) return r;
            wi = normalize(wi);
            if (dot(wi, normal) * cosO >= 0.0f) return r;
            if (!roughTransmission(m, p, normal, wo, wi, wh, eta, etaI, etaT, f, pdfTrans)) return r;
        }
    }
    const float absCosI = std::fabs(dot(normal, wi));
    if (absCosI <= 0.0f) return r;
    const float pdf = p.pSpec * pdfSpec + p.pDiff * pdfDiffuse + p.pTrans * pdfTrans;
    if (pdf <= 0.0f) return r;
    r.direction = wi;
    r.pdf = pdf;
    r.isDelta = isDelta;
    r.weight = vmax(f * absCosI / pdf, V3());
    return r;
}

}  // namespace

BsdfSample sampleBsdf(const PtrMaterial& m, V3 position, V3 normal, V3 wo, V3 incidentDir, bool frontFace, Rng& rng,
                      const ClampParams& cp) {  // E:1493-1918
    BsdfSample r;
    const uint32_t type = matType(m);
    if (type == PTR_MAT_PBR && cp.metalPbr) return samplePbrMetal(m, normal, wo, incidentDir, rng, cp);
    if (type == PTR_MAT_SUBSURFACE && cp.metalSss && sampleSeparableSss(m, position, normal, wo, rng, cp, r)) return r;

    if (type == PTR_MAT_LAMBERTIAN || type == PTR_MAT_SUBSURFACE) {
        float pdf = 0.0f;
        const V3 wi = sampleCosineHemisphere(rng, normal, pdf);
        const float cosI = dot(normal, wi);
        if (pdf <= 0.0f || cosI <= 0.0f) return r;
        const V3 f = baseColor(m) / kPi;
        const V3 weight = f * cosI / pdf;
        if (!finite3(weight)) return r;
        r.direction = wi;
        r.weight = vmax(weight, V3());
        r.pdf = pdf;
        return r;
    }

    if (type == PTR_MAT_PBR) {
        const V3 base = baseColor(m);
        const float metallic = clampf(m.pbrParams[0], 0.0f, 1.0f);
        const float roughness = clampf(m.baseColorRoughness[3], 0.0f, 1.0f);
        const float df0 = dielectricF0FromIor(m.typeEta[1]);
        const V3 f0 = base * metallic + splat(df0) * (1.0f - metallic);
        const V3 diffuseColor = base * (1.0f - metallic);
        const float specWeight = pbrSpecularWeight(f0);
        const float diffuseWeight = 1.0f - specWeight;
        V3 wi, f;
        float pdfSpec = 0.0f, pdfDiffuse = 0.0f;
        if (rng.nextFloat() < specWeight) {
            if (roughness <= 1.0e-3f) {
                wi = normalize(reflect(incidentDir, normal));
                if (dot(normal, wi) <= 0.0f) return r;
                f = schlickFresnel(f0, std::max(dot(normal, wo), 0.0f));
                pdfSpec = 1.0f;
                r.isDelta = true;
            } else {
                const float alpha = roughness * roughness;
                const V3 wh = sampleGgxHalfVector(rng, alpha, normal);
                if (dot(wh, normal) <= 0.0f) return r;
                wi = normalize(reflect(-wo, wh));
                const float cosI = std::max(dot(normal, wi), 0.0f);
                const float cosO = std::max(dot(normal, wo), 0.0f);
                if (cosI <= 0.0f || cosO <= 0.0f) return r;
                f = specTerm(schlickFresnel(f0, dot(wi, wh)), alpha, normal, wh, cosO, cosI);
                f = clampSpecularTail(f, roughness, f0, cp);
                pdfSpec = ggxPdf(alpha, normal, wo, wi);
            }
        } else {
            float pdf = 0.0f;
            wi = sampleCosineHemisphere(rng, normal, pdf);
            pdfDiffuse = pdf;
            if (pdfDiffuse <= 0.0f || std::max(dot(normal, wi), 0.0f) <= 0.0f) return r;
            f = diffuseColor / kPi;
        }
        const float cosI = std::max(dot(normal, wi), 0.0f);
        const float specPdfClamped = (pdfSpec > 0.0f) ? clampSpecularPdf(pdfSpec, cp) : 0.0f;
        const float pdf = specWeight * specPdfClamped + diffuseWeight * pdfDiffuse;
        if (pdf <= 0.0f || cosI <= 0.0f) return r;
        const V3 weight = f * cosI / pdf;
        if (!finite3(weight)) return r;
        r.direction = wi;
        r.weight = vmax(weight, V3());
        r.pdf = pdf;
        return r;
    }

    if (type == PTR_MAT_PLASTIC) {
        const float cr = coatRoughness(m);
        const float alpha = cr * cr;
        const float pCoat = coatSampleWeight(m);
        V3 wi;
        if (rng.nextFloat() < pCoat) {
            const V3 wh = sampleGgxHalfVector(rng, alpha, normal);
            if (dot(wh, normal) <= 0.0f) return r;
            wi = normalize(reflect(-wo, wh));
        } else {
            float pdf = 0.0f;
            wi = sampleCosineHemisphere(rng, normal, pdf);
        }
        const float cosI = std::max(dot(normal, wi), 0.0f);
        const float cosO = std::max(dot(normal, wo), 0.0f);
        if (cosI <= 0.0f || cosO <= 0.0f) return r;
        V3 specular, diffuse;
        float specularPdf;
        plasticLobes(m, normal, wo, wi, cosO, cosI, cp, specular, specularPdf, diffuse);
        const float pdfDiffuse = lambertPdf(normal, wi);
        const float pdf = pCoat * specularPdf + (1.0f - pCoat) * pdfDiffuse;
        if (pdf <= 0.0f) return r;
        const V3 weight = (specular + diffuse) * cosI / pdf;
        if (!finite3(weight)) return r;
        r.direction = wi;
        r.weight = vmax(weight, V3());
        r.pdf = pdf;
        return r;
    }

    if (type == PTR_MAT_METAL) {
        const float roughness = clampf(m.baseColorRoughness[3], 0.0f, 1.0f);
        const V3 f0 = conductorF0(m);
        if (roughness <= 1.0e-3f) {
            const V3 wi = normalize(reflect(incidentDir, normal));
            if (dot(normal, wi) <= 0.0f) return r;
            r.direction = wi;
            r.weight = metalFresnel(m, f0, std::max(dot(normal, wo), 0.0f));
            r.pdf = 1.0f;
            r.isDelta = true;
            return r;
        }
        const float alpha = roughness * roughness;
        // Metal-only: visible-normal sampling (M:5228), otherwise plain half-vector sampling
        const V3 wh = cp.metalSpecular ? sampleGgxVndf(rng, roughness, normal, wo) : sampleGgxHalfVector(rng, alpha, normal);
        if (dot(wh, normal) <= 0.0f) return r;
        const V3 wi = normalize(reflect(-wo, wh));
        const float cosI = dot(normal, wi);
        const float cosO = dot(normal, wo);
        if (cosI <= 0.0f || cosO <= 0.0f) return r;
        const float dotWoWh = dot(wo, wh);
        if (dotWoWh <= 0.0f) return r;
        const V3 F = metalFresnel(m, f0, dot(wi, wh));
        const float D = ggxDistribution(alpha, dot(normal, wh));
        const float G = ggxG1(alpha, cosO) * ggxG1(alpha, cosI);
        const float denom = 4.0f * cosO * cosI;
        V3 f = F * (D * G / std::max(denom, 1.0e-6f));
        if (cp.metalSpecular) f = f * specularEnergyCompensation(f0, roughness, cosO);   // M:5262
        f = clampSpecularTail(f, roughness, f0, cp);
        const float pdf = cp.metalSpecular ? ggxPdfVisible(alpha, normal, wo, wi)
                                           : D * std::max(dot(normal, wh), 0.0f) / std::max(4.0f * dotWoWh, 1.0e-6f);
        if (pdf <= 0.0f) return r;
        const float clampedPdf = clampSpecularPdf(pdf, cp);
        const V3 weight = f * cosI / clampedPdf;
        if (!finite3(weight)) return r;
        r.direction = wi;
        r.weight = vmax(weight, V3());
        r.pdf = clampedPdf;
        return r;
    }

    if (type == PTR_MAT_CARPAINT) {
        float pCoat, pFlake, pBase;
        carpaintLobeWeights(m, pCoat, pFlake, pBase);
        const float pick = rng.nextFloat();
        uint32_t lobe = 0u;  // 0 base, 1 flake, 2 coat
        if (pCoat > 0.0f && pick < pCoat) {
            lobe = 2u;
        } else if (pFlake > 0.0f && pick < pCoat + pFlake) {
            lobe = 1u;
        } else if (pBase <= 1.0e-6f) {
            if (pFlake > pCoat && pFlake > 0.0f) {
                lobe = 1u;
            } else if (pCoat > 0.0f) {
                lobe = 2u;
            }
        }
        V3 wi;
        if (lobe == 2u) {
            const V3 wh = sampleGgxVndf(rng, coatRoughness(m), normal, wo);
            if (dot(wh, normal) <= 0.0f) return r;
            wi = normalize(reflect(-wo, wh));
        } else if (lobe == 1u) {
            const float fr = std::max(cpFlakeRoughness(m), 1.0e-3f);
            const V3 fn = carpaintFlakeNormal(m, position, normal);
            const V3 wh = sampleGgxHalfVector(rng, fr * fr, fn);
            if (dot(wh, fn) <= 0.0f) return r;
            wi = normalize(reflect(-wo, wh));
        } else {
            const float metallic = cpBaseMetallic(m);
            const float diffuseWeight = std::max(1.0f - metallic, 0.0f);
            const float specWeight = std::max(metallic, 0.0f);
            const float weightSum = diffuseWeight + specWeight;
            const float choose = rng.nextFloat();
            const bool sampleSpec = (specWeight > 0.0f) && (weightSum > 0.0f) && (choose < specWeight / std::max(weightSum, 1.0e-6f));
            if (sampleSpec) {
                const float br = std::max(cpBaseRoughness(m), 1.0e-3f);
                const V3 wh = sampleGgxHalfVector(rng, br * br, normal);
                if (dot(wh, normal) <= 0.0f) return r;
                wi = normalize(reflect(-wo, wh));
            } else {
                float pdf = 0.0f;
                wi = sampleCosineHemisphere(rng, normal, pdf);
            }
        }
        if (!finite3(wi) || dot(normal, wi) <= 0.0f) return r;
        const LobeResult coat = carpaintEvalCoat(m, normal, wo, wi, cp);
        const LobeResult flake = carpaintEvalFlake(m, position, normal, wo, wi, cp);
        const LobeResult base = carpaintEvalBase(m, normal, wo, wi, cp);
        const float combinedPdf = (pBase * base.pdf + pFlake * flake.pdf) + pCoat * coat.pdf;
        if (combinedPdf <= 0.0f) return r;
        const LobeResult& sel = (lobe == 1u) ? flake : (lobe == 2u ? coat : base);
        if (sel.pdf <= 0.0f || !(sel.value.x > 0.0f || sel.value.y > 0.0f || sel.value.z > 0.0f)) return r;
        const float cosI = std::max(dot(normal, wi), 0.0f);
        if (cosI <= 0.0f) return r;
        const V3 weight = sel.value * cosI / combinedPdf;
        if (!finite3(weight)) return r;
        r.direction = wi;
        r.weight = vmax(weight, V3());
        r.pdf = combinedPdf;
        return r;
    }

    if (type == PTR_MAT_DIELECTRIC) {
        r.isDelta = true;
        const float refIdx = std::max(m.typeEta[1], 1.0f);
        float etaI = 1.0f, etaT = refIdx;
        const float cosO = clampf(dot(-incidentDir, normal), -1.0f, 1.0f);
        // thin-walled glass is a Metal-only notion (typeEta.w > 0.5): both faces see air -> glass, no medium event
        const bool thin = cp.thinDielectrics && m.typeEta[3] > 0.5f;
        if (!thin && !frontFace) {
            etaI = refIdx;
            etaT = 1.0f;
        }
        const float relativeEta = etaI / etaT;
        float cosT = 0.0f;
        const float Fr = fresnelDielectricExact(cosO, etaI, etaT, cosT);
        V3 direction, weight;
        V3 refracted;
        if (rng.nextFloat() < Fr) {
            direction = reflect(incidentDir, normal);
            weight = splat(Fr);
        } else if (!refract(incidentDir, normal, relativeEta, refracted) || dot(refracted, refracted) <= 0.0f) {
            direction = reflect(incidentDir, normal);
            weight = splat(Fr);
        } else {
            // note: weight is NOT divided by the selection probability (reference quirk, Appendix A row 8)
            direction = normalize(refracted);
            const float etaScale = (etaT * etaT) / (etaI * etaI);
            const float directionScale = etaScale * (std::fabs(cosT) / std::max(std::fabs(cosO), 1.0e-6f));
            weight = splat(std::max(1.0f - Fr, 0.0f) * directionScale);
            if (!thin) r.mediumEvent = frontFace ? 1 : -1;
        }
        r.direction = normalize(direction);
        r.weight = weight;
        r.pdf = 1.0f;
        return r;
    }

    float pdf = 0.0f;
    const V3 wi = sampleCosineHemisphere(rng, normal, pdf);
    if (pdf <= 0.0f) return r;
    const V3 weight = baseColor(m);
    if (!finite3(weight)) return r;
    r.direction = wi;
    r.weight = vmax(weight, V3());
    r.pdf = pdf;
    return r;
}

// ---------------------------------------------------------------------------------------------
// EnvImportanceSampler.mm
// ---------------------------------------------------------------------------------------------
namespace {

void buildAliasTable(const std::vector<float>& probabilities, std::vector<uint32_t>& alias, std::vector<float>& threshold) {  // :16-66
    const size_t count = probabilities.size();
    alias.assign(count, 0u);
    threshold.assign(count, 0.0f);
    if (count == 0) return;
    std::vector<float> scaled(count);
    std::vector<size_t> small, large;
    for (size_t i = 0; i < count; ++i) {
        scaled[i] = probabilities[i] * static_cast<float>(count);
        (scaled[i] < 1.0f ? small : large).push_back(i);
    }
    while (!small.empty() && !large.empty()) {
        const size_t s = small.back();
        small.pop_back();
        const size_t l = large.back();
        threshold[s] = clampf(scaled[s], 0.0f, 1.0f);
        alias[s] = static_cast<uint32_t>(l);
        scaled[l] = (scaled[l] + scaled[s]) - 1.0f;
        if (scaled[l] < 1.0f - 1e-7f) {
            large.pop_back();
            small.push_back(l);
        }
    }
    for (const std::vector<size_t>* rest : {&small, &large}) {
        for (size_t i : *rest) {
            threshold[i] = 1.0f;
            alias[i] = static_cast<uint32_t>(i);
        }
    }
}

}  // namespace

bool buildEnvDistribution(const float* rgba, uint32_t width, uint32_t height, EnvDistribution& out) {  // :70-171
    out = EnvDistribution{};
    if (!rgba || width == 0 || height == 0) return false;
    const size_t texels = static_cast<size_t>(width) * height;
    const float dTheta = kPi / static_cast<float>(height);
    const float dPhi = (2.0f * kPi) / static_cast<float>(width);
    std::vector<float> weights(texels, 0.0f), rowWeights(height, 0.0f);
    float totalWeight = 0.0f;
    for (uint32_t y = 0; y < height; ++y) {
        const float sinTheta = std::sin((static_cast<float>(y) + 0.5f) * dTheta);
        const float cell = std::max(sinTheta, 0.0f) * dTheta * dPhi;
        for (uint32_t x = 0; x < width; ++x) {
            const size_t i = static_cast<size_t>(y) * width + x;
            const float* px = rgba + i * 4u;
            const float lum = (0.2126f * px[0] + 0.7152f * px[1]) + 0.0722f * px[2];
            const float w = std::max(lum, 0.0f) * cell;
            weights[i] = w;
            rowWeights[y] += w;
            totalWeight += w;
        }
    }
    if (totalWeight <= 0.0f) return false;
    out.width = width;
    out.height = height;
    out.aliasCount = static_cast<uint32_t>(texels);
    out.totalWeight = totalWeight;
    std::vector<float> marginal(height);
    for (uint32_t y = 0; y < height; ++y) marginal[y] = rowWeights[y] > 0.0f ? (rowWeights[y] / totalWeight) : 0.0f;
    buildAliasTable(marginal, out.marginalAlias, out.marginalThreshold);
    out.conditionalAlias.assign(texels, 0u);
    out.conditionalThreshold.assign(texels, 0.0f);
    std::vector<float> cond(width);
    std::vector<uint32_t> aliasRow;
    std::vector<float> thresholdRow;
    for (uint32_t y = 0; y < height; ++y) {
        const size_t off = static_cast<size_t>(y) * width;
        if (rowWeights[y] > 0.0f) {
            const float inv = 1.0f / rowWeights[y];
            for (uint32_t x = 0; x < width; ++x) cond[x] = weights[off + x] * inv;
        } else {
            std::fill(cond.begin(), cond.end(), 1.0f / static_cast<float>(width));
        }
        buildAliasTable(cond, aliasRow, thresholdRow);
        for (uint32_t x = 0; x < width; ++x) {
            out.conditionalAlias[off + x] = aliasRow[x];
            out.conditionalThreshold[off + x] = thresholdRow[x];
        }
    }
    out.texelPdf.assign(texels, 0.0f);
    for (uint32_t y = 0; y < height; ++y) {
        const float sinTheta = std::sin((static_cast<float>(y) + 0.5f) * dTheta);
        const float cell = std::max(sinTheta, 0.0f) * dTheta * dPhi;
        for (uint32_t x = 0; x < width; ++x) {
            const size_t i = static_cast<size_t>(y) * width + x;
            const float probability = weights[i] / totalWeight;
            out.texelPdf[i] = (cell > 0.0f) ? (probability / cell) : 0.0f;
        }
    }
    return true;
}

EnvSample sampleEnvironmentCpu(const EnvDistribution& dist, float uMarginal, float uConditional, float uJitter,
                               float rotation, float intensity, const float* rgba) {  // :173-236
    EnvSample s;
    if (!rgba || dist.width == 0 || dist.height == 0 || dist.aliasCount == 0) return s;
    uMarginal = clampf(uMarginal, 0.0f, 0.99999994f);
    uConditional = clampf(uConditional, 0.0f, 0.99999994f);
    uJitter = clampf(uJitter, 0.0f, 0.99999994f);
    const float rowChoice = uMarginal * static_cast<float>(dist.height);
    uint32_t row = std::min(static_cast<uint32_t>(rowChoice), dist.height - 1u);
    const float rowFrac = rowChoice - static_cast<float>(row);
    if (rowFrac >= dist.marginalThreshold[row]) row = std::min(dist.marginalAlias[row], dist.height - 1u);
    const float colChoice = uConditional * static_cast<float>(dist.width);
    uint32_t col = std::min(static_cast<uint32_t>(colChoice), dist.width - 1u);
    const float colFrac = colChoice - static_cast<float>(col);
    const size_t rowOffset = static_cast<size_t>(row) * dist.width;
    if (colFrac >= dist.conditionalThreshold[rowOffset + col]) col = std::min(dist.conditionalAlias[rowOffset + col], dist.width - 1u);
    const float jitterX = uConditional - std::floor(uConditional);
    const float fx = (static_cast<float>(col) + jitterX) / static_cast<float>(dist.width);
    const float fy = (static_cast<float>(row) + uJitter) / static_cast<float>(dist.height);
    const float theta = fy * kPi;
    const float phi = fx * (2.0f * kPi);
    const float sinTheta = std::sin(theta), cosTheta = std::cos(theta);
    const V3 mapDir(sinTheta * std::cos(phi), cosTheta, sinTheta * std::sin(phi));
    const float cosRot = std::cos(rotation), sinRot = std::sin(rotation);
    s.direction = V3(mapDir.x * cosRot + mapDir.z * sinRot, mapDir.y, -mapDir.x * sinRot + mapDir.z * cosRot);
    const size_t texel = rowOffset + col;
    s.pdf = texel < dist.texelPdf.size() ? dist.texelPdf[texel] : 0.0f;
    s.radiance = V3(rgba + texel * 4u) * intensity;
    return s;
}

// ---------------------------------------------------------------------------------------------
// scene queries                                                                       E:2302-2433
// ---------------------------------------------------------------------------------------------
V3 nextRayOrigin(const HitInfo& hit, V3 direction) { return offsetRayOrigin(hit, direction); }

bool intersectScene(const Scene& scene, const Ray& ray, HitInfo& out, Counters* counters) {
    RayHit rh;
    if (!scene.intersect(ray.origin, ray.direction, kEpsilon, kInf, rh, false, counters)) return false;
    const Geom& g = scene.geoms[rh.geom];
    out.t = rh.t;
    out.position = ray.origin + out.t * ray.direction;
    if (dot(rh.ng, rh.ng) > 0.0f) out.normal = normalize(rh.ng);
    out.frontFace = dot(ray.direction, out.normal) < 0.0f;
    V3 adjusted = out.frontFace ? out.normal : -out.normal;
    V3 shading = adjusted;
    uint32_t material = g.materialIndex;
    out.primitiveType = g.type;
    out.geom = rh.geom;
    out.bu = rh.u;
    out.bv = rh.v;
    out.primitiveIndex = rh.primId;
    out.geomIndex = g.type == GeomType::Mesh ? g.meshIndex : 0u;
    out.twoSided = false;
    if (g.type == GeomType::Mesh && !g.indices.empty() && !g.normals.empty()) {
        const uint32_t base = rh.primId * 3u;
        if (base + 2u < g.indices.size()) {
            const float u = rh.u, v = rh.v, w = 1.0f - u - v;
            const V3 interp = (w * g.normals[g.indices[base]] + u * g.normals[g.indices[base + 1u]]) + v * g.normals[g.indices[base + 2u]];
            if (dot(interp, interp) > 0.0f) {
                shading = normalize(interp);
                if (dot(shading, adjusted) < 0.0f) shading = -shading;
            }
        }
    } else if (g.type == GeomType::Rectangles && !g.normals.empty()) {
        const uint32_t index = rh.primId * 3u;
        if (index < g.indices.size()) {
            shading = g.normals[g.indices[index]];
            if (dot(shading, adjusted) < 0.0f) shading = -shading;
        }
        uint32_t rectIndex = rh.primId;
        if (rh.primId < g.triToRect.size()) rectIndex = g.triToRect[rh.primId];
        out.primitiveIndex = rectIndex;
        if (scene.rects && rectIndex < scene.rectCount) out.twoSided = scene.rects[rectIndex].materialTwoSided[1] != 0u;
        if (!g.primMaterial.empty()) material = g.primMaterial[rh.primId];
    } else if (g.type == GeomType::Spheres) {
        if (rh.primId < scene.spheres.size()) {
            const V3 center(scene.spheres[rh.primId].centerRadius);
            const V3 normal = normalize(out.position - center);
            shading = normal;
            out.normal = normal;
            out.frontFace = dot(ray.direction, normal) < 0.0f;
            out.twoSided = true;
            if (!g.primMaterial.empty()) material = g.primMaterial[rh.primId];
        }
    }
    out.shadingNormal = shading;
    out.materialIndex = material;
    return true;
}

// ---------------------------------------------------------------------------------------------
// material textures of the Metal metallic-roughness model                  M:108-198, 583-940, 2923-3216, 5919-6400
// ---------------------------------------------------------------------------------------------
// The reference samples through the GPU's texture units; a software path has to fix a filtering rule.  This is the rule of
// csrc/kernels/texture.h, restated: linear RGBA floats with a full mip chain (2x2 box, ((a+b)+(c+d))*0.25, second tap clamped at
// odd edges), ray-cone level of detail, bilinear inside a level (texel centres at (i+0.5)/W, glTF wrap modes), linear between the
// two nearest levels, NEAREST samplers = one texel of the rounded level.  No first-hit ray differentials.
struct C4 {
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
};

struct OracleTexture {
    uint32_t width = 0, height = 0, levels = 0, wrapS = 0, wrapT = 0, filter = 1;
    std::vector<std::vector<float>> mips;   // RGBA per level
};

static std::vector<OracleTexture> buildTextures(const PtrSceneDesc& desc) {
    std::vector<OracleTexture> out;
    if (!desc.textures) return out;
    for (uint32_t i = 0; i < desc.textureCount; ++i) {
        const PtrTexture& t = desc.textures[i];
        OracleTexture o;
        o.width = t.width;
        o.height = t.height;
        o.wrapS = t.wrapS & 3u;
        o.wrapT = t.wrapT & 3u;
        o.filter = t.filter ? 1u : 0u;
        o.mips.emplace_back(t.rgba, t.rgba + static_cast<size_t>(t.width) * t.height * 4u);
        uint32_t w = t.width, h = t.height;
        while ((w > 1u || h > 1u) && o.mips.size() < 16u) {
            const uint32_t nw = std::max(w / 2u, 1u), nh = std::max(h / 2u, 1u);
            const std::vector<float>& src = o.mips.back();
            std::vector<float> dst(static_cast<size_t>(nw) * nh * 4u);
            for (uint32_t y = 0; y < nh; ++y) {
                const uint32_t y0 = std::min(2u * y, h - 1u), y1 = std::min(2u * y + 1u, h - 1u);
                for (uint32_t x = 0; x < nw; ++x) {
                    const uint32_t x0 = std::min(2u * x, w - 1u), x1 = std::min(2u * x + 1u, w - 1u);
                    for (int c = 0; c < 4; ++c) {
                        const float a = src[(static_cast<size_t>(y0) * w + x0) * 4u + c], b = src[(static_cast<size_t>(y0) * w + x1) * 4u + c];
                        const float cc = src[(static_cast<size_t>(y1) * w + x0) * 4u + c], d = src[(static_cast<size_t>(y1) * w + x1) * 4u + c];
                        dst[(static_cast<size_t>(y) * nw + x) * 4u + c] = ((a + b) + (cc + d)) * 0.25f;
                    }
                }
            }
            o.mips.push_back(std::move(dst));
            w = nw;
            h = nh;
        }
        o.levels = static_cast<uint32_t>(o.mips.size());
        out.push_back(std::move(o));
    }
    return out;
}

static int texWrap(int i, int n, uint32_t mode) {
    if (mode == 1u) return std::min(std::max(i, 0), n - 1);
    if (mode == 2u) {
        const int period = 2 * n;
        int j = i % period;
        if (j < 0) j += period;
        return j < n ? j : period - 1 - j;
    }
    int j = i % n;
    return j < 0 ? j + n : j;
}

static C4 texFetch(const OracleTexture& t, uint32_t level, int x, int y) {
    const uint32_t w = std::max(t.width >> level, 1u);
    const float* p = &t.mips[level][(static_cast<size_t>(y) * w + static_cast<size_t>(x)) * 4u];
    return C4{p[0], p[1], p[2], p[3]};
}

static C4 texBilinear(const OracleTexture& t, uint32_t level, float u, float v) {
    const int W = static_cast<int>(std::max(t.width >> level, 1u)), H = static_cast<int>(std::max(t.height >> level, 1u));
    if (t.filter == 0u) {
        return texFetch(t, level, texWrap(static_cast<int>(std::floor(u * static_cast<float>(W))), W, t.wrapS),
                        texWrap(static_cast<int>(std::floor(v * static_cast<float>(H))), H, t.wrapT));
    }
    const float fx = u * static_cast<float>(W) - 0.5f, fy = v * static_cast<float>(H) - 0.5f;
    const float x0f = std::floor(fx), y0f = std::floor(fy);
    const float tx = fx - x0f, ty = fy - y0f;
    const int x0 = texWrap(static_cast<int>(x0f), W, t.wrapS), x1 = texWrap(static_cast<int>(x0f) + 1, W, t.wrapS);
    const int y0 = texWrap(static_cast<int>(y0f), H, t.wrapT), y1 = texWrap(static_cast<int>(y0f) + 1, H, t.wrapT);
    const C4 c00 = texFetch(t, level, x0, y0), c10 = texFetch(t, level, x1, y0), c01 = texFetch(t, level, x0, y1), c11 = texFetch(t, level, x1, y1);
    const float ix = 1.0f - tx, iy = 1.0f - ty;
    return C4{(c00.x * ix + c10.x * tx) * iy + (c01.x * ix + c11.x * tx) * ty, (c00.y * ix + c10.y * tx) * iy + (c01.y * ix + c11.y * tx) * ty,
              (c00.z * ix + c10.z * tx) * iy + (c01.z * ix + c11.z * tx) * ty, (c00.w * ix + c10.w * tx) * iy + (c01.w * ix + c11.w * tx) * ty};
}

static C4 texSample(const std::vector<OracleTexture>& textures, uint32_t index, float u, float v, float lod, C4 fallback) {
    if (index == 0xFFFFFFFFu || index >= textures.size()) return fallback;
    const OracleTexture& t = textures[index];
    const float maxMip = static_cast<float>(t.levels - 1u);
    const float l = std::min(std::max(lod, 0.0f), maxMip);
    if (t.filter == 0u) return texBilinear(t, static_cast<uint32_t>(std::floor(l + 0.5f)), u, v);
    const float l0f = std::floor(l);
    const uint32_t l0 = static_cast<uint32_t>(l0f), l1 = std::min(l0 + 1u, t.levels - 1u);
    const float f = l - l0f;
    const C4 a = texBilinear(t, l0, u, v);
    if (!(f > 0.0f) || l1 == l0) return a;
    const C4 b = texBilinear(t, l1, u, v);
    return C4{a.x + (b.x - a.x) * f, a.y + (b.y - a.y) * f, a.z + (b.z - a.z) * f, a.w + (b.w - a.w) * f};
}

void sampleTextures(const PtrSceneDesc& desc, uint32_t texture, const float* in, uint64_t n, float* out) {
    const std::vector<OracleTexture> textures = buildTextures(desc);
    for (uint64_t i = 0; i < n; ++i) {
        const C4 c = texSample(textures, texture, in[i * 3], in[i * 3 + 1], in[i * 3 + 2], C4{-1.0f, -1.0f, -1.0f, -1.0f});
        out[i * 4] = c.x;
        out[i * 4 + 1] = c.y;
        out[i * 4 + 2] = c.z;
        out[i * 4 + 3] = c.w;
    }
}

static float texLod(const std::vector<OracleTexture>& textures, uint32_t index, float uvPerWorld, float footprintWorld) {   // M:162-176
    if (index == 0xFFFFFFFFu || index >= textures.size()) return 0.0f;
    const OracleTexture& t = textures[index];
    if (t.width == 0u || t.height == 0u) return 0.0f;
    if (t.levels <= 1u || uvPerWorld <= 0.0f || footprintWorld <= 0.0f) return 0.0f;
    const float maxRes = std::max(static_cast<float>(t.width), static_cast<float>(t.height));
    const float texelFootprint = footprintWorld * uvPerWorld * maxRes;
    const float lod = std::log2(std::max(texelFootprint, 1.0e-7f));
    return std::min(std::max(lod, 0.0f), static_cast<float>(t.levels - 1u));
}

static V3 decodeNormalMap(V3 s, float normalScale, float& outLength) {   // M:108-127
    V3 n = s * 2.0f - splat(1.0f);
    n.x *= normalScale;
    n.y *= normalScale;
    outLength = length(n);
    const float xyLen2 = n.x * n.x + n.y * n.y;
    n.z = std::sqrt(std::max(1.0f - xyLen2, 0.0f));
    const float len2 = dot(n, n);
    if (len2 > 1.0e-12f) {
        n = n * (1.0f / std::sqrt(len2));
    } else {
        n = V3(0.0f, 0.0f, 1.0f);
    }
    return n;
}

struct TexSlot {
    float u = 0.0f, v = 0.0f, uvPerWorld = 0.0f;
};

static TexSlot texSlot(const PtrMaterial& m, uint32_t slot, uint32_t uvSet, const float uv0[2], const float uv1[2], float perWorld0, float perWorld1) {
    V3 row0(m.textureTransform[2 * slot]), row1(m.textureTransform[2 * slot + 1]);
    const float linearSum = (std::fabs(row0.x) + std::fabs(row0.y)) + (std::fabs(row1.x) + std::fabs(row1.y));
    if (!finite3(row0) || !finite3(row1) || !(linearSum > 1.0e-8f)) {   // M:2942-2983
        row0 = V3(1.0f, 0.0f, 0.0f);
        row1 = V3(0.0f, 1.0f, 0.0f);
    }
    const float* uv = uvSet == 0u ? uv0 : uv1;
    TexSlot t;
    t.u = (row0.x * uv[0] + row0.y * uv[1]) + row0.z;
    t.v = (row1.x * uv[0] + row1.y * uv[1]) + row1.z;
    const float sx = std::sqrt(row0.x * row0.x + row1.x * row1.x), sy = std::sqrt(row0.y * row0.y + row1.y * row1.y);
    t.uvPerWorld = (uvSet == 0u ? perWorld0 : perWorld1) * std::max(std::max(sx, sy), 1.0e-6f);   // M:3002-3009
    return t;
}

// uv-per-world of one set over a triangle (triangle_surface_partials, M:741-820) in world space
static float triangleUvPerWorld(V3 edge1, V3 edge2, float du1, float dv1, float du2, float dv2) {
    const float det = du1 * dv2 - dv1 * du2;
    if (std::fabs(det) > 1.0e-9f) {
        const float inv = 1.0f / det;
        const V3 dPdu = (edge1 * dv2 - edge2 * dv1) * inv, dPdv = (edge2 * du1 - edge1 * du2) * inv;
        const float lenU = length(dPdu), lenV = length(dPdv);
        if (lenU > 1.0e-8f && lenV > 1.0e-8f) {
            const float perWorld = std::max(1.0f / lenU, 1.0f / lenV);
            if (std::isfinite(perWorld) && perWorld > 0.0f) return perWorld;
        }
    }
    const float worldArea = length(cross(edge1, edge2)), uvArea = std::fabs(det);
    const float perWorld = (worldArea > 1.0e-12f && uvArea > 1.0e-12f) ? std::sqrt(uvArea / worldArea) : 0.0f;
    return std::isfinite(perWorld) ? perWorld : 0.0f;
}

// The per-hit material of the textured model (M:5919-6400).  `material` is the hit's own copy and is modified in place;
// returns true when the alpha test discards the hit.
static bool applyPbrTextures(const Scene& scene, const std::vector<OracleTexture>& textures, const HitInfo& hit, PtrMaterial& material, V3 wo,
                             float coneWidth, float coneSpread, float hitDistance, Rng& rng, V3& shadingNormalOut, bool& twoSidedOut) {
    const Geom& g = scene.geoms[hit.geom];
    const uint32_t base = hit.primitiveIndex * 3u;
    const uint32_t i0 = g.indices[base], i1 = g.indices[base + 1u], i2 = g.indices[base + 2u];
    // saturated barycentric weights (M:583-591)
    V3 w = vmax(V3(1.0f - hit.bu - hit.bv, hit.bu, hit.bv), V3());
    const float wsum = (w.x + w.y) + w.z;
    w = (wsum > 1.0e-8f) ? w / wsum : V3(1.0f, 0.0f, 0.0f);
    auto uvOf = [&](const std::vector<float>& set, uint32_t v, int c) { return set.empty() ? 0.0f : set[2 * v + c]; };
    const float a0[2] = {uvOf(g.uv0, i0, 0), uvOf(g.uv0, i0, 1)}, b0[2] = {uvOf(g.uv0, i1, 0), uvOf(g.uv0, i1, 1)}, c0[2] = {uvOf(g.uv0, i2, 0), uvOf(g.uv0, i2, 1)};
    const float a1[2] = {uvOf(g.uv1, i0, 0), uvOf(g.uv1, i0, 1)}, b1[2] = {uvOf(g.uv1, i1, 0), uvOf(g.uv1, i1, 1)}, c1[2] = {uvOf(g.uv1, i2, 0), uvOf(g.uv1, i2, 1)};
    const float uv0[2] = {(a0[0] * w.x + b0[0] * w.y) + c0[0] * w.z, (a0[1] * w.x + b0[1] * w.y) + c0[1] * w.z};
    const float uv1[2] = {(a1[0] * w.x + b1[0] * w.y) + c1[0] * w.z, (a1[1] * w.x + b1[1] * w.y) + c1[1] * w.z};
    const V3 edge1 = g.positions[i1] - g.positions[i0], edge2 = g.positions[i2] - g.positions[i0];
    const float perWorld0 = triangleUvPerWorld(edge1, edge2, b0[0] - a0[0], b0[1] - a0[1], c0[0] - a0[0], c0[1] - a0[1]);
    const float perWorld1 = triangleUvPerWorld(edge1, edge2, b1[0] - a1[0], b1[1] - a1[1], c1[0] - a1[0], c1[1] - a1[1]);
    const float coneFootprint = std::max(coneWidth + coneSpread * std::max(hitDistance, 0.0f), 1.0e-7f);   // M:158-160
    const float surfaceFootprint = coneFootprint / std::max(std::fabs(dot(normalize(hit.normal), normalize(wo))), 1.0e-3f);   // M:178-185
    const uint32_t uvSetOf[6] = {std::min(material.textureUvSet0[0], 1u), std::min(material.textureUvSet0[1], 1u), std::min(material.textureUvSet0[2], 1u),
                                 std::min(material.textureUvSet0[3], 1u), std::min(material.textureUvSet1[0], 1u), std::min(material.textureUvSet1[1], 1u)};
    auto slot = [&](uint32_t k) { return texSlot(material, k, uvSetOf[k], uv0, uv1, perWorld0, perWorld1); };
    auto lodOf = [&](uint32_t tex, const TexSlot& t) { return texLod(textures, tex, t.uvPerWorld, surfaceFootprint); };
    auto valid = [&](uint32_t tex) { return tex != 0xFFFFFFFFu && tex < textures.size(); };
    const C4 one{1.0f, 1.0f, 1.0f, 1.0f};
    const uint32_t texBase = material.textureIndices0[0], texOrm = material.textureIndices0[1], texNormal = material.textureIndices0[2],
                   texOcc = material.textureIndices0[3], texEmissive = material.textureIndices1[0], texTrans = material.textureIndices1[1];

    const TexSlot sBase = slot(0u);
    const C4 baseSample = texSample(textures, texBase, sBase.u, sBase.v, lodOf(texBase, sBase), one);
    const V3 baseColor = V3(material.baseColorRoughness) * V3(baseSample.x, baseSample.y, baseSample.z);
    float metallic = clampf(material.pbrParams[0], 0.0f, 1.0f), roughness = clampf(material.pbrParams[1], 0.0f, 1.0f);
    const bool disableOrm = (material.materialFlags & 1u) != 0u;
    if (!disableOrm && valid(texOrm)) {
        const TexSlot sOrm = slot(1u);
        const C4 mr = texSample(textures, texOrm, sOrm.u, sOrm.v, lodOf(texOrm, sOrm), one);
        metallic = clampf(mr.z * metallic, 0.0f, 1.0f);
        roughness = clampf(mr.y * roughness, 0.0f, 1.0f);
    }
    float transmission = clampf(material.pbrExtras[2], 0.0f, 1.0f);
    if (valid(texTrans)) {
        const TexSlot sT = slot(5u);
        transmission = clampf(transmission * texSample(textures, texTrans, sT.u, sT.v, lodOf(texTrans, sT), one).x, 0.0f, 1.0f);
    }
    transmission *= (1.0f - metallic);
    const float alpha = clampf(clampf(material.pbrExtras[0], 0.0f, 1.0f) * baseSample.w, 0.0f, 1.0f);   // M:6196-6217
    if (material.pbrExtras[3] > 0.5f) {
        const bool discard = material.pbrExtras[3] < 1.5f ? (alpha < clampf(material.pbrExtras[1], 0.0f, 1.0f)) : (rng.nextFloat() > alpha);
        if (discard) return true;
    }
    float occlusion = 1.0f;
    if (!disableOrm && valid(texOcc)) {
        const TexSlot sO = slot(3u);
        const float occ = texSample(textures, texOcc, sO.u, sO.v, lodOf(texOcc, sO), one).x;
        const float strength = clampf(material.pbrParams[2], 0.0f, 1.0f);
        occlusion = 1.0f + (occ - 1.0f) * strength;
    }
    V3 emissive(material.emission);
    if (valid(texEmissive)) {
        const TexSlot sE = slot(4u);
        const C4 e = texSample(textures, texEmissive, sE.u, sE.v, lodOf(texEmissive, sE), one);
        emissive *= V3(e.x, e.y, e.z);
    }
    V3 shadingNormal = hit.shadingNormal;
    if (dot(shadingNormal, shadingNormal) <= 0.0f) shadingNormal = hit.normal;
    shadingNormal = normalize(shadingNormal);
    const float normalScale = material.pbrParams[3];
    if (valid(texNormal) && normalScale > 1.0e-4f) {   // M:6281-6346
        const TexSlot sN = slot(2u);
        const C4 ns = texSample(textures, texNormal, sN.u, sN.v, lodOf(texNormal, sN), C4{0.5f, 0.5f, 1.0f, 1.0f});
        float normalLength = 1.0f;
        const V3 nts = decodeNormalMap(V3(ns.x, ns.y, ns.z), normalScale, normalLength);
        V3 t(1.0f, 0.0f, 0.0f), bt;
        bool hasBasis = false;
        if (!g.tangents.empty()) {
            const float* t0 = &g.tangents[4 * i0];
            const float* t1 = &g.tangents[4 * i1];
            const float* t2 = &g.tangents[4 * i2];
            V3 tw = (V3(t0) * w.x + V3(t1) * w.y) + V3(t2) * w.z;
            const float tsign = (t0[3] * w.x + t1[3] * w.y) + t2[3] * w.z;
            const float len2 = dot(tw, tw);
            tw = (finite3(tw) && len2 > 1.0e-12f) ? tw * (1.0f / std::sqrt(len2)) : V3(1.0f, 0.0f, 0.0f);
            if (std::fabs(tsign) > 0.5f) {
                t = tw - shadingNormal * dot(shadingNormal, tw);
                if (finite3(t) && dot(t, t) > 1.0e-6f) {
                    t = normalize(t);
                    bt = normalize(cross(shadingNormal, t)) * (tsign < 0.0f ? -1.0f : 1.0f);
                    hasBasis = finite3(bt) && dot(bt, bt) > 1.0e-6f;
                }
            }
        }
        if (!hasBasis) {   // M:843-911, in world space
            const bool set1 = uvSetOf[2] != 0u;
            const float* q0 = set1 ? a1 : a0;
            const float* q1 = set1 ? b1 : b0;
            const float* q2 = set1 ? c1 : c0;
            const float du1 = q1[0] - q0[0], dv1 = q1[1] - q0[1], du2 = q2[0] - q0[0], dv2 = q2[1] - q0[1];
            const float denom = du1 * dv2 - dv1 * du2;
            if (std::fabs(denom) >= 1.0e-8f) {
                const float r = 1.0f / denom;
                // the device reads the edges from its triangle record (v0 - v1 negated, v2 - v0): the same differences
                const Prim tri = [&] {
                    Prim p{};
                    p.e1 = g.positions[i0] - g.positions[i1];
                    p.e2 = g.positions[i2] - g.positions[i0];
                    return p;
                }();
                const V3 e1 = -tri.e1, e2 = tri.e2;
                V3 tangentW = (e1 * dv2 - e2 * dv1) * r;
                V3 bitangentW = (e2 * du1 - e1 * du2) * r;
                const float tl = dot(tangentW, tangentW), bl = dot(bitangentW, bitangentW);
                if (finite3(tangentW) && tl > 1.0e-12f && finite3(bitangentW) && bl > 1.0e-12f) {
                    tangentW = tangentW * (1.0f / std::sqrt(tl));
                    bitangentW = bitangentW * (1.0f / std::sqrt(bl));
                    t = tangentW - shadingNormal * dot(shadingNormal, tangentW);
                    if (finite3(t) && dot(t, t) > 1.0e-6f) {
                        t = normalize(t);
                        const float handed = dot(cross(shadingNormal, t), bitangentW) < 0.0f ? -1.0f : 1.0f;
                        bt = normalize(cross(shadingNormal, t)) * (handed * (g.detSign < 0.0f ? -1.0f : 1.0f));
                        hasBasis = true;
                    }
                }
            }
        }
        if (!hasBasis) {   // M:934-940
            const V3 up = std::fabs(shadingNormal.z) < 0.999f ? V3(0.0f, 0.0f, 1.0f) : V3(1.0f, 0.0f, 0.0f);
            t = normalize(cross(up, shadingNormal));
            bt = cross(shadingNormal, t);
        }
        V3 mapped = normalize((t * nts.x + bt * nts.y) + shadingNormal * nts.z);
        if (dot(mapped, hit.normal) < 0.0f) mapped = -mapped;
        shadingNormal = mapped;
        const float tok = std::max((1.0f - normalLength) / std::max(normalLength, 1.0e-6f), 0.0f);   // M:6348-6388 without the gradient term
        roughness = clampf(std::sqrt(roughness * roughness + tok), 0.0f, 1.0f);
    }
    material.baseColorRoughness[0] = baseColor.x;
    material.baseColorRoughness[1] = baseColor.y;
    material.baseColorRoughness[2] = baseColor.z;
    material.baseColorRoughness[3] = roughness;
    material.pbrParams[0] = metallic;
    material.pbrExtras[2] = transmission;
    material.emission[0] = emissive.x;
    material.emission[1] = emissive.y;
    material.emission[2] = emissive.z;
    material.emission[3] = 0.0f;
    std::memcpy(&material.materialPad[0], &occlusion, 4);
    material.materialPad[1] = 0x4F43434Cu;
    shadingNormalOut = shadingNormal;
    twoSidedOut = hit.twoSided || material.typeEta[2] > 0.5f;
    return false;
}

// ---------------------------------------------------------------------------------------------
// render loop                                                                         E:2443-3214
// ---------------------------------------------------------------------------------------------
// Path signature of a sample (the HIP counting build computes the same word, csrc/kernels/device_types.h kSig*):
// bits 0..15: bit d set when the rectangle-light sample taken at path vertex d contributed; bits 16..31: hash chain over the
// primitives hit, vertex by vertex (a miss included).
static uint32_t sigHashStep(uint32_t h, uint32_t primType, uint32_t geomIndex, uint32_t primIndex) {
    uint32_t x = (h * 0x9e3779b1u) ^ (primType * 0x85ebca6bu) ^ (geomIndex * 0xc2b2ae35u) ^ primIndex;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    return x & 0xFFFFu;
}

void render(const Scene& scene, const PtrSceneDesc& desc, const PtrSettings& settings, uint32_t spp, uint32_t threads,
            uint32_t yBegin, uint32_t yEnd, float* outRgb, RenderCounters* outCounters, uint32_t* outSignature, uint8_t* outMarginal) {
    const uint32_t width = settings.width, height = settings.height;
    const PtrMaterial* materials = desc.materials;
    const uint32_t materialCount = desc.materialCount;
    const PtrRect* rectangles = desc.rects;
    const uint32_t rectangleCount = desc.rectCount;
    const float emissionScale = (settings.emissionScale > 0.0f && std::isfinite(settings.emissionScale)) ? settings.emissionScale : 1.0f;

    EnvMap envMap;
    EnvMap* env = nullptr;
    if (desc.envRgba && desc.envWidth > 0 && desc.envHeight > 0) {
        envMap.rgba = desc.envRgba;
        envMap.width = desc.envWidth;
        envMap.height = desc.envHeight;
        envMap.hasDistribution = buildEnvDistribution(desc.envRgba, desc.envWidth, desc.envHeight, envMap.dist);
        env = &envMap;
    }

    // rectangle lights: DiffuseLight rectangles with non-zero emission (E:2484-2522)
    std::vector<RectLight> rectLights;
    std::vector<int32_t> lightIndexByRect;
    if (rectangles && rectangleCount > 0 && materials && materialCount > 0) {
        lightIndexByRect.assign(rectangleCount, -1);
        for (uint32_t i = 0; i < rectangleCount; ++i) {
            const PtrRect& rect = rectangles[i];
            const PtrMaterial& mat = materials[std::min(rect.materialTwoSided[0], materialCount - 1u)];
            if (matType(mat) != PTR_MAT_DIFFUSE_LIGHT) continue;
            const V3 baseEmission = V3(mat.emission) * emissionScale;
            if (!(dot(baseEmission, baseEmission) > 0.0f)) continue;
            RectLight li;
            li.rectIndex = i;
            li.corner = V3(rect.corner);
            li.edgeU = V3(rect.edgeU);
            li.edgeV = V3(rect.edgeV);
            li.normal = normalize(V3(rect.normalAndPlane));
            li.twoSided = rect.materialTwoSided[1] != 0u;
            li.baseEmission = baseEmission;
            li.emissionUsesEnv = false;  // emitEnv is disabled for NEE on this path (E:2515-2516)
            li.area = length(cross(li.edgeU, li.edgeV));
            lightIndexByRect[i] = static_cast<int32_t>(rectLights.size());
            rectLights.push_back(li);
        }
    }
    const uint32_t rectLightCount = static_cast<uint32_t>(rectLights.size());
    const bool envMapAvailable = env && env->width > 0 && env->height > 0 && env->rgba;
    const bool envSampling = env && env->hasDistribution && envMapAvailable;
    const ClampParams cp = makeClampParams(settings);
    const Camera camera = buildCamera(settings);
    const uint32_t targetSamples = std::max<uint32_t>(1u, spp);
    const uint32_t seedBase = settings.seed != 0 ? settings.seed : 0x9e3779b9u;

    std::atomic<uint64_t> cExtend{0}, cShadow{0}, cNodes{0}, cPrims{0}, cShaded{0}, cTri{0};
    const bool counting = outCounters != nullptr;

    // test knob (PtrSettings.debugShadowSlack): 0 = the reference's shadow-ray length, quirk Q9
    const float shadowSlack = (settings.debugShadowSlack > 0.0f && settings.debugShadowSlack < 1.0f) ? settings.debugShadowSlack : 0.0f;
    const bool wantSignature = outSignature != nullptr || outMarginal != nullptr;
    // material textures: sampled by the Metal metallic-roughness model only (the Embree backend never reads them)
    const std::vector<OracleTexture> textures = cp.metalPbr ? buildTextures(desc) : std::vector<OracleTexture>();
    const bool texturedScene = !textures.empty();
    // make_primary_ray_cone (M:141-152)
    float primaryConeWidth = 0.0f, primaryConeSpread = 0.0f;
    {
        const float pixelX = length(camera.horizontal) / std::max(static_cast<float>(width), 1.0f);
        const float pixelY = length(camera.vertical) / std::max(static_cast<float>(height), 1.0f);
        const float pixelFootprint = std::max(std::max(pixelX, pixelY), 1.0e-6f);
        const V3 centre = (camera.lowerLeft + 0.5f * camera.horizontal) + 0.5f * camera.vertical;
        const float focus = length(centre - camera.origin);
        primaryConeWidth = std::max(2.0f * camera.lensRadius, 0.0f);
        primaryConeSpread = pixelFootprint / std::max(focus, 1.0e-6f);
    }

    auto renderPixel = [&](uint32_t x, uint32_t y, RenderCounters& rc) -> V3 {
        V3 pixelRadiance;
        const uint32_t pixelIndex = y * width + x;
        uint32_t sig = 0u;        // signature of the pixel's last sample
        bool marginal = false;    // some rectangle-light shadow test of that sample flips within +-2e-6 (relative) of its ray length
        Counters tc;
        Counters* tcp = counting ? &tc : nullptr;
        auto trace = [&](const Ray& r, HitInfo& h) {
            if (counting) ++rc.extendRays;
            return intersectScene(scene, r, h, tcp);
        };
        auto occluded = [&](V3 o, V3 d, float tMax) {
            if (counting) ++rc.shadowRays;
            return scene.occluded(o, d, kEpsilon, tMax, false, tcp);
        };

        for (uint32_t s = 0; s < targetSamples; ++s) {
            Rng rng;
            rng.state = Rng::hash(seedBase ^ pixelIndex ^ (s * 0x9e3779b9u));
            Ray ray = generateCameraRay(camera, width, height, x, y, rng);
            V3 throughput(1.0f, 1.0f, 1.0f), radiance;
            sig = 0u;
            marginal = false;
            float lastBsdfPdf = 1.0f;
            bool lastScatterWasDelta = true;
            uint32_t specularDepth = 0;
            // Metal-only media semantics (settings.metalSemantics & PTR_METAL_MEDIA): stack of the dielectrics the path
            // is inside of, shaders/pathtrace.metal:5768-5773
            constexpr uint32_t kMaxMediumStack = 8;
            uint32_t mediumStack[kMaxMediumStack] = {0};
            uint32_t mediumDepth = 0;
            const bool media = (settings.metalSemantics & PTR_METAL_MEDIA) != 0u;
            float coneWidth = primaryConeWidth, coneSpread = primaryConeSpread;   // ray cone of the path (textured scenes)

            for (uint32_t depth = 0; depth < settings.maxDepth; ++depth) {
                HitInfo hit;
                if (!trace(ray, hit)) {
                    if (wantSignature) sig = (sig & 0xFFFFu) | (sigHashStep(sig >> 16, 7u, 0u, 0u) << 16);
                    const V3 background = evaluateBackground(settings, env, ray.direction);
                    float misWeight = 1.0f;
                    const bool useSpecularMis = (!lastScatterWasDelta) || settings.enableSpecularNee || settings.enableMnee;
                    if (useSpecularMis && envSampling) {
                        const float lightPdf = environmentPdf(*env, settings.environmentRotation, ray.direction);
                        const float denom = lastBsdfPdf + lightPdf;
                        if (denom > 0.0f) misWeight = lastBsdfPdf / denom;
                        misWeight = clampf(misWeight, kMisWeightClampMin, kMisWeightClampMax);
                    }
                    radiance += clampFireflyContribution(throughput, background * misWeight, cp);
                    break;
                }
                if (!materials || materialCount == 0) break;
                if (counting) {
                    ++rc.shadedHits;
                    if (hit.primitiveType == GeomType::Mesh) ++rc.triangleHits;
                }
                if (wantSignature) {
                    const uint32_t ptype = hit.primitiveType == GeomType::Mesh ? 0u : (hit.primitiveType == GeomType::Spheres ? 1u : 2u);
                    sig = (sig & 0xFFFFu) | (sigHashStep(sig >> 16, ptype, ptype == 0u ? hit.geomIndex : 0u, hit.primitiveIndex) << 16);
                }
                if (media && mediumDepth > 0) {  // Beer-Lambert over the segment inside the innermost medium, M:5869-5876
                    const PtrMaterial& inside = materials[std::min(mediumStack[mediumDepth - 1], materialCount - 1)];
                    const V3 sigma = vmax(V3(inside.dielectricSigmaA), V3());
                    if (sigma.x > 0.0f || sigma.y > 0.0f || sigma.z > 0.0f) {
                        const float segment = std::max(hit.t, 0.0f);
                        throughput *= V3(std::exp(-sigma.x * segment), std::exp(-sigma.y * segment), std::exp(-sigma.z * segment));
                    }
                }

                PtrMaterial texturedMaterial;   // the hit's own copy when textures modulate it (M:6391-6394)
                const PtrMaterial* materialPtr = &materials[std::min(hit.materialIndex, materialCount - 1)];
                const uint32_t type = matType(*materialPtr);
                const V3 incidentDir = normalize(ray.direction);
                const V3 wo = -incidentDir;
                V3 shadingNormal = hit.shadingNormal;
                if (dot(shadingNormal, shadingNormal) <= 0.0f) shadingNormal = hit.normal;
                if (type == PTR_MAT_DIELECTRIC) {
                    shadingNormal = hit.normal;  // dielectrics shade with the geometric normal, as stored (E:2641-2644)
                    // Metal-only: the normal faces the incoming ray (set_face_normal, shaders/pathtrace.metal:1187-1191)
                    if ((settings.metalSemantics & PTR_METAL_FACE_NORMAL) && !hit.frontFace) shadingNormal = -shadingNormal;
                }
                shadingNormal = normalize(shadingNormal);
                if (cp.metalPbr && type == PTR_MAT_PBR) {
                    bool passThrough = false;
                    if (texturedScene && hit.primitiveType == GeomType::Mesh && !scene.geoms[hit.geom].positions.empty()) {
                        texturedMaterial = *materialPtr;
                        V3 mapped;
                        bool twoSided = hit.twoSided;
                        passThrough = applyPbrTextures(scene, textures, hit, texturedMaterial, wo, coneWidth, coneSpread, hit.t, rng, mapped, twoSided);
                        if (!passThrough) {
                            materialPtr = &texturedMaterial;
                            shadingNormal = mapped;
                            hit.twoSided = twoSided;
                        }
                    }
                    if (passThrough) {   // M:6206-6216: the alpha test discarded the hit, the ray carries on
                        ray.origin = offsetRayOrigin(hit, ray.direction);
                        lastBsdfPdf = 1.0f;
                        lastScatterWasDelta = true;
                        specularDepth += 1u;
                        continue;
                    }
                    // an emissive metallic-roughness surface adds its emission and the path goes on (M:6437-6442)
                    const V3 emission(materialPtr->emission);
                    if ((emission.x != 0.0f || emission.y != 0.0f || emission.z != 0.0f) && (hit.frontFace || hit.twoSided)) {
                        radiance += clampFireflyContribution(throughput, emission, cp);
                    }
                }
                const PtrMaterial& material = *materialPtr;

                if (type == PTR_MAT_DIFFUSE_LIGHT) {  // E:2660-2706
                    V3 emission = V3(material.emission) * emissionScale;
                    if (material.emission[3] > 0.0f && envMapAvailable && hit.frontFace) {
                        emission *= sampleEnvironment(*env, -shadingNormal, settings.environmentRotation, settings.environmentIntensity);
                    }
                    if ((dot(emission, emission) > 0.0f) && (hit.frontFace || hit.twoSided)) {
                        float misWeight = 1.0f;
                        const bool useSpecularMis = (!lastScatterWasDelta) || settings.enableSpecularNee || settings.enableMnee;
                        if (useSpecularMis && rectLightCount > 0) {
                            const float lightPdf = rectLightPdfForHit(lightIndexByRect, rectangles, rectangleCount, rectLightCount, hit, ray.origin);
                            const float denom = lastBsdfPdf + lightPdf;
                            if (denom > 0.0f) misWeight = lastBsdfPdf / denom;
                            misWeight = clampf(misWeight, kMisWeightClampMin, kMisWeightClampMax);
                        }
                        radiance += clampFireflyContribution(throughput, emission * misWeight, cp);
                    }
                    break;
                }

                bool surfaceIsDelta = materialIsDelta(material);
                if (cp.metalPbr && matType(material) == PTR_MAT_PBR) {   // M:4578-4581
                    surfaceIsDelta = clampf(material.baseColorRoughness[3], 0.0f, 1.0f) <= 1.0e-3f;
                }

                if (!surfaceIsDelta && rectLightCount > 0) {  // rect-light NEE, E:2710-2772
                    RectLightSample ls;
                    if (sampleRectLight(rectLights, env, settings, hit, rng, ls)) {
                        const float nDotL = std::max(dot(shadingNormal, ls.direction), 0.0f);
                        if (ls.pdf > 0.0f && nDotL > 0.0f) {
                            const float shadowMax = std::max(ls.distance * (1.0f - shadowSlack) - kEpsilon, kEpsilon);
                            if (outMarginal) {
                                const V3 so = offsetRayOrigin(hit, ls.direction);
                                if (scene.occluded(so, ls.direction, kEpsilon, shadowMax * (1.0f - 2.0e-6f), false, nullptr) !=
                                    scene.occluded(so, ls.direction, kEpsilon, shadowMax * (1.0f + 2.0e-6f), false, nullptr)) {
                                    marginal = true;
                                }
                            }
                            if (!occluded(offsetRayOrigin(hit, ls.direction), ls.direction, shadowMax)) {
                                const BsdfEval be = evaluateBsdf(material, hit.position, shadingNormal, wo, ls.direction, cp);
                                if (neeContributes(be, cp)) {
                                    const float weight = neeWeight(ls.pdf, be.pdf, cp);
                                    V3 contribution = (ls.emission * be.value) * nDotL;
                                    contribution *= weight / ls.pdf;
                                    if (finite3(contribution)) {
                                        const V3 clamped = clampFireflyContribution(throughput, contribution, cp);
                                        radiance += clamped;
                                        if (wantSignature && depth < 16u && (clamped.x > 0.0f || clamped.y > 0.0f || clamped.z > 0.0f)) sig |= 1u << depth;
                                    }
                                }
                            }
                        }
                    }
                }

                if (!surfaceIsDelta && envSampling) {  // environment NEE, E:2774-2811
                    // draw into named locals: argument evaluation order is unspecified (quirk Q10);
                    // the reference binary (Apple clang) evaluates left to right
                    const float uMarginal = rng.nextFloat();
                    const float uConditional = rng.nextFloat();
                    const float uJitter = rng.nextFloat();
                    const EnvSample es = sampleEnvironmentCpu(env->dist, uMarginal, uConditional, uJitter,
                                                              settings.environmentRotation, settings.environmentIntensity, env->rgba);
                    const float nDotL = std::max(dot(shadingNormal, es.direction), 0.0f);
                    if (es.pdf > 0.0f && nDotL > 0.0f) {
                        if (!occluded(offsetRayOrigin(hit, es.direction), es.direction, kInf)) {
                            const V3 envRadiance = sampleEnvironment(*env, es.direction, settings.environmentRotation, settings.environmentIntensity);
                            const BsdfEval be = evaluateBsdf(material, hit.position, shadingNormal, wo, es.direction, cp);
                            if (neeContributes(be, cp)) {
                                const float weight = neeWeight(es.pdf, be.pdf, cp);
                                V3 contribution = (envRadiance * be.value) * nDotL;
                                contribution *= weight / es.pdf;
                                if (finite3(contribution)) radiance += clampFireflyContribution(throughput, contribution, cp);
                            }
                        }
                    }
                }

                BsdfSample bs;
                bool usedRandomWalk = false;   // M:6650-6676
                if (cp.metalSss && cp.sssMode == 2u && matType(material) == PTR_MAT_SUBSURFACE && material.sssParams[1] >= 0.5f && hit.frontFace) {
                    SssWalk walk;
                    WalkOutcome outcome = sssWalkBegin(material, hit.position, hit.normal, wo, incidentDir, rng, cp, bs, walk);
                    while (outcome == WalkOutcome::Walking) {
                        HitInfo boundary;
                        const bool hitBoundary = trace(Ray{walk.position, walk.direction}, boundary);
                        outcome = sssWalkStep(material, cp.sssMaxSteps, walk, hitBoundary, boundary.t, boundary.position, boundary.normal, rng, bs);
                    }
                    usedRandomWalk = outcome == WalkOutcome::Sample;
                }
                if (!usedRandomWalk) bs = sampleBsdf(material, hit.position, shadingNormal, wo, incidentDir, hit.frontFace, rng, cp);
                if (bs.pdf <= 0.0f || dot(bs.direction, bs.direction) <= 0.0f || !finite3(bs.weight)) break;

                if (media && bs.mediumEvent != 0) {  // M:6694-6709
                    if (bs.mediumEvent > 0) {
                        const uint32_t entered = std::min(hit.materialIndex, materialCount - 1);
                        if (mediumDepth < kMaxMediumStack) {
                            mediumStack[mediumDepth++] = entered;
                        } else {
                            mediumStack[kMaxMediumStack - 1] = entered;
                        }
                    } else if (mediumDepth > 0) {
                        --mediumDepth;
                    }
                }

                const uint32_t nextSpecularDepth = bs.isDelta ? (specularDepth + 1u) : 0u;
                specularDepth = nextSpecularDepth;

                const bool specDirectionValid = (dot(bs.direction, bs.direction) > 0.0f) && finite3(bs.direction);
                const bool mneeEligible = settings.enableMnee && bs.isDelta && specDirectionValid &&
                                          type == PTR_MAT_DIELECTRIC && nextSpecularDepth == 1u;
                const bool specNeeEligible = settings.enableSpecularNee && bs.isDelta && specDirectionValid && !mneeEligible;

                // Extra "NEE along the specular direction" rays.  The specular-NEE (E:2856-2917) and MNEE
                // (E:2919-2980) blocks are the same computation behind different gates.
                auto envAlong = [&](V3 origin, V3 dir, V3 weight, float bsdfPdfFloorInput) {
                    if (occluded(origin, dir, kInf)) return;
                    const float envPdf = std::max(environmentPdf(*env, settings.environmentRotation, dir), kSpecularNeePdfFloor);
                    const float invEnvPdf = std::min(1.0f / envPdf, kSpecularNeeInvPdfClamp);
                    const float bsdfPdf = std::max(bsdfPdfFloorInput, kSpecularNeePdfFloor);
                    const float denom = envPdf + bsdfPdf;
                    float misWeight = denom > 0.0f ? (envPdf / denom) : 0.0f;
                    misWeight = clampf(misWeight, kMisWeightClampMin, kMisWeightClampMax);
                    const V3 envColor = sampleEnvironment(*env, dir, settings.environmentRotation, settings.environmentIntensity);
                    const V3 c = (weight * envColor) * (misWeight * invEnvPdf);
                    if (finite3(c)) radiance += clampFireflyContribution(throughput, c, cp);
                };
                auto rectAlong = [&](V3 origin, V3 dir, V3 weight, float bsdfPdfFloorInput) {
                    HitInfo lightHit;
                    if (!trace(Ray{origin, dir}, lightHit)) return;
                    RectLightHit rh;
                    if (!rectLightHitInfo(lightIndexByRect, rectLights, rectangles, rectangleCount, env, settings, lightHit, origin, rh)) return;
                    const float lightPdf = std::max(rh.pdf, kSpecularNeePdfFloor);
                    const float invLightPdf = std::min(1.0f / lightPdf, kSpecularNeeInvPdfClamp);
                    const float bsdfPdf = std::max(bsdfPdfFloorInput, kSpecularNeePdfFloor);
                    const float denom = lightPdf + bsdfPdf;
                    float misWeight = denom > 0.0f ? (lightPdf / denom) : 0.0f;
                    misWeight = clampf(misWeight, kMisWeightClampMin, kMisWeightClampMax);
                    const V3 c = (weight * rh.emission) * (misWeight * invLightPdf);
                    if (finite3(c)) radiance += clampFireflyContribution(throughput, c, cp);
                };

                if (specNeeEligible || mneeEligible) {
                    const V3 dir = normalize(bs.direction);
                    const V3 origin = offsetRayOrigin(hit, dir);
                    if (envSampling) envAlong(origin, dir, bs.weight, bs.pdf);
                    if (rectLightCount > 0) rectAlong(origin, dir, bs.weight, bs.pdf);
                }

                if (mneeEligible && settings.enableMneeSecondary) {  // two-bounce specular chain, E:2982-3096
                    const V3 chainDir = normalize(bs.direction);
                    const Ray chainRay{offsetRayOrigin(hit, chainDir), chainDir};
                    HitInfo chainHit;
                    if (trace(chainRay, chainHit)) {
                        bool chainHitIsLight = false;
                        if (rectLightCount > 0) {
                            RectLightHit tmp;
                            chainHitIsLight = rectLightHitInfo(lightIndexByRect, rectLights, rectangles, rectangleCount, env, settings,
                                                               chainHit, chainRay.origin, tmp);
                        }
                        if (!chainHitIsLight) {
                            const PtrMaterial& chainMaterial = materials[std::min(chainHit.materialIndex, materialCount - 1)];
                            if (materialIsDelta(chainMaterial)) {
                                V3 chainNormal = chainHit.normal;
                                if (dot(chainNormal, chainNormal) <= 0.0f) chainNormal = V3(0.0f, 1.0f, 0.0f);
                                chainNormal = normalize(chainNormal);
                                const V3 chainIncident = normalize(chainRay.direction);
                                Rng chainRng = rng;  // copy: the main stream does not advance
                                const BsdfSample cs = sampleBsdf(chainMaterial, chainHit.position, chainNormal, -chainIncident,
                                                                 chainIncident, chainHit.frontFace, chainRng, cp);
                                if (cs.pdf > 0.0f && cs.isDelta && dot(cs.direction, cs.direction) > 0.0f && finite3(cs.weight)) {
                                    const V3 secondDir = normalize(cs.direction);
                                    const V3 secondOrigin = offsetRayOrigin(chainHit, secondDir);
                                    const V3 combinedWeight = bs.weight * cs.weight;
                                    const float chainPdf = bs.pdf * cs.pdf;
                                    if (envSampling) envAlong(secondOrigin, secondDir, combinedWeight, chainPdf);
                                    if (rectLightCount > 0) rectAlong(secondOrigin, secondDir, combinedWeight, chainPdf);
                                }
                            }
                        }
                    }
                }

                throughput *= bs.weight;
                throughput = clampPathThroughput(throughput, cp);
                if (!finite3(throughput)) break;
                const float maxComp = std::max(std::max(throughput.x, throughput.y), throughput.z);
                if (maxComp <= 0.0f) break;

                if (texturedScene) {   // the path's ray cone (M:7262-7267, 5703-5715)
                    coneWidth = std::max(coneWidth + coneSpread * std::max(hit.t, 0.0f), 1.0e-7f);
                    float inc = 0.0f;
                    if (!bs.isDelta) {
                        const bool pbr = cp.metalPbr && type == PTR_MAT_PBR;
                        const int lobe = pbr ? bs.lobe : ((type == PTR_MAT_LAMBERTIAN || type == PTR_MAT_SUBSURFACE) ? 0 : 1);
                        const float r = clampf(pbr ? bs.lobeRoughness : material.baseColorRoughness[3], 0.0f, 1.0f);
                        inc = lobe == 0 ? 0.55f : (lobe == 1 ? 0.03f + (0.45f - 0.03f) * r : 0.10f + (0.60f - 0.10f) * r);
                    }
                    coneSpread = std::min(coneSpread + inc, 1.5f);
                }
                lastBsdfPdf = bs.pdf > 0.0f ? bs.pdf : lastBsdfPdf;
                lastScatterWasDelta = bs.isDelta;
                ray.origin = bs.hasExitPoint ? sssExitOrigin(bs.exitPoint, dot(bs.exitNormal, bs.exitNormal) > 0.0f ? bs.exitNormal : shadingNormal, bs.direction)
                                            : offsetRayOrigin(hit, bs.direction);
                ray.direction = bs.direction;

                if (settings.enableRussianRoulette && depth >= 5) {
                    const float rrProb = clampf(maxComp, 0.05f, 0.95f);
                    if (rng.nextFloat() > rrProb) break;
                    throughput /= rrProb;
                }
            }
            pixelRadiance += radiance;
        }
        if (counting) {
            rc.nodes += tc.nodes;
            rc.prims += tc.prims;
        }
        if (outSignature) outSignature[pixelIndex] = sig;
        if (outMarginal) outMarginal[pixelIndex] = marginal ? 1u : 0u;
        return pixelRadiance / static_cast<float>(targetSamples);
    };

    // 16x16 tiles pulled from an atomic counter by a std::thread pool (E:2538-2571, 3159-3187)
    constexpr uint32_t kTile = 16u;
    yEnd = std::min(yEnd, height);
    const uint32_t tilesX = (width + kTile - 1u) / kTile;
    const uint32_t tileY0 = yBegin / kTile, tileY1 = (yEnd + kTile - 1u) / kTile;
    const uint32_t totalTiles = tilesX * (tileY1 > tileY0 ? tileY1 - tileY0 : 0u);
    std::atomic<uint32_t> nextTile{0};
    auto worker = [&]() {
        RenderCounters rc;
        while (true) {
            const uint32_t tile = nextTile.fetch_add(1, std::memory_order_relaxed);
            if (tile >= totalTiles) break;
            const uint32_t ty = tileY0 + tile / tilesX, tx = tile % tilesX;
            const uint32_t x0 = tx * kTile, x1 = std::min(x0 + kTile, width);
            const uint32_t y0 = std::max(ty * kTile, yBegin), y1 = std::min(ty * kTile + kTile, yEnd);
            for (uint32_t y = y0; y < y1; ++y) {
                for (uint32_t x = x0; x < x1; ++x) {
                    const V3 avg = renderPixel(x, y, rc);
                    float* px = outRgb + (static_cast<size_t>(y) * width + x) * 3u;
                    px[0] = avg.x;
                    px[1] = avg.y;
                    px[2] = avg.z;
                }
            }
        }
        if (counting) {
            cExtend += rc.extendRays;
            cShadow += rc.shadowRays;
            cNodes += rc.nodes;
            cPrims += rc.prims;
            cShaded += rc.shadedHits;
            cTri += rc.triangleHits;
        }
    };
    uint32_t workerCount = threads;
    if (workerCount == 0) workerCount = std::max(1u, std::thread::hardware_concurrency());
    if (workerCount == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (uint32_t i = 0; i < workerCount; ++i) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    if (outCounters) {
        outCounters->extendRays = cExtend;
        outCounters->shadowRays = cShadow;
        outCounters->nodes = cNodes;
        outCounters->prims = cPrims;
        outCounters->shadedHits = cShaded;
        outCounters->triangleHits = cTri;
    }
}

}  // namespace oracle
