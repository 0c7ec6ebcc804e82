// TEST INFRASTRUCTURE — not part of the product.  See oracle/README.md.
// Minimal float3 arithmetic with the evaluation order of Apple's <simd/simd.h> as used by the reference's
// Embree path (dot = (x*x' + y*y') + z*z'; normalize(v) = v * (1/sqrt(dot(v,v))); length = sqrt(dot)).
// Compiled with -ffp-contract=off so every operation rounds once, in source order.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>

namespace oracle {

struct V3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    V3() = default;
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
};

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline V3& operator-=(V3& a, V3 b) { a = a - b; return a; }
inline V3& operator*=(V3& a, V3 b) { a = a * b; return a; }
inline V3& operator*=(V3& a, float s) { a = a * s; return a; }
inline V3& operator/=(V3& a, float s) { a = a / s; return a; }

inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vsqrt(V3 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)}; }
inline V3 splat(float s) { return {s, s, s}; }
inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }  // == std::clamp for lo<=hi
inline V3 vclamp(V3 a, float lo, float hi) { return {clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)}; }
inline bool finite3(V3 a) { return std::isfinite(a.x) && std::isfinite(a.y) && std::isfinite(a.z); }

constexpr float kPi = 3.14159265358979323846f;

}  // namespace oracle
