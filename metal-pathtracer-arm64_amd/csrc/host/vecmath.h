// Small float vector/matrix types for the host scene layer (stands in for Apple <simd/simd.h>,
// which the reference's SceneResources/SceneManager use).  Column-major float4x4 like simd.
#pragma once

#include <cmath>
#include <cstdint>

namespace ptr {

struct float2 {
    float x = 0.0f, y = 0.0f;
};

struct float3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    float3() = default;
    float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct float4 {
    float x = 0.0f, y = 0.0f, z = 0.0f, w = 0.0f;
    float4() = default;
    float4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    float4(const float3& v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
};

inline float3 operator+(const float3& a, const float3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 operator-(const float3& a, const float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 operator-(const float3& a) { return {-a.x, -a.y, -a.z}; }
inline float3 operator*(const float3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator*(float s, const float3& a) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 operator/(const float3& a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(const float3& a, const float3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float3 cross(const float3& a, const float3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(const float3& a) { return std::sqrt(dot(a, a)); }
inline float3 normalize(const float3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct float4x4 {
    float4 columns[4];
    static float4x4 identity() {
        float4x4 m;
        m.columns[0] = {1, 0, 0, 0};
        m.columns[1] = {0, 1, 0, 0};
        m.columns[2] = {0, 0, 1, 0};
        m.columns[3] = {0, 0, 0, 1};
        return m;
    }
    float at(int row, int col) const { return (&columns[col].x)[row]; }
    float& at(int row, int col) { return (&columns[col].x)[row]; }
};

inline float4 mul(const float4x4& m, const float4& v) {
    float4 r;
    r.x = m.columns[0].x * v.x + m.columns[1].x * v.y + m.columns[2].x * v.z + m.columns[3].x * v.w;
    r.y = m.columns[0].y * v.x + m.columns[1].y * v.y + m.columns[2].y * v.z + m.columns[3].y * v.w;
    r.z = m.columns[0].z * v.x + m.columns[1].z * v.y + m.columns[2].z * v.z + m.columns[3].z * v.w;
    r.w = m.columns[0].w * v.x + m.columns[1].w * v.y + m.columns[2].w * v.z + m.columns[3].w * v.w;
    return r;
}

inline float4x4 mul(const float4x4& a, const float4x4& b) {
    float4x4 r;
    for (int c = 0; c < 4; ++c) {
        r.columns[c] = mul(a, b.columns[c]);
    }
    return r;
}

}  // namespace ptr
