// Device float3 arithmetic.  Evaluation order is fixed (dot = (xx'+yy')+zz', normalize = v * (1/sqrt(dot)))
// and the translation unit is compiled with -ffp-contract=off, so products and sums round once each in
// source order — the same convention the CPU oracle uses, which keeps per-path arithmetic comparable.
#pragma once

#include <hip/hip_runtime.h>

namespace ptrk {

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 mk3(float s) { return f3{s, s, s}; }
__device__ __forceinline__ f3 mk3(const float4& v) { return f3{v.x, v.y, v.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
__device__ __forceinline__ f3& operator-=(f3& a, f3 b) { a = a - b; return a; }
__device__ __forceinline__ f3& operator*=(f3& a, f3 b) { a = a * b; return a; }
__device__ __forceinline__ f3& operator*=(f3& a, float s) { a = a * s; return a; }
__device__ __forceinline__ f3& operator/=(f3& a, float s) { a = a / s; return a; }

__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float length(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 normalize(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
// std::max / std::min semantics (first argument wins on NaN/equality) — NOT fmaxf
__device__ __forceinline__ float smax(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float smin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ f3 vmax(f3 a, f3 b) { return f3{smax(a.x, b.x), smax(a.y, b.y), smax(a.z, b.z)}; }
__device__ __forceinline__ f3 vmax0(f3 a) { return f3{smax(a.x, 0.0f), smax(a.y, 0.0f), smax(a.z, 0.0f)}; }
__device__ __forceinline__ f3 vsqrt(f3 a) { return f3{sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)}; }
__device__ __forceinline__ f3 vmaxs(f3 a, float s) { return f3{smax(a.x, s), smax(a.y, s), smax(a.z, s)}; }
__device__ __forceinline__ f3 vexp(f3 a) { return f3{expf(a.x), expf(a.y), expf(a.z)}; }
// std::clamp semantics for lo <= hi (NaN passes through)
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }
__device__ __forceinline__ f3 vclamp(f3 a, float lo, float hi) {
    return f3{clampf(a.x, lo, hi), clampf(a.y, lo, hi), clampf(a.z, lo, hi)};
}
__device__ __forceinline__ bool finite3(f3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }
__device__ __forceinline__ float4 mk4(f3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

constexpr float kPi = 3.14159265358979323846f;

}  // namespace ptrk
