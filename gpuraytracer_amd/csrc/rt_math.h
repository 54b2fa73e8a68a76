// Device-side float helpers shared by the render kernels. Every function is one written operation
// per rounding (the library is built with -ffp-contract=off), so results are reproducible against
// a host restatement of the same expressions.
//
// HLSL -> HIP rules: dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z; normalize(v) = v / sqrt(dot(v,v));
// saturate(NaN) = 0; min/max = v_min_f32 / v_max_f32 (non-NaN operand wins, -0 < +0); round = rintf
// (ties to even); pow(x, 5) = ((x*x)*(x*x))*x; float -> uint truncates, negatives and NaN give 0;
// sin/cos = range reduction + 11/10-degree minimax polynomials.
#ifndef GPURT_MATH_H
#define GPURT_MATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtm {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 mul(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 scale(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 divs(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize3(f3 a) { return divs(a, sqrtf(dot3(a, a))); }
__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
__device__ __forceinline__ uint32_t f2u(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}

// shader/Random.hlsli:3-9, 63-66 of the reference
__device__ __forceinline__ uint32_t xorshift(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ float rnd(uint32_t& s) { return saturatef((float)xorshift(s) * 2.3283064e-10f); }

__device__ __forceinline__ void sincos_poly(float x, float* s, float* c) {
    float q = 0.15915494f * x;
    q = (x >= 0.0f) ? (float)(int)(q + 0.5f) : (float)(int)(q - 0.5f);
    float y = x - 6.2831855f * q;
    float sign;
    if (y > 1.5707964f) { y = 3.1415927f - y; sign = -1.0f; }
    else if (y < -1.5707964f) { y = -3.1415927f - y; sign = -1.0f; }
    else sign = 1.0f;
    float y2 = y * y;
    *s = (((((-2.3889859e-08f * y2 + 2.7525562e-06f) * y2 - 0.00019840874f) * y2 + 0.0083333310f) * y2 - 0.16666667f) * y2 + 1.0f) * y;
    float p = ((((-2.6051615e-07f * y2 + 2.4760495e-05f) * y2 - 0.0013888378f) * y2 + 0.041666638f) * y2 - 0.5f) * y2 + 1.0f;
    *c = sign * p;
}

}  // namespace rtm

#endif
