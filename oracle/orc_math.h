// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call it.
//
// orc_math.h: scalar/vector arithmetic conventions of the restatement.
//
// The reference's device math is sutil/vec_math.h + CUDA libm compiled with -use_fast_math
// (LumenPT/CMakeLists.txt:35,114), so its last-ulp behaviour is unpinned.  The oracle fixes
// one exact arithmetic so that a second implementation can be compared with it bit for bit:
//   * every operation is IEEE-754 binary32, evaluated exactly as written, left to right, NO contraction
//     (build with -ffp-contract=off); a fused multiply-add happens only where fmaf() is written;
//   * dot/cross/normalize/length/reflect and the float3 operators follow the operation order of
//     LumenPT/vendor/Include/sutil/vec_math.h:415-561 (dot = x*x' + y*y' + z*z'; normalize = v * (1/sqrt(dot));
//     v / s = v * (1/s));
//   * sin/cos/log/exp/pow are the fixed polynomial routines below (Cephes single-precision coefficients),
//     not libm, so they can be reproduced on any device.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

static inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
static inline f3 mk3(float s) { return f3{s, s, s}; }
static inline f3 mk3(const f4& v) { return f3{v.x, v.y, v.z}; }
static inline f4 mk4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
static inline f4 mk4(const f3& v, float w) { return f4{v.x, v.y, v.z, w}; }

static inline f3 operator+(const f3& a, const f3& b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline f3 operator-(const f3& a, const f3& b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline f3 operator*(const f3& a, const f3& b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline f3 operator*(const f3& a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
static inline f3 operator*(float s, const f3& a) { return f3{a.x * s, a.y * s, a.z * s}; }
static inline f3 operator+(const f3& a, float s) { return f3{a.x + s, a.y + s, a.z + s}; }
static inline f3 operator+(float s, const f3& a) { return f3{s + a.x, s + a.y, s + a.z}; }
static inline f3 operator-(const f3& a) { return f3{-a.x, -a.y, -a.z}; }
static inline f3 operator/(const f3& a, float s) { float inv = 1.0f / s; return a * inv; }   // vec_math.h:480-484
static inline f3& operator+=(f3& a, const f3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline f3& operator*=(f3& a, const f3& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
static inline f3& operator*=(f3& a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
static inline f3& operator/=(f3& a, float s) { float inv = 1.0f / s; a *= inv; return a; }

static inline f4 operator*(const f4& a, float s) { return f4{a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline f4 operator*(const f4& a, const f4& b) { return f4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline f4 operator+(const f4& a, const f4& b) { return f4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline f2 operator*(const f2& a, float s) { return f2{a.x * s, a.y * s}; }
static inline f2 operator+(const f2& a, const f2& b) { return f2{a.x + b.x, a.y + b.y}; }

static inline float dot(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline f3 cross(const f3& a, const f3& b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float length(const f3& v) { return sqrtf(dot(v, v)); }
static inline f3 normalize(const f3& v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
static inline f3 reflect(const f3& i, const f3& n) { return i - 2.0f * n * dot(n, i); }     // vec_math.h:558-561
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }     // vec_math.h:119-122
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }            // bsdf_math.cuh:16-19
static inline float saturatef(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
static inline float sqr(float a) { return a * a; }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---------------------------------------------------------------------------------------------------------
// Fixed transcendental routines (spec shared with the device implementation, see DESIGN.md "arithmetic").
// ---------------------------------------------------------------------------------------------------------
static inline void det_sincosf(float x, float* s, float* c)
{
    const float kf = rintf(x * 0.636619772367581343f);        // nearest multiple of pi/2 (ties to even)
    const int k = (int)kf;
    float y = fmaf(kf, -1.5703125f, x);                       // Cody-Waite 3-term reduction
    y = fmaf(kf, -4.837512969970703125e-4f, y);
    y = fmaf(kf, -7.54978995489188216e-8f, y);
    const float z = y * y;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sp = fmaf(ps * z, y, y);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cp = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
    case 0: *s = sp;  *c = cp;  break;
    case 1: *s = cp;  *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

// natural log, x > 0 and normal
static inline float det_logf(float x)
{
    const uint32_t bits = f2u(x);
    int e = (int)((bits >> 23) & 255u) - 126;
    float m = u2f((bits & 0x007fffffu) | 0x3f000000u);        // mantissa in [0.5, 1)
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = fmaf(p, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float y = p * m * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

// e^x; returns 0 below -87, +inf above 88.7
static inline float det_expf(float x)
{
    if (!(x >= -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.7f) return INFINITY;
    const float n = floorf(fmaf(1.44269504088896341f, x, 0.5f));
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float v = fmaf(p, z, r) + 1.0f;
    const int ni = (int)n;                                    // in [-126, 128]
    // scale by 2^ni in two exact steps so that ni = 128 and ni = -126 stay representable
    const int h = ni / 2;
    return v * u2f((uint32_t)(h + 127) << 23) * u2f((uint32_t)(ni - h + 127) << 23);
}

static inline float det_powf(float a, float b)               // a >= 0
{
    if (a == 0.0f) return (b == 0.0f) ? 1.0f : 0.0f;
    return det_expf(b * det_logf(a));
}

// ---------------------------------------------------------------------------------------------------------
// binary16 <-> binary32, round to nearest even (what __float22half2_rn / __half2float do; used for the
// barycentrics of IntersectionData.h:90 and the motion vectors of MotionVectors.cu:42)
// ---------------------------------------------------------------------------------------------------------
static inline uint16_t f32_to_f16(float f)
{
    const uint32_t x = f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x0200u : 0u));   // inf / nan
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                        // overflow -> inf
    if (ax < 0x33000001u) return (uint16_t)sign;                                                      // underflow -> 0
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    int shift;
    uint32_t he;
    if (e < -14) { shift = 13 + (-14 - e); he = 0; } else { shift = 13; he = (uint32_t)(e + 15); }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    uint32_t h;
    if (he == 0) h = hm;                       // subnormal (hm may carry into the exponent: still correct)
    else h = ((he << 10) + (hm - 0x400u));     // hm in [0x400, 0x800]; carry propagates into the exponent
    return (uint16_t)(sign | h);
}

static inline float f16_to_f32(uint16_t h)
{
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u;
    const uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        // subnormal: m * 2^-24
        float v = (float)m * 5.9604644775390625e-8f;
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
static inline float quantize_f16(float f) { return f16_to_f32(f32_to_f16(f)); }

// ---------------------------------------------------------------------------------------------------------
// RNG — LumenPT/src/CUDAKernels/RandomUtilities.cuh:5-18
// ---------------------------------------------------------------------------------------------------------
static inline uint32_t wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}
static inline uint32_t random_int(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
static inline float random_float(uint32_t& s) { return (float)random_int(s) * 2.3283064365387e-10f; }

// Halton radical inverse with the reference's "++index" — GPUGeneratePrimRay.cu:8-26
static inline float halton(uint32_t index, uint32_t base)
{
    ++index;
    float f = 1.f, r = 0.f;
    while (index > 0) {
        f = f / (float)base;
        r = r + f * (float)(index % base);
        index = index / base;
    }
    return r;
}

}  // namespace orc
