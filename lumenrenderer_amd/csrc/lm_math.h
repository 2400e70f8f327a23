// lm_math.h — arithmetic conventions of the MI355X path tracer (device + host).
//
// Every stage of the path is specified as exact IEEE-754 binary32 arithmetic, evaluated as written with no
// contraction (the whole library is built with -ffp-contract=off; an FMA happens only where fmaf is written),
// so that results are reproducible across launches, wave orders and devices, and can be checked bit for bit.
// Vector helpers keep the operation order of the reference's sutil/vec_math.h (LumenPT/vendor/Include/sutil/
// vec_math.h:415-561): dot = x*x' + y*y' + z*z', normalize = v * (1/sqrt(dot)), v / s = v * (1/s).
// Transcendentals are fixed polynomial routines (Cephes single-precision coefficients) instead of OCML, whose
// results are not specified to the bit; DESIGN.md "Arithmetic" documents them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LM_HD __host__ __device__ __forceinline__

struct lf2 { float x, y; };
struct lf3 { float x, y, z; };

LM_HD lf3 v3(float x, float y, float z) { lf3 r; r.x = x; r.y = y; r.z = z; return r; }
LM_HD lf3 v3(float s) { return v3(s, s, s); }
LM_HD lf3 v3(const float4& v) { return v3(v.x, v.y, v.z); }
LM_HD float4 v4(const lf3& v, float w) { return make_float4(v.x, v.y, v.z, w); }
// float4 arithmetic uses HIP's own element-wise operators (hip/amd_detail/amd_hip_vector_types.h)

LM_HD lf3 operator+(const lf3& a, const lf3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LM_HD lf3 operator-(const lf3& a, const lf3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LM_HD lf3 operator*(const lf3& a, const lf3& b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
LM_HD lf3 operator*(const lf3& a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
LM_HD lf3 operator*(float s, const lf3& a) { return v3(a.x * s, a.y * s, a.z * s); }
LM_HD lf3 operator+(const lf3& a, float s) { return v3(a.x + s, a.y + s, a.z + s); }
LM_HD lf3 operator+(float s, const lf3& a) { return v3(s + a.x, s + a.y, s + a.z); }
LM_HD lf3 operator-(const lf3& a, float s) { return v3(a.x - s, a.y - s, a.z - s); }
LM_HD lf3 operator-(const lf3& a) { return v3(-a.x, -a.y, -a.z); }
LM_HD lf3 operator/(const lf3& a, float s) { const float inv = 1.0f / s; return a * inv; }

LM_HD float dot3(const lf3& a, const lf3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LM_HD lf3 cross3(const lf3& a, const lf3& b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
LM_HD float length3(const lf3& v) { return sqrtf(dot3(v, v)); }
LM_HD lf3 normalize3(const lf3& v) { const float inv = 1.0f / sqrtf(dot3(v, v)); return v * inv; }
LM_HD lf3 reflect3(const lf3& i, const lf3& n) { return i - 2.0f * n * dot3(n, i); }
LM_HD float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
LM_HD float lerpf(float a, float b, float t) { return a + t * (b - a); }
LM_HD float saturatef(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
LM_HD float sqrf(float a) { return a * a; }

LM_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
LM_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

// ---- fixed transcendentals ------------------------------------------------------------------------------
LM_HD void lm_sincosf(float x, float* s, float* c)
{
    const float kf = rintf(x * 0.636619772367581343f);
    const int k = (int)kf;
    float y = fmaf(kf, -1.5703125f, x);
    y = fmaf(kf, -4.837512969970703125e-4f, y);
    y = fmaf(kf, -7.54978995489188216e-8f, y);
    const float z = y * y;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sp = fmaf(ps * z, y, y);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cp = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = k & 3;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = (q == 1 || q == 2) ? -cc : cc;
}
LM_HD float lm_logf(float x)
{
    const uint32_t bits = f2u(x);
    int e = (int)((bits >> 23) & 255u) - 126;
    float m = u2f((bits & 0x007fffffu) | 0x3f000000u);
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
LM_HD float lm_expf(float x)
{
    if (!(x >= -87.0f)) return (x != x) ? x : 0.0f;
    if (x > 88.7f) return u2f(0x7f800000u);
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
    const int ni = (int)n;
    const int h = ni / 2;
    return v * u2f((uint32_t)(h + 127) << 23) * u2f((uint32_t)(ni - h + 127) << 23);
}
LM_HD float lm_powf(float a, float b)
{
    if (a == 0.0f) return (b == 0.0f) ? 1.0f : 0.0f;
    return lm_expf(b * lm_logf(a));
}

// ---- binary16 <-> binary32, round to nearest even ---------------------------------------------------------
LM_HD uint32_t lm_f32_to_f16(float f)
{
    const uint32_t x = f2u(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x0200u : 0u);
    if (ax >= 0x477ff000u) return sign | 0x7c00u;
    if (ax < 0x33000001u) return sign;
    const int e = (int)(ax >> 23) - 127;
    const uint32_t m = (ax & 0x007fffffu) | 0x00800000u;
    int shift; uint32_t he;
    if (e < -14) { shift = 13 + (-14 - e); he = 0; } else { shift = 13; he = (uint32_t)(e + 15); }
    uint32_t hm = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    const uint32_t h = (he == 0) ? hm : ((he << 10) + (hm - 0x400u));
    return sign | h;
}
LM_HD float lm_f16_to_f32(uint32_t h)
{
    const uint32_t sign = (h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 31u;
    const uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        const float v = (float)m * 5.9604644775390625e-8f;
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

// ---- RNG (reference: LumenPT/src/CUDAKernels/RandomUtilities.cuh:5-18) ------------------------------------
LM_HD uint32_t lm_wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}
LM_HD uint32_t lm_random_int(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
LM_HD float lm_random_float(uint32_t& s) { return (float)lm_random_int(s) * 2.3283064365387e-10f; }
// (int)roundf(x) for x >= 0 (round half away from zero): x - trunc(x) is exact, so this IS roundf there, without the sign handling
// of the general function (5 instead of 8 instructions in the 32-candidate loop of the pick)
LM_HD int lm_round_nonneg(float x) { const float t = truncf(x); return (int)t + ((x - t) >= 0.5f ? 1 : 0); }
// Halton radical inverse, index pre-incremented (reference: GPUGeneratePrimRay.cu:8-26): f = f / base; r = r + f * (index % base); index /= base.
// The primary-ray kernel spent 31 M wave instructions per frame in these two loops (an IEEE division sequence per digit), so bases 2 and 3
// take shortcuts that reproduce the loop's roundings exactly (pinned: tests/golden/ref_kat.npz rows `halt`, and the loop itself below):
//   base 2: f = 2^-k is exact and every partial sum has at most k significant bits, so below 2^24 the result is the bit-reversed index
//           scaled by 2^-32, exactly; larger indices (never reached by a 4K frame) take the loop;
//   base 3: the loop's f values do not depend on the index — f_k = fl(f_(k-1) / 3) — so they are compile-time constants, the digits come
//           from a multiply-high, and a zero digit above the top one adds +0 (r + 0 = r) where the loop has already stopped.
LM_HD float lm_halton_loop(uint32_t index, uint32_t base)
{
    ++index;
    float f = 1.f, r = 0.f;
    const float fb = (float)base;
    while (index > 0) {
        f = f / fb;
        r = r + f * (float)(index % base);
        index = index / base;
    }
    return r;
}
struct LmThirds { float f[21]; };
constexpr LmThirds lm_thirds() { LmThirds t{}; float f = 1.f; for (int k = 0; k < 21; k++) { f = f / 3.f; t.f[k] = f; } return t; }      // 3^21 > 2^32
LM_HD uint32_t lm_brev32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1); v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4); v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
LM_HD float lm_halton(uint32_t index, uint32_t base)
{
    if (base == 2u) {
        const uint32_t i1 = index + 1u;
        if (i1 < (1u << 24)) return (float)lm_brev32(i1) * 2.3283064365386963e-10f;      // 2^-32; at most 24 significant bits: both steps exact
        return lm_halton_loop(index, 2u);
    }
    if (base == 3u) {
        constexpr LmThirds T = lm_thirds();
        uint32_t i1 = index + 1u;
        float r = 0.f;
        for (int k = 0; k < 21; k++) {
            if (i1 == 0u) break;
            const uint32_t q = i1 / 3u;
            r = r + T.f[k] * (float)(i1 - 3u * q);
            i1 = q;
        }
        return r;
    }
    return lm_halton_loop(index, base);
}
