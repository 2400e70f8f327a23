// glm_min.h — the dozen glm types and functions include/lumen_mi_renderer.hpp and examples/sandbox_driver.cpp touch, so that both build
// where neither glm nor the reference tree exists (the GPU box).  Column-major matrices and (w, x, y, z) quaternions as in glm; nothing else
// of glm is modelled.  Test scaffolding for tests/test_gpu_parity.py::test_reference_shaped_adapter_renders_the_c_example_picture.
#pragma once
#include <cmath>

namespace glm
{
    struct vec2 { float x = 0, y = 0; vec2() = default; vec2(float a, float b) : x(a), y(b) {} };
    struct uvec2 { unsigned x = 0, y = 0; uvec2() = default; uvec2(unsigned a, unsigned b) : x(a), y(b) {} };
    struct vec3
    {
        float x = 0, y = 0, z = 0;
        vec3() = default; explicit vec3(float s) : x(s), y(s), z(s) {} vec3(float a, float b, float c) : x(a), y(b), z(c) {}
        float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; }
    };
    struct vec4
    {
        float x = 0, y = 0, z = 0, w = 0;
        vec4() = default; vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {} vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
        float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; }
    };
    struct mat3 { vec3 c[3]; mat3() = default; mat3(const vec3& a, const vec3& b, const vec3& d) { c[0] = a; c[1] = b; c[2] = d; } vec3& operator[](int i) { return c[i]; } const vec3& operator[](int i) const { return c[i]; } };
    struct mat4
    {
        vec4 c[4];
        mat4() = default;
        explicit mat4(float d) { c[0].x = d; c[1].y = d; c[2].z = d; c[3].w = d; }
        vec4& operator[](int i) { return c[i]; } const vec4& operator[](int i) const { return c[i]; }
    };
    struct quat { float w = 1, x = 0, y = 0, z = 0; quat() = default; quat(float a, float b, float c, float d) : w(a), x(b), y(c), z(d) {} };

    inline mat4 transpose(const mat4& m) { mat4 t; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) t[i][j] = m[j][i]; return t; }
    inline const float* value_ptr(const mat4& m) { return &m.c[0].x; }
    // rotation matrix -> unit quaternion (largest-component branch), and back
    inline quat quat_cast(const mat3& m)
    {
        const float fx = m[0][0] - m[1][1] - m[2][2], fy = m[1][1] - m[0][0] - m[2][2], fz = m[2][2] - m[0][0] - m[1][1], fw = m[0][0] + m[1][1] + m[2][2];
        int big = 0; float best = fw;
        if (fx > best) { best = fx; big = 1; } if (fy > best) { best = fy; big = 2; } if (fz > best) { best = fz; big = 3; }
        const float v = std::sqrt(best + 1.0f) * 0.5f, k = 0.25f / v;
        switch (big) {
        case 0: return quat(v, (m[1][2] - m[2][1]) * k, (m[2][0] - m[0][2]) * k, (m[0][1] - m[1][0]) * k);
        case 1: return quat((m[1][2] - m[2][1]) * k, v, (m[0][1] + m[1][0]) * k, (m[2][0] + m[0][2]) * k);
        case 2: return quat((m[2][0] - m[0][2]) * k, (m[0][1] + m[1][0]) * k, v, (m[1][2] + m[2][1]) * k);
        default: return quat((m[0][1] - m[1][0]) * k, (m[2][0] + m[0][2]) * k, (m[1][2] + m[2][1]) * k, v);
        }
    }
    inline mat4 toMat4(const quat& q)
    {
        const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
        mat4 r(1.0f);
        r[0][0] = 1.0f - 2.0f * (yy + zz); r[0][1] = 2.0f * (xy + wz); r[0][2] = 2.0f * (xz - wy);
        r[1][0] = 2.0f * (xy - wz); r[1][1] = 1.0f - 2.0f * (xx + zz); r[1][2] = 2.0f * (yz + wx);
        r[2][0] = 2.0f * (xz + wy); r[2][1] = 2.0f * (yz - wx); r[2][2] = 1.0f - 2.0f * (xx + yy);
        return r;
    }
}
