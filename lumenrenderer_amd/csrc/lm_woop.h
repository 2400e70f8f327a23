// lm_woop.h — Woop unit-triangle packet from a world-space triangle; shared by the host BVH builder (bvh.cpp) and the GPU
// refit kernel (kernels.hip) so that both produce the same bits: IEEE double arithmetic, no contraction, one rounding to
// fp32 per coefficient.  The packet maps the triangle to the unit triangle: rows r0/r1/r2 give (u, v, w) of a point.
#pragma once
#include "lm_layout.h"

__host__ __device__ inline LmWoop lm_make_woop(const float* t)
{
    const double v0[3] = {t[0], t[1], t[2]};
    const double e1[3] = {(double)t[3] - t[0], (double)t[4] - t[1], (double)t[5] - t[2]};
    const double e2[3] = {(double)t[6] - t[0], (double)t[7] - t[1], (double)t[8] - t[2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double det = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    LmWoop w;
    w.r0 = make_float4(0.f, 0.f, 0.f, 0.f); w.r1 = w.r0; w.r2 = w.r0;
    if (!(det > 0.0) || !(det - det == 0.0)) return w;         // degenerate (or non-finite) triangle: the zero packet never reports a hit
    const double ru[3] = {e2[1] * n[2] - e2[2] * n[1], e2[2] * n[0] - e2[0] * n[2], e2[0] * n[1] - e2[1] * n[0]};
    const double rv[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};
    float r0[4], r1[4], r2[4];
    double du = 0, dv = 0, dw = 0;
    for (int i = 0; i < 3; i++) {
        r0[i] = (float)(ru[i] / det); r1[i] = (float)(rv[i] / det); r2[i] = (float)(n[i] / det);
        du -= ru[i] / det * v0[i]; dv -= rv[i] / det * v0[i]; dw -= n[i] / det * v0[i];
    }
    r0[3] = (float)du; r1[3] = (float)dv; r2[3] = (float)dw;
    w.r0 = make_float4(r0[0], r0[1], r0[2], r0[3]);
    w.r1 = make_float4(r1[0], r1[1], r1[2], r1[3]);
    w.r2 = make_float4(r2[0], r2[1], r2[2], r2[3]);
    return w;
}
