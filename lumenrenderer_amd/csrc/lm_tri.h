// lm_tri.h — triangle packet of the traversal from a world-space triangle; shared by the host BVH builder (bvh.cpp) and the GPU refit
// kernel (kernels.hip).  The watertight ray / triangle test (lm_traverse.h lm_tri_test: Woop, Benthin, Wald, JCGT 2013) works on the
// vertices themselves, so that two triangles which share a vertex evaluate the very same numbers: the packet IS the three vertices.
// Each vertex is stored as FIVE floats x y z x y, because the test permutes the axes cyclically per ray — (kx, ky, kz) = (0,1,2), (1,2,0)
// or (2,0,1) — and the three consecutive floats that start at float kx of the record are exactly (v[kx], v[ky], v[kz]): a ray reads its
// permuted vertex with one 12-byte load at a per-ray offset and spends no instruction on the permutation.  3 x 5 floats + one unused =
// 64 bytes: one triangle = one aligned cache line.
#pragma once
#include "lm_layout.h"

__host__ __device__ inline LmTriPacket lm_make_packet(const float* t)
{
    LmTriPacket w;
    for (int k = 0; k < 3; k++) {
        w.f[5 * k + 0] = t[3 * k + 0]; w.f[5 * k + 1] = t[3 * k + 1]; w.f[5 * k + 2] = t[3 * k + 2];
        w.f[5 * k + 3] = t[3 * k + 0]; w.f[5 * k + 4] = t[3 * k + 1];
    }
    w.f[15] = 0.f;
    return w;
}
