// gen_kat3.cpp — known-answer generator, third translation unit: the reference's CAMERA, compiled from its own source file.
// Built by make_kat.py together with /root/reference/Lumen_Engine/Lumen/src/Lumen/Renderer/Camera.cpp (plain C++ on the vendored glm),
// plus the vendored sutil/Matrix.h for the matrix the motion-vector pass receives.  Rows:
//   cam  position(3) quaternion wxyz(4) aspect | right(3) up(3) forward(3) | eye(3) U(3) V(3) W(3)            Camera::GetVectorData (Camera.cpp:79-93)
//   mvm  previous camera world matrix, row major(16) aspect | M(16), row major                                 = projection * inverse(previous)
//        as WaveFrontRenderer.cpp:763-776 hands it to GenerateMotionVectors (CPUShadingKernels.cu:39); sutil stores row major, glm column major
// Container-only; contains no reference source text — it only #includes / links it from /root/reference.
#include <cstdio>
#include <random>
#include <cuda_runtime.h>
#include <sutil/vec_math.h>
#include <sutil/Matrix.h>
#include "Lumen/Renderer/Camera.h"

static sutil::Matrix4x4 toSutil(const glm::mat4& m)      // glm is column major, sutil row major: element (row, column)
{
    float d[16];
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) d[r * 4 + c] = m[c][r];
    return sutil::Matrix4x4(d);
}

int main()
{
    std::mt19937 rng(20240607u);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    auto quat = [&]() { glm::quat q(U(rng), U(rng), U(rng), U(rng)); return glm::normalize(q); };
    for (int i = 0; i < 400; i++) {
        const glm::vec3 p0(30.f * U(rng), 10.f * U(rng), 30.f * U(rng)), p1 = p0 + glm::vec3(0.3f * U(rng), 0.3f * U(rng), 0.3f * U(rng));
        const glm::quat q0 = quat(), q1 = (i % 3 == 0) ? q0 : glm::normalize(glm::quat(q0.w + 0.02f * U(rng), q0.x + 0.02f * U(rng), q0.y + 0.02f * U(rng), q0.z + 0.02f * U(rng)));
        const float aspect = i % 4 == 0 ? 16.f / 9.f : 0.4f + 2.2f * (0.5f + 0.5f * U(rng));
        Camera cam;
        cam.SetRotation(q0); cam.SetPosition(p0); cam.SetAspectRatio(aspect);
        glm::vec3 eye, u, v, w;
        cam.GetVectorData(eye, u, v, w);
        glm::mat4 prev, cur;
        cam.GetMatrixData(prev, cur);
        std::printf("cam %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g", p0.x, p0.y, p0.z, q0.w, q0.x, q0.y, q0.z, aspect);
        for (int c = 0; c < 3; c++) std::printf(" %.9g %.9g %.9g", cur[c][0], cur[c][1], cur[c][2]);
        std::printf(" %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", eye.x, eye.y, eye.z, u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z);
        // the frame ends (previous = current), the camera moves, the next frame asks for both matrices
        cam.UpdatePreviousFrameMatrix();
        cam.SetRotation(q1); cam.SetPosition(p1);
        cam.GetMatrixData(prev, cur);
        const sutil::Matrix4x4 M = toSutil(cam.GetProjectionMatrix()) * toSutil(prev).inverse();
        std::printf("mvm");
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) std::printf(" %.9g", prev[c][r]);
        std::printf(" %.9g", aspect);
        for (int k = 0; k < 16; k++) std::printf(" %.9g", M.getData()[k]);
        std::printf("\n");
    }
    return 0;
}
