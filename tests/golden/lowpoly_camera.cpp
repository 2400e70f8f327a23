// lowpoly_camera.cpp — container-only generator behind tests/golden/make_lowpoly_fixture.py: the Sandbox's camera for its default model, produced by the
// reference's OWN Camera class (compiled from /root/reference/.../Lumen/Renderer/Camera.cpp on the vendored glm; nothing of it is copied here).
// Pose 0 is what Application.cpp:145-146 sets (position (-150, 300, 150), rotation quatLookAtRH(normalize(-1, 0.5, 1), +Y)); poses 1.. replay the input
// handling of OutputLayer.cpp:512-559 with "W held" (300 / 60 = 5 units per frame along W) and a mouse drag with the left button down
// (sensitivity 0.2 degrees per pixel: IncrementYaw(-radians(dx * 0.2)), IncrementPitch(radians(dy * 0.2))), one pose per rendered frame.
// Row per pose: eye(3) right(3) up(3) forward(3) = position and columns 0 / 1 / 2 of the camera matrix (Camera.cpp:128-140).
#include <cstdio>
#include <cmath>
#include "Lumen/Renderer/Camera.h"

int main()
{
    Camera cam;
    cam.SetAspectRatio(1280.f / 720.f);
    cam.SetPosition(glm::vec3{-150.f, 300.f, 150.f});
    cam.SetRotation(glm::quatLookAtRH(glm::normalize(glm::vec3{-1.f, 0.5f, 1.f}), glm::vec3{0.f, 1.f, 0.f}));
    const float sens = 0.2f, speed = 300.f / 60.f;
    for (int k = 0; k < 64; k++) {
        glm::mat4 prev, cur;
        cam.GetMatrixData(prev, cur);
        std::printf("%.9g %.9g %.9g", cur[3][0], cur[3][1], cur[3][2]);
        for (int c = 0; c < 3; c++) std::printf(" %.9g %.9g %.9g", cur[c][0], cur[c][1], cur[c][2]);
        std::printf("\n");
        cam.UpdatePreviousFrameMatrix();
        // next frame's input: mouse drag (dx = 3 px, dy = +-1 px in a slow wave), then W held
        const float dx = 3.f, dy = (k / 8) % 2 ? -1.f : 1.f;
        cam.IncrementYaw(-glm::radians(dx * sens));
        cam.IncrementPitch(glm::radians(dy * sens));
        glm::vec3 eye, U, V, W;
        cam.GetVectorData(eye, U, V, W);
        glm::vec3 dir = glm::normalize(W) * speed;
        cam.SetPosition(eye + glm::normalize(dir) * speed);
    }
    return 0;
}
