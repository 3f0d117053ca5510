// HostOps.h — host-only operations of the reference's app shell that survive as API:
//   moveCamera      CudaTracer/CudaTracer.cu:822-870 (WASDQE translate by 0.2, TFGH rotate by 10 degrees)
//   saveScreenshot  CudaTracer/CudaTracer.cu:795-813 (TGA type 2, 24-bit BGR, bottom-up) — here fed
//                   from a host copy of the RGBA display buffer instead of glReadPixels.
#pragma once
#include "RenderStructs.h"

bool moveCamera(Camera& camera, unsigned char key);
bool writeTga(const char* filename, const ptss_uchar4* rgba, int width, int height);
// glm::quat(vec3 eulerAngles) — pitch (x), yaw (y), roll (z), radians.
quat quatFromEuler(vec3 euler);
