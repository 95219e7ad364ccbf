// ray.h — a ray of the facade (reference: VerStarting/ray.h:12-24).  The
// reciprocal direction the reference caches inside the Ray is a device-side
// detail here (mt_trace.h RayRegs), so the host type is just origin+direction.
#pragma once
#include "math3d.h"

namespace raytracer {
using math3d::V3D;

class Ray {
 public:
  Ray(V3D org, V3D dir) : origin(org), direction(dir) {}
  V3D origin;
  V3D direction;  // expected to be normalised
};

}  // namespace raytracer
