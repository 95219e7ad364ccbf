// primitive_triangle.cc — host part of Triangle (reference:
// VerStarting/primitive_triangle.cc:14-24,145-155).  Intersection and
// interpolation live in the HIP kernels (csrc/mt_trace.h, csrc/mt_shade.h).
#include "primitive_triangle.h"

namespace raytracer {

Triangle::~Triangle() {}

AABB Triangle::GetAABB() const { return cached_aabb; }

void Triangle::CacheAABB() {
  AABB box{vertex[0], vertex[0]};
  box.Extend(vertex[1]);
  box.Extend(vertex[2]);
  cached_aabb = box;
}

// Both are unimplemented stubs in the reference as well.
std::string Triangle::Serialize() const { return "nope"; }

bool Triangle::Deserialize(std::unique_ptr<Triangle>*, const std::string&) { return false; }

}  // namespace raytracer
