// aabb.h — axis-aligned box (reference: VerStarting/aabb.h:7-17, aabb.cc).
#pragma once
#include <utility>
#include "math3d.h"

namespace raytracer {
using math3d::V3D;

class AABB {
 public:
  // closed-interval point test, aabb.cc:29-33
  bool Contains(const V3D& point) const;
  // overlap test by centre distance, aabb.cc:9-27 (unused by the hot path)
  bool Contains(const AABB& aabb) const;
  // both corners inside, aabb.cc:5-7 — what the octree split uses
  bool FullyContains(const AABB& aabb) const;
  void Extend(const AABB& aabb);
  void Extend(const V3D& point);
  std::pair<V3D, V3D> GetCenterWHD() const;

  V3D min, max;
};

}  // namespace raytracer
