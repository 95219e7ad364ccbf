// aabb.cc — reference: VerStarting/aabb.cc.  std::min/std::max keep the
// reference's operand order (it decides what happens with NaN coordinates).
#include "aabb.h"

#include <algorithm>
#include <cmath>

namespace raytracer {

bool AABB::Contains(const V3D& p) const {
  for (int axis = 0; axis < 3; axis++) {
    if (!(p.v[axis] >= min.v[axis] && p.v[axis] <= max.v[axis])) return false;
  }
  return true;
}

bool AABB::FullyContains(const AABB& other) const {
  return Contains(other.min) && Contains(other.max);
}

bool AABB::Contains(const AABB& other) const {
  const auto mine = GetCenterWHD();
  const auto theirs = other.GetCenterWHD();
  for (int axis = 0; axis < 3; axis++) {
    const double gap = std::fabs(mine.first.v[axis] - theirs.first.v[axis]) * 2.0;
    if (!(gap <= mine.second.v[axis] + theirs.second.v[axis])) return false;
  }
  return true;
}

void AABB::Extend(const V3D& p) {
  for (int axis = 0; axis < 3; axis++) {
    min.v[axis] = std::min(min.v[axis], p.v[axis]);
    max.v[axis] = std::max(max.v[axis], p.v[axis]);
  }
}

void AABB::Extend(const AABB& other) {
  for (int axis = 0; axis < 3; axis++) {
    min.v[axis] = std::min(min.v[axis], other.min.v[axis]);
    max.v[axis] = std::max(max.v[axis], other.max.v[axis]);
  }
}

std::pair<V3D, V3D> AABB::GetCenterWHD() const {
  const V3D extent = max - min;
  return {min + extent / 2, extent};
}

}  // namespace raytracer
