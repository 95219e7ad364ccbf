// primitive_triangle.h — the only primitive kind (reference:
// VerStarting/primitive_triangle.h:10-30).
#pragma once
#include <memory>
#include <string>
#include "primitive.h"

namespace raytracer {

class Triangle : public Primitive {
 public:
  ~Triangle() override;
  AABB GetAABB() const override;  // the cached box
  std::string Serialize() const override;
  static bool Deserialize(std::unique_ptr<Triangle>* primitive, const std::string& data);

  // Must be called after the vertices are set and before AddPrimitive
  // (objreader.cc:182-183 does; octtree_test.cc forgets to).
  void CacheAABB();

  V3D vertex[3]{};
  V3D normal[3]{};
  V3D uvw[3]{};
  AABB cached_aabb;
};

}  // namespace raytracer
