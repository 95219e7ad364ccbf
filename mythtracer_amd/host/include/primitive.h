// primitive.h — base of scene primitives (reference: VerStarting/primitive.h).
//
// In the reference a Primitive answers IntersectRay/GetNormal/GetUVW through
// virtual calls on the CPU.  Here those three run inside the HIP kernels
// (mt_trace.h, mt_shade.h) on flattened triangle streams, so the host class
// keeps only what the loaders and the octree builder need.
#pragma once
#include <string>
#include "aabb.h"
#include "material.h"
#include "math3d.h"
#include "ray.h"

namespace raytracer {

class Primitive {
 public:
  virtual ~Primitive() {}
  virtual AABB GetAABB() const = 0;
  virtual std::string Serialize() const = 0;

  Material* mtl = nullptr;  // not owned
  int debug_line_no = 0;    // line of the input file that defined it
};

}  // namespace raytracer
