// octtree.h — the loose octree of the facade (reference:
// VerStarting/octtree.h:14-69, octtree.cc).
//
// Same public interface; different inside.  Finalize() builds the tree straight
// into flat, breadth-first arrays (the layout libmythtracer_hip.so consumes,
// include/mythtracer_hip.h mt_scene_desc) instead of a pointer tree, and
// IntersectRay() runs on the GPU.  The split rule, child boxes, first-fit
// child order and in-node primitive order reproduce octtree.cc:52-135 exactly:
// traversal results depend on them.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
#include "math3d.h"
#include "primitive.h"

struct mt_scene;

namespace raytracer {
using math3d::V3D;

class Triangle;

// Flattened tree + triangle streams (host copy of what lives in HBM).
struct FlatTree {
  int depth = 0;  // root = 1
  std::vector<double> node_aabb;      // 6 per node
  std::vector<double> node_center;    // 3 per node
  std::vector<int32_t> first_child;   // 0 = leaf
  std::vector<int32_t> prim_begin, prim_count;
  std::vector<int32_t> tri_id;        // stream position -> AddPrimitive index
  size_t NodeCount() const { return first_child.size(); }
};

class OctTree {
 public:
  OctTree();
  ~OctTree();
  OctTree(const OctTree&) = delete;
  OctTree& operator=(const OctTree&) = delete;

  // Takes ownership.  Only Triangle primitives are supported; anything else
  // is rejected (deleted, message on stderr).  Not allowed after Finalize().
  void AddPrimitive(Primitive* p);

  // Builds the tree.  Prints "Triangles: N" like the reference.
  void Finalize();
  bool IsFinalized() const { return finalized_; }
  // extension: no "Triangles: N" line on stdout
  void SetQuiet(bool quiet) { quiet_ = quiet; }
  // extension: HIP device of the standalone IntersectRay(s) (MythTracer::SetDevice forwards to it)
  void SetDevice(int hip_device) { device_ = hip_device; }

  // Closest hit of one ray, on the GPU (a batch of one; see IntersectRays for
  // the efficient form).  nullptr when nothing is hit or no GPU is usable (the
  // reason is then available from LastError()).
  const Primitive* IntersectRay(const Ray& ray, V3D* point, V3D::basetype* distance) const;

  // Batch form: rays = n x {origin xyz, direction xyz}.  Outputs may be null.
  // Returns false on a device error.
  bool IntersectRays(int n, const double* rays, const Primitive** prims, double* distances,
                     double* points) const;

  AABB GetAABB() const;

  // --- used by MythTracer / tests
  const FlatTree& Flat() const { return flat_; }
  size_t PrimitiveCount() const { return prims_.size(); }
  const Triangle* GetTriangle(size_t add_index) const;
  const char* LastError() const { return error_.c_str(); }
  static const int SPLIT_BOUNDARY = 16;  // octtree.h:43

 private:
  AABB root_aabb_;  // starts as {0,0,0}-{0,0,0} and only grows (octtree.cc:8-14)
  std::vector<std::unique_ptr<Primitive>> prims_;
  FlatTree flat_;
  bool finalized_ = false;
  bool quiet_ = false;
  int device_ = 0;
  mutable mt_scene* geometry_only_ = nullptr;  // for standalone IntersectRay
  mutable std::string error_;
};

}  // namespace raytracer
