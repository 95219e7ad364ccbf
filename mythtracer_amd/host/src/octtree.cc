// octtree.cc — octree construction on the host, traversal on the GPU
// (reference: VerStarting/octtree.cc).
#include "octtree.h"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "mythtracer_hip.h"
#include "primitive_triangle.h"

namespace raytracer {

OctTree::OctTree() {}

OctTree::~OctTree() {
  if (geometry_only_) mt_scene_destroy(geometry_only_);
}

void OctTree::AddPrimitive(Primitive* p) {
  if (dynamic_cast<Triangle*>(p) == nullptr) {
    fprintf(stderr, "error: OctTree::AddPrimitive: only Triangle primitives are supported\n");
    delete p;
    return;
  }
  if (finalized_) {
    fprintf(stderr, "error: OctTree::AddPrimitive after Finalize is not allowed\n");
    delete p;
    return;
  }
  prims_.emplace_back(p);
  root_aabb_.Extend(p->GetAABB());  // octtree.cc:12-13
}

AABB OctTree::GetAABB() const { return root_aabb_; }

const Triangle* OctTree::GetTriangle(size_t i) const {
  return static_cast<const Triangle*>(prims_[i].get());
}

namespace {

// One node while the tree is being built: its box and its (ordered) members.
struct BuildNode {
  AABB box;
  V3D center;
  std::vector<int32_t> members;
  int32_t first_child = 0;
  int depth = 1;
};

// The eight octants of `box` around `c`, in the reference's numbering
// (octtree.cc:61-100): bit 0 = upper x half, bit 1 = upper z half, bit 2 =
// upper y half.
AABB Octant(const AABB& box, const V3D& c, int k) {
  const bool xh = k & 1, zh = k & 2, yh = k & 4;
  AABB o;
  o.min = {xh ? c.v[0] : box.min.v[0], yh ? c.v[1] : box.min.v[1], zh ? c.v[2] : box.min.v[2]};
  o.max = {xh ? box.max.v[0] : c.v[0], yh ? box.max.v[1] : c.v[1], zh ? box.max.v[2] : c.v[2]};
  return o;
}

const int kBuilderDepthLimit = 4096;

}  // namespace

void OctTree::Finalize() {
  if (finalized_) return;
  if (!quiet_) printf("Triangles: %u\n", (unsigned int)prims_.size());

  // Breadth-first construction.  A node's split depends only on its own box
  // and member list, so the result equals the reference's depth-first
  // AttemptSplit recursion (octtree.cc:52-135), node for node.
  std::vector<BuildNode> nodes(1);
  nodes[0].box = root_aabb_;
  nodes[0].members.resize(prims_.size());
  for (size_t i = 0; i < prims_.size(); i++) nodes[0].members[i] = (int32_t)i;
  int depth = 1;
  for (size_t at = 0; at < nodes.size(); at++) {
    if ((int)nodes[at].members.size() < SPLIT_BOUNDARY) continue;
    if (nodes[at].depth >= kBuilderDepthLimit) {
      // The reference would keep recursing (and overflow its stack) on >= 16
      // primitives that never separate; stop splitting instead.
      fprintf(stderr, "warning: octree depth limit %d reached, node left unsplit\n", kBuilderDepthLimit);
      continue;
    }
    const AABB box = nodes[at].box;
    V3D c;
    for (int axis = 0; axis < 3; axis++) {  // Node::CalcCenter, octtree.cc:46-50
      c.v[axis] = box.min.v[axis] + (box.max.v[axis] - box.min.v[axis]) / 2.0;
    }
    const int32_t first = (int32_t)nodes.size();
    const int child_depth = nodes[at].depth + 1;
    nodes.resize(nodes.size() + 8);  // may move nodes[at]
    for (int k = 0; k < 8; k++) {
      nodes[first + k].box = Octant(box, c, k);
      nodes[first + k].depth = child_depth;
    }
    std::vector<int32_t> stay;
    for (int32_t id : nodes[at].members) {
      const AABB pb = prims_[id]->GetAABB();
      int home = -1;
      for (int k = 0; k < 8 && home < 0; k++) {  // first child that fully contains it
        if (nodes[first + k].box.FullyContains(pb)) home = k;
      }
      if (home >= 0) nodes[first + home].members.push_back(id);
      else stay.push_back(id);  // straddlers stay, order preserved
    }
    nodes[at].members.swap(stay);
    nodes[at].center = c;
    nodes[at].first_child = first;
    if (child_depth > depth) depth = child_depth;
  }

  // Flatten: nodes in BFS order, triangles as one stream, node after node.
  FlatTree& f = flat_;
  const size_t n = nodes.size();
  f.depth = depth;
  f.node_aabb.resize(n * 6);
  f.node_center.resize(n * 3);
  f.first_child.resize(n);
  f.prim_begin.resize(n);
  f.prim_count.resize(n);
  f.tri_id.clear();
  f.tri_id.reserve(prims_.size());
  for (size_t i = 0; i < n; i++) {
    memcpy(&f.node_aabb[i * 6], nodes[i].box.min.v, 24);
    memcpy(&f.node_aabb[i * 6 + 3], nodes[i].box.max.v, 24);
    memcpy(&f.node_center[i * 3], nodes[i].center.v, 24);
    f.first_child[i] = nodes[i].first_child;
    f.prim_begin[i] = (int32_t)f.tri_id.size();
    f.prim_count[i] = (int32_t)nodes[i].members.size();
    f.tri_id.insert(f.tri_id.end(), nodes[i].members.begin(), nodes[i].members.end());
  }
  finalized_ = true;
}

bool OctTree::IntersectRays(int n, const double* rays, const Primitive** prims, double* distances,
                            double* points) const {
  if (!finalized_) {
    error_ = "OctTree::IntersectRays before Finalize";
    return false;
  }
  if (geometry_only_ == nullptr) {
    // geometry-only upload: materials are irrelevant to closest-hit queries
    const FlatTree& f = flat_;
    const size_t nt = f.tri_id.size();
    std::vector<double> vtx(nt * 9), nrm(nt * 9, 0.0), uvw(nt * 9, 0.0), box(nt * 6);
    std::vector<int32_t> mtl(nt, -1), line(nt);
    for (size_t s = 0; s < nt; s++) {
      const Triangle* t = GetTriangle((size_t)f.tri_id[s]);
      memcpy(&vtx[s * 9], t->vertex, 72);
      memcpy(&box[s * 6], t->cached_aabb.min.v, 24);
      memcpy(&box[s * 6 + 3], t->cached_aabb.max.v, 24);
      line[s] = t->debug_line_no;
    }
    mt_scene_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = sizeof d;
    d.abi_version = MT_ABI_VERSION;
    d.device = device_;
    d.n_nodes = (int32_t)f.NodeCount();
    d.n_tris = (int32_t)nt;
    d.tree_depth = f.depth;
    d.node_aabb = f.node_aabb.data();
    d.node_center = f.node_center.data();
    d.node_first_child = f.first_child.data();
    d.node_prim_begin = f.prim_begin.data();
    d.node_prim_count = f.prim_count.data();
    d.tri_vertex = vtx.data();
    d.tri_normal = nrm.data();
    d.tri_uvw = uvw.data();
    d.tri_aabb = box.data();
    d.tri_material = mtl.data();
    d.tri_line_no = line.data();
    d.tri_id = f.tri_id.data();
    geometry_only_ = mt_scene_create(&d);
    if (geometry_only_ == nullptr) {
      error_ = mt_last_error();
      return false;
    }
  }
  std::vector<int32_t> hit((size_t)n);
  if (mt_intersect_rays(geometry_only_, n, rays, hit.data(), nullptr, distances, points, nullptr) != MT_OK) {
    error_ = mt_last_error();
    return false;
  }
  if (prims) {
    for (int i = 0; i < n; i++) {
      prims[i] = hit[i] < 0 ? nullptr : prims_[(size_t)flat_.tri_id[(size_t)hit[i]]].get();
    }
  }
  return true;
}

const Primitive* OctTree::IntersectRay(const Ray& ray, V3D* point, V3D::basetype* distance) const {
  const double r[6] = {ray.origin.v[0],    ray.origin.v[1],    ray.origin.v[2],
                       ray.direction.v[0], ray.direction.v[1], ray.direction.v[2]};
  const Primitive* p = nullptr;
  double t, pt[3];
  if (!IntersectRays(1, r, &p, &t, pt)) {
    fprintf(stderr, "error: OctTree::IntersectRay: %s\n", error_.c_str());
    return nullptr;
  }
  if (p == nullptr) return nullptr;  // outputs untouched on a miss, like octtree.cc:250-252
  *point = {pt[0], pt[1], pt[2]};
  *distance = t;
  return p;
}

}  // namespace raytracer
