// flatten.h — Scene -> the flat arrays of mt_scene_desc (include/mythtracer_hip.h).
// Internal to the facade; exposed to tests through host_capi.cc.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mythtracer_hip.h"
#include "scene.h"

namespace raytracer {

struct FlatScene {
  // triangle streams in node-stream order (FlatTree::tri_id)
  std::vector<double> vertex, normal, uvw, aabb;
  std::vector<int32_t> material, line_no;
  std::vector<mt_material> materials;
  std::vector<mt_texture> textures;            // texels point into the two below / the Scene
  std::vector<std::vector<uint8_t>> rgb8;      // packed 8-bit texels, one per texture (or empty)
  std::string error;

  // Fills everything from a finalized scene.  Returns false (error set) on
  // inconsistent input.
  bool Build(const Scene& scene);
  // A descriptor whose pointers refer to this object and to scene.tree.Flat().
  mt_scene_desc Describe(const Scene& scene, int device) const;
};

}  // namespace raytracer
