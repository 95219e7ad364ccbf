// scene.h — reference: VerStarting/scene.h:9-15.
#pragma once
#include <vector>
#include "light.h"
#include "material.h"
#include "octtree.h"

namespace raytracer {

class Scene {
 public:
  OctTree tree;
  MaterialMap materials;
  TextureMap textures;
  std::vector<Light> lights;
};

}  // namespace raytracer
