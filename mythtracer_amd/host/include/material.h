// material.h — surface description (reference: VerStarting/material.h:12-50).
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include "math3d.h"
#include "texture.h"

namespace raytracer {
using math3d::V3D;

class Material {
 public:
  V3D ambient{}, diffuse{}, specular{};   // MTL Ka, Kd, Ks
  Texture* tex = nullptr;                 // MTL map_Ka; not owned
  V3D::basetype specular_exp = 0.0;       // MTL Ns
  V3D::basetype reflectance = 0.0;        // MTL Refl (non-standard)
  V3D::basetype transparency = 0.0;       // MTL Tr
  V3D transmission_filter{};              // MTL Tf
  V3D::basetype refraction_index = 0.0;   // MTL Ni
};

typedef std::unordered_map<std::string, std::unique_ptr<Material>> MaterialMap;

}  // namespace raytracer
