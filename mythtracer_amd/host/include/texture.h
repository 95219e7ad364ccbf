// texture.h — texture of the facade (reference: VerStarting/texture.h:13-25).
// Sampling (Texture::GetColorAt, texture.cc:11-58) runs on the GPU
// (mt_shade.h texture_color_at); the host type only carries the texels.
#pragma once
#include <cstddef>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "math3d.h"

namespace raytracer {
using math3d::V3D;

class Texture {
 public:
  // Dependency-free loaders, chosen by the file's magic bytes: PNG (8-bit,
  // non-interlaced; own inflate), baseline JPEG (libjpeg's default integer
  // arithmetic restated), binary PPM (P6), 24/32-bit BMP, uncompressed
  // true-colour TGA.  (The reference decodes through SDL2_image,
  // texture.cc:60-109, which this build does not link.)  Colour = byte/255.0,
  // texture.cc:100-104.
  static Texture* LoadFromFile(const char* fname);

  size_t width = 0;
  size_t height = 0;
  std::vector<V3D> colors;
};

typedef std::unordered_map<std::string, std::unique_ptr<Texture>> TextureMap;

}  // namespace raytracer
