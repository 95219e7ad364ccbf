// objreader.h — Wavefront .obj/.mtl loading (reference:
// VerStarting/objreader.h, objreader.cc).  Same two classes and entry points;
// the parser is a from-scratch rewrite that reproduces the reference's
// observable behaviour, quirks included (see src/objreader.cc).
#pragma once
#include <string>
#include <vector>
#include "math3d.h"
#include "scene.h"

namespace raytracer {
using math3d::V3D;

class ObjFileReader {
 public:
  bool ReadObjFile(Scene* scene, const char* fname);

 private:
  bool Face(const char* line);
  std::string dir_;
  Scene* scene_ = nullptr;
  std::vector<V3D> positions_, texcoords_, normals_;
  Material* current_ = nullptr;
  int line_no_ = 0;
};

class MtlFileReader {
 public:
  bool ReadMtlFile(Scene* scene, const char* fname);

 private:
  void Commit();
  Texture* FindOrLoadTexture(const char* fname);
  std::string dir_;
  Scene* scene_ = nullptr;
  std::unique_ptr<Material> pending_;
  std::string pending_name_;
};

}  // namespace raytracer
