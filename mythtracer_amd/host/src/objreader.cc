// objreader.cc — Wavefront .obj / .mtl loading for the facade.
//
// A from-scratch parser whose *observable behaviour* matches the reference's
// ObjFileReader / MtlFileReader (VerStarting/objreader.cc), because the same
// file has to yield the same triangles, in the same order, with the same
// line numbers.  The behaviours that matter (all deliberate):
//   * input is consumed in chunks of at most 127 characters (fgets into a
//     128-byte buffer, objreader.cc:234-235); a longer line continues as a new
//     "line", and every chunk counts for debug_line_no (0-based);
//   * the text after the LAST '\r' and after the last '\n' is cut
//     (objreader.cc:239-247);
//   * the keyword is the first whitespace-delimited word; '#' starts a comment;
//     unknown keywords are skipped with a warning; a malformed known statement
//     aborts the load;
//   * statement bodies are matched with scanf semantics from the first
//     character of the chunk ("v %lf %lf %lf"): leading blanks before the
//     keyword make OBJ statements fail, while MTL statements tolerate them
//     (their patterns start with a blank), except newmtl;
//   * a face index group is only accepted when followed by whitespace: the
//     reference reads groups with `stream >> token` and stops at end-of-stream
//     BEFORE using the token (objreader.cc:111-115), so "f 1 2 3" has two
//     usable groups (-> load fails) and "f 1 2 3 " has three;
//   * index groups are tried as v/vt/vn, v//vn, v/vt, v with %i conversions
//     (hex/octal accepted) sharing their output variables across attempts;
//   * 3 or 4 groups only; a quad becomes (0,1,2) and (2,3,0);
//   * normals / texcoords are taken only if all three corners have an index.
// Out-of-range indices are undefined behaviour in the reference; here they
// abort the load with a message.
#include "objreader.h"

#include <cstdio>
#include <cstring>
#include <memory>

#include "primitive_triangle.h"

namespace raytracer {
namespace {

struct Closer {
  void operator()(FILE* f) const { fclose(f); }
};
using File = std::unique_ptr<FILE, Closer>;

std::string DirName(const std::string& path) {
  const size_t cut = path.find_last_of("/\\");
  return cut == std::string::npos ? std::string() : path.substr(0, cut);
}

std::string Join(const std::string& dir, const char* name) {
  return dir.empty() ? std::string(name) : dir + "/" + name;
}

// Reads the next chunk (<= 127 chars), trims it the way the reference does and
// extracts the keyword.  Returns false at end of file.
bool NextChunk(FILE* f, char (&chunk)[128], char (&keyword)[16]) {
  if (fgets(chunk, sizeof chunk, f) == nullptr) return false;
  if (char* cr = strrchr(chunk, '\r')) *cr = '\0';
  if (char* nl = strrchr(chunk, '\n')) *nl = '\0';
  memset(keyword, 0, sizeof keyword);
  if (sscanf(chunk, "%15s", keyword) != 1) keyword[0] = '\0';
  return true;
}

bool IsBlank(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

bool Triple(const char* chunk, const char* pattern, V3D* out) {
  double a, b, c;
  if (sscanf(chunk, pattern, &a, &b, &c) != 3) return false;
  *out = {a, b, c};
  return true;
}

bool Single(const char* chunk, const char* pattern, double* out) {
  return sscanf(chunk, pattern, out) == 1;
}

}  // namespace

// ---------------------------------------------------------------- .obj

bool ObjFileReader::Face(const char* line) {
  int vi[5], ti[5], ni[5];
  size_t groups = 0;

  const char* p = line;
  while (IsBlank(*p)) p++;
  while (*p && !IsBlank(*p)) p++;  // the "f" itself
  // A stream that reached its end while reading "f" is no longer good().
  bool stream_good = (*p != '\0');
  while (stream_good) {
    while (IsBlank(*p)) p++;
    if (*p == '\0') break;  // nothing left: extraction fails at end-of-stream
    const char* begin = p;
    while (*p && !IsBlank(*p)) p++;
    if (*p == '\0') break;  // group touched end-of-stream: dropped (the quirk)
    const std::string group(begin, p);

    int v = 0, vt = 0, vn = 0;
    const char* g = group.c_str();
    if (sscanf(g, "%i/%i/%i", &v, &vt, &vn) != 3 && sscanf(g, "%i//%i", &v, &vn) != 2 &&
        sscanf(g, "%i/%i", &v, &vt) != 2 && sscanf(g, "%i", &v) != 1) {
      fprintf(stderr, "warning: unsupported face format \"%s\"\n", g);
      return false;
    }
    if (groups < 5) {
      vi[groups] = v - 1;  // file indices are 1-based
      ti[groups] = vt - 1;
      ni[groups] = vn - 1;
    }
    groups++;
  }

  if (groups != 3 && groups != 4) {
    fprintf(stderr, "warning: unsupported face count (%i)\n  %s\n", (int)groups, line);
    return false;
  }
  if (groups == 4) {  // close the quad: corners 0 1 2 3 0
    vi[4] = vi[0];
    ti[4] = ti[0];
    ni[4] = ni[0];
  }
  const size_t corners = (groups == 4) ? 5 : 3;
  for (size_t first = 0; first + 3 <= corners; first += 2) {  // (0,1,2) then (2,3,0)
    std::unique_ptr<Triangle> tr(new Triangle());
    for (size_t k = 0; k < 3; k++) {
      const int idx = vi[first + k];
      if (idx < 0 || (size_t)idx >= positions_.size()) {
        fprintf(stderr, "error: face vertex index %i out of range (line %i)\n", idx + 1, line_no_);
        return false;
      }
      tr->vertex[k] = positions_[(size_t)idx];
    }
    if (ni[first] != -1 && ni[first + 1] != -1 && ni[first + 2] != -1) {
      for (size_t k = 0; k < 3; k++) {
        const int idx = ni[first + k];
        if (idx < 0 || (size_t)idx >= normals_.size()) {
          fprintf(stderr, "error: face normal index %i out of range (line %i)\n", idx + 1, line_no_);
          return false;
        }
        tr->normal[k] = normals_[(size_t)idx];
      }
    }
    if (ti[first] != -1 && ti[first + 1] != -1 && ti[first + 2] != -1) {
      for (size_t k = 0; k < 3; k++) {
        const int idx = ti[first + k];
        if (idx < 0 || (size_t)idx >= texcoords_.size()) {
          fprintf(stderr, "error: face texcoord index %i out of range (line %i)\n", idx + 1, line_no_);
          return false;
        }
        tr->uvw[k] = texcoords_[(size_t)idx];
      }
    }
    tr->mtl = current_;
    tr->debug_line_no = line_no_;
    tr->CacheAABB();
    scene_->tree.AddPrimitive(tr.release());
  }
  return true;
}

bool ObjFileReader::ReadObjFile(Scene* scene, const char* fname) {
  positions_.clear();
  texcoords_.clear();
  normals_.clear();
  scene_ = scene;
  dir_ = DirName(fname);
  current_ = nullptr;

  File f(fopen(fname, "r"));
  if (!f) {
    fprintf(stderr, "error: file \"%s\" not found\n", fname);
    return false;
  }
  char chunk[128], key[16];
  for (line_no_ = 0; NextChunk(f.get(), chunk, key); line_no_++) {
    if (key[0] == '\0' || key[0] == '#') continue;
    bool ok = true;
    V3D val;
    if (!strcmp(key, "v")) {
      if ((ok = Triple(chunk, "v %lf %lf %lf", &val))) positions_.push_back(val);
      else fprintf(stderr, "warning: unsupported vertex format \"%s\"\n", chunk);
    } else if (!strcmp(key, "vn")) {
      if ((ok = Triple(chunk, "vn %lf %lf %lf", &val))) normals_.push_back(val);
      else fprintf(stderr, "warning: unsupported normal format \"%s\"\n", chunk);
    } else if (!strcmp(key, "vt")) {
      double u, v, w = 0.0;  // w is optional
      if ((ok = sscanf(chunk, "vt %lf %lf %lf", &u, &v, &w) >= 2)) texcoords_.push_back({u, v, w});
      else fprintf(stderr, "warning: unsupported texcoord format \"%s\"\n", chunk);
    } else if (!strcmp(key, "f")) {
      ok = Face(chunk);
    } else if (!strcmp(key, "mtllib")) {
      char name[256];
      if ((ok = sscanf(chunk, "mtllib %255[^\n]", name) == 1)) {
        MtlFileReader mtl;
        ok = mtl.ReadMtlFile(scene_, Join(dir_, name).c_str());
      } else {
        fprintf(stderr, "warning: unsupported mtllib format \"%s\"\n", chunk);
      }
    } else if (!strcmp(key, "usemtl")) {
      char name[128];
      if ((ok = sscanf(chunk, "usemtl %127s", name) == 1)) {
        auto it = scene_->materials.find(name);
        if (it == scene_->materials.end()) {
          fprintf(stderr, "warning: material \"%s\" not found\n", name);
          current_ = nullptr;  // keep parsing; following faces have no material
        } else {
          current_ = it->second.get();
        }
      } else {
        fprintf(stderr, "warning: unsupported usemtl format\n");
      }
    } else if (!strcmp(key, "s") || !strcmp(key, "g") || !strcmp(key, "o")) {
      // smoothing groups, groups, objects: accepted and ignored
    } else {
      fprintf(stderr, "warning: unknown OBJ feature \"%s\"\n", key);
    }
    if (!ok) return false;
  }
  return true;
}

// ---------------------------------------------------------------- .mtl

void MtlFileReader::Commit() {
  if (pending_ == nullptr) return;
  // The reference assigns a fresh object into the map (objreader.cc:279), which
  // frees a previous material of the same name while triangles may still point
  // at it.  Overwriting in place keeps those pointers valid.
  auto it = scene_->materials.find(pending_name_);
  if (it != scene_->materials.end()) {
    *it->second = *pending_;
    pending_.reset();
  } else {
    scene_->materials[pending_name_] = std::move(pending_);
  }
  pending_name_.clear();
}

Texture* MtlFileReader::FindOrLoadTexture(const char* fname) {
  auto it = scene_->textures.find(fname);
  if (it != scene_->textures.end()) return it->second.get();
  Texture* tex = Texture::LoadFromFile(Join(dir_, fname).c_str());
  if (tex == nullptr) {
    fprintf(stderr, "error: cannot load texture \"%s\"\n", fname);
    return nullptr;
  }
  scene_->textures[fname].reset(tex);
  return tex;
}

bool MtlFileReader::ReadMtlFile(Scene* scene, const char* fname) {
  pending_.reset();
  pending_name_.clear();
  scene_ = scene;
  dir_ = DirName(fname);

  File f(fopen(fname, "r"));
  if (!f) {
    fprintf(stderr, "error: file \"%s\" not found\n", fname);
    return false;
  }
  // statements that set one field of the material being defined
  struct ColorField { const char* key; const char* pattern; V3D Material::*field; };
  struct ScalarField { const char* key; const char* pattern; double Material::*field; };
  static const ColorField kColors[] = {
      {"Ka", " Ka %lf %lf %lf", &Material::ambient},
      {"Kd", " Kd %lf %lf %lf", &Material::diffuse},
      {"Ks", " Ks %lf %lf %lf", &Material::specular},
      {"Tf", " Tf %lf %lf %lf", &Material::transmission_filter},
  };
  static const ScalarField kScalars[] = {
      {"Ns", " Ns %lf", &Material::specular_exp},
      {"Ni", " Ni %lf", &Material::refraction_index},
      {"Tr", " Tr %lf", &Material::transparency},
      {"Refl", " Refl %lf", &Material::reflectance},
  };
  static const char* const kIgnored[] = {"d", "illum", "Ke", "map_Kd"};

  char chunk[128], key[16];
  while (NextChunk(f.get(), chunk, key)) {
    if (key[0] == '\0' || key[0] == '#') continue;
    if (!strcmp(key, "newmtl")) {
      Commit();
      char name[128];
      if (sscanf(chunk, "newmtl %127s", name) != 1) {
        fprintf(stderr, "warning: unsupported newmtl format\n");
        return false;
      }
      pending_.reset(new Material);
      pending_name_ = name;
      continue;
    }
    bool handled = false;
    for (const ColorField& c : kColors) {
      if (strcmp(key, c.key)) continue;
      handled = true;
      if (pending_ == nullptr) {
        fprintf(stderr, "warning: material not ready; missing newmtl\n");
        return false;
      }
      if (!Triple(chunk, c.pattern, &((*pending_).*(c.field)))) {
        fprintf(stderr, "warning: unsupported %s format \"%s\"\n", c.key, chunk);
        return false;
      }
    }
    for (const ScalarField& s : kScalars) {
      if (strcmp(key, s.key)) continue;
      handled = true;
      if (pending_ == nullptr) {
        fprintf(stderr, "warning: material not ready; missing newmtl\n");
        return false;
      }
      if (!Single(chunk, s.pattern, &((*pending_).*(s.field)))) {
        fprintf(stderr, "warning: unsupported %s format \"%s\"\n", s.key, chunk);
        return false;
      }
    }
    if (!strcmp(key, "map_Ka")) {
      handled = true;
      if (pending_ == nullptr) {
        fprintf(stderr, "warning: material not ready; missing newmtl\n");
        return false;
      }
      char name[256];
      if (sscanf(chunk, " map_Ka %255[^\n]", name) != 1) {
        fprintf(stderr, "warning: unsupported map_ka format \"%s\"\n", chunk);
        return false;
      }
      pending_->tex = FindOrLoadTexture(name);
      if (pending_->tex == nullptr) return false;  // a missing texture fails the .mtl
    }
    for (const char* ig : kIgnored) {
      if (!strcmp(key, ig)) handled = true;
    }
    if (!handled) fprintf(stderr, "warning: unknown MTL feature \"%s\"\n", key);
  }
  Commit();
  return true;
}

}  // namespace raytracer
