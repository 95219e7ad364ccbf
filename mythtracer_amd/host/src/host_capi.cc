// host_capi.cc — a thin extern "C" shim over the C++ facade so that Python
// (ctypes) can drive it: tests, bench.py and __graft_entry__.py.  It adds no
// behaviour of its own; every call forwards to the raytracer:: classes.
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "flatten.h"
#include "mythtracer.h"
#include "mythtracer_hip.h"
#include "primitive_triangle.h"

using raytracer::Camera;
using raytracer::Light;
using raytracer::Material;
using raytracer::MythTracer;
using raytracer::PerPixelDebugInfo;
using raytracer::Texture;
using raytracer::Triangle;
using raytracer::WorkChunk;
using math3d::V3D;

namespace {

struct Handle {
  MythTracer mt;
  std::vector<Material*> mtl_by_index;   // programmatic materials
  std::vector<Texture*> tex_by_index;
  std::string error;
};

Camera MakeCamera(const double* c) { return Camera{{c[0], c[1], c[2]}, c[3], c[4], c[5], c[6]}; }

}  // namespace

extern "C" {

void* mth_new(int device, int quiet) {
  Handle* h = new Handle;
  h->mt.SetDevice(device);
  h->mt.SetQuiet(quiet != 0);
  return h;
}

void mth_free(void* p) { delete static_cast<Handle*>(p); }

void mth_set_devices(void* p, const int* devices, int n) {
  static_cast<Handle*>(p)->mt.SetDevices(std::vector<int>(devices, devices + n));
}

const char* mth_last_error(void* p) {
  Handle* h = static_cast<Handle*>(p);
  h->error = h->mt.LastError();
  return h->error.c_str();
}

int mth_load_obj(void* p, const char* path) { return static_cast<Handle*>(p)->mt.LoadObj(path) ? 1 : 0; }

int mth_add_material(void* p, const char* name, const double* ka, const double* kd, const double* ks,
                     double ns, double refl, double tr, const double* tf, double ni) {
  Handle* h = static_cast<Handle*>(p);
  std::unique_ptr<Material> m(new Material);
  m->ambient = {ka[0], ka[1], ka[2]};
  m->diffuse = {kd[0], kd[1], kd[2]};
  m->specular = {ks[0], ks[1], ks[2]};
  m->transmission_filter = {tf[0], tf[1], tf[2]};
  m->specular_exp = ns;
  m->reflectance = refl;
  m->transparency = tr;
  m->refraction_index = ni;
  h->mtl_by_index.push_back(m.get());
  h->mt.GetScene()->materials[name] = std::move(m);
  return (int)h->mtl_by_index.size() - 1;
}

int mth_add_texture(void* p, const char* name, int w, int hgt, const double* rgb) {
  Handle* h = static_cast<Handle*>(p);
  std::unique_ptr<Texture> t(new Texture);
  t->width = (size_t)w;
  t->height = (size_t)hgt;
  t->colors.resize((size_t)w * hgt);
  for (size_t i = 0; i < t->colors.size(); i++) t->colors[i] = {rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]};
  h->tex_by_index.push_back(t.get());
  h->mt.GetScene()->textures[name] = std::move(t);
  return (int)h->tex_by_index.size() - 1;
}

int mth_material_set_texture(void* p, int mtl, int tex) {
  Handle* h = static_cast<Handle*>(p);
  if (mtl < 0 || (size_t)mtl >= h->mtl_by_index.size() || tex < -1 || tex >= (int)h->tex_by_index.size()) return 0;
  h->mtl_by_index[(size_t)mtl]->tex = tex < 0 ? nullptr : h->tex_by_index[(size_t)tex];
  return 1;
}

int mth_add_triangle(void* p, const double* v, const double* n, const double* uvw, int mtl, int line_no) {
  Handle* h = static_cast<Handle*>(p);
  Triangle* t = new Triangle();
  for (int k = 0; k < 3; k++) {
    t->vertex[k] = {v[k * 3], v[k * 3 + 1], v[k * 3 + 2]};
    if (n) t->normal[k] = {n[k * 3], n[k * 3 + 1], n[k * 3 + 2]};
    if (uvw) t->uvw[k] = {uvw[k * 3], uvw[k * 3 + 1], uvw[k * 3 + 2]};
  }
  t->mtl = (mtl >= 0 && (size_t)mtl < h->mtl_by_index.size()) ? h->mtl_by_index[(size_t)mtl] : nullptr;
  t->debug_line_no = line_no;
  t->CacheAABB();
  h->mt.GetScene()->tree.AddPrimitive(t);
  return (int)h->mt.GetScene()->tree.PrimitiveCount() - 1;
}

void mth_set_lights(void* p, const double* l, int n) {
  auto& lights = static_cast<Handle*>(p)->mt.GetScene()->lights;
  lights.clear();
  for (int i = 0; i < n; i++) {
    const double* q = l + 12 * i;
    lights.push_back(Light{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}, {q[6], q[7], q[8]}, {q[9], q[10], q[11]}});
  }
}

void mth_set_max_level(void* p, int level) { static_cast<Handle*>(p)->mt.SetMaxRecursionLevel(level); }

// host-only: builds the octree (no GPU needed)
void mth_finalize(void* p) { static_cast<Handle*>(p)->mt.GetScene()->tree.Finalize(); }

// finalize + upload to the GPU
int mth_prepare(void* p) { return static_cast<Handle*>(p)->mt.Prepare() ? 1 : 0; }

void* mth_device_scene(void* p) { return static_cast<Handle*>(p)->mt.DeviceScene(); }

void mth_root_aabb(void* p, double* out6) {
  const raytracer::AABB b = static_cast<Handle*>(p)->mt.GetScene()->tree.GetAABB();
  memcpy(out6, b.min.v, 24);
  memcpy(out6 + 3, b.max.v, 24);
}

void mth_tree_info(void* p, int* n_nodes, int* n_tris, int* depth) {
  const auto& tree = static_cast<Handle*>(p)->mt.GetScene()->tree;
  *n_nodes = (int)tree.Flat().NodeCount();
  *n_tris = (int)tree.PrimitiveCount();
  *depth = tree.Flat().depth;
}

void mth_tree_dump(void* p, double* aabb, double* center, int32_t* first_child, int32_t* prim_begin,
                   int32_t* prim_count, int32_t* prim_ids) {
  const raytracer::FlatTree& f = static_cast<Handle*>(p)->mt.GetScene()->tree.Flat();
  memcpy(aabb, f.node_aabb.data(), f.node_aabb.size() * 8);
  memcpy(center, f.node_center.data(), f.node_center.size() * 8);
  memcpy(first_child, f.first_child.data(), f.first_child.size() * 4);
  memcpy(prim_begin, f.prim_begin.data(), f.prim_begin.size() * 4);
  memcpy(prim_count, f.prim_count.data(), f.prim_count.size() * 4);
  memcpy(prim_ids, f.tri_id.data(), f.tri_id.size() * 4);
}

// vertex(9) normal(9) uvw(9) aabb(6) per triangle in AddPrimitive order;
// mtl_name_hash is not exported: material identity is checked through renders.
void mth_triangles(void* p, double* out33, int32_t* line_no, int32_t* has_mtl) {
  const auto& tree = static_cast<Handle*>(p)->mt.GetScene()->tree;
  for (size_t i = 0; i < tree.PrimitiveCount(); i++) {
    const Triangle* t = tree.GetTriangle(i);
    double* o = out33 + i * 33;
    memcpy(o, t->vertex, 72);
    memcpy(o + 9, t->normal, 72);
    memcpy(o + 18, t->uvw, 72);
    memcpy(o + 27, t->cached_aabb.min.v, 24);
    memcpy(o + 30, t->cached_aabb.max.v, 24);
    line_no[i] = t->debug_line_no;
    has_mtl[i] = t->mtl != nullptr;
  }
}

// The flattened scene exactly as MythTracer::Prepare hands it to
// mt_scene_create, for tests that call the C ABI directly.  Two-step protocol:
// call with NULL outputs to get the counts, then with buffers.
int mth_flatten(void* p, int* n_tris, int* n_materials, int* n_textures, double* vertex,
                double* normal, double* uvw, double* aabb, int32_t* material, int32_t* line_no,
                double* mats17 /* 16 values + tex index per material */,
                int32_t* tex_whf /* width,height,format per texture */) {
  Handle* h = static_cast<Handle*>(p);
  auto* scene = h->mt.GetScene();
  if (!scene->tree.IsFinalized()) scene->tree.Finalize();
  raytracer::FlatScene f;
  if (!f.Build(*scene)) {
    h->error = f.error;
    return 0;
  }
  *n_tris = (int)f.material.size();
  *n_materials = (int)f.materials.size();
  *n_textures = (int)f.textures.size();
  if (vertex == nullptr) return 1;
  memcpy(vertex, f.vertex.data(), f.vertex.size() * 8);
  memcpy(normal, f.normal.data(), f.normal.size() * 8);
  memcpy(uvw, f.uvw.data(), f.uvw.size() * 8);
  memcpy(aabb, f.aabb.data(), f.aabb.size() * 8);
  memcpy(material, f.material.data(), f.material.size() * 4);
  memcpy(line_no, f.line_no.data(), f.line_no.size() * 4);
  for (size_t i = 0; i < f.materials.size(); i++) {
    const mt_material& m = f.materials[i];
    double* o = mats17 + i * 17;
    memcpy(o, m.ambient, 24);
    memcpy(o + 3, m.diffuse, 24);
    memcpy(o + 6, m.specular, 24);
    o[9] = m.specular_exp;
    o[10] = m.reflectance;
    o[11] = m.transparency;
    memcpy(o + 12, m.transmission_filter, 24);
    o[15] = m.refraction_index;
    o[16] = (double)m.tex;
  }
  for (size_t i = 0; i < f.textures.size(); i++) {
    tex_whf[i * 3] = f.textures[i].width;
    tex_whf[i * 3 + 1] = f.textures[i].height;
    tex_whf[i * 3 + 2] = f.textures[i].format;
  }
  return 1;
}

// texels of texture i of the flattened scene: RGB8 bytes or doubles
int mth_flatten_texels(void* p, int i, void* out) {
  Handle* h = static_cast<Handle*>(p);
  raytracer::FlatScene f;
  if (!f.Build(*h->mt.GetScene()) || i < 0 || (size_t)i >= f.textures.size()) return 0;
  const mt_texture& t = f.textures[(size_t)i];
  memcpy(out, t.texels, (size_t)t.width * t.height * (t.format == MT_TEX_RGB8 ? 3 : 24));
  return 1;
}

int mth_num_materials(void* p) { return (int)static_cast<Handle*>(p)->mt.GetScene()->materials.size(); }

// ka kd ks ns refl tr tf ni (16 doubles) of the named material; 0 if absent
int mth_get_material(void* p, const char* name, double* out16, int* has_tex) {
  auto& mats = static_cast<Handle*>(p)->mt.GetScene()->materials;
  auto it = mats.find(name);
  if (it == mats.end()) return 0;
  const Material& m = *it->second;
  memcpy(out16, m.ambient.v, 24);
  memcpy(out16 + 3, m.diffuse.v, 24);
  memcpy(out16 + 6, m.specular.v, 24);
  out16[9] = m.specular_exp;
  out16[10] = m.reflectance;
  out16[11] = m.transparency;
  memcpy(out16 + 12, m.transmission_filter.v, 24);
  out16[15] = m.refraction_index;
  *has_tex = m.tex != nullptr;
  return 1;
}

void mth_sensor(const double* cam7, int w, int hgt, double* out12) {
  const Camera cam = MakeCamera(cam7);
  const Camera::Sensor s = cam.GetSensor(w, hgt);
  memcpy(out12, cam.origin.v, 24);
  memcpy(out12 + 3, s.StartPoint().v, 24);
  memcpy(out12 + 6, s.DeltaScanline().v, 24);
  memcpy(out12 + 9, s.DeltaPixel().v, 24);
}

void mth_sensor_ray(const double* cam7, int w, int hgt, int x, int y, double* dir3) {
  const Camera cam = MakeCamera(cam7);
  const raytracer::Ray r = cam.GetSensor(w, hgt).GetRay(x, y);
  memcpy(dir3, r.direction.v, 24);
}

// MythTracer::RayTrace(WorkChunk*).  stats10: rays p/s/sh, box, node, tri, mt,
// shaded, then kernel_ms and total_ms as doubles in ms2.
int mth_render_chunk(void* p, const double* cam7, int iw, int ih, int cx, int cy, int cw, int ch,
                     uint8_t* rgb, int32_t* dbg_line, double* dbg_point, uint64_t* stats8, double* ms2) {
  Handle* h = static_cast<Handle*>(p);
  WorkChunk chunk{iw, ih, cx, cy, cw, ch, MakeCamera(cam7), {}, {}};
  const size_t npx = (size_t)(cw > 0 ? cw : 0) * (size_t)(ch > 0 ? ch : 0);
  chunk.output_bitmap.resize(npx * 3);
  if (dbg_line) chunk.output_debug.resize(npx);
  if (!h->mt.RayTrace(&chunk)) return 0;
  memcpy(rgb, chunk.output_bitmap.data(), npx * 3);
  for (size_t i = 0; dbg_line && i < npx; i++) {
    dbg_line[i] = chunk.output_debug[i].line_no;
    if (dbg_point) memcpy(dbg_point + i * 3, chunk.output_debug[i].point.v, 24);
  }
  const raytracer::RenderStats& s = h->mt.LastStats();
  if (stats8) {
    const uint64_t v[8] = {s.rays_primary, s.rays_secondary, s.rays_shadow, s.box_tests,
                           s.node_visits,  s.tri_tests,      s.mt_tests,    s.shaded_hits};
    memcpy(stats8, v, sizeof v);
  }
  if (ms2) {
    ms2[0] = s.kernel_ms;
    ms2[1] = s.total_ms;
  }
  return 1;
}

// MythTracer::RayTrace(int, int, Camera*, vector*)
int mth_render_image(void* p, const double* cam7, int iw, int ih, uint8_t* rgb) {
  Handle* h = static_cast<Handle*>(p);
  Camera cam = MakeCamera(cam7);
  std::vector<uint8_t> bmp;
  if (!h->mt.RayTrace(iw, ih, &cam, &bmp)) return 0;
  memcpy(rgb, bmp.data(), bmp.size());
  return 1;
}

// The frame loop of main_local.cc:51-132 as a driver written against the facade runs it: per frame the lights are
// cleared and pushed again (:79-110), a Camera is aggregate-initialised with the frame's yaw (:72-76, `dyaw` degrees
// per frame, a triangular pan of +-4 steps around cam7's yaw that ends on it) and RayTrace(W, H, &cam, &bitmap)
// fills ONE vector that lives outside the loop (:122).  ms[i] = wall time of frame i's RayTrace call including the
// light upload and the copy of the frame into `bitmap` -- what the drop-in caller waits for; rgb = the last frame.
int mth_frame_loop(void* p, const double* cam7, int iw, int ih, int n_frames, double dyaw, int collect_stats,
                   double* ms, uint8_t* rgb) {
  Handle* h = static_cast<Handle*>(p);
  h->mt.SetCollectStats(collect_stats != 0);
  const std::vector<raytracer::Light> lights = h->mt.GetScene()->lights;
  std::vector<uint8_t> bitmap;
  for (int i = 0; i < n_frames; i++) {
    auto& L = h->mt.GetScene()->lights;
    L.clear();
    for (const raytracer::Light& l : lights) L.push_back(l);
    const int j = ((n_frames - 1 - i) % 16 + 16) % 16;
    const int tri = j <= 4 ? j : (j <= 12 ? 8 - j : j - 16);
    double c[7];
    memcpy(c, cam7, sizeof c);
    c[4] += dyaw * tri;
    Camera cam = MakeCamera(c);
    const auto t0 = std::chrono::steady_clock::now();
    if (!h->mt.RayTrace(iw, ih, &cam, &bitmap)) {
      h->mt.SetCollectStats(true);
      return 0;
    }
    ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  h->mt.SetCollectStats(true);
  if (rgb && !bitmap.empty()) memcpy(rgb, bitmap.data(), bitmap.size());
  return 1;
}

// OctTree::IntersectRays; prim = AddPrimitive index or -1
int mth_intersect(void* p, int n, const double* rays, int32_t* prim, int32_t* line, double* t, double* point) {
  Handle* h = static_cast<Handle*>(p);
  auto& tree = h->mt.GetScene()->tree;
  if (!tree.IsFinalized()) tree.Finalize();
  std::vector<const raytracer::Primitive*> hits((size_t)n);
  if (!tree.IntersectRays(n, rays, hits.data(), t, point)) {
    h->error = tree.LastError();
    return 0;
  }
  for (int i = 0; i < n; i++) {
    int idx = -1;
    if (hits[(size_t)i]) {
      for (size_t k = 0; k < tree.PrimitiveCount(); k++) {
        if (tree.GetTriangle(k) == hits[(size_t)i]) { idx = (int)k; break; }
      }
    }
    if (prim) prim[i] = idx;
    if (line) line[i] = hits[(size_t)i] ? hits[(size_t)i]->debug_line_no : -1;
  }
  return 1;
}

// --- wire formats (WorkChunk / Camera), for the serialisation tests
int mth_chunk_serialize_input(const int32_t* f6, uint8_t* out24) {
  WorkChunk c{f6[0], f6[1], f6[2], f6[3], f6[4], f6[5], Camera{}, {}, {}};
  std::vector<uint8_t> b;
  c.SerializeInput(&b);
  memcpy(out24, b.data(), b.size());
  return (int)b.size();
}

int mth_chunk_deserialize_input(const uint8_t* bytes, int n, int32_t* f6) {
  WorkChunk c{};
  if (!c.DeserializeInput(std::vector<uint8_t>(bytes, bytes + n))) return 0;
  const int32_t v[6] = {c.image_width, c.image_height, c.chunk_x, c.chunk_y, c.chunk_width, c.chunk_height};
  memcpy(f6, v, sizeof v);
  return 1;
}

int mth_chunk_output_roundtrip(int cw, int ch, const uint8_t* rgb, int n_rgb, uint8_t* packet,
                               int packet_cap, uint8_t* back) {
  WorkChunk c{};
  c.chunk_width = cw;
  c.chunk_height = ch;
  c.output_bitmap.assign(rgb, rgb + n_rgb);
  std::vector<uint8_t> b;
  if (!c.SerializeOutput(&b) || (int)b.size() > packet_cap) return -1;
  memcpy(packet, b.data(), b.size());
  WorkChunk d{};
  d.chunk_width = cw;
  d.chunk_height = ch;
  if (!d.DeserializeOutput(b)) return -2;
  memcpy(back, d.output_bitmap.data(), d.output_bitmap.size());
  return (int)b.size();
}

int mth_chunk_deserialize_output(int cw, int ch, const uint8_t* packet, int n) {
  WorkChunk d{};
  d.chunk_width = cw;
  d.chunk_height = ch;
  return d.DeserializeOutput(std::vector<uint8_t>(packet, packet + n)) ? 1 : 0;
}

int mth_camera_roundtrip(const double* cam7, uint8_t* out56, double* back7) {
  Camera cam = MakeCamera(cam7);
  std::vector<uint8_t> b;
  cam.Serialize(&b);
  memcpy(out56, b.data(), b.size());
  Camera c2{};
  if (!c2.Deserialize(b)) return 0;
  const double v[7] = {c2.origin.v[0], c2.origin.v[1], c2.origin.v[2], c2.pitch, c2.yaw, c2.roll, c2.aov};
  memcpy(back7, v, sizeof v);
  return (int)b.size();
}

}  // extern "C"
