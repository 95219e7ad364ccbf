// mythtracer.cc — MythTracer / WorkChunk of the facade (reference:
// VerStarting/mythtracer.cc:243-429).  Everything under the reference's pixel
// loop (mythtracer.cc:292-305) is one kernel launch behind mt_render_chunk.
#include "mythtracer.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <limits>
#include <unordered_map>

#include "flatten.h"
#include "mythtracer_hip.h"
#include "primitive_triangle.h"

namespace raytracer {

MythTracer::MythTracer() {}

MythTracer::~MythTracer() { DropDeviceScenes(); }

void MythTracer::DropDeviceScenes() {
  if (dev_) mt_scene_destroy(dev_);
  dev_ = nullptr;
  for (mt_scene* r : replicas_) mt_scene_destroy(r);
  replicas_.clear();
}

void MythTracer::SetDevices(const std::vector<int>& hip_devices) {
  DropDeviceScenes();  // uploaded again, where they are wanted, by the next Prepare
  devices_ = hip_devices;
  if (!devices_.empty()) SetDevice(devices_[0]);
}

Scene* MythTracer::GetScene() { return &scene; }

bool MythTracer::LoadObj(const char* fname) {
  if (!quiet_) puts("Reading .OBJ file.");
  ObjFileReader reader;
  if (!reader.ReadObjFile(&scene, fname)) return false;
  was_scene_finalized = false;
  return true;
}

namespace {

// Textures whose texels all came from 8-bit data (colour == k / 255.0, which
// is what both the reference's loader and ours produce) travel as RGB8; the
// kernel re-creates the identical double with byte / 255.0.
bool PackRgb8(const Texture& t, std::vector<uint8_t>* out) {
  out->resize(t.colors.size() * 3);
  for (size_t i = 0; i < t.colors.size(); i++) {
    for (int c = 0; c < 3; c++) {
      const double v = t.colors[i].v[c];
      const int k = (int)(v * 255.0 + 0.5);
      if (!(v >= 0.0 && v <= 1.0) || k < 0 || k > 255 || (double)k / 255.0 != v) {
        out->clear();
        return false;
      }
      (*out)[i * 3 + c] = (uint8_t)k;
    }
  }
  return true;
}

}  // namespace

bool FlatScene::Build(const Scene& scene) {
  const OctTree& tree = scene.tree;
  if (!tree.IsFinalized()) {
    error = "scene tree is not finalized";
    return false;
  }
  const FlatTree& f = tree.Flat();
  const size_t nt = f.tri_id.size();

  // materials / textures -> dense tables, in order of first use
  std::unordered_map<const Material*, int32_t> mtl_index;
  std::unordered_map<const Texture*, int32_t> tex_index;
  std::vector<const Material*> mtls;
  std::vector<const Texture*> texs;
  auto intern_material = [&](const Material* m) -> int32_t {
    if (m == nullptr) return -1;
    auto it = mtl_index.find(m);
    if (it != mtl_index.end()) return it->second;
    const int32_t id = (int32_t)mtls.size();
    mtl_index[m] = id;
    mtls.push_back(m);
    if (m->tex != nullptr && tex_index.find(m->tex) == tex_index.end()) {
      tex_index[m->tex] = (int32_t)texs.size();
      texs.push_back(m->tex);
    }
    return id;
  };

  vertex.resize(nt * 9);
  normal.resize(nt * 9);
  uvw.resize(nt * 9);
  aabb.resize(nt * 6);
  material.resize(nt);
  line_no.resize(nt);
  for (size_t s = 0; s < nt; s++) {
    const Triangle* t = tree.GetTriangle((size_t)f.tri_id[s]);
    memcpy(&vertex[s * 9], t->vertex, 72);
    memcpy(&normal[s * 9], t->normal, 72);
    memcpy(&uvw[s * 9], t->uvw, 72);
    memcpy(&aabb[s * 6], t->cached_aabb.min.v, 24);
    memcpy(&aabb[s * 6 + 3], t->cached_aabb.max.v, 24);
    material[s] = intern_material(t->mtl);
    line_no[s] = t->debug_line_no;
  }
  materials.assign(mtls.size(), mt_material{});
  for (size_t i = 0; i < mtls.size(); i++) {
    const Material& m = *mtls[i];
    mt_material& o = materials[i];
    memcpy(o.ambient, m.ambient.v, 24);
    memcpy(o.diffuse, m.diffuse.v, 24);
    memcpy(o.specular, m.specular.v, 24);
    memcpy(o.transmission_filter, m.transmission_filter.v, 24);
    o.specular_exp = m.specular_exp;
    o.reflectance = m.reflectance;
    o.transparency = m.transparency;
    o.refraction_index = m.refraction_index;
    o.tex = m.tex ? tex_index[m.tex] : -1;
  }
  textures.assign(texs.size(), mt_texture{});
  rgb8.assign(texs.size(), {});
  for (size_t i = 0; i < texs.size(); i++) {
    const Texture& t = *texs[i];
    if (t.colors.size() != t.width * t.height || t.width == 0 || t.height == 0) {
      error = "texture with inconsistent size";
      return false;
    }
    mt_texture& o = textures[i];
    o.width = (int32_t)t.width;
    o.height = (int32_t)t.height;
    if (PackRgb8(t, &rgb8[i])) {
      o.format = MT_TEX_RGB8;
      o.texels = rgb8[i].data();
    } else {
      o.format = MT_TEX_F64;
      o.texels = t.colors.data();  // V3D is three packed doubles
    }
  }
  return true;
}

mt_scene_desc FlatScene::Describe(const Scene& scene, int device) const {
  const FlatTree& f = scene.tree.Flat();
  mt_scene_desc d;
  memset(&d, 0, sizeof d);
  d.struct_size = sizeof d;
  d.abi_version = MT_ABI_VERSION;
  d.device = device;
  d.n_nodes = (int32_t)f.NodeCount();
  d.n_tris = (int32_t)f.tri_id.size();
  d.n_materials = (int32_t)materials.size();
  d.n_textures = (int32_t)textures.size();
  d.tree_depth = f.depth;
  d.node_aabb = f.node_aabb.data();
  d.node_center = f.node_center.data();
  d.node_first_child = f.first_child.data();
  d.node_prim_begin = f.prim_begin.data();
  d.node_prim_count = f.prim_count.data();
  d.tri_vertex = vertex.data();
  d.tri_normal = normal.data();
  d.tri_uvw = uvw.data();
  d.tri_aabb = aabb.data();
  d.tri_material = material.data();
  d.tri_line_no = line_no.data();
  d.tri_id = f.tri_id.data();
  d.materials = materials.data();
  d.textures = textures.data();
  return d;
}

bool MythTracer::Prepare() {
  if (!was_scene_finalized) {
    if (!quiet_) puts("Finalizing tree.");
    scene.tree.Finalize();
    was_scene_finalized = true;
    DropDeviceScenes();  // geometry changed (LoadObj after a render): upload again
  }
  if (dev_) return true;
  FlatScene flat;
  if (!flat.Build(scene)) {
    error_ = flat.error;
    return false;
  }
  // one replica per listed device: every worker of the reference loads its own copy of the scene
  // (main_net_worker.cc:29-32)
  const size_t n = devices_.empty() ? 1 : devices_.size();
  for (size_t r = 0; r < n; r++) {
    const mt_scene_desc d = flat.Describe(scene, devices_.empty() ? device_ : devices_[r]);
    mt_scene* s = mt_scene_create(&d);
    if (s == nullptr) {
      error_ = mt_last_error();
      fprintf(stderr, "error: cannot create the device scene: %s\n", error_.c_str());
      DropDeviceScenes();
      return false;
    }
    if (r == 0) dev_ = s;
    else replicas_.push_back(s);
  }
  return true;
}

bool MythTracer::RayTrace(int image_width, int image_height, Camera* camera,
                          std::vector<uint8_t>* output_bitmap) {
  if (devices_.size() > 1) {
    // the whole frame on all listed GPUs (mt_render_frame_multi); 64x64 tiles, finer than the master's 128x128
    // chunks (main_net_master.cc:24-25): a pixel's cost varies 60-fold across the frame
    if (!Prepare()) return false;
    if (!quiet_) puts("Rendering.");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<mt_scene*> all{dev_};
    all.insert(all.end(), replicas_.begin(), replicas_.end());
    for (mt_scene* s : all) {
      if (mt_scene_set_lights(s, reinterpret_cast<const mt_light*>(scene.lights.data()), (int)scene.lights.size()) != MT_OK) {
        error_ = mt_last_error();
        return false;
      }
    }
    const Camera::Sensor sensor = camera->GetSensor(image_width, image_height);
    mt_sensor ms;
    memcpy(ms.origin, camera->origin.v, 24);
    memcpy(ms.start_point, sensor.StartPoint().v, 24);
    memcpy(ms.delta_scanline, sensor.DeltaScanline().v, 24);
    memcpy(ms.delta_pixel, sensor.DeltaPixel().v, 24);
    output_bitmap->resize((size_t)image_width * image_height * 3);
    std::vector<mt_stats> st(all.size());
    if (mt_render_frame_multi(all.data(), (int)all.size(), &ms, image_width, image_height, 64, 64, max_level_,
                              output_bitmap->data(), st.data()) != MT_OK) {
      error_ = mt_last_error();
      fprintf(stderr, "error: render failed: %s\n", error_.c_str());
      return false;
    }
    stats_ = RenderStats{};
    for (const mt_stats& q : st) {
      stats_.rays_primary += q.rays_primary;
      stats_.rays_secondary += q.rays_secondary;
      stats_.rays_shadow += q.rays_shadow;
      stats_.box_tests += q.box_tests;
      stats_.node_visits += q.node_visits;
      stats_.tri_tests += q.tri_tests;
      stats_.mt_tests += q.mt_tests;
      stats_.shaded_hits += q.shaded_hits;
      if (q.kernel_ms > stats_.kernel_ms) stats_.kernel_ms = q.kernel_ms;  // the slowest replica
      stats_.total_ms = q.total_ms;
    }
    if (!quiet_) printf("%.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return true;
  }
  // (mythtracer.cc:258-278 wraps a full-frame WorkChunk and resizes the caller's vector.)  The chunk borrows the caller's
  // vector: a frame loop that hands the same vector in again (main_local.cc:51-132) keeps its storage -- no 6 MB
  // allocation with its page faults per frame.
  WorkChunk chunk{image_width, image_height, 0, 0, image_width, image_height, *camera, {}, {}};
  chunk.output_bitmap.swap(*output_bitmap);
  chunk.output_bitmap.resize((size_t)image_width * image_height * 3);
  const bool ok = RayTrace(&chunk);
  output_bitmap->swap(chunk.output_bitmap);
  return ok;
}

bool MythTracer::RayTrace(WorkChunk* chunk) {
  if (!Prepare()) return false;
  if (!quiet_) puts("Rendering.");
  const auto t0 = std::chrono::steady_clock::now();

  // Lights are re-read on every call: callers rewrite scene.lights between
  // frames (main_local.cc:79-110).
  static_assert(sizeof(Light) == sizeof(mt_light), "Light must match mt_light");
  if (mt_scene_set_lights(dev_, reinterpret_cast<const mt_light*>(scene.lights.data()),
                          (int)scene.lights.size()) != MT_OK) {
    error_ = mt_last_error();
    return false;
  }
  const Camera::Sensor sensor = chunk->camera.GetSensor(chunk->image_width, chunk->image_height);
  mt_sensor ms;
  memcpy(ms.origin, chunk->camera.origin.v, 24);
  memcpy(ms.start_point, sensor.StartPoint().v, 24);
  memcpy(ms.delta_scanline, sensor.DeltaScanline().v, 24);
  memcpy(ms.delta_pixel, sensor.DeltaPixel().v, 24);

  const size_t npx = (size_t)chunk->chunk_width * (size_t)chunk->chunk_height;
  if (chunk->output_bitmap.size() < npx * 3) {
    error_ = "WorkChunk::output_bitmap is smaller than chunk_width*chunk_height*3";
    fprintf(stderr, "error: %s\n", error_.c_str());
    return false;
  }
  std::vector<mt_debug_px> dbg;
  const bool want_debug = !chunk->output_debug.empty();
  if (want_debug) {
    if (chunk->output_debug.size() < npx) {
      error_ = "WorkChunk::output_debug is smaller than chunk_width*chunk_height";
      return false;
    }
    dbg.resize(npx);
  }
  mt_stats st;
  memset(&st, 0, sizeof st);
  (void)mt_scene_set_stats(dev_, collect_stats_ ? 1 : 0);
  if (mt_render_chunk(dev_, &ms, chunk->image_width, chunk->image_height, chunk->chunk_x,
                      chunk->chunk_y, chunk->chunk_width, chunk->chunk_height, max_level_,
                      chunk->output_bitmap.data(), want_debug ? dbg.data() : nullptr,
                      collect_stats_ ? &st : nullptr) != MT_OK) {
    error_ = mt_last_error();
    fprintf(stderr, "error: render failed: %s\n", error_.c_str());
    return false;
  }
  for (size_t i = 0; want_debug && i < npx; i++) {
    chunk->output_debug[i].line_no = dbg[i].line_no;
    chunk->output_debug[i].point = {dbg[i].point[0], dbg[i].point[1], dbg[i].point[2]};
  }
  stats_.rays_primary = st.rays_primary;
  stats_.rays_secondary = st.rays_secondary;
  stats_.rays_shadow = st.rays_shadow;
  stats_.box_tests = st.box_tests;
  stats_.node_visits = st.node_visits;
  stats_.tri_tests = st.tri_tests;
  stats_.mt_tests = st.mt_tests;
  stats_.shaded_hits = st.shaded_hits;
  stats_.kernel_ms = st.kernel_ms;
  stats_.total_ms = st.total_ms;
  if (!quiet_) {
    // wall-clock, unlike the reference's clock() (process CPU time)
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%.3fs\n", sec);
  }
  return true;
}

// ---- wire format of a chunk (mythtracer.cc:314-429): six little-endian u32
// in, u32 byte count + RGB bytes out.

void WorkChunk::SerializeInput(std::vector<uint8_t>* bytes) {
  const uint32_t f[6] = {(uint32_t)image_width, (uint32_t)image_height, (uint32_t)chunk_x,
                         (uint32_t)chunk_y,     (uint32_t)chunk_width,  (uint32_t)chunk_height};
  bytes->resize(kSerializedInputSize);
  memcpy(bytes->data(), f, sizeof f);
}

bool WorkChunk::DeserializeInput(const std::vector<uint8_t>& bytes) {
  if (bytes.size() != kSerializedInputSize) return false;
  uint32_t f[6];
  memcpy(f, bytes.data(), sizeof f);
  const uint32_t iw = f[0], ih = f[1], cx = f[2], cy = f[3], cw = f[4], ch = f[5];
  const bool sane = iw >= 1 && ih >= 1 && cw >= 1 && ch >= 1 && iw <= 100000 && ih <= 100000 &&
                    cx <= iw && cy <= ih && cw <= iw && ch <= ih && cx + cw <= iw && cy + ch <= ih;
  if (!sane) return false;
  image_width = (int)iw;
  image_height = (int)ih;
  chunk_x = (int)cx;
  chunk_y = (int)cy;
  chunk_width = (int)cw;
  chunk_height = (int)ch;
  return true;
}

bool WorkChunk::SerializeOutput(std::vector<uint8_t>* bytes) {
  if (output_bitmap.size() > std::numeric_limits<uint32_t>::max()) {
    fprintf(stderr, "error: too large WorkerChunk, cannot serialize\n");
    return false;
  }
  const uint32_t n = (uint32_t)output_bitmap.size();
  bytes->resize(sizeof n + n);
  memcpy(bytes->data(), &n, sizeof n);
  if (n) memcpy(bytes->data() + sizeof n, output_bitmap.data(), n);
  return true;
}

bool WorkChunk::DeserializeOutput(const std::vector<uint8_t>& bytes) {
  if (bytes.size() < kSerializedOutputMinimumSize) return false;
  uint32_t n;
  memcpy(&n, bytes.data(), sizeof n);
  const uint64_t want = (uint64_t)chunk_width * (uint64_t)chunk_height * 3;
  if ((uint64_t)n != want) return false;
  // The reference trusts the count and reads n bytes whatever the packet
  // holds (mythtracer.cc:425-426); a short packet is rejected here.
  if (bytes.size() - sizeof n < n) return false;
  output_bitmap.assign(bytes.begin() + sizeof n, bytes.begin() + sizeof n + n);
  return true;
}

}  // namespace raytracer
