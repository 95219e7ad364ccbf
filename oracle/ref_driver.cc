// ref_driver — TEST INFRASTRUCTURE ONLY (not product code).
//
// A small job runner around the *real* MythTracer reference, compiled from the
// sources where they lie under /root/reference/VerStarting (see oracle/Makefile;
// nothing from the reference is copied into this repository).  It drives the
// reference's public C++ API exactly as VerStarting/main_local.cc:34-122 does
// (LoadObj -> GetScene()->lights -> Camera{} -> RayTrace) and dumps what comes
// out, so that golden vectors under tests/golden/ and the "reference" CPU
// baseline in bench.py are produced by the reference itself.
//
// The reference's texture.cc needs <SDL2/SDL_image.h>, which this image lacks;
// it is therefore NOT built (no stand-in is written) and the two symbols it
// defines (Texture::LoadFromFile, Texture::GetColorAt) stay unresolved in
// oracle/_ref/libmythtracer_ref.so.  They are bound lazily, so jobs whose .mtl
// has no map_Ka line never touch them.  Textured scenes cannot be run here.
//
// Job file: whitespace separated directives, one per line:
//   obj <path>
//   image <W> <H>
//   chunk <x> <y> <w> <h>            (default: the whole image)
//   camera <ox> <oy> <oz> <pitch> <yaw> <roll> <aov>
//   light <px py pz> <ar ag ab> <dr dg db> <sr sg sb>     (repeatable)
//   out_rgb <file>                   raw RGB8 of the chunk
//   out_debug <file>                 per pixel: int32 line_no + 3 x f64 point
//   rays <in.bin> <out.bin>          in: n x 6 f64 (origin, direction);
//                                    out per ray: i32 line_no(-1 miss), f64 t,
//                                    3 f64 P, 3 f64 normal, 3 f64 uvw
//   sensor <out.bin>                 ray direction of every pixel of the chunk
//                                    (3 f64 each) via Camera::Sensor::GetRay
//   repeat <n>                       time n renders, report the best
//   out_time <file>                  JSON with wall-clock seconds of RayTrace
//   tree <out.bin>                   the finalized octree, breadth-first (the
//                                    8 children of a node consecutive, root
//                                    first).  Header: i32 n_nodes; per node:
//                                    6 f64 aabb (min, max), 3 f64 center,
//                                    i32 has_children, i32 n_prims, then
//                                    n_prims x i32 debug_line_no in the order
//                                    of Node::primitives.  (OctTree::root is
//                                    private: octtree.h is included with
//                                    `private` defined away, which does not
//                                    change the layout of the class.)
//   wire <out.bin>                   bytes of WorkChunk::SerializeInput for the
//                                    job's image/chunk (24), Camera::Serialize
//                                    for its camera (56) and, if out_rgb was
//                                    rendered, WorkChunk::SerializeOutput
//   wire_in <file>                   one candidate WorkChunk input blob per
//                                    24 bytes; appends to the `wire` output one
//                                    byte per blob: DeserializeInput's verdict
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <omp.h>

#include <deque>
#include <list>
#include <memory>
#include <utility>

// test-only access to OctTree::root for the `tree` dump (see above)
#define private public
#include "octtree.h"
#undef private
#include "mythtracer.h"
#include "camera.h"

using math3d::V3D;
using raytracer::Camera;
using raytracer::Light;
using raytracer::MythTracer;
using raytracer::PerPixelDebugInfo;
using raytracer::Primitive;
using raytracer::Ray;
using raytracer::WorkChunk;

namespace {

struct Job {
  std::string obj;
  int W = 0, H = 0;
  int cx = 0, cy = 0, cw = -1, ch = -1;
  double cam[7] = {0, 0, 0, 0, 0, 0, 60};
  std::vector<Light> lights;
  std::string out_rgb, out_debug, rays_in, rays_out, sensor_out, out_time, tree_out, wire_out, wire_in;
  int repeat = 1;
};

bool ParseJob(const char *path, Job *job) {
  std::ifstream f(path);
  if (!f) {
    fprintf(stderr, "ref_driver: cannot open job %s\n", path);
    return false;
  }
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream s(line);
    std::string key;
    if (!(s >> key) || key[0] == '#') continue;
    if (key == "obj") {
      std::getline(s >> std::ws, job->obj);
    } else if (key == "image") {
      s >> job->W >> job->H;
    } else if (key == "chunk") {
      s >> job->cx >> job->cy >> job->cw >> job->ch;
    } else if (key == "camera") {
      for (double &d : job->cam) s >> d;
    } else if (key == "light") {
      double d[12];
      for (double &x : d) s >> x;
      job->lights.push_back(Light{{d[0], d[1], d[2]},
                                  {d[3], d[4], d[5]},
                                  {d[6], d[7], d[8]},
                                  {d[9], d[10], d[11]}});
    } else if (key == "out_rgb") {
      s >> job->out_rgb;
    } else if (key == "out_debug") {
      s >> job->out_debug;
    } else if (key == "rays") {
      s >> job->rays_in >> job->rays_out;
    } else if (key == "sensor") {
      s >> job->sensor_out;
    } else if (key == "tree") {
      s >> job->tree_out;
    } else if (key == "wire") {
      s >> job->wire_out;
    } else if (key == "wire_in") {
      s >> job->wire_in;
    } else if (key == "repeat") {
      s >> job->repeat;
    } else if (key == "out_time") {
      s >> job->out_time;
    } else {
      fprintf(stderr, "ref_driver: unknown directive %s\n", key.c_str());
      return false;
    }
    if (s.fail()) {
      fprintf(stderr, "ref_driver: bad arguments for %s\n", key.c_str());
      return false;
    }
  }
  if (job->cw < 0) {
    job->cx = job->cy = 0;
    job->cw = job->W;
    job->ch = job->H;
  }
  return true;
}

bool WriteFile(const std::string &path, const void *data, size_t n) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) {
    fprintf(stderr, "ref_driver: cannot write %s\n", path.c_str());
    return false;
  }
  size_t w = n ? fwrite(data, 1, n, f) : 0;
  fclose(f);
  return w == n;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc != 2) {
    fprintf(stderr, "usage: ref_driver <job file>\n");
    return 2;
  }
  Job job;
  if (!ParseJob(argv[1], &job)) return 2;

  // The reference chats on stdout (dots per row, timings); keep it away from
  // whoever parses our output.
  if (!freopen("/dev/null", "w", stdout)) return 2;

  MythTracer mt;
  if (!mt.LoadObj(job.obj.c_str())) {
    fprintf(stderr, "ref_driver: LoadObj failed\n");
    return 1;
  }
  mt.GetScene()->lights = job.lights;

  Camera cam{{job.cam[0], job.cam[1], job.cam[2]},
             job.cam[3], job.cam[4], job.cam[5], job.cam[6]};

  // First call finalizes the tree (mythtracer.cc:281-285).  Do it on a 1x1
  // chunk so that the timed render below is the pixel loop only.
  if (job.W > 0 && job.H > 0) {
    WorkChunk warm{job.W, job.H, 0, 0, 1, 1, cam, {}, {}};
    warm.output_bitmap.resize(3);
    mt.RayTrace(&warm);
  } else {
    mt.GetScene()->tree.Finalize();
  }

  std::vector<uint8_t> wire_output;  // SerializeOutput of the rendered chunk
  if (job.W > 0 && (!job.out_rgb.empty() || !job.out_debug.empty() ||
                    !job.out_time.empty())) {
    double best = 1e300;
    WorkChunk chunk{job.W, job.H, job.cx, job.cy, job.cw, job.ch, cam, {}, {}};
    for (int r = 0; r < job.repeat; r++) {
      chunk.output_bitmap.assign((size_t)job.cw * job.ch * 3, 0);
      if (!job.out_debug.empty()) {
        chunk.output_debug.assign((size_t)job.cw * job.ch, PerPixelDebugInfo{});
      }
      auto t0 = std::chrono::steady_clock::now();
      mt.RayTrace(&chunk);
      auto t1 = std::chrono::steady_clock::now();
      best = std::min(best, std::chrono::duration<double>(t1 - t0).count());
    }
    if (!job.out_rgb.empty() &&
        !WriteFile(job.out_rgb, chunk.output_bitmap.data(),
                   chunk.output_bitmap.size())) {
      return 1;
    }
    if (!job.wire_out.empty() && !job.out_rgb.empty()) {
      if (!chunk.SerializeOutput(&wire_output)) return 1;
    }
    if (!job.out_debug.empty()) {
      std::vector<uint8_t> buf(chunk.output_debug.size() * 28);
      uint8_t *p = buf.data();
      for (const auto &d : chunk.output_debug) {
        int32_t ln = d.line_no;
        memcpy(p, &ln, 4);
        memcpy(p + 4, d.point.v, 24);
        p += 28;
      }
      if (!WriteFile(job.out_debug, buf.data(), buf.size())) return 1;
    }
    if (!job.out_time.empty()) {
      char js[256];
      int n = snprintf(js, sizeof js,
                       "{\"seconds\": %.6f, \"threads\": %d, \"pixels\": %lld}\n",
                       best, omp_get_max_threads(),
                       (long long)job.cw * job.ch);
      if (!WriteFile(job.out_time, js, (size_t)n)) return 1;
    }
  }

  if (!job.tree_out.empty()) {
    using Node = raytracer::OctTree::Node;
    std::vector<uint8_t> out;
    auto put = [&out](const void *p, size_t n) {
      const uint8_t *b = (const uint8_t *)p;
      out.insert(out.end(), b, b + n);
    };
    std::deque<const Node *> queue{&mt.GetScene()->tree.root};
    std::vector<const Node *> order;
    while (!queue.empty()) {
      const Node *n = queue.front();
      queue.pop_front();
      order.push_back(n);
      for (const Node &c : n->nodes) queue.push_back(&c);
    }
    int32_t nn = (int32_t)order.size();
    put(&nn, 4);
    for (const Node *n : order) {
      put(n->aabb.min.v, 24);
      put(n->aabb.max.v, 24);
      put(n->center.v, 24);
      int32_t hc = n->nodes.empty() ? 0 : 1, np = (int32_t)n->primitives.size();
      put(&hc, 4);
      put(&np, 4);
      for (const Primitive *p : n->primitives) {
        int32_t ln = p->debug_line_no;
        put(&ln, 4);
      }
    }
    if (!WriteFile(job.tree_out, out.data(), out.size())) return 1;
  }

  if (!job.wire_out.empty()) {
    std::vector<uint8_t> out, b;
    WorkChunk chunk{job.W, job.H, job.cx, job.cy, job.cw, job.ch, cam, {}, {}};
    chunk.SerializeInput(&b);
    out.insert(out.end(), b.begin(), b.end());
    cam.Serialize(&b);
    out.insert(out.end(), b.begin(), b.end());
    out.insert(out.end(), wire_output.begin(), wire_output.end());
    if (!job.wire_in.empty()) {
      std::ifstream in(job.wire_in, std::ios::binary);
      std::vector<char> raw((std::istreambuf_iterator<char>(in)),
                            std::istreambuf_iterator<char>());
      for (size_t i = 0; i + 24 <= raw.size(); i += 24) {
        WorkChunk w{};
        std::vector<uint8_t> blob(raw.begin() + i, raw.begin() + i + 24);
        out.push_back(w.DeserializeInput(blob) ? 1 : 0);
      }
    }
    if (!WriteFile(job.wire_out, out.data(), out.size())) return 1;
  }

  if (!job.rays_in.empty()) {
    std::ifstream in(job.rays_in, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(in)),
                          std::istreambuf_iterator<char>());
    size_t n = raw.size() / 48;
    std::vector<uint8_t> out(n * 84);
    for (size_t i = 0; i < n; i++) {
      double r[6];
      memcpy(r, raw.data() + i * 48, 48);
      Ray ray({r[0], r[1], r[2]}, {r[3], r[4], r[5]});
      V3D point{NAN, NAN, NAN};
      double t = NAN;
      const Primitive *p = mt.GetScene()->tree.IntersectRay(ray, &point, &t);
      int32_t ln = -1;
      V3D nrm{NAN, NAN, NAN}, uvw{NAN, NAN, NAN};
      if (p != nullptr) {
        ln = p->debug_line_no;
        nrm = p->GetNormal(point);
        uvw = p->GetUVW(point);
      }
      uint8_t *o = out.data() + i * 84;
      memcpy(o, &ln, 4);
      memcpy(o + 4, &t, 8);
      memcpy(o + 12, point.v, 24);
      memcpy(o + 36, nrm.v, 24);
      memcpy(o + 60, uvw.v, 24);
    }
    if (!WriteFile(job.rays_out, out.data(), out.size())) return 1;
  }

  if (!job.sensor_out.empty()) {
    Camera::Sensor sensor = cam.GetSensor(job.W, job.H);
    std::vector<double> dirs((size_t)job.cw * job.ch * 3);
    size_t k = 0;
    for (int j = 0; j < job.ch; j++) {
      for (int i = 0; i < job.cw; i++) {
        Ray r = sensor.GetRay(job.cx + i, job.cy + j);
        dirs[k++] = r.direction.v[0];
        dirs[k++] = r.direction.v[1];
        dirs[k++] = r.direction.v[2];
      }
    }
    if (!WriteFile(job.sensor_out, dirs.data(), dirs.size() * 8)) return 1;
  }
  return 0;
}
