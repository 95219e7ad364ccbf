// mythtracer.h — the renderer class: the drop-in seam (reference:
// VerStarting/mythtracer.h:11-79).  Same names, fields and call contract as the
// reference so that main_local.cc / main_net_worker.cc build unchanged; the
// pixel loop behind RayTrace runs on an MI355X through libmythtracer_hip.so.
#pragma once
#include <stdint.h>
#include <cstddef>
#include <string>
#include <vector>
#include "camera.h"
#include "objreader.h"
#include "octtree.h"

struct mt_scene;
struct mt_stats;

namespace raytracer {
using math3d::V3D;

const int MAX_RECURSION_LEVEL = 5;

struct PerPixelDebugInfo {
  int line_no;
  V3D point;
};

class WorkChunk {
 public:
  // input
  int image_width, image_height;
  int chunk_x, chunk_y;
  int chunk_width, chunk_height;
  Camera camera;  // not part of the serialized input

  static const size_t kSerializedInputSize = 6 * sizeof(uint32_t);
  void SerializeInput(std::vector<uint8_t>* bytes);
  bool DeserializeInput(const std::vector<uint8_t>& bytes);

  // output: chunk-local row-major RGB8; optional per-pixel first-hit info
  std::vector<uint8_t> output_bitmap;
  std::vector<PerPixelDebugInfo> output_debug;

  static const size_t kSerializedOutputMinimumSize = sizeof(uint32_t);
  bool SerializeOutput(std::vector<uint8_t>* bytes);
  bool DeserializeOutput(const std::vector<uint8_t>& bytes);
};

// Work counters of the last RayTrace call (extension; see mt_stats).
struct RenderStats {
  uint64_t rays_primary = 0, rays_secondary = 0, rays_shadow = 0;
  uint64_t box_tests = 0, node_visits = 0, tri_tests = 0, mt_tests = 0, shaded_hits = 0;
  double kernel_ms = 0, total_ms = 0;
};

class MythTracer {
 public:
  MythTracer();
  ~MythTracer();
  MythTracer(const MythTracer&) = delete;
  MythTracer& operator=(const MythTracer&) = delete;

  Scene* GetScene();
  bool LoadObj(const char* fname);
  bool RayTrace(int image_width, int image_height, Camera* camera,
                std::vector<uint8_t>* output_bitmap);
  bool RayTrace(WorkChunk* chunk);

  // --- extensions (not in the reference)
  void SetDevice(int hip_device) {
    device_ = hip_device;
    scene.tree.SetDevice(hip_device);
  }
  // Several GPUs of this process for the W x H overload of RayTrace: a scene replica per listed HIP device
  // (a device may be listed more than once), tiles k = r (mod N) of the frame rendered side by side, gathered
  // and blitted on the first device -- mt_render_frame_multi; the master/worker farm of main_net_master.cc:195-236
  // in one process.  RayTrace(WorkChunk*) -- the unit a worker renders -- stays on the first device.
  void SetDevices(const std::vector<int>& hip_devices);
  void SetMaxRecursionLevel(int level) { max_level_ = level; }  // default MAX_RECURSION_LEVEL
  void SetQuiet(bool quiet) {                                   // no progress text on stdout
    quiet_ = quiet;
    scene.tree.SetQuiet(quiet);
  }
  // Work counters of a render (LastStats).  On by default (the tests read them); a driver that only wants the frame
  // switches them off: the kernels built without the counting are about a tenth faster (mt_scene_set_stats).
  void SetCollectStats(bool on) { collect_stats_ = on; }
  const RenderStats& LastStats() const { return stats_; }
  const char* LastError() const { return error_.c_str(); }
  // Finalizes the tree if needed and uploads the scene; RayTrace does this
  // lazily on its first call exactly like the reference finalizes lazily.
  bool Prepare();
  mt_scene* DeviceScene() { return dev_; }

 private:
  Scene scene;
  bool was_scene_finalized = false;
  mt_scene* dev_ = nullptr;            // replica on devices_[0] (= device_)
  std::vector<mt_scene*> replicas_;    // replicas on devices_[1..]
  std::vector<int> devices_;           // empty = {device_}
  int device_ = 0;
  void DropDeviceScenes();
  int max_level_ = MAX_RECURSION_LEVEL;
  bool quiet_ = false;
  bool collect_stats_ = true;
  RenderStats stats_;
  std::string error_;
};

}  // namespace raytracer
