// mt_capi.hip — host side of libmythtracer_hip.so: the C ABI declared in
// include/mythtracer_hip.h.  Validates the flattened scene (so that the kernel
// can index it without bounds checks), keeps it resident in HBM and launches
// the kernels of mt_render.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <limits>
#include <utility>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>

// Single translation unit: the kernels are compiled together with their host
// side so that no relocatable device code is needed.
#include "mt_render.hip"

using namespace mt;

namespace {

thread_local char g_err[1024] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(MT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                  __FILE__, __LINE__);                                                  \
  } while (0)

constexpr size_t kLdsBudget = 160 * 1024;

// Tuning constants of the work order and of the engine choice, with the values the sweeps of DESIGN.md section 5
// settled on.  Changed through mt_scene_set_tuning only (tests, experiment scripts): the library reads no
// environment variable on the launch path.  (A -DMT_DEBUG_KNOBS build additionally reads its DUMP facilities --
// item cycles, unit counts, heartbeat, time line -- from the environment, once, in mt_scene_create.)
struct Tuning {
  double v[MT_TUNE_COUNT];
  Tuning() {
    for (double &x : v) x = 0.0;
    v[MT_TUNE_POOL_BELOW] = 9.0;          // blocks per resident wave below which a launch is taken to be tail-bound
    v[MT_TUNE_POOL_CAP] = 0.0;            // 0 = default capacity of a wave's ray pool
    v[MT_TUNE_PACKED_STACK] = 1.0;
    v[MT_TUNE_BLOCKS_PER_CU] = 0.0;       // 0 = as many as fit
    v[MT_TUNE_FORECAST_RADIUS] = -1.0;    // < 0 = 1 block, 2 when the origin moved
    v[MT_TUNE_BLEND] = 0.9;
    v[MT_TUNE_FORMS] = 1.0;
    v[MT_TUNE_POOL_CUT_SHARE] = -1.0;     // < 0 = 1.0 with history, 0.3 without
    v[MT_TUNE_POOL_PIECE_TIME1] = 0.35; v[MT_TUNE_POOL_PIECE_TIME2] = 0.12;
    v[MT_TUNE_POOL_PIECE_WORK1] = 1.1;  v[MT_TUNE_POOL_PIECE_WORK2] = 3.0;
    v[MT_TUNE_POOL_CELL_FACTOR] = 1.0;  // (round 4, swept on the panning 4K frame at N = 8: 3.0 / 2.0 / 1.5 / 1.2 / 1.0 / 0.85 / 0.7 / 0.5 -> 3.39 / 3.41 / 3.37 / 3.35 / 3.20 / 3.11 / 3.29 / 4.37 ms for the slowest rank; with HYBRID_WORK2 2.6: 3.06)
    v[MT_TUNE_QUAD_SHARE] = 0.95; v[MT_TUNE_QUAD_SHARE_MOVING] = 0.7;
    v[MT_TUNE_QUAD_KEEP] = 1.0;
    v[MT_TUNE_QUAD_WORK] = 1.7;   v[MT_TUNE_QUAD_WORK_MOVING] = 1.5;
    v[MT_TUNE_POOL_SCRATCH_MB] = 4096.0;  // automatic mode: above this the state machine renders (explicit engine 2: the cap shrinks)
    // hybrid launches: blocks above POOL_SHARE of an even split go to the ray pool in pieces, blocks above QUAD_SHARE to
    // the state machine as quarters with four lanes per pixel (swept on one rank's share of the 4K frame at N = 8,
    // scripts/hybrid_sweep.py: 0.8 / 0.9 / 1.0 / 1.15 / 1.3 -> slowest rank 3.90 / 3.65 / 3.65 / 3.72 / 3.92 ms; with the
    // quarters' threshold below the pool's the state machine's 1.7x work per quartered block comes back: +0.3 ms)
    v[MT_TUNE_HYBRID_POOL_SHARE] = 1.3;   // (swept with HYBRID_CELL_FACTOR at N = 8, panning: (1.0, 1.0) 3.03 ms, (1.3, 0.85) 2.88: fewer blocks through the pool as quarters, the cells' threshold where it was)
    v[MT_TUNE_HYBRID_CELL_FACTOR] = 0.85;
    v[MT_TUNE_HYBRID_QUAD_SHARE] = 1.0;
    v[MT_TUNE_FORECAST_STEP] = 8.0;       // pixels between the old-image positions a re-projected forecast takes its maximum over
    v[MT_TUNE_HYBRID_WORK1] = 1.3; v[MT_TUNE_HYBRID_WORK2] = 2.6;  // pool quarters / cells: summed cost over the state machine's whole-block cost
    v[MT_TUNE_HYBRID_STARTER_SHARE] = 0.33;  // hybrid launches: state-machine units above this share of an even split start with the launch (hybrid_kernel)
    v[MT_TUNE_DEEP_LAYOUT] = 1.0;
    v[MT_TUNE_MULTI_FORCE_PEER_COPY] = 0.0;
    v[MT_TUNE_MULTI_BALANCE] = 1.0;
    v[MT_TUNE_SM_CELL_SHARE] = 0.8;  // (x the quarters' cutting threshold; blocks with zero-component rays only: mt_order.h)
    v[MT_TUNE_SM_CELL_TIME] = 0.2;
    v[MT_TUNE_SM_CELL_WORK] = 3.0;
    v[MT_TUNE_ORDER_GROUPS] = (double)kOrdGroups;  // workgroups of order_kernel (mt_order.h)
    v[MT_TUNE_XCD_QUEUES] = 2.0;  // one work order per XCD over a 4 x 2 grid of regions of equal forecast cost (L2 hit rate 0.82 -> 0.92 room, 0.66 -> 0.82 loft)
  }
};

// process-wide default of mt_scene_set_engine for scenes created from now on (mt_set_default_engine)
std::atomic<int> g_default_engine{0};

}  // namespace

struct mt_scene {
  int device = 0;
  DevScene dev{};
  std::vector<void *> allocs;  // everything to hipFree
  mt_light *d_lights = nullptr;
  int lights_cap = 0;
  unsigned long long *d_counters = nullptr;
  unsigned int *d_work = nullptr;
  unsigned int *d_queues = nullptr;  // kQueueWords: the per-XCD work orders' counters and bounds (RenderParams::queues)
  unsigned int *d_order_ctl = nullptr;  // kOrdWords: the order kernels' sums, histograms, grids (zero at creation, never reset by the host)
  unsigned int *d_order_whist = nullptr; // [kOrdGroupsMax][kOrdKeysMax]
  unsigned int *d_item_unit = nullptr;
  size_t item_unit_bytes = 0;
  unsigned order_epoch = 0;             // order_kernel launches of this scene so far
  mt_sensor irr_sensor{};               // the sensor `irr_sensor_has` was found for (image irr_w x irr_h)
  int irr_w = 0, irr_h = 0, irr_sensor_has = 0;
  bool irr_sensor_valid = false;
  DevScene dev_uploaded;                // what d_dev holds
  bool dev_uploaded_valid = false;
  unsigned short *d_item_cell = nullptr;
  size_t item_cell_bytes = 0;
  double *d_frames = nullptr;      // throughput engine: recursion frames
  size_t frames_bytes = 0;
  int32_t *d_hit_prim = nullptr;   // launch 1 -> launch 2 hand-off (per pixel)
  size_t hit_prim_bytes = 0;
  double *d_hit_t = nullptr;
  size_t hit_t_bytes = 0;
  unsigned int *d_class_list = nullptr;  // [3][n_items]
  size_t class_list_bytes = 0;
  char *d_pool = nullptr;          // latency engine: ray pool scratch of every wave (mt_pool.h)
  size_t pool_bytes = 0;
  int engine = 0;                  // 0 = automatic, 1 = throughput (state machine), 2 = latency (ray pool)
  int last_engine = 0;             // engine of the previous launch (cost histories are per engine)
  // cost feedback (schedule_kernel): valid for launches of the same geometry
  unsigned int *d_item_cost = nullptr;   // [n_items]
  size_t item_cost_bytes = 0;
  unsigned int *d_item_forecast = nullptr;  // [n_items]
  size_t item_forecast_bytes = 0;
  unsigned int *d_item_forms = nullptr;     // [2 n_items] cost of a block as one unit / as four quarters (forecast_kernel)
  size_t item_forms_bytes = 0;
  unsigned char *d_item_form = nullptr;     // [n_items] hybrid launches: how each block was rendered (hybrid_schedule_kernel)
  size_t item_form_bytes = 0;
  mt_sensor cost_sensor{};               // camera of the launch that measured the costs
  unsigned int *d_order_item = nullptr;  // [4 n_items]
  size_t order_item_bytes = 0;
  signed char *d_order_sub = nullptr;    // [4 n_items]
  size_t order_sub_bytes = 0;
  unsigned long long cost_signature = 0;  // 0 = no history
  bool use_history = true;
  uint8_t *d_rgb = nullptr;
  size_t rgb_bytes = 0;
  mt_debug_px *d_debug = nullptr;
  size_t debug_bytes = 0;
  std::vector<mt_light> lights_host;  // what d_lights holds
  int forecasts_in_a_row = 0;  // launches with this geometry and camera whose work order came from a forecast
  int waves_per_block = 4;
  int deep = 0;                    // which DEEP instantiations of the kernels: 0, 1, 2 (mt_device.h, deep_layout)
  char *d_deep = nullptr;          // their per-wave areas
  size_t deep_bytes = 0;
  size_t lds_bytes = 0;
  int grid_blocks = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // per-launch kernel timing (mt_scene_kernel_times): [i][0] before the primary
  // kernel, [1] between the two kernels, [2] after the render kernel
  static constexpr int kTimedLaunches = 64;
  hipEvent_t ev_k[kTimedLaunches][3] = {};
  unsigned long long launches_timed = 0, launches_read = 0;
  int n_cu = 0;
  bool stats_enabled = true;
  unsigned long long *hb_host = nullptr;  // MT_DEBUG_HEARTBEAT: pinned, device-visible
  unsigned long long *d_prof = nullptr;   // -DMT_PROF build: phase cycle sums
  DevScene *d_dev = nullptr;              // device copy of `dev` (DevScene::self)
  Tuning tune;
  // multi-GPU frames with a moving camera: every rank's costs of the previous frame (mt_scene_import_costs_device)
  RenderParams last_P{};                  // geometry of the last launch (mt_scene_export_costs_device)
  bool last_P_valid = false;
  unsigned int *d_cost_map = nullptr;
  size_t cost_map_bytes = 0;
  int cost_map_w = 0, cost_map_h = 0;
  unsigned long long cost_map_for_launch = ~0ull;  // the map describes the frame of launch number ... (launches_timed then)
  int tree_depth_levels = 0, n_tris_total = 0, n_nodes_total = 0;  // for MT_TUNE_PACKED_STACK
  // -DMT_DEBUG_KNOBS builds only (read from the environment once, in mt_scene_create)
  std::string dbg_item_cycles, dbg_timeline;
  int dbg_print_units = 0;
  // mt_render_frame_multi: this scene's tile buffer and stream, the frame + gathered tiles on the first scene's GPU
  uint8_t *d_multi_tiles = nullptr;
  size_t multi_tiles_bytes = 0;
  uint8_t *d_multi_gather = nullptr;
  size_t multi_gather_bytes = 0;
  uint8_t *d_multi_frame = nullptr;
  size_t multi_frame_bytes = 0;
  hipStream_t multi_stream = nullptr;
  hipEvent_t multi_done = nullptr;
  unsigned int *d_multi_map = nullptr;    // this replica's block costs / the combined map on its way back
  size_t multi_map_bytes = 0;
  unsigned int *d_multi_maps = nullptr;   // first replica: all replicas' maps
  size_t multi_maps_bytes = 0;
  hipEvent_t multi_comb_done = nullptr;
  // mt_render_chunk (host buffers): page-locked staging for the frame on its way to the caller's buffer (copied in
  // pieces, each piece's host copy under the next piece's DMA), pinned words for the counters
  uint8_t *h_stage = nullptr;
  size_t stage_bytes = 0;
  static constexpr int kStagePieces = 16;
  hipEvent_t ev_stage[kStagePieces] = {};
  unsigned long long *h_counters = nullptr;
  // tile-list launches (mt_render_tile_list_device): the launch's own copy of the list, and tile -> slot
  int32_t *d_tile_list = nullptr;
  size_t tile_list_bytes = 0;
  int32_t *d_tile_slot = nullptr;
  size_t tile_slot_bytes = 0;
  // mt_order_tiles_device: summed block costs per tile
  unsigned long long *d_tile_cost = nullptr;
  size_t tile_cost_bytes = 0;
  // mt_render_frame_multi, cost-balanced ownership: this replica's order and list; on the first replica every replica's list
  int32_t *d_multi_order = nullptr;
  size_t multi_order_bytes = 0;
  int32_t *d_multi_list = nullptr;
  size_t multi_list_bytes = 0;
  int32_t *d_multi_lists = nullptr;
  size_t multi_lists_bytes = 0;
  unsigned long long multi_geom = 0;      // geometry (image, tiles, depth, replicas) the state below belongs to
  unsigned long long multi_list_id = 0;   // changes whenever the tiles are dealt out anew
  bool multi_have_map = false;            // d_multi_map holds the combined costs of the previous frame of multi_geom
  int multi_frames_at_rest = 0;
  int multi_rank = -1;
  mt_sensor multi_sensor{};
};

namespace {

template <typename T>
int upload(mt_scene *s, const T *host, size_t count, const T **dev_out) {
  T *d = nullptr;
  const size_t bytes = (count ? count : 1) * sizeof(T);
  HIP_TRY(hipMalloc((void **)&d, bytes));
  s->allocs.push_back(d);
  if (count) HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
  *dev_out = d;
  return MT_OK;
}

bool finite3(const double *p, size_t n) {
  for (size_t i = 0; i < n; i++) {
    if (!std::isfinite(p[i])) return false;
  }
  return true;
}

// Launch geometry: workgroups of 4 waves -- or of 2 or 1 where that puts more waves on a CU: a deep octree's traversal
// frames (27 KB per wave at 16 levels) let one 4-wave workgroup fill two thirds of the LDS and leave room for a fifth
// wave only as a workgroup of its own.
int configure_launch(mt_scene *s) {
  s->deep = s->tune.v[MT_TUNE_DEEP_LAYOUT] != 0.0 ? deep_layout(s->dev.tree_depth) : 0;
  const size_t per_wave = wave_stack_bytes(s->dev.tree_depth, s->dev.pack_shift != 0, s->deep != 0);
  if (per_wave > kLdsBudget) {
    return fail(MT_ERR_UNSUPPORTED, "octree depth %d needs %zu B of LDS per wave (> %zu)",
                s->dev.tree_depth, per_wave, kLdsBudget);
  }
  // The attribute is per function AND per device: keep it, for every device, at the largest size any scene
  // of this process needs there (a shallower scene must not lower it; scenes may be created from several threads).
  auto set_attribute = [&](size_t bytes) -> int {
    static std::mutex mu;
    static std::map<int, size_t> lds_attr;  // device -> bytes set
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = lds_attr[s->device];
    if (bytes > have) {
#define MT_KERNELS_OF(D_)                                                                                  \
  (const void *)render_kernel<true, D_>, (const void *)render_kernel<false, D_>,                              \
  (const void *)primary_kernel<true, D_>, (const void *)primary_kernel<false, D_>,                            \
  (const void *)pool_kernel<true, D_>, (const void *)pool_kernel<false, D_>,                                  \
  (const void *)hybrid_kernel<true, D_>, (const void *)hybrid_kernel<false, D_>,                              \
  (const void *)probe_kernel<D_>, (const void *)intersect_kernel<D_>
      const void *kernels[] = {MT_KERNELS_OF(0), MT_KERNELS_OF(1), MT_KERNELS_OF(2)};
#undef MT_KERNELS_OF
      for (const void *k : kernels) {
        HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      }
      have = bytes;
    }
    return MT_OK;
  };
  int best_wpb = 0, best_per_cu = 0;
  for (int wpb = 4; wpb >= 1; wpb >>= 1) {
    if (per_wave * wpb > kLdsBudget) continue;
    int rc = set_attribute(per_wave * wpb);
    if (rc != MT_OK) return rc;
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, s->deep == 2 ? (const void *)render_kernel<true, 2> : (s->deep == 1 ? (const void *)render_kernel<true, 1> : (const void *)render_kernel<true, 0>), wpb * 64, per_wave * wpb));
    if (per_cu < 1) per_cu = 1;
    if (per_cu * wpb > 16) per_cu = 16 / wpb;  // more waves only add divergence state
    if (per_cu * wpb > best_per_cu * best_wpb) {  // (ties: the larger workgroup, tried first)
      best_wpb = wpb;
      best_per_cu = per_cu;
    }
  }
  int per_cu = best_per_cu;
  s->waves_per_block = best_wpb;
  s->lds_bytes = per_wave * best_wpb;
  {  // occupancy experiments
    const int v = (int)s->tune.v[MT_TUNE_BLOCKS_PER_CU];
    if (v >= 1 && v < per_cu) per_cu = v;
  }
  s->grid_blocks = s->n_cu * per_cu;
  return MT_OK;
}

int ensure_bytes(void **ptr, size_t *have, size_t need) {
  if (*have >= need && *ptr) return MT_OK;
  if (*ptr) HIP_TRY(hipFree(*ptr));
  *ptr = nullptr;
  *have = 0;
  HIP_TRY(hipMalloc(ptr, need ? need : 1));
  *have = need;
  return MT_OK;
}

// launches kernel<STATS, DEEP> (or kernel<DEEP>) for the scene's layout
#define MT_LAUNCH_SD(kernel, stats, grid_, block_, lds_, stream_, ...)                                              \
  do {                                                                                                                \
    if (s->deep == 2) {                                                                                               \
      if (stats) hipLaunchKernelGGL((kernel<true, 2>), grid_, block_, lds_, stream_, __VA_ARGS__);                    \
      else hipLaunchKernelGGL((kernel<false, 2>), grid_, block_, lds_, stream_, __VA_ARGS__);                         \
    } else if (s->deep == 1) {                                                                                        \
      if (stats) hipLaunchKernelGGL((kernel<true, 1>), grid_, block_, lds_, stream_, __VA_ARGS__);                    \
      else hipLaunchKernelGGL((kernel<false, 1>), grid_, block_, lds_, stream_, __VA_ARGS__);                         \
    } else {                                                                                                          \
      if (stats) hipLaunchKernelGGL((kernel<true, 0>), grid_, block_, lds_, stream_, __VA_ARGS__);                    \
      else hipLaunchKernelGGL((kernel<false, 0>), grid_, block_, lds_, stream_, __VA_ARGS__);                         \
    }                                                                                                                 \
  } while (0)
#define MT_LAUNCH_D(kernel, grid_, block_, lds_, stream_, ...)                                                      \
  do {                                                                                                                \
    if (s->deep == 2) hipLaunchKernelGGL((kernel<2>), grid_, block_, lds_, stream_, __VA_ARGS__);                     \
    else if (s->deep == 1) hipLaunchKernelGGL((kernel<1>), grid_, block_, lds_, stream_, __VA_ARGS__);                \
    else hipLaunchKernelGGL((kernel<0>), grid_, block_, lds_, stream_, __VA_ARGS__);                                  \
  } while (0)

// the per-wave global areas of the DEEP instantiations, for a launch of `waves` waves
int ensure_deep(mt_scene *s, size_t waves) {
  if (!s->deep) return MT_OK;
  const size_t stride = (wave_deep_bytes(s->dev.tree_depth) + 255) & ~(size_t)255;
  int rc = ensure_bytes((void **)&s->d_deep, &s->deep_bytes, stride * waves);
  if (rc != MT_OK) return rc;
  s->dev.deep_base = s->d_deep;
  s->dev.deep_stride = stride;
  return MT_OK;
}

int launch_render(mt_scene *s, const mt_sensor *sensor, int image_w, int image_h, int rx, int ry,
                  int rw, int rh, int tile_w, int tile_h, int first_tile, int tile_stride,
                  int n_tiles, int max_depth, uint8_t *d_rgb, mt_debug_px *d_debug,
                  hipStream_t stream, const int32_t *d_list = nullptr, unsigned long long list_id = 0) {
  if (max_depth < 0 || max_depth > MT_MAX_RECURSION) {
    return fail(MT_ERR_ARG, "max_depth %d outside [0, %d]", max_depth, MT_MAX_RECURSION);
  }
  RenderParams P{};
  P.sensor = *sensor;
  P.image_w = image_w;
  P.image_h = image_h;
  P.region_x = rx; P.region_y = ry; P.region_w = rw; P.region_h = rh;
  P.tile_w = tile_w; P.tile_h = tile_h;
  P.tiles_x = (rw + tile_w - 1) / tile_w;
  P.first_tile = first_tile; P.tile_stride = tile_stride; P.n_tiles = n_tiles;
  P.blocks_x = (tile_w + 7) / 8;
  P.blocks_y = (tile_h + 7) / 8;
  P.max_depth = max_depth;
  const unsigned long long items = (unsigned long long)n_tiles * P.blocks_x * P.blocks_y;
  if (items > 0xfffffff0ull) return fail(MT_ERR_ARG, "too many work items (%llu)", items);
  P.n_items = (unsigned)items;
  P.out_rgb = d_rgb;
  P.out_debug = d_debug;
  P.counters = s->d_counters;
  P.work_counter = s->d_work;
  const size_t waves = (size_t)s->grid_blocks * s->waves_per_block;
  // The block costs of the previous launch are a valid forecast when that
  // launch had the same geometry (an animation frame, main_local.cc:79-110, or a
  // repeated benchmark step) -- whichever engine measured them.  Then the blocks
  // are handed out longest first, the longest in pieces.  Otherwise: the ray
  // pool forecasts from 1/16 of the primary rays (probe_kernel); the state
  // machine classifies the blocks by material in a launch of its own
  // (primary_kernel).
  unsigned long long sig = 1469598103934665603ull;
  {
    const long long key[] = {image_w, image_h, rx, ry, rw, rh, tile_w, tile_h, first_tile, tile_stride,
                             n_tiles, max_depth, s->dev.n_lights, d_list ? 1 : 0, d_list ? (long long)list_id : 0};
    for (long long v : key) {
      sig = (sig ^ (unsigned long long)v) * 1099511628211ull;
    }
    if (sig == 0) sig = 1;
  }
  // (a tile list promises to be the previous launch's list by its non-zero list_id only)
  bool have_costs = s->use_history && s->cost_signature == sig && !(d_list != nullptr && list_id == 0);
  // A list launch with ANOTHER list (the tiles were dealt out anew): the slots' cost words belong to other tiles, but the
  // frame-wide map imported since the previous launch has every block's cost by image position.
  const bool map_ready = s->d_cost_map != nullptr && s->cost_map_for_launch == s->launches_timed &&
                         s->cost_map_w >= (image_w + 7) / 8 && s->cost_map_h >= (image_h + 7) / 8;
  const bool from_map = s->use_history && d_list != nullptr && !have_costs && map_ready && s->last_P_valid &&
                        (tile_w & 7) == 0 && (tile_h & 7) == 0 && (rx & 7) == 0 && (ry & 7) == 0 &&
                        s->last_P.image_w == image_w && s->last_P.image_h == image_h && s->last_P.max_depth == max_depth;
  if (from_map) have_costs = true;
  // ---- which engine?  Both compute every pixel with the same operations in the
  // same order (tests render through both).  The state machine (one lane per
  // pixel, its context in registers) has the lower cost per ray and is the
  // default; the ray pool has the shorter chain of dependent passes per pixel
  // and takes over when a launch is bound by its longest work unit rather than
  // by its amount of work, i.e. when there are few blocks per wave (a rank's
  // share of a multi-GPU frame, a small chunk).
  int engine = s->engine;
  const bool engine_auto = engine != 1 && engine != 2 && engine != 3;
  // Ray pool: records per wave.  64 * (2^(max_depth+1) - 1) is every call of every pixel's recursion tree at once;
  // beyond kPoolCapMax the kernel's throttle keeps the pool within the capacity (depth first).  A record grows with
  // the number of lights (160 + 80 n bytes), so the capacity shrinks with it -- down to the floor the throttle needs
  // -- to keep the scratch of all resident waves within MT_TUNE_POOL_SCRATCH_MB.
  const int n_l = s->dev.n_lights;
  constexpr int kPoolMaxLights = 254;  // a pool entry has 8 bits for (light + 1)
  long long pool_cap = 0;
  size_t pool_stride = 0;
  bool pool_fits = n_l <= kPoolMaxLights, pool_roomy = pool_fits;
  if (pool_fits) {
    constexpr long long kPoolCapMax = 1024;
    const long long all = 64ll * ((2ll << max_depth) - 1);
    const long long floor_cap = 64 + 128 + 4ll * (max_depth + 1) + 64;
    pool_cap = all < kPoolCapMax ? all : kPoolCapMax;
    const size_t rec_bytes = (size_t)(kRecFixed + kLightSlot * n_l) * sizeof(double);
    const size_t per_rec = rec_bytes + (size_t)(n_l > 0 ? n_l : 1) * 4 + 4;
    const double budget = s->tune.v[MT_TUNE_POOL_SCRATCH_MB] * 1048576.0;
    const long long fit = (long long)(budget / ((double)per_rec * (double)(waves ? waves : 1)));
    if (pool_cap > fit) {  // automatic mode: such a launch goes to the state machine; engine 2 by request: a smaller pool
      pool_roomy = false;
      pool_cap = fit;
    }
    if ((long long)s->tune.v[MT_TUNE_POOL_CAP] > 0) pool_cap = (long long)s->tune.v[MT_TUNE_POOL_CAP];  // tests: force the depth-first throttle
    if (pool_cap > 65535) pool_cap = 65535;  // a pool entry holds the record number in 16 bits
    if (pool_cap < floor_cap) {
      pool_fits = (double)floor_cap * (double)per_rec * (double)(waves ? waves : 1) <= 4.0 * budget;
      pool_cap = floor_cap;
    }
    pool_stride = ((size_t)pool_cap * per_rec + 255) & ~(size_t)255;
  }
  if (engine_auto) {
    // blocks per wave below which a launch is taken to be tail-bound (one rank's share of the 4K frame
    // at N = 8 has 7.9 per wave: ray pool 4.4 ms, state machine 4.7; at N = 4, 15.8: 7.3 against 6.6)
    const float per_wave = (float)s->tune.v[MT_TUNE_POOL_BELOW];
    // ... and the first frame of a geometry: without measured costs the order
    // of the work is a guess, and the pool's short pixel chains forgive a bad
    // guess (12 ms against the state machine's 14.5 on the 1080p frame).
    // The state machine has no limit on lights or scratch: it takes what the pool cannot hold.
    // With measured costs such a launch goes to the HYBRID kernel (engine 3): only its longest blocks go through the
    // pool, in pieces; the rest keeps the state machine's cost per ray (one rank's share of the 4K frame at N = 8:
    // 3.63 ms against the pool's 3.88 and the state machine's 4.36; at N = 4 the state machine alone is ahead, 5.71
    // against 5.80).  It has no debug-buffer path: such launches stay with the pool.
    const bool small = (float)P.n_items < per_wave * (float)waves;
    engine = !(pool_fits && pool_roomy) ? 1 : (!have_costs ? 2 : (small ? (d_debug == nullptr ? 3 : 2) : 1));
  }
  // Engine 3 (hybrid: the longest blocks through the ray pool in pieces, the rest through the state machine, one
  // kernel) needs measured costs to tell the two kinds apart and has no debug-buffer path; a launch without either
  // is rendered by the ray pool (or the state machine where the pool does not fit).
  if (engine == 3 && !(have_costs && pool_fits && d_debug == nullptr)) engine = pool_fits ? 2 : 1;
  if (engine == 2 && !pool_fits) {
    return fail(MT_ERR_UNSUPPORTED, "the ray pool (engine 2) holds at most %d lights within its scratch budget; "
                "%d were set -- engine 0 (automatic) or 1 renders such scenes", kPoolMaxLights, n_l);
  }
  const bool pool_engine = engine == 2, hybrid = engine == 3;
  const bool history = have_costs && (pool_engine || d_debug == nullptr);  // (hybrid: both hold, see above)
  P.from_primary = history ? 0 : 1;
  {
    int rc = ensure_bytes((void **)&s->d_item_cost, &s->item_cost_bytes, (size_t)P.n_items * 4);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_item_forecast, &s->item_forecast_bytes, (size_t)P.n_items * 4);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_item_forms, &s->item_forms_bytes, (size_t)P.n_items * 8);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_order_item, &s->order_item_bytes, (size_t)P.n_items * 64);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_order_sub, &s->order_sub_bytes, (size_t)P.n_items * 16);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_item_form, &s->item_form_bytes, (size_t)P.n_items);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_item_unit, &s->item_unit_bytes, (size_t)P.n_items * 4);
    if (rc != MT_OK) return rc;
  }
  P.item_cost = s->d_item_cost;
  P.item_forecast = s->d_item_forecast;
  P.item_whole = s->d_item_forms;
  P.item_qsum = s->d_item_forms + P.n_items;
  if (s->tune.v[MT_TUNE_FORMS] == 0.0) P.item_whole = P.item_qsum = nullptr;
  P.order_item = s->d_order_item;
  P.order_sub = s->d_order_sub;
  P.n_work = s->d_work + 7;
  // the combined cost map of all ranks, if one was imported after the previous launch (else nullptr: own costs only)
  P.cost_map = (s->d_cost_map != nullptr && s->cost_map_for_launch == s->launches_timed) ? s->d_cost_map : nullptr;
  P.cost_map_w = s->cost_map_w;
  P.cost_map_h = s->cost_map_h;
  P.from_map = from_map ? 1 : 0;
  if (from_map) P.item_whole = P.item_qsum = nullptr;  // (the two measured forms of a block are kept per slot)
  if (d_list != nullptr && n_tiles > 0) {
    const int tiles_total = P.tiles_x * ((rh + tile_h - 1) / tile_h);
    int rc = ensure_bytes((void **)&s->d_tile_list, &s->tile_list_bytes, (size_t)n_tiles * 4);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_tile_slot, &s->tile_slot_bytes, (size_t)tiles_total * 4);
    if (rc != MT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_tile_list, d_list, (size_t)n_tiles * 4, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(tile_slot_kernel, dim3((tiles_total + 255) / 256), dim3(256), 0, stream, s->d_tile_list, n_tiles, s->d_tile_slot, tiles_total, 0);
    hipLaunchKernelGGL(tile_slot_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, s->d_tile_list, n_tiles, s->d_tile_slot, tiles_total, 1);
    HIP_TRY(hipGetLastError());
    P.tile_list = s->d_tile_list;
    P.tile_slot = s->d_tile_slot;
  }
  if (pool_engine || hybrid) {
    int rc = ensure_bytes((void **)&s->d_pool, &s->pool_bytes, pool_stride * waves);
    if (rc != MT_OK) return rc;
    P.pool_scratch = s->d_pool;
    P.pool_stride = pool_stride;
    P.pool_cap = (int)pool_cap;
    P.prio_units = (unsigned)(s->n_cu * 4);  // one per SIMD
  }
  if (!pool_engine) {
    size_t fbytes = waves * ((size_t)(max_depth > 0 ? max_depth : 1) * kFrameSlots + kParkSlots) * 64 * sizeof(double);
    const size_t slots_px = (size_t)n_tiles * (size_t)tile_w * (size_t)tile_h;
    int rc = ensure_bytes((void **)&s->d_frames, &s->frames_bytes, fbytes);
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_hit_prim, &s->hit_prim_bytes, slots_px * sizeof(int32_t));
    if (rc == MT_OK) rc = ensure_bytes((void **)&s->d_hit_t, &s->hit_t_bytes, slots_px * sizeof(double));
    if (rc == MT_OK) {
      rc = ensure_bytes((void **)&s->d_class_list, &s->class_list_bytes,
                        3 * (size_t)P.n_items * sizeof(unsigned int));
    }
    if (rc != MT_OK) return rc;
    P.frames = s->d_frames;
    P.hit_prim = s->d_hit_prim;
    P.hit_t = s->d_hit_t;
    P.class_list = s->d_class_list;
    P.class_count = s->d_work + 4;  // d_work: [0..1] work counters, [4..6] class counts
  }
  P.item_cycles = nullptr;
  unsigned long long *d_item = nullptr;
  const char *item_dump = s->dbg_item_cycles.empty() ? nullptr : s->dbg_item_cycles.c_str();
  if (item_dump && P.n_items > 0) {
    HIP_TRY(hipMalloc((void **)&d_item, (size_t)P.n_items * 16 * 16 * 4));
    HIP_TRY(hipMemset(d_item, 0, (size_t)P.n_items * 16 * 16 * 4));
    P.item_cycles = d_item;
  }
  if (P.n_items == 0) return MT_OK;
  // (launches with a work order: order_kernel zeroes the counters on its way)
  if (!history && !pool_engine) HIP_TRY(hipMemsetAsync(s->d_work, 0, 16 * sizeof(unsigned), stream));
  P.order_ctl = s->d_order_ctl;
  P.order_whist = s->d_order_whist;
  P.item_unit = s->d_item_unit;
  // one work order per XCD: state-machine launches with a cost history only (the other engines keep the one order)
  // (measured and left out: first frames through the ray pool -- room 8.4 -> 8.9 ms, loft 19.7 -> 20.9: the probe's guess
  // balances the regions too roughly, and such a frame ends with its longest units either way --; the state machine's
  // part of hybrid launches, i.e. a rank's share of a frame -- mean of eight ranks' 4K shares 2.71 -> 2.77 ms)
  P.queues = (s->tune.v[MT_TUNE_XCD_QUEUES] != 0.0 && history && !pool_engine && !hybrid) ? s->d_queues : nullptr;
  if (P.queues) {
    int rc = ensure_bytes((void **)&s->d_item_cell, &s->item_cell_bytes, (size_t)P.n_items * 2);
    if (rc != MT_OK) return rc;
    P.item_cell = s->d_item_cell;
  }
  const dim3 grid(s->grid_blocks), block(s->waves_per_block * 64);
  {  // (the probe's grid follows the number of blocks: 16 of them per wave)
    const size_t probe_waves = ((size_t)4 * P.n_items + block.x - 1) / block.x * s->waves_per_block;
    int rc = ensure_deep(s, std::max(waves, probe_waves));
    if (rc != MT_OK) return rc;
  }
  if (!s->dev_uploaded_valid || memcmp(&s->dev_uploaded, &s->dev, sizeof(DevScene)) != 0) {  // (nearly never: the scene description changes with the lights, the traversal mode, the layout)
    HIP_TRY(hipMemcpyAsync(s->d_dev, &s->dev, sizeof(DevScene), hipMemcpyHostToDevice, stream));
    memcpy(&s->dev_uploaded, &s->dev, sizeof(DevScene));
    s->dev_uploaded_valid = true;
  }
  // events: [0] -> [1] forecast / classification + work order; [1] -> [2] the frame kernel
  hipEvent_t *ek = s->ev_k[s->launches_timed % mt_scene::kTimedLaunches];
  for (int i = 0; i < 3; i++) {
    if (!ek[i]) HIP_TRY(hipEventCreate(&ek[i]));
  }
  HIP_TRY(hipEventRecord(ek[0], stream));
  // Has the camera moved since the costs were measured?  Then forecast_kernel
  // re-projects them (radius 1 block; 2 when the origin moved too: parallax).
  int reproject = 0, radius = 0;
  float blend = 0.0f;  // see forecast_kernel
  // Did the frame that measured the costs have pixels whose primary rays had a zero direction component?  Found on the
  // host, exactly: per scanline and component the direction is r + dp x with r = start + ds y (the kernels' own
  // expression, Sensor::GetRay), zero for at most the pixels next to -r / dp -- a whole column or row for a camera on
  // an axis, isolated pixels for one with roll or pitch.  Those blocks' costs are skipped by a re-projected forecast
  // (forecast_kernel).
  auto has_zero_component_pixel = [&](const mt_sensor &o) -> int {
    for (int k = 0; k < 3; k++) {
      for (int y = 0; y < image_h; y++) {
        const double r = o.start_point[k] + o.delta_scanline[k] * (double)y;
        if (o.delta_pixel[k] == 0.0 || !std::isfinite(r / o.delta_pixel[k])) {
          if (r + o.delta_pixel[k] * 0.0 == 0.0) return 1;
          continue;
        }
        const double x0 = std::nearbyint(-r / o.delta_pixel[k]);
        for (int dx = -1; dx <= 1; dx++) {
          const double x = x0 + dx;
          if (x >= 0.0 && x < (double)image_w && r + o.delta_pixel[k] * x == 0.0) return 1;
        }
      }
    }
    return 0;
  };
  int old_irr = 0, new_irr = 0;
  if (history) {  // (kept per sensor: a camera at rest is looked at once)
    if (!s->irr_sensor_valid || memcmp(&s->irr_sensor, sensor, sizeof(mt_sensor)) != 0 || s->irr_w != image_w || s->irr_h != image_h) {
      s->irr_sensor = *sensor; s->irr_w = image_w; s->irr_h = image_h;
      s->irr_sensor_has = has_zero_component_pixel(*sensor);
      s->irr_sensor_valid = true;
    }
    new_irr = s->irr_sensor_has;
  }
  if (history && memcmp(&s->cost_sensor, sensor, sizeof(mt_sensor)) != 0) {
    old_irr = has_zero_component_pixel(s->cost_sensor);
    reproject = 1;
    radius = memcmp(s->cost_sensor.origin, sensor->origin, sizeof sensor->origin) != 0 ? 2 : 1;
    if (s->tune.v[MT_TUNE_FORECAST_RADIUS] >= 0.0) radius = (int)s->tune.v[MT_TUNE_FORECAST_RADIUS];
  }
  if (history && !reproject && !from_map && s->forecasts_in_a_row > 0) {
    // swept (scripts/blend_sweep.py, state machine, 64 frames): 0 -> every other frame 6 % slower (mean 7.09 ms), 0.5 -> one
    // in three (7.03), 0.9 -> one in eight (7.01); a frozen forecast (1.0) repeats its frame time to 0.2 % (scripts/alternation.py)
    // (a running mean of the measurements first -- 1/2, 2/3, ... -- so that the first frames' costs, measured under a
    // guessed order, do not linger)
    const float cap = (float)s->tune.v[MT_TUNE_BLEND];
    blend = std::min(cap, (float)s->forecasts_in_a_row / (float)(s->forecasts_in_a_row + 1));
  }
  // (the two measurements of a block belong to ONE camera, geometry and set of lights)
  if (s->forecasts_in_a_row == 0) HIP_TRY(hipMemsetAsync(s->d_item_forms, 0, (size_t)P.n_items * 8, stream));
  // (a forecast made from the OTHER engine's costs -- the frame after a first frame -- does not count: the next one
  // starts the running mean with this engine's own measurement)
  s->forecasts_in_a_row = (history && !reproject && !from_map && s->last_engine == engine) ? s->forecasts_in_a_row + 1 : 0;
  if (pool_engine) {
    if (!history) {
      MT_LAUNCH_D(probe_kernel, dim3((4 * P.n_items + block.x - 1) / block.x), block, s->lds_bytes, stream, s->dev, P);
      HIP_TRY(hipGetLastError());
    }
    // blocks above cut_share of an even split of the frame are handed out in pieces
    const double *tv = s->tune.v;
    SchedParams sp{history ? 1.0f : 0.3f, {1.0f, (float)tv[MT_TUNE_POOL_PIECE_TIME1], (float)tv[MT_TUNE_POOL_PIECE_TIME2]},
                   {1.0f, (float)tv[MT_TUNE_POOL_PIECE_WORK1], (float)tv[MT_TUNE_POOL_PIECE_WORK2]},
                   (float)tv[MT_TUNE_POOL_CELL_FACTOR],
                   (!from_map && (!history || s->last_engine == 2)) ? 1 : 0};  // a forecast is cut more eagerly (own_costs: granularity in bits 30-31)
    if (tv[MT_TUNE_POOL_CUT_SHARE] >= 0.0) sp.cut_share = (float)tv[MT_TUNE_POOL_CUT_SHARE];
    ForecastArgs fa{s->cost_sensor, reproject, radius, (history && (s->last_engine == 1 || s->last_engine == 3)) ? 0 : 1,
                    (history && s->last_engine == 1) ? 1.7f : sp.piece_work[1], sp.piece_work[2], 16000u, blend,
                    (history && s->last_engine == 3) ? s->d_item_form : nullptr, (float)tv[MT_TUNE_HYBRID_WORK1],
                    (float)tv[MT_TUNE_HYBRID_WORK2], (float)tv[MT_TUNE_FORECAST_STEP], (float)tv[MT_TUNE_SM_CELL_WORK], old_irr, new_irr};
    OrderArgs oa{};
    oa.n_waves = s->grid_blocks * s->waves_per_block;
    oa.epoch = s->order_epoch++;
    const int ord_groups = std::max(1, std::min((int)s->tune.v[MT_TUNE_ORDER_GROUPS], kOrdGroupsMax));
    oa.sp = sp;
    hipLaunchKernelGGL(order_forecast_kernel<2>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, fa, oa);
        hipLaunchKernelGGL(order_count_kernel<2>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, oa);
        hipLaunchKernelGGL(order_scatter_kernel<2>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, oa);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ek[1], stream));
    MT_LAUNCH_SD(pool_kernel, s->stats_enabled, grid, block, s->lds_bytes, stream, s->dev, P);
  } else {
    if (history) {
      // blocks above this share of an even split are cut into quarters; a re-projected forecast (moving camera) is
      // cut more eagerly -- it is a neighbourhood maximum of stale costs (swept, scripts/quad_sweep.py: repeated frame
      // 0.6 / 0.8 / 1.0 -> 7.18 / 6.84 / 7.28 ms, moving camera 6.69 / 7.20 / 9.89; with work 1.5: share 0.7 -> 6.40)
      // (with the per-block ratio of the two forms' costs -- forecast_kernel -- the repeated frame no longer alternates,
      // and re-swept: 0.8 / 0.9 / 0.95 / 1.0 / 1.05 / 1.1 -> 6.67 / 6.49 / 6.47 / 6.46 / 6.56 / 6.93 ms)
      const float quad_share = (float)s->tune.v[reproject ? MT_TUNE_QUAD_SHARE_MOVING : MT_TUNE_QUAD_SHARE];  // 0.7 / 0.95
      // ... and stay so above this fraction of that threshold (1 = no hysteresis: swept, scripts/quad_sweep.py --
      // settings that steady the repeated frame cost the moving camera 50 %)
      const float quad_keep = (float)s->tune.v[MT_TUNE_QUAD_KEEP];
      // work of a block rendered as quarters / rendered whole (swept with the share): 1.5 / 1.7
      const float quad_work = (float)s->tune.v[reproject ? MT_TUNE_QUAD_WORK_MOVING : MT_TUNE_QUAD_WORK];
      const double *tv = s->tune.v;
      ForecastArgs fa{s->cost_sensor, reproject, radius, s->last_engine == 2 ? 1 : 0, s->last_engine == 2 ? 1.1f : quad_work, 3.0f,
                      16000u, blend, s->last_engine == 3 ? s->d_item_form : nullptr, (float)tv[MT_TUNE_HYBRID_WORK1],
                      (float)tv[MT_TUNE_HYBRID_WORK2], (float)tv[MT_TUNE_FORECAST_STEP], (float)tv[MT_TUNE_SM_CELL_WORK], old_irr, new_irr};
      OrderArgs oa{};
      oa.n_waves = s->grid_blocks * s->waves_per_block;
      oa.epoch = s->order_epoch++;
    const int ord_groups = std::max(1, std::min((int)s->tune.v[MT_TUNE_ORDER_GROUPS], kOrdGroupsMax));
      if (hybrid) {
        const float k = reproject ? (float)(tv[MT_TUNE_QUAD_SHARE_MOVING] / tv[MT_TUNE_QUAD_SHARE]) : 1.0f;  // a re-projected forecast is cut more eagerly
        oa.quad_share = k * (float)tv[MT_TUNE_HYBRID_QUAD_SHARE];
        oa.pool_share = k * (float)tv[MT_TUNE_HYBRID_POOL_SHARE];
        oa.piece_time1 = (float)tv[MT_TUNE_POOL_PIECE_TIME1];
        oa.piece_time2 = (float)tv[MT_TUNE_POOL_PIECE_TIME2];
        oa.cell_factor = (float)tv[MT_TUNE_HYBRID_CELL_FACTOR];
        oa.form_out = s->d_item_form;
        oa.starter_share = (float)tv[MT_TUNE_HYBRID_STARTER_SHARE];
        oa.max_starters = (unsigned)std::min(s->grid_blocks, (int)(0.25 * s->grid_blocks * s->waves_per_block));
        hipLaunchKernelGGL(order_forecast_kernel<1>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, fa, oa);
        hipLaunchKernelGGL(order_count_kernel<1>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, oa);
        hipLaunchKernelGGL(order_scatter_kernel<1>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, oa);
      } else {
        oa.quad_share = quad_share;
        oa.quad_keep = quad_keep;
        oa.cell_share = (float)tv[MT_TUNE_SM_CELL_SHARE];
        // (cells only on MEASURED costs: a re-projected forecast of such a block is a guess -- thirty times a mean block --
        // that cannot tell the loft's column, 1.4 frames long as quarters, from the room's, 0.87: room panning +0.5 % with
        // the guess trusted, loft -2 %)
        oa.new_irr = reproject ? 0 : new_irr;
        oa.cell_time = (float)tv[MT_TUNE_SM_CELL_TIME];
        oa.queue_mode = (int)tv[MT_TUNE_XCD_QUEUES];
        hipLaunchKernelGGL(order_forecast_kernel<0>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, fa, oa);
        hipLaunchKernelGGL(order_count_kernel<0>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, oa);
        hipLaunchKernelGGL(order_scatter_kernel<0>, dim3(ord_groups), dim3(kOrdThreads), 0, stream, P, oa);
      }
    } else {
      MT_LAUNCH_SD(primary_kernel, s->stats_enabled, grid, block, s->lds_bytes, stream, s->dev, P);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ek[1], stream));
    if (hybrid) {
      MT_LAUNCH_SD(hybrid_kernel, s->stats_enabled, grid, block, s->lds_bytes, stream, s->dev, P);
    } else {
      MT_LAUNCH_SD(render_kernel, s->stats_enabled, grid, block, s->lds_bytes, stream, s->dev, P);
    }
  }
  HIP_TRY(hipEventRecord(ek[2], stream));
  HIP_TRY(hipGetLastError());
  s->launches_timed++;
  s->last_P = P;
  s->last_P_valid = true;
  s->cost_signature = (d_list != nullptr && list_id == 0) ? 0 : sig;  // the costs now in d_item_cost belong to this geometry and engine
  s->last_engine = engine;
  s->cost_sensor = *sensor;
  if (s->dbg_print_units) {  // -DMT_DEBUG_KNOBS: how many work units did the order have?
    if (s->dbg_print_units == 2) {  // without synchronising: kept in a ring, printed every 16th launch
      static unsigned *ring = nullptr;
      static unsigned long long n = 0;
      if (!ring) HIP_TRY(hipMalloc((void **)&ring, 16 * sizeof(unsigned)));
      HIP_TRY(hipMemcpyAsync(ring + (n % 16), s->d_work + 7, sizeof(unsigned), hipMemcpyDeviceToDevice, stream));
      if (++n % 16 == 0) {
        unsigned host[16];
        HIP_TRY(hipMemcpy(host, ring, sizeof host, hipMemcpyDeviceToHost));
        fprintf(stderr, "[mt units]");
        for (int i = 0; i < 16; i++) fprintf(stderr, " %u", host[i]);
        fprintf(stderr, "\n");
      }
    } else {  // (synchronises!)
      unsigned nw = 0;
      HIP_TRY(hipMemcpy(&nw, s->d_work + 7, sizeof nw, hipMemcpyDeviceToHost));
      fprintf(stderr, "[mt units] %u units for %u blocks\n", nw, P.n_items);
    }
  }
  if (d_item) {  // debug: dump per-item durations (synchronises!)
    std::vector<unsigned long long> host((size_t)P.n_items * 16 * 2 * 4);
    HIP_TRY(hipMemcpy(host.data(), d_item, host.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d_item));
    if (FILE *f = fopen(item_dump, "wb")) {
      fwrite(host.data(), 8, host.size(), f);
      fclose(f);
    }
  }
  if (s->hb_host) {
    // Debug mode: watch the launch from the host and report where it is stuck.
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(stream) == hipErrorNotReady) {
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (sec > 10.0) {
        fprintf(stderr, "[mt heartbeat] kernel still running after %.0f s\n", sec);
        int hist[8] = {0};
        const int nw = s->grid_blocks * s->waves_per_block;
        for (int w = 0; w < nw; w++) {
          const unsigned long long *h = s->hb_host + (size_t)w * 4;
          hist[h[0] & 7]++;
          if ((h[0] & 255) != 5 && (h[0] & 255) != 0) {
            fprintf(stderr, "  wave %d: stage %llu arg %llu exec@fetch %llx alive %llx exec %llx\n", w, h[0] & 255,
                    h[0] >> 8, h[1], h[2], h[3]);
          }
        }
        fprintf(stderr, "  stage histogram:");
        for (int i = 0; i < 8; i++) fprintf(stderr, " %d:%d", i, hist[i]);
        fprintf(stderr, "\n");
        fflush(stderr);
        _exit(86);
      }
    }
  }
  return MT_OK;
}

int check_status(const unsigned long long *c) {
  if (c[ST_STATUS] == DEV_OK) return MT_OK;
  return fail(MT_ERR_INTERNAL, "device loop bound tripped (code %llu): kernel logic error", c[ST_STATUS]);
}

void fill_stats(const unsigned long long *c, mt_stats *st) {
  st->rays_primary = c[ST_RAYS_PRIMARY];
  st->rays_secondary = c[ST_RAYS_SECONDARY];
  st->rays_shadow = c[ST_RAYS_SHADOW];
  st->box_tests = c[ST_BOX_TESTS];
  st->node_visits = c[ST_NODE_VISITS];
  st->tri_tests = c[ST_TRI_TESTS];
  st->mt_tests = c[ST_MT_TESTS];
  st->shaded_hits = c[ST_SHADED_HITS];
  st->wave_node_steps = c[ST_WAVE_NODE_STEPS];
  st->wave_tri_steps = c[ST_WAVE_TRI_STEPS];
  st->bytes_scalar = c[ST_BYTES_SCALAR];
  st->bytes_vector = c[ST_BYTES_VECTOR];
}

int check_image_args(const mt_scene *s, const mt_sensor *sensor, int image_w, int image_h) {
  if (!s) return fail(MT_ERR_ARG, "scene is NULL");
  if (!sensor) return fail(MT_ERR_ARG, "sensor is NULL");
  if (image_w <= 0 || image_h <= 0 || image_w > 100000 || image_h > 100000) {
    return fail(MT_ERR_ARG, "image size %dx%d out of range", image_w, image_h);
  }
  return MT_OK;
}

}  // namespace

extern "C" {

const char *mt_last_error(void) { return g_err; }
int mt_abi_version(void) { return MT_ABI_VERSION; }

int mt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(MT_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

void mt_scene_destroy(mt_scene *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (void *p : s->allocs) (void)hipFree(p);
  if (s->d_pool) (void)hipFree(s->d_pool);
  if (s->d_frames) (void)hipFree(s->d_frames);
  if (s->d_hit_prim) (void)hipFree(s->d_hit_prim);
  if (s->d_hit_t) (void)hipFree(s->d_hit_t);
  if (s->d_class_list) (void)hipFree(s->d_class_list);
  if (s->d_item_cost) (void)hipFree(s->d_item_cost);
  if (s->d_item_forecast) (void)hipFree(s->d_item_forecast);
  if (s->d_item_forms) (void)hipFree(s->d_item_forms);
  if (s->d_item_form) (void)hipFree(s->d_item_form);
  if (s->d_cost_map) (void)hipFree(s->d_cost_map);
  if (s->d_order_item) (void)hipFree(s->d_order_item);
  if (s->d_item_cell) (void)hipFree(s->d_item_cell);
  if (s->d_item_unit) (void)hipFree(s->d_item_unit);
  if (s->d_order_sub) (void)hipFree(s->d_order_sub);
  if (s->d_rgb) (void)hipFree(s->d_rgb);
  if (s->d_debug) (void)hipFree(s->d_debug);
  if (s->d_lights) (void)hipFree(s->d_lights);
  if (s->h_stage) (void)hipHostFree(s->h_stage);
  for (hipEvent_t e : s->ev_stage) {
    if (e) (void)hipEventDestroy(e);
  }
  if (s->h_counters) (void)hipHostFree(s->h_counters);
  if (s->d_deep) (void)hipFree(s->d_deep);
  if (s->d_tile_list) (void)hipFree(s->d_tile_list);
  if (s->d_tile_slot) (void)hipFree(s->d_tile_slot);
  if (s->d_tile_cost) (void)hipFree(s->d_tile_cost);
  if (s->d_multi_order) (void)hipFree(s->d_multi_order);
  if (s->d_multi_list) (void)hipFree(s->d_multi_list);
  if (s->d_multi_lists) (void)hipFree(s->d_multi_lists);
  if (s->d_multi_tiles) (void)hipFree(s->d_multi_tiles);
  if (s->d_multi_map) (void)hipFree(s->d_multi_map);
  if (s->d_multi_maps) (void)hipFree(s->d_multi_maps);
  if (s->multi_comb_done) (void)hipEventDestroy(s->multi_comb_done);
  if (s->d_multi_gather) (void)hipFree(s->d_multi_gather);
  if (s->d_multi_frame) (void)hipFree(s->d_multi_frame);
  if (s->multi_stream) (void)hipStreamDestroy(s->multi_stream);
  if (s->multi_done) (void)hipEventDestroy(s->multi_done);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  for (auto &tri : s->ev_k) {
    for (hipEvent_t e : tri) {
      if (e) (void)hipEventDestroy(e);
    }
  }
  delete s;
}

static int scene_create_impl(mt_scene *s, const mt_scene_desc *d) {
  if (d->struct_size != sizeof(mt_scene_desc) || d->abi_version != MT_ABI_VERSION) {
    return fail(MT_ERR_ARG, "mt_scene_desc size/version mismatch (%u/%u, want %zu/%d)",
                d->struct_size, d->abi_version, sizeof(mt_scene_desc), MT_ABI_VERSION);
  }
  if (d->n_nodes < 1 || d->n_tris < 0 || d->n_materials < 0 || d->n_textures < 0) {
    return fail(MT_ERR_ARG, "negative or empty counts");
  }
  if (!d->node_aabb || !d->node_center || !d->node_first_child || !d->node_prim_begin ||
      !d->node_prim_count) {
    return fail(MT_ERR_ARG, "node arrays missing");
  }
  if (d->n_tris > 0 && (!d->tri_vertex || !d->tri_normal || !d->tri_uvw || !d->tri_aabb ||
                        !d->tri_material || !d->tri_line_no)) {
    return fail(MT_ERR_ARG, "triangle arrays missing");
  }
  if ((d->n_materials > 0 && !d->materials) || (d->n_textures > 0 && !d->textures)) {
    return fail(MT_ERR_ARG, "material/texture arrays missing");
  }
  // --- structural validation: the kernel indexes all of this unchecked.
  const int nn = d->n_nodes;
  std::vector<int> depth_of((size_t)nn, 0);
  depth_of[0] = 1;
  int max_depth = 1;
  long long prim_total = 0;
  for (int i = 0; i < nn; i++) {
    const int fc = d->node_first_child[i];
    const int pb = d->node_prim_begin[i], pc = d->node_prim_count[i];
    if (pb < 0 || pc < 0 || (long long)pb + pc > d->n_tris) {
      return fail(MT_ERR_ARG, "node %d: primitive range [%d,+%d) outside 0..%d", i, pb, pc, d->n_tris);
    }
    prim_total += pc;
    if (depth_of[i] == 0) return fail(MT_ERR_ARG, "node %d is not reachable in BFS order", i);
    if (fc != 0) {
      if (fc <= i || (long long)fc + 8 > nn) {
        return fail(MT_ERR_ARG, "node %d: first_child %d invalid (n_nodes %d)", i, fc, nn);
      }
      for (int k = 0; k < 8; k++) {
        if (depth_of[fc + k] != 0) return fail(MT_ERR_ARG, "node %d has two parents", fc + k);
        depth_of[fc + k] = depth_of[i] + 1;
      }
      if (depth_of[i] + 1 > max_depth) max_depth = depth_of[i] + 1;
    }
  }
  if (prim_total != d->n_tris) {
    return fail(MT_ERR_ARG, "nodes reference %lld primitives, scene has %d", prim_total, d->n_tris);
  }
  if (max_depth > MT_MAX_TREE_DEPTH) {
    return fail(MT_ERR_UNSUPPORTED, "octree depth %d exceeds MT_MAX_TREE_DEPTH %d", max_depth,
                MT_MAX_TREE_DEPTH);
  }
  for (int i = 0; i < d->n_tris; i++) {
    const int m = d->tri_material[i];
    if (m < -1 || m >= d->n_materials) {
      return fail(MT_ERR_ARG, "triangle %d: material index %d outside -1..%d", i, m, d->n_materials - 1);
    }
  }
  for (int i = 0; i < d->n_materials; i++) {
    const int t = d->materials[i].tex;
    if (t < -1 || t >= d->n_textures) {
      return fail(MT_ERR_ARG, "material %d: texture index %d outside -1..%d", i, t, d->n_textures - 1);
    }
  }
  for (int i = 0; i < d->n_textures; i++) {
    const mt_texture &t = d->textures[i];
    if (t.width <= 0 || t.height <= 0 || t.width > 30000 || t.height > 30000 || !t.texels ||
        (t.format != MT_TEX_RGB8 && t.format != MT_TEX_F64)) {
      return fail(MT_ERR_ARG, "texture %d: bad size/format", i);
    }
  }
  // --- is the min/max-instruction path admissible for this scene?
  bool regular = finite3(d->node_aabb, (size_t)nn * 6) && finite3(d->node_center, (size_t)nn * 3) &&
                 finite3(d->tri_aabb, (size_t)d->n_tris * 6);
  for (int i = 0; regular && i < nn; i++) {
    for (int k = 0; k < 3; k++) {
      const double lo = d->node_aabb[i * 6 + k], hi = d->node_aabb[i * 6 + 3 + k];
      const double c = d->node_center[i * 3 + k];
      if (!(lo <= hi)) regular = false;
      if (d->node_first_child[i] != 0 && !(lo <= c && c <= hi)) regular = false;
    }
  }
  for (int i = 0; regular && i < d->n_tris; i++) {
    for (int k = 0; k < 3; k++) {
      if (!(d->tri_aabb[i * 6 + k] <= d->tri_aabb[i * 6 + 3 + k])) regular = false;
    }
  }

  HIP_TRY(hipSetDevice(d->device));
  s->device = d->device;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, d->device));
  s->n_cu = prop.multiProcessorCount;

  std::vector<NodeRec> recs((size_t)nn);
  for (int i = 0; i < nn; i++) {
    NodeRec &r = recs[i];
    memset(&r, 0, sizeof r);
    for (int k = 0; k < 3; k++) {
      r.lo[k] = d->node_aabb[i * 6 + k];
      r.hi[k] = d->node_aabb[i * 6 + 3 + k];
      r.c[k] = d->node_center[i * 3 + k];
    }
    r.first_child = d->node_first_child[i];
    r.prim_begin = d->node_prim_begin[i];
    r.prim_count = d->node_prim_count[i];
    r.level = depth_of[i] - 1;
  }
  // Block boxes (mt_device.h kGroupTris): union of the fp64 boxes of each
  // block of kGroupTris consecutive stream triangles, then rounded to fp32
  // exactly like the per-triangle copies (the filter's margin covers it).
  std::vector<float> groups;
  for (int t0 = 0; t0 < d->n_tris; t0 += kGroupTris) {
    const int e = std::min(t0 + kGroupTris, d->n_tris);
    double u[6];
    for (int k = 0; k < 6; k++) u[k] = d->tri_aabb[(size_t)t0 * 6 + k];
    for (int t = t0 + 1; t < e; t++) {
      const double *b = d->tri_aabb + (size_t)t * 6;
      for (int k = 0; k < 3; k++) {
        u[k] = std::min(u[k], b[k]);
        u[3 + k] = std::max(u[3 + k], b[3 + k]);
      }
    }
    for (int k = 0; k < 6; k++) groups.push_back((float)u[k]);
  }
  // Subtree boxes (mt_trace.h tight_keep_mask): union of the fp64 boxes of all
  // triangles stored in a node or below it, rounded to fp32 like the others.
  // Breadth-first order: children have larger indices than their parent.  An
  // empty subtree gets an inverted box, which every ray misses.
  std::vector<float> subs((size_t)nn * 6 + 8 * 6, 0.0f);
  {
    std::vector<double> sb((size_t)nn * 6);
    for (int i = nn - 1; i >= 0; i--) {
      double *u = &sb[(size_t)i * 6];
      for (int k = 0; k < 3; k++) {
        u[k] = 3.0e38;
        u[3 + k] = -3.0e38;
      }
      const NodeRec &r = recs[i];
      for (int t = r.prim_begin; t < r.prim_begin + r.prim_count; t++) {
        const double *b = d->tri_aabb + (size_t)t * 6;
        for (int k = 0; k < 3; k++) {
          u[k] = std::min(u[k], b[k]);
          u[3 + k] = std::max(u[3 + k], b[3 + k]);
        }
      }
      if (r.first_child != 0) {
        for (int c = 0; c < 8; c++) {
          const double *b = &sb[(size_t)(r.first_child + c) * 6];
          for (int k = 0; k < 3; k++) {
            u[k] = std::min(u[k], b[k]);
            u[3 + k] = std::max(u[3 + k], b[3 + k]);
          }
        }
      }
      for (int k = 0; k < 6; k++) subs[(size_t)i * 6 + k] = (float)u[k];
      if (r.first_child != 0) {
        int mask = 0;
        for (int c = 0; c < 8; c++) {
          const double *b = &sb[(size_t)(r.first_child + c) * 6];
          if (b[0] <= b[3]) mask |= 1 << c;  // an empty subtree keeps the inverted box
        }
        recs[i].child_mask = mask;
      }
    }
  }
  // Records of the hit-set traversal (mt_device.h HsRec).
  static_assert(offsetof(HsRec, kid) == 16 && offsetof(HsRec, own) == kHsRecOwn && offsetof(HsRec, planes) == kHsRecPlanes &&
                    offsetof(HsRec, sl_begin) == kHsRecSl && offsetof(HsRec, ll_begin) == kHsRecLl, "the walk reads the record at these offsets");
  std::vector<HsRec> hsr((size_t)nn + 1);
  memset(hsr.data(), 0, hsr.size() * sizeof(HsRec));
  std::vector<float> sl_box;  // DevScene::sl_box32
  for (int i = 0; i < nn; i++) {
    HsRec &h = hsr[i];
    const NodeRec &r = recs[i];
    h.first_child = r.first_child;
    h.prim_begin = r.prim_begin;
    h.prim_count = r.prim_count;
    h.child_mask = r.first_child != 0 ? (r.child_mask & 0xff) : 0;
    for (int c = 0; c < 8; c++) {
      for (int a = 0; a < 3; a++) {
        const float lo = r.first_child != 0 ? subs[(size_t)(r.first_child + c) * 6 + a] : 3.0e38f;
        const float hi = r.first_child != 0 ? subs[(size_t)(r.first_child + c) * 6 + 3 + a] : -3.0e38f;
        h.kid[a][c] = lo;
        h.kid[a][8 + c] = hi;
        h.kid[a][16 + c] = lo;
      }
    }
    double u[6] = {3.0e38, 3.0e38, 3.0e38, -3.0e38, -3.0e38, -3.0e38};
    for (int t = r.prim_begin; t < r.prim_begin + r.prim_count; t++) {
      const double *b = d->tri_aabb + (size_t)t * 6;
      for (int k = 0; k < 3; k++) {
        u[k] = std::min(u[k], b[k]);
        u[3 + k] = std::max(u[3 + k], b[3 + k]);
      }
    }
    for (int a = 0; a < 3; a++) {
      h.own[a][0] = h.own[a][2] = (float)u[a];
      h.own[a][1] = (float)u[3 + a];
    }
    h.sl_begin = -1;
    if (r.prim_count >= 1 && r.prim_count <= kHsShortList) {
      h.sl_begin = (int32_t)(sl_box.size() / kSlQuadFloats);
      for (int q = 0; q < r.prim_count; q += 4) {
        float quad[kSlQuadFloats];
        for (int j = 0; j < 4; j++) {
          for (int a = 0; a < 3; a++) {
            const bool real = q + j < r.prim_count;
            const float lo = real ? (float)d->tri_aabb[(size_t)(r.prim_begin + q + j) * 6 + a] : 3.0e38f;
            const float hi = real ? (float)d->tri_aabb[(size_t)(r.prim_begin + q + j) * 6 + 3 + a] : -3.0e38f;
            quad[a * 12 + j] = lo;
            quad[a * 12 + 4 + j] = hi;
            quad[a * 12 + 8 + j] = lo;
          }
        }
        sl_box.insert(sl_box.end(), quad, quad + kSlQuadFloats);
      }
    }
    for (int k = 0; k < 3; k++) {
      h.planes[k] = r.lo[k];
      h.planes[3 + k] = r.c[k];
      h.planes[6 + k] = r.hi[k];
    }
  }
  // Spatially sorted copies of the long lists (DevScene::ll_*): a median-split tree over the boxes' centres, cut on
  // the longest axis of the centres' extent at a multiple of 64 (16 below 64) entries, so that 16 consecutive entries
  // -- one block box -- and 64 -- one super box -- are spatial neighbours.  Every list is padded to a multiple of kLlPad.
  std::vector<int32_t> ll_tri;
  std::vector<float> ll_box, ll_grp, ll_sup;
  std::vector<double> ll_exact;
  {
    const float inv[6] = {3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    std::vector<int32_t> order, work;
    for (int i = 0; i < nn; i++) {
      HsRec &h = hsr[i];
      h.ll_begin = -1;
      const int pb = recs[i].prim_begin, pc = recs[i].prim_count;
      if (pc <= kHsShortList) continue;
      order.resize((size_t)pc);
      for (int k = 0; k < pc; k++) order[(size_t)k] = pb + k;
      // iterative median split over [lo, hi) of `order`
      std::vector<std::pair<int, int>> todo{{0, pc}};
      while (!todo.empty()) {
        const int lo = todo.back().first, hi = todo.back().second;
        todo.pop_back();
        const int n = hi - lo;
        if (n <= 16) continue;
        double cmin[3] = {1e300, 1e300, 1e300}, cmax[3] = {-1e300, -1e300, -1e300};
        for (int k = lo; k < hi; k++) {
          const double *b = d->tri_aabb + (size_t)order[(size_t)k] * 6;
          for (int a = 0; a < 3; a++) {
            const double c = b[a] * 0.5 + b[3 + a] * 0.5;
            cmin[a] = std::min(cmin[a], c);
            cmax[a] = std::max(cmax[a], c);
          }
        }
        int ax = 0;
        for (int a = 1; a < 3; a++) {
          if (cmax[a] - cmin[a] > cmax[ax] - cmin[ax]) ax = a;
        }
        const int unit = n > 64 ? 64 : 16;
        int left = ((n / 2 + unit - 1) / unit) * unit;
        if (left >= n) left = n - (n % unit ? n % unit : unit);
        if (left <= 0 || left >= n) continue;
        auto key = [&](int32_t t) {
          const double *b = d->tri_aabb + (size_t)t * 6;
          return b[ax] * 0.5 + b[3 + ax] * 0.5;
        };
        std::nth_element(order.begin() + lo, order.begin() + lo + left, order.begin() + hi,
                         [&](int32_t x, int32_t y) { const double kx = key(x), ky = key(y); return kx < ky || (kx == ky && x < y); });
        todo.push_back({lo, lo + left});
        todo.push_back({lo + left, hi});
      }
      h.ll_begin = (int32_t)ll_tri.size();
      const int padded = ((pc + kLlPad - 1) / kLlPad) * kLlPad;
      for (int k = 0; k < padded; k++) {
        if (k < pc) {
          const int32_t t = order[(size_t)k];
          ll_tri.push_back(t);
          for (int q = 0; q < 6; q++) ll_box.push_back((float)d->tri_aabb[(size_t)t * 6 + q]);
          for (int q = 0; q < 6; q++) ll_exact.push_back(d->tri_aabb[(size_t)t * 6 + q]);
          for (int q = 0; q < 9; q++) ll_exact.push_back(d->tri_vertex[(size_t)t * 9 + q]);
        } else {
          ll_tri.push_back(-1);
          for (int q = 0; q < 6; q++) ll_box.push_back(inv[q]);
          for (int q = 0; q < 15; q++) ll_exact.push_back(0.0);
        }
      }
    }
    auto unions = [&](const std::vector<float> &src, size_t per, std::vector<float> *dst) {
      const size_t n = src.size() / 6;
      for (size_t b = 0; b < n; b += per) {
        float u[6] = {inv[0], inv[1], inv[2], inv[3], inv[4], inv[5]};
        for (size_t j = b; j < std::min(b + per, n); j++) {
          if (!(src[j * 6] <= src[j * 6 + 3])) continue;  // padding (inverted)
          for (int q = 0; q < 3; q++) {
            u[q] = std::min(u[q], src[j * 6 + q]);
            u[3 + q] = std::max(u[3 + q], src[j * 6 + 3 + q]);
          }
        }
        for (int q = 0; q < 6; q++) dst->push_back(u[q]);
      }
    };
    unions(ll_box, 16, &ll_grp);
    unions(ll_grp, 4, &ll_sup);
    ll_tri.resize(ll_tri.size() + 64, -1);  // (never empty)
    ll_exact.resize(ll_exact.size() + 15, 0.0);
  }
  // Second level (mt_device.h kSuperBlocks): one fp32 union box per 8 consecutive blocks, for the long lists.
  std::vector<float> supers;
  {
    const size_t nblk = groups.size() / 6;
    for (size_t b = 0; b < nblk; b += kSuperBlocks) {
      float u[6];
      for (int k = 0; k < 6; k++) u[k] = groups[b * 6 + k];
      for (size_t j = b + 1; j < std::min(b + (size_t)kSuperBlocks, nblk); j++) {
        for (int k = 0; k < 3; k++) {
          u[k] = std::min(u[k], groups[j * 6 + k]);
          u[3 + k] = std::max(u[3 + k], groups[j * 6 + 3 + k]);
        }
      }
      for (int k = 0; k < 6; k++) supers.push_back(u[k]);
    }
    supers.resize(supers.size() + 8 * 6, 0.0f);  // look-ahead padding
  }
  groups.resize(groups.size() + 8 * 6, 0.0f);  // the scan looks four boxes ahead
  int rc;
  if ((rc = upload(s, groups.data(), groups.size(), &s->dev.grp_aabb32)) != MT_OK) return rc;
  if ((rc = upload(s, supers.data(), supers.size(), &s->dev.sup_aabb32)) != MT_OK) return rc;
  if ((rc = upload(s, subs.data(), subs.size(), &s->dev.sub_aabb32)) != MT_OK) return rc;
  if ((rc = upload(s, hsr.data(), hsr.size(), &s->dev.hs_rec)) != MT_OK) return rc;
  if ((rc = upload(s, ll_tri.data(), ll_tri.size(), &s->dev.ll_tri)) != MT_OK) return rc;
  if ((rc = upload(s, ll_exact.data(), ll_exact.size(), &s->dev.ll_exact)) != MT_OK) return rc;
  {  // the three levels once more as quads for the walk's per-lane reads (DevScene::ll_*_q; layout of sl_box32)
    auto quads = [&](const std::vector<float> &src, size_t n_boxes) {
      std::vector<float> q(((n_boxes + 3) / 4 + 1) * (size_t)kSlQuadFloats, 0.0f);
      for (size_t i = 0; i < ((n_boxes + 3) / 4) * 4; i++) {
        for (int a = 0; a < 3; a++) {
          const float lo = i < n_boxes ? src[i * 6 + a] : 3.0e38f, hi = i < n_boxes ? src[i * 6 + 3 + a] : -3.0e38f;
          float *d4 = &q[(i / 4) * kSlQuadFloats + (size_t)a * 12 + (i & 3)];
          d4[0] = lo;
          d4[4] = hi;
          d4[8] = lo;
        }
      }
      return q;
    };
    const size_t n_entries = ll_tri.size() - 64;
    const std::vector<float> qb = quads(ll_box, n_entries), qg = quads(ll_grp, n_entries / 16), qs = quads(ll_sup, n_entries / 64);
    if ((rc = upload(s, qb.data(), qb.size(), &s->dev.ll_box_q)) != MT_OK) return rc;
    if ((rc = upload(s, qg.data(), qg.size(), &s->dev.ll_grp_q)) != MT_OK) return rc;
    if ((rc = upload(s, qs.data(), qs.size(), &s->dev.ll_sup_q)) != MT_OK) return rc;
  }
  sl_box.resize(sl_box.size() + 2 * kSlQuadFloats, 0.0f);  // (the copies are whole 16-byte pieces; never empty)
  if ((rc = upload(s, sl_box.data(), sl_box.size(), &s->dev.sl_box32)) != MT_OK) return rc;
  if ((rc = upload(s, recs.data(), recs.size(), &s->dev.nodes)) != MT_OK) return rc;
  const size_t nt = (size_t)d->n_tris;
  {
    // The scan loop looks two boxes ahead (mt_trace.h): pad the stream.
    std::vector<double> boxes(nt * 6 + 4 * 6, 0.0);
    if (nt) memcpy(boxes.data(), d->tri_aabb, nt * 6 * sizeof(double));
    if ((rc = upload(s, boxes.data(), boxes.size(), &s->dev.tri_aabb)) != MT_OK) return rc;
    // fp32 copy for the conservative pre-filter (mt_trace.h Filter32); padded by
    // 8 boxes because that loop looks four boxes ahead.
    std::vector<float> boxes32(nt * 6 + 64 * 6, 0.0f);  // (padding: the look-ahead of the scans; the hit-set traversal copies whole 16-byte pieces)
    double bmax[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < nt * 6; i++) {
      boxes32[i] = (float)boxes[i];
      const double a = std::fabs(boxes[i]);
      if (a > bmax[i % 3]) bmax[i % 3] = a;
    }
    if ((rc = upload(s, boxes32.data(), boxes32.size(), &s->dev.tri_aabb32)) != MT_OK) return rc;
    for (int k = 0; k < 3; k++) s->dev.bmax[k] = bmax[k];
  }
  if ((rc = upload(s, d->tri_vertex, nt * 9, &s->dev.tri_vertex)) != MT_OK) return rc;
  if ((rc = upload(s, d->tri_normal, nt * 9, &s->dev.tri_normal)) != MT_OK) return rc;
  if ((rc = upload(s, d->tri_uvw, nt * 9, &s->dev.tri_uvw)) != MT_OK) return rc;
  if ((rc = upload(s, d->tri_material, nt, &s->dev.tri_mtl)) != MT_OK) return rc;
  if ((rc = upload(s, d->tri_line_no, nt, &s->dev.tri_line)) != MT_OK) return rc;
  if ((rc = upload(s, d->materials, (size_t)d->n_materials, &s->dev.mtls)) != MT_OK) return rc;
  std::vector<DevTexture> texs((size_t)d->n_textures);
  for (int i = 0; i < d->n_textures; i++) {
    const mt_texture &t = d->textures[i];
    const size_t texel_bytes = (t.format == MT_TEX_RGB8 ? 3 : 24);
    const uint8_t *dev_texels = nullptr;
    if ((rc = upload(s, (const uint8_t *)t.texels, (size_t)t.width * t.height * texel_bytes,
                     &dev_texels)) != MT_OK) {
      return rc;
    }
    texs[i] = DevTexture{dev_texels, t.width, t.height, t.format, 0};
  }
  if ((rc = upload(s, texs.data(), texs.size(), &s->dev.texs)) != MT_OK) return rc;
  s->dev.n_tris = d->n_tris;
  s->dev.n_nodes = nn;
  s->dev.tree_depth = max_depth;
  s->dev.force_mode = 0;
  s->dev.scene_regular = regular ? 1 : 0;
  {
    // 16-byte traversal stack frames when "first child" and "best triangle + 1"
    // share one word: a quarter less LDS per wave
    int tri_bits = 1;
    while (tri_bits < 31 && ((long long)d->n_tris + 1) > (1ll << tri_bits)) tri_bits++;
    int node_bits = 1;
    while (node_bits < 31 && (long long)nn > (1ll << node_bits)) node_bits++;
    s->dev.pack_shift = (tri_bits + node_bits <= 32) ? tri_bits : 0;  // (MT_TUNE_PACKED_STACK = 0 switches it off)
  }
  s->dev.n_lights = 0;
  s->dev.lights = nullptr;
  HIP_TRY(hipMalloc((void **)&s->d_counters, ST_COUNT * sizeof(unsigned long long)));
  s->allocs.push_back(s->d_counters);
  HIP_TRY(hipMemset(s->d_counters, 0, ST_COUNT * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc((void **)&s->d_work, 64));
  s->allocs.push_back(s->d_work);
  HIP_TRY(hipMalloc((void **)&s->d_queues, kQueueWords * sizeof(unsigned)));
  s->allocs.push_back(s->d_queues);
  HIP_TRY(hipMalloc((void **)&s->d_order_ctl, kOrdWords * sizeof(unsigned)));
  s->allocs.push_back(s->d_order_ctl);
  HIP_TRY(hipMemset(s->d_order_ctl, 0, kOrdWords * sizeof(unsigned)));
  HIP_TRY(hipMalloc((void **)&s->d_order_whist, (size_t)kOrdGroupsMax * kOrdKeysMax * sizeof(unsigned)));
  s->allocs.push_back(s->d_order_whist);
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));
  if ((rc = mt_scene_set_lights(s, nullptr, 0)) != MT_OK) return rc;
  s->dev.hb = nullptr;
  s->dev.prof = nullptr;
  HIP_TRY(hipMalloc((void **)&s->d_dev, sizeof(DevScene)));
  s->allocs.push_back(s->d_dev);
  s->dev.self = s->d_dev;
#ifdef MT_PROF
  // (behind the phase sums: a time line of one wave, -DMT_PROF builds only -- kProfTimeline stamps)
  HIP_TRY(hipMalloc((void **)&s->d_prof, (PROF_COUNT + 1 + kProfTimeline) * sizeof(unsigned long long)));
  s->allocs.push_back(s->d_prof);
  HIP_TRY(hipMemset(s->d_prof, 0, (PROF_COUNT + 1 + kProfTimeline) * sizeof(unsigned long long)));
  s->dev.prof = s->d_prof;
#endif
#ifdef MT_DEBUG_KNOBS
  if (const char *e = getenv("MT_DEBUG_ITEM_CYCLES")) s->dbg_item_cycles = e;
  if (const char *e = getenv("MT_DEBUG_TIMELINE")) s->dbg_timeline = e;
  if (const char *e = getenv("MT_DEBUG_PRINT_UNITS")) s->dbg_print_units = atoi(e);
  const bool heartbeat = getenv("MT_DEBUG_HEARTBEAT") != nullptr;
#else
  const bool heartbeat = false;
#endif
  if (heartbeat) {
    HIP_TRY(hipHostMalloc((void **)&s->hb_host, 65536 * sizeof(unsigned long long), hipHostMallocMapped));
    memset(s->hb_host, 0, 65536 * sizeof(unsigned long long));
    void *dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, s->hb_host, 0));
    s->dev.hb = (volatile unsigned long long *)dp;
  }
  return configure_launch(s);
}

mt_scene *mt_scene_create(const mt_scene_desc *d) {
  if (!d) {
    fail(MT_ERR_ARG, "desc is NULL");
    return nullptr;
  }
  mt_scene *s = new mt_scene();
  s->engine = g_default_engine.load();
  if (scene_create_impl(s, d) != MT_OK) {
    mt_scene_destroy(s);
    return nullptr;
  }
  return s;
}

int mt_scene_set_lights(mt_scene *s, const mt_light *lights, int n) {
  if (!s || n < 0 || (n > 0 && !lights)) return fail(MT_ERR_ARG, "bad lights argument");
  HIP_TRY(hipSetDevice(s->device));
  if (n > s->lights_cap || !s->d_lights) {
    if (s->d_lights) HIP_TRY(hipFree(s->d_lights));
    s->d_lights = nullptr;
    const int cap = n > 8 ? n : 8;
    HIP_TRY(hipMalloc((void **)&s->d_lights, (size_t)cap * sizeof(mt_light)));
    s->lights_cap = cap;
  }
  // The reference's drivers push the same lights again before every frame (main_local.cc:79-110): what the device
  // holds already is not uploaded again.  Other lights, other costs: the damped forecast starts over (the old costs
  // remain its first guess).
  if ((size_t)n != s->lights_host.size() || s->dev.lights != s->d_lights ||
      (n && memcmp(s->lights_host.data(), lights, (size_t)n * sizeof(mt_light)) != 0)) {
    if (n) HIP_TRY(hipMemcpy(s->d_lights, lights, (size_t)n * sizeof(mt_light), hipMemcpyHostToDevice));
    s->lights_host.assign(lights, lights + n);
    s->forecasts_in_a_row = 0;
  }
  s->dev.lights = s->d_lights;
  s->dev.n_lights = n;
  return MT_OK;
}

int mt_scene_set_traversal_mode(mt_scene *s, int mode) {
  if (!s || mode < 0 || mode > 7) return fail(MT_ERR_ARG, "mode must be 0..7");
  s->dev.force_mode = mode;
  return MT_OK;
}

int mt_scene_set_scheduling(mt_scene *s, int use_cost_history) {
  if (!s || (use_cost_history != 0 && use_cost_history != 1)) return fail(MT_ERR_ARG, "bad scheduling argument");
  s->use_history = use_cost_history != 0;
  s->cost_signature = 0;
  return MT_OK;
}

int mt_scene_set_stats(mt_scene *s, int enabled) {
  if (!s || (enabled != 0 && enabled != 1)) return fail(MT_ERR_ARG, "bad stats argument");
  s->stats_enabled = enabled != 0;
  return MT_OK;
}

int mt_scene_set_engine(mt_scene *s, int engine) {
  if (!s || engine < 0 || engine > 3) return fail(MT_ERR_ARG, "engine must be 0 (automatic), 1, 2 or 3");
  s->engine = engine;
  s->cost_signature = 0;
  return MT_OK;
}

int mt_set_default_engine(int engine) {
  if (engine < 0 || engine > 3) return fail(MT_ERR_ARG, "engine must be 0 (automatic), 1, 2 or 3");
  g_default_engine.store(engine);
  return MT_OK;
}

int mt_scene_set_tuning(mt_scene *s, int knob, double value) {
  if (!s || knob < 0 || knob >= MT_TUNE_COUNT || !std::isfinite(value)) return fail(MT_ERR_ARG, "bad tuning argument");
  // The kernels divide by the work factors and scale an even share by the shares: those must be positive (and small
  // enough for their products to stay finite in float); counts and budgets must not be negative or beyond what the
  // conversions to integers hold.
  switch (knob) {
    case MT_TUNE_POOL_PIECE_TIME1: case MT_TUNE_POOL_PIECE_TIME2: case MT_TUNE_POOL_PIECE_WORK1: case MT_TUNE_POOL_PIECE_WORK2:
    case MT_TUNE_POOL_CELL_FACTOR: case MT_TUNE_QUAD_SHARE: case MT_TUNE_QUAD_SHARE_MOVING: case MT_TUNE_QUAD_KEEP:
    case MT_TUNE_QUAD_WORK: case MT_TUNE_QUAD_WORK_MOVING: case MT_TUNE_HYBRID_POOL_SHARE: case MT_TUNE_HYBRID_QUAD_SHARE:
    case MT_TUNE_SM_CELL_SHARE: case MT_TUNE_SM_CELL_TIME: case MT_TUNE_SM_CELL_WORK: case MT_TUNE_HYBRID_CELL_FACTOR:
    case MT_TUNE_HYBRID_WORK1: case MT_TUNE_HYBRID_WORK2: case MT_TUNE_FORECAST_STEP: case MT_TUNE_HYBRID_STARTER_SHARE:
      if (!(value >= 1e-6 && value <= 1e6)) return fail(MT_ERR_ARG, "tuning knob %d must lie in [1e-6, 1e6]", knob);
      break;
    case MT_TUNE_POOL_BELOW: case MT_TUNE_POOL_CAP: case MT_TUNE_BLOCKS_PER_CU:
      if (!(value >= 0.0 && value <= 1e9)) return fail(MT_ERR_ARG, "tuning knob %d must lie in [0, 1e9]", knob);
      break;
    case MT_TUNE_POOL_SCRATCH_MB:
      if (!(value >= 1.0 && value <= 1e9)) return fail(MT_ERR_ARG, "the ray pool's scratch budget must lie in [1, 1e9] MB");
      break;
    case MT_TUNE_BLEND:
      if (!(value >= 0.0 && value <= 1.0)) return fail(MT_ERR_ARG, "the forecast's damping must lie in [0, 1]");
      break;
    case MT_TUNE_FORECAST_RADIUS:
      if (!(value <= 8.0)) return fail(MT_ERR_ARG, "the forecast's radius must be at most 8 blocks (< 0 = automatic)");
      break;
    case MT_TUNE_POOL_CUT_SHARE:
      if (!(value < 0.0 || (value >= 1e-6 && value <= 1e6))) return fail(MT_ERR_ARG, "the pool's cutting share must be negative (automatic) or lie in [1e-6, 1e6]");
      break;
    case MT_TUNE_ORDER_GROUPS:
      if (!(value >= 1.0 && value <= (double)kOrdGroupsMax)) return fail(MT_ERR_ARG, "the work-order kernel runs on 1 .. %d workgroups", kOrdGroupsMax);
      break;
    case MT_TUNE_XCD_QUEUES:
      if (!(value == 0.0 || value == 1.0 || value == 2.0)) return fail(MT_ERR_ARG, "XCD queues: 0 (off), 1 (stripes) or 2 (grid)");
      break;
    default: break;  // switches: any finite value (0 / non-zero)
  }
  s->tune.v[knob] = value;
  s->cost_signature = 0;  // other constants, other order: start from a first frame
  if (knob == MT_TUNE_PACKED_STACK || knob == MT_TUNE_BLOCKS_PER_CU || knob == MT_TUNE_DEEP_LAYOUT) {
    if (knob == MT_TUNE_PACKED_STACK) {
      int tri_bits = 1;
      while (tri_bits < 31 && ((long long)s->dev.n_tris + 1) > (1ll << tri_bits)) tri_bits++;
      int node_bits = 1;
      while (node_bits < 31 && (long long)s->dev.n_nodes > (1ll << node_bits)) node_bits++;
      s->dev.pack_shift = (value != 0.0 && tri_bits + node_bits <= 32) ? tri_bits : 0;
    }
    HIP_TRY(hipSetDevice(s->device));
    return configure_launch(s);
  }
  return MT_OK;
}

int mt_scene_export_costs_device(mt_scene *s, void *d_map, int map_w, int map_h, void *stream) {
  if (!s || !d_map || map_w <= 0 || map_h <= 0) return fail(MT_ERR_ARG, "bad cost map arguments");
  if (!s->last_P_valid) return fail(MT_ERR_ARG, "no launch to export the costs of");
  const RenderParams &P = s->last_P;
  if (map_w < (P.image_w + 7) / 8 || map_h < (P.image_h + 7) / 8 || (P.tile_w & 7) || (P.tile_h & 7) || (P.region_x & 7) || (P.region_y & 7)) {
    return fail(MT_ERR_ARG, "cost map smaller than the image's 8x8 blocks, or tiles not on the 8-pixel grid");
  }
  HIP_TRY(hipSetDevice(s->device));
  if (P.n_items == 0) return MT_OK;
  const double *tv = s->tune.v;
  const bool hy = s->last_engine == 3;
  hipLaunchKernelGGL(export_costs_kernel, dim3((P.n_items + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, s->last_engine,
                     (float)(hy ? tv[MT_TUNE_HYBRID_WORK1] : tv[MT_TUNE_POOL_PIECE_WORK1]),
                     (float)(hy ? tv[MT_TUNE_HYBRID_WORK2] : tv[MT_TUNE_POOL_PIECE_WORK2]), (float)tv[MT_TUNE_SM_CELL_WORK],
                     (const unsigned char *)s->d_item_form, (unsigned *)d_map, map_w, map_h);
  HIP_TRY(hipGetLastError());
  return MT_OK;
}

int mt_scene_import_costs_device(mt_scene *s, const void *d_map, int map_w, int map_h, void *stream) {
  if (!s || !d_map || map_w <= 0 || map_h <= 0) return fail(MT_ERR_ARG, "bad cost map arguments");
  HIP_TRY(hipSetDevice(s->device));
  const size_t bytes = (size_t)map_w * (size_t)map_h * sizeof(unsigned);
  int rc = ensure_bytes((void **)&s->d_cost_map, &s->cost_map_bytes, bytes);
  if (rc != MT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(s->d_cost_map, d_map, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  s->cost_map_w = map_w;
  s->cost_map_h = map_h;
  s->cost_map_for_launch = s->launches_timed;  // valid for the NEXT launch only
  return MT_OK;
}

int mt_scene_read_stats(mt_scene *s, mt_stats *st) {
  if (!s || !st) return fail(MT_ERR_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(s->device));
  unsigned long long c[ST_COUNT];
  HIP_TRY(hipMemcpy(c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(s->d_counters, 0, sizeof c));
  memset(st, 0, sizeof *st);
  fill_stats(c, st);
#ifdef MT_PROF
  {
    unsigned long long pr[PROF_COUNT];
    HIP_TRY(hipMemcpy(pr, s->d_prof, sizeof pr, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(s->d_prof, 0, sizeof pr));
    if (const char *tl = s->dbg_timeline.empty() ? nullptr : s->dbg_timeline.c_str()) {  // dump and reset the time line
      std::vector<unsigned long long> host(1 + kProfTimeline);
      HIP_TRY(hipMemcpy(host.data(), s->d_prof + PROF_COUNT, host.size() * 8, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemset(s->d_prof + PROF_COUNT, 0, host.size() * 8));
      if (FILE *f = fopen(tl, "wb")) {
        fwrite(host.data(), 8, host.size(), f);
        fclose(f);
      }
    }
    static const char *names[PROF_COUNT] = {"trace_cycles", "scan_raypar_cycles", "scan_transposed_cycles",
                                            "children_unwind_cycles", "n_raypar_scans", "n_transposed_scans",
                                            "n_transposed_chunks", "n_raypar_tris", "n_traces", "lane_phase_cycles",
                                            "scan_m2f", "scan_m2", "scan_m1", "scan_m0", "n_m2f", "n_m2", "n_m1", "n_m0",
                                            "tris_m2f", "tris_m1", "tris_transposed", "g_groups", "g_live", "g_ranges", "g_range_tris", "nin_sum", "nin_lt8", "nin_lt24", "want_sum",
                                            "grp_mask_t", "grp_ranges_t", "tr_blocks_t", "tr_tris_t", "tr_bcast_t", "tr_rays", "m2f_call_t",
                                            "hs_rec_t", "hs_big_t", "hs_small_t", "hs_trans_t", "hs_kids_t", "hs_ret_t", "hs_close_t", "hs_n_enter", "hs_n_big", "hs_n_small", "hs_n_empty", "hs_n_trans", "hs_n_ret", "hs_n_rethit", "hs_lanes", "hs_big_tris", "hs_small_tris"};
    fprintf(stderr, "[mt prof]");
    for (int i = 0; i < PROF_COUNT; i++) fprintf(stderr, " %s=%llu", names[i], pr[i]);
    fprintf(stderr, "\n");
  }
#endif
  return check_status(c);
}

int mt_scene_kernel_times(mt_scene *s, int max_n, double *primary_ms, double *render_ms) {
  if (!s || max_n < 0 || (max_n > 0 && (!primary_ms || !render_ms))) {
    return fail(MT_ERR_ARG, "bad kernel_times arguments");
  }
  HIP_TRY(hipSetDevice(s->device));
  unsigned long long first = s->launches_read;
  if (s->launches_timed - first > (unsigned long long)mt_scene::kTimedLaunches) {
    first = s->launches_timed - mt_scene::kTimedLaunches;  // older ones were overwritten
  }
  if (s->launches_timed - first > (unsigned long long)max_n) first = s->launches_timed - max_n;
  int n = 0;
  for (unsigned long long i = first; i < s->launches_timed; i++, n++) {
    hipEvent_t *ek = s->ev_k[i % mt_scene::kTimedLaunches];
    HIP_TRY(hipEventSynchronize(ek[2]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, ek[0], ek[1]));
    HIP_TRY(hipEventElapsedTime(&b, ek[1], ek[2]));
    primary_ms[n] = a;
    render_ms[n] = b;
  }
  s->launches_read = s->launches_timed;
  return n;
}

int mt_render_chunk_device(mt_scene *s, const mt_sensor *sensor, int image_w, int image_h,
                           int chunk_x, int chunk_y, int chunk_w, int chunk_h, int max_depth,
                           void *d_rgb, void *d_debug, void *stream) {
  int rc = check_image_args(s, sensor, image_w, image_h);
  if (rc != MT_OK) return rc;
  // WorkChunk::DeserializeInput's constraints, mythtracer.cc:358-371
  if (chunk_x < 0 || chunk_y < 0 || chunk_w <= 0 || chunk_h <= 0 ||
      (long long)chunk_x + chunk_w > image_w || (long long)chunk_y + chunk_h > image_h) {
    return fail(MT_ERR_ARG, "chunk %d,%d %dx%d outside image %dx%d", chunk_x, chunk_y, chunk_w,
                chunk_h, image_w, image_h);
  }
  if (!d_rgb) return fail(MT_ERR_ARG, "d_rgb is NULL");
  HIP_TRY(hipSetDevice(s->device));
  return launch_render(s, sensor, image_w, image_h, chunk_x, chunk_y, chunk_w, chunk_h, chunk_w,
                       chunk_h, 0, 1, 1, max_depth, (uint8_t *)d_rgb, (mt_debug_px *)d_debug,
                       (hipStream_t)stream);
}

int mt_render_tiles_device(mt_scene *s, const mt_sensor *sensor, int image_w, int image_h,
                           int tile_w, int tile_h, int first_tile, int tile_stride, int n_tiles,
                           int max_depth, void *d_rgb, void *stream) {
  int rc = check_image_args(s, sensor, image_w, image_h);
  if (rc != MT_OK) return rc;
  if (tile_w <= 0 || tile_h <= 0 || first_tile < 0 || tile_stride <= 0 || n_tiles < 0) {
    return fail(MT_ERR_ARG, "bad tiling arguments");
  }
  const long long tiles_total =
      (long long)((image_w + tile_w - 1) / tile_w) * ((image_h + tile_h - 1) / tile_h);
  if (n_tiles > 0 && (long long)first_tile + (long long)(n_tiles - 1) * tile_stride >= tiles_total) {
    return fail(MT_ERR_ARG, "tile selection exceeds the %lld tiles of the image", tiles_total);
  }
  if (!d_rgb && n_tiles > 0) return fail(MT_ERR_ARG, "d_rgb is NULL");
  HIP_TRY(hipSetDevice(s->device));
  return launch_render(s, sensor, image_w, image_h, 0, 0, image_w, image_h, tile_w, tile_h,
                       first_tile, tile_stride, n_tiles, max_depth, (uint8_t *)d_rgb, nullptr,
                       (hipStream_t)stream);
}

int mt_blit_tiles_device(mt_scene *s, int image_w, int image_h, int tile_w, int tile_h,
                         int first_tile, int tile_stride, int n_tiles, const void *d_tiles,
                         void *d_image, void *stream) {
  if (!s || !d_tiles || !d_image || image_w <= 0 || image_h <= 0 || tile_w <= 0 || tile_h <= 0 ||
      first_tile < 0 || tile_stride <= 0 || n_tiles < 0) {
    return fail(MT_ERR_ARG, "bad blit arguments");
  }
  const int tiles_x = (image_w + tile_w - 1) / tile_w;
  const long long tiles_total = (long long)tiles_x * ((image_h + tile_h - 1) / tile_h);
  if (n_tiles > 0 && (long long)first_tile + (long long)(n_tiles - 1) * tile_stride >= tiles_total) {
    return fail(MT_ERR_ARG, "tile selection exceeds the %lld tiles of the image", tiles_total);
  }
  if (n_tiles == 0) return MT_OK;
  HIP_TRY(hipSetDevice(s->device));
  hipLaunchKernelGGL(blit_tiles_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, image_w,
                     image_h, tile_w, tile_h, tiles_x, first_tile, tile_stride, n_tiles, (const int32_t *)nullptr,
                     (const uint8_t *)d_tiles, (uint8_t *)d_image);
  HIP_TRY(hipGetLastError());
  return MT_OK;
}

int mt_order_tiles_device(mt_scene *s, const void *d_cost_map, int map_w, int map_h, int image_w, int image_h,
                          int tile_w, int tile_h, void *d_order, void *stream) {
  if (!s || !d_cost_map || !d_order || map_w <= 0 || map_h <= 0 || image_w <= 0 || image_h <= 0 || tile_w <= 0 || tile_h <= 0 ||
      image_w > 100000 || image_h > 100000) {
    return fail(MT_ERR_ARG, "bad tile order arguments");
  }
  if (map_w < (image_w + 7) / 8 || map_h < (image_h + 7) / 8) return fail(MT_ERR_ARG, "cost map smaller than the image's 8x8 blocks");
  const int tiles_x = (image_w + tile_w - 1) / tile_w;
  const long long total = (long long)tiles_x * ((image_h + tile_h - 1) / tile_h);
  if (total > (1ll << 20)) return fail(MT_ERR_ARG, "too many tiles to order (%lld)", total);
  HIP_TRY(hipSetDevice(s->device));
  int rc = ensure_bytes((void **)&s->d_tile_cost, &s->tile_cost_bytes, (size_t)total * 8);
  if (rc != MT_OK) return rc;
  const int n = (int)total;
  hipLaunchKernelGGL(tile_cost_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned *)d_cost_map,
                     map_w, map_h, image_w, image_h, tile_w, tile_h, tiles_x, n, s->d_tile_cost);
  hipLaunchKernelGGL(tile_order_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, s->d_tile_cost, n, (int32_t *)d_order);
  HIP_TRY(hipGetLastError());
  return MT_OK;
}

int mt_dealt_tile_count(int image_w, int image_h, int tile_w, int tile_h, int world, int rank) {
  if (image_w <= 0 || image_h <= 0 || tile_w <= 0 || tile_h <= 0 || world < 1 || rank < 0 || rank >= world) {
    return fail(MT_ERR_ARG, "bad tile count arguments");
  }
  const long long total = (long long)((image_w + tile_w - 1) / tile_w) * ((image_h + tile_h - 1) / tile_h);
  if (total > 0x7fffffffll) return fail(MT_ERR_ARG, "too many tiles");
  return dealt_tile_count((int)total, world, rank);
}

int mt_deal_tiles_device(mt_scene *s, const void *d_order, int image_w, int image_h, int tile_w, int tile_h, int world,
                         int rank, void *d_list, void *stream) {
  if (!s || !d_list) return fail(MT_ERR_ARG, "bad deal arguments");
  const int n = mt_dealt_tile_count(image_w, image_h, tile_w, tile_h, world, rank);
  if (n <= 0) return n;
  const int total = ((image_w + tile_w - 1) / tile_w) * ((image_h + tile_h - 1) / tile_h);
  HIP_TRY(hipSetDevice(s->device));
  hipLaunchKernelGGL(deal_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const int32_t *)d_order, total,
                     world, rank, n, (int32_t *)d_list);
  HIP_TRY(hipGetLastError());
  return n;
}

int mt_render_tile_list_device(mt_scene *s, const mt_sensor *sensor, int image_w, int image_h, int tile_w, int tile_h,
                               const void *d_list, int n_tiles, uint64_t list_id, int max_depth, void *d_rgb,
                               void *stream) {
  int rc = check_image_args(s, sensor, image_w, image_h);
  if (rc != MT_OK) return rc;
  if (tile_w <= 0 || tile_h <= 0 || n_tiles < 0) return fail(MT_ERR_ARG, "bad tiling arguments");
  const long long tiles_total = (long long)((image_w + tile_w - 1) / tile_w) * ((image_h + tile_h - 1) / tile_h);
  if (n_tiles > tiles_total) return fail(MT_ERR_ARG, "%d tiles listed, the image has %lld", n_tiles, tiles_total);
  if (n_tiles > 0 && (!d_rgb || !d_list)) return fail(MT_ERR_ARG, "d_rgb or d_list is NULL");
  HIP_TRY(hipSetDevice(s->device));
  return launch_render(s, sensor, image_w, image_h, 0, 0, image_w, image_h, tile_w, tile_h, 0, 1, n_tiles, max_depth,
                       (uint8_t *)d_rgb, nullptr, (hipStream_t)stream, n_tiles > 0 ? (const int32_t *)d_list : nullptr,
                       (unsigned long long)list_id);
}

int mt_blit_tile_list_device(mt_scene *s, int image_w, int image_h, int tile_w, int tile_h, const void *d_list,
                             int n_tiles, const void *d_tiles, void *d_image, void *stream) {
  if (!s || !d_tiles || !d_image || !d_list || image_w <= 0 || image_h <= 0 || tile_w <= 0 || tile_h <= 0 || n_tiles < 0) {
    return fail(MT_ERR_ARG, "bad blit arguments");
  }
  if (n_tiles == 0) return MT_OK;
  HIP_TRY(hipSetDevice(s->device));
  hipLaunchKernelGGL(blit_tiles_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, image_w, image_h, tile_w, tile_h,
                     (image_w + tile_w - 1) / tile_w, 0, 1, n_tiles, (const int32_t *)d_list, (const uint8_t *)d_tiles,
                     (uint8_t *)d_image);
  HIP_TRY(hipGetLastError());
  return MT_OK;
}

int mt_render_chunk(mt_scene *s, const mt_sensor *sensor, int image_w, int image_h, int chunk_x,
                    int chunk_y, int chunk_w, int chunk_h, int max_depth, uint8_t *out_rgb,
                    mt_debug_px *out_debug, mt_stats *stats) {
  if (!out_rgb) return fail(MT_ERR_ARG, "out_rgb is NULL");
  int rc = check_image_args(s, sensor, image_w, image_h);
  if (rc != MT_OK) return rc;
  const auto w0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(s->device));
  const size_t npx = (size_t)(chunk_w > 0 ? chunk_w : 0) * (size_t)(chunk_h > 0 ? chunk_h : 0);
  if ((rc = ensure_bytes((void **)&s->d_rgb, &s->rgb_bytes, npx * 3)) != MT_OK) return rc;
  if (out_debug) {
    if ((rc = ensure_bytes((void **)&s->d_debug, &s->debug_bytes, npx * sizeof(mt_debug_px))) != MT_OK) {
      return rc;
    }
  }
  if (!s->h_counters) HIP_TRY(hipHostMalloc((void **)&s->h_counters, ST_COUNT * sizeof(unsigned long long), hipHostMallocDefault));
  // How the frame reaches the caller's (pageable) buffer, measured on the box (scripts/ubench/d2h_paths.hip, 6.2 MB of
  // a 1080p frame / 24.9 MB of a 4K one): plain hipMemcpy 1.17 / 1.22 ms; a page-locked staging buffer + memcpy 0.42 /
  // 1.72 (the memcpy alone 0.30 / 1.28); registering the caller's buffer per call 0.82 / 1.22 (the registration 0.7);
  // a copy into memory that IS registered 0.12 / 0.45 -- but keeping a caller's buffer registered across calls is not
  // safe (a vector freed and allocated again at the same address would receive its frame in the OLD pages).  So: the
  // staging buffer, in pieces, every piece's memcpy under the next piece's DMA: about the memcpy's time.
  const size_t out_bytes = npx * 3;
  if (s->stage_bytes < out_bytes) {
    if (s->h_stage) HIP_TRY(hipHostFree(s->h_stage));
    s->h_stage = nullptr;
    s->stage_bytes = 0;
    HIP_TRY(hipHostMalloc((void **)&s->h_stage, out_bytes, hipHostMallocDefault));
    s->stage_bytes = out_bytes;
  }
  for (hipEvent_t &e : s->ev_stage) {
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  hipStream_t stream = nullptr;
  HIP_TRY(hipMemsetAsync(s->d_counters, 0, ST_COUNT * sizeof(unsigned long long), stream));
  HIP_TRY(hipEventRecord(s->ev0, stream));
  const bool counters_were = s->stats_enabled;
  if (stats) s->stats_enabled = true;  // the caller asked for them
  rc = mt_render_chunk_device(s, sensor, image_w, image_h, chunk_x, chunk_y, chunk_w, chunk_h,
                              max_depth, s->d_rgb, out_debug ? s->d_debug : nullptr, stream);
  s->stats_enabled = counters_were;
  if (rc != MT_OK) return rc;
  HIP_TRY(hipEventRecord(s->ev1, stream));
  // everything that comes back is queued behind the kernels: the counters (device status), the frame in pieces
  HIP_TRY(hipMemcpyAsync(s->h_counters, s->d_counters, ST_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemsetAsync(s->d_counters, 0, ST_COUNT * sizeof(unsigned long long), stream));
  const int n_pieces = out_bytes < (1u << 20) ? 1 : (out_bytes < (8u << 20) ? 4 : mt_scene::kStagePieces);
  const size_t piece = ((out_bytes + n_pieces - 1) / n_pieces + 4095) & ~(size_t)4095;
  for (int k = 0; k < n_pieces; k++) {
    const size_t off = (size_t)k * piece;
    if (off < out_bytes) {
      HIP_TRY(hipMemcpyAsync(s->h_stage + off, s->d_rgb + off, std::min(piece, out_bytes - off), hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipEventRecord(s->ev_stage[k], stream));
  }
  if (out_debug) {
    HIP_TRY(hipMemcpyAsync(out_debug, s->d_debug, npx * sizeof(mt_debug_px), hipMemcpyDeviceToHost, stream));
  }
  for (int k = 0; k < n_pieces; k++) {
    const size_t off = (size_t)k * piece;
    HIP_TRY(hipEventSynchronize(s->ev_stage[k]));
    if (k == 0 && check_status(s->h_counters) != MT_OK) {  // (the counters came first: no frame of a failed launch)
      (void)hipStreamSynchronize(stream);
      return check_status(s->h_counters);
    }
    if (off < out_bytes) memcpy(out_rgb + off, s->h_stage + off, std::min(piece, out_bytes - off));
  }
  HIP_TRY(hipStreamSynchronize(stream));
  if ((rc = check_status(s->h_counters)) != MT_OK) return rc;
  if (stats) {
    memset(stats, 0, sizeof *stats);
    fill_stats(s->h_counters, stats);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->total_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  }
#ifdef MT_PROF
  {  // (the phase profile is printed by mt_scene_read_stats)
    mt_stats dummy;
    (void)mt_scene_read_stats(s, &dummy);
  }
#endif
  return MT_OK;
}

// ---- one frame on several GPUs of this process (main_net_master.cc:195-236) --------------------------
// Replica r renders ITS tiles -- dealt out by cost, see mt_order_tiles_device; by tile number while no frame of this
// geometry has been measured -- into its own tile buffer on its own device and stream (the launches are made from one
// host thread per replica, so that no device waits for another's launch calls); the first replica's stream then waits
// for each replica's event, pulls its buffer over xGMI (hipMemcpyPeerAsync; the buffers of replicas on the SAME device
// are read in place) and blits it into the frame; one D2H copy ends the call.
int mt_render_frame_multi(mt_scene *const *scenes, int n, const mt_sensor *sensor, int image_w, int image_h,
                          int tile_w, int tile_h, int max_depth, uint8_t *out_rgb, mt_stats *stats) {
  if (!scenes || n < 1 || n > 1024) return fail(MT_ERR_ARG, "bad scene list");
  for (int r = 0; r < n; r++) {
    if (!scenes[r]) return fail(MT_ERR_ARG, "scenes[%d] is NULL", r);
    for (int q = 0; q < r; q++) {
      if (scenes[q] == scenes[r]) return fail(MT_ERR_ARG, "scenes[%d] and scenes[%d] are the same replica", q, r);
    }
  }
  if (!out_rgb) return fail(MT_ERR_ARG, "out_rgb is NULL");
  int rc = check_image_args(scenes[0], sensor, image_w, image_h);
  if (rc != MT_OK) return rc;
  if (tile_w <= 0 || tile_h <= 0) return fail(MT_ERR_ARG, "bad tile size");
  const auto w0 = std::chrono::steady_clock::now();
  const long long tiles_total_ll = (long long)((image_w + tile_w - 1) / tile_w) * ((image_h + tile_h - 1) / tile_h);
  if (tiles_total_ll > (1ll << 20)) return fail(MT_ERR_ARG, "too many tiles (%lld)", tiles_total_ll);
  const int tiles_total = (int)tiles_total_ll;
  const size_t slot = (size_t)tile_w * tile_h * 3;
  const int n_max = (tiles_total + n - 1) / n;
  auto tiles_of = [&](int r) -> int { return dealt_tile_count(tiles_total, n, r); };
  mt_scene *root = scenes[0];
  // the replicas' block costs are exchanged after every frame (mt_scene_export_costs_device: a moving camera's next
  // frame is ordered by the costs of ALL tiles of this one, and the tiles are dealt out by them)
  const bool share_costs = n > 1 && (tile_w & 7) == 0 && (tile_h & 7) == 0;
  const int map_w = (image_w + 7) / 8, map_h = (image_h + 7) / 8;
  const size_t map_bytes = (size_t)map_w * map_h * sizeof(unsigned);

  // ---- who renders what.  State of the previous call (kept on every replica, decided on the first one's): the same
  // geometry and replica list?  Then the tiles are dealt out anew by the combined costs of the previous frame -- every
  // replica holds that map and orders it by itself -- unless the camera has been at rest for two frames: from then on
  // the assignment is kept, and with it the per-slot cost history (running means, the blocks' two measured forms).
  unsigned long long geom = 1469598103934665603ull;
  for (long long v : {(long long)image_w, (long long)image_h, (long long)tile_w, (long long)tile_h, (long long)max_depth, (long long)n}) {
    geom = (geom ^ (unsigned long long)v) * 1099511628211ull;
  }
  if (geom == 0) geom = 1;
  bool same_geom = true;
  for (int r = 0; r < n; r++) {
    same_geom = same_geom && scenes[r]->multi_geom == geom && scenes[r]->multi_rank == r &&
                scenes[r]->multi_list_id == root->multi_list_id;
  }
  const bool balance = share_costs && root->tune.v[MT_TUNE_MULTI_BALANCE] != 0.0;
  const bool at_rest = same_geom && memcmp(&root->multi_sensor, sensor, sizeof(mt_sensor)) == 0;
  const int frames_at_rest = at_rest ? root->multi_frames_at_rest + 1 : 0;
  const bool by_map = same_geom && balance && root->multi_have_map && (!at_rest || frames_at_rest < 2);
  const bool redeal = !same_geom || by_map;
  static std::atomic<unsigned long long> next_list_id{1};
  const unsigned long long list_id = redeal ? next_list_id.fetch_add(1) : root->multi_list_id;
  auto forget = [&]() {
    for (int q = 0; q < n; q++) scenes[q]->multi_geom = 0;
  };
  auto drain = [&]() {  // let whatever was launched finish before the caller touches its buffers or the scenes again
    for (int q = 0; q < n; q++) {
      if (scenes[q]->multi_stream) {
        (void)hipSetDevice(scenes[q]->device);
        (void)hipStreamSynchronize(scenes[q]->multi_stream);
      }
    }
    (void)hipSetDevice(root->device);
  };

  // ---- phase 1: every replica renders its tiles
  std::vector<int> rcs((size_t)n, MT_OK);
  std::vector<std::string> errs((size_t)n);
  auto launch_one = [&](int r) {
    mt_scene *s = scenes[r];
    auto body = [&]() -> int {
      HIP_TRY(hipSetDevice(s->device));
      if (!s->multi_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->multi_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&s->multi_done, hipEventDisableTiming));
      }
      int rc2 = ensure_bytes((void **)&s->d_multi_tiles, &s->multi_tiles_bytes, (size_t)n_max * slot);
      if (rc2 == MT_OK) rc2 = ensure_bytes((void **)&s->d_multi_list, &s->multi_list_bytes, (size_t)n_max * 4);
      if (rc2 != MT_OK) return rc2;
      if (redeal) {
        const int32_t *order = nullptr;
        if (by_map) {
          if ((rc2 = ensure_bytes((void **)&s->d_multi_order, &s->multi_order_bytes, (size_t)tiles_total * 4)) != MT_OK) return rc2;
          if ((rc2 = mt_order_tiles_device(s, s->d_multi_map, map_w, map_h, image_w, image_h, tile_w, tile_h,
                                           s->d_multi_order, s->multi_stream)) != MT_OK) return rc2;
          order = s->d_multi_order;
        }
        if ((rc2 = mt_deal_tiles_device(s, order, image_w, image_h, tile_w, tile_h, n, r, s->d_multi_list, s->multi_stream)) < 0) return rc2;
      }
      if (stats) HIP_TRY(hipMemsetAsync(s->d_counters, 0, ST_COUNT * sizeof(unsigned long long), s->multi_stream));
      const bool counters_were = s->stats_enabled;
      if (stats) s->stats_enabled = true;
      rc2 = launch_render(s, sensor, image_w, image_h, 0, 0, image_w, image_h, tile_w, tile_h, 0, 1, tiles_of(r),
                          max_depth, s->d_multi_tiles, nullptr, s->multi_stream, tiles_of(r) > 0 ? s->d_multi_list : nullptr,
                          list_id);
      s->stats_enabled = counters_were;
      if (rc2 != MT_OK) return rc2;
      if (share_costs) {
        if ((rc2 = ensure_bytes((void **)&s->d_multi_map, &s->multi_map_bytes, map_bytes)) != MT_OK) return rc2;
        HIP_TRY(hipMemsetAsync(s->d_multi_map, 0, map_bytes, s->multi_stream));
        if (tiles_of(r) > 0 && (rc2 = mt_scene_export_costs_device(s, s->d_multi_map, map_w, map_h, s->multi_stream)) != MT_OK) return rc2;
      }
      HIP_TRY(hipEventRecord(s->multi_done, s->multi_stream));
      return MT_OK;
    };
    rcs[(size_t)r] = body();
    if (rcs[(size_t)r] != MT_OK) errs[(size_t)r] = g_err;  // (thread-local text: hand it to the caller's thread)
  };
  if (n == 1) {
    launch_one(0);
  } else {
    std::vector<std::thread> th;
    th.reserve((size_t)n);
    for (int r = 0; r < n; r++) th.emplace_back(launch_one, r);
    for (auto &t : th) t.join();
  }
  for (int r = 0; r < n; r++) {
    if (rcs[(size_t)r] != MT_OK) {
      drain();
      forget();
      return fail(rcs[(size_t)r], "replica %d: %s", r, errs[(size_t)r].c_str());
    }
  }

  // ---- phase 2: gather on the first replica's device, blit, one copy to the host.  (One exit: a failure half way
  // must not return while copies into out_rgb or kernels on the replicas' buffers are still queued.)
  const bool force_peer = root->tune.v[MT_TUNE_MULTI_FORCE_PEER_COPY] != 0.0;
  auto phase2 = [&]() -> int {
    HIP_TRY(hipSetDevice(root->device));
    int rc2 = ensure_bytes((void **)&root->d_multi_frame, &root->multi_frame_bytes, (size_t)image_w * image_h * 3);
    if (rc2 == MT_OK) rc2 = ensure_bytes((void **)&root->d_multi_gather, &root->multi_gather_bytes, (size_t)n * (size_t)n_max * slot);
    if (rc2 == MT_OK) rc2 = ensure_bytes((void **)&root->d_multi_lists, &root->multi_lists_bytes, (size_t)n * (size_t)n_max * 4);
    if (rc2 != MT_OK) return rc2;
    if (redeal) {  // every replica's list, for the blit: the same order, dealt out for every rank
      for (int r = 0; r < n; r++) {
        if ((rc2 = mt_deal_tiles_device(root, by_map ? root->d_multi_order : nullptr, image_w, image_h, tile_w, tile_h, n, r,
                                        root->d_multi_lists + (size_t)r * n_max, root->multi_stream)) < 0) return rc2;
      }
    }
    for (int r = 1; r < n; r++) HIP_TRY(hipStreamWaitEvent(root->multi_stream, scenes[r]->multi_done, 0));  // (every replica: also one without tiles has a map on its way)
    for (int r = 0; r < n; r++) {
      mt_scene *s = scenes[r];
      const int n_r = tiles_of(r);
      if (n_r == 0) continue;
      const uint8_t *src = s->d_multi_tiles;
      if (s->device != root->device || force_peer) {
        if (s->device != root->device) {  // direct xGMI reads where the platform offers them (once per pair; the copy works without, staged)
          static std::mutex mu;
          static std::map<std::pair<int, int>, bool> tried;
          std::lock_guard<std::mutex> lock(mu);
          bool &done = tried[std::make_pair(root->device, s->device)];
          if (!done) {
            done = true;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, root->device, s->device) == hipSuccess && can) {
              if (hipDeviceEnablePeerAccess(s->device, 0) != hipSuccess) (void)hipGetLastError();  // (already enabled)
            } else {
              (void)hipGetLastError();
            }
          }
        }
        uint8_t *dst = root->d_multi_gather + (size_t)r * (size_t)n_max * slot;
        HIP_TRY(hipMemcpyPeerAsync(dst, root->device, s->d_multi_tiles, s->device, (size_t)n_r * slot, root->multi_stream));
        src = dst;
      }
      if ((rc2 = mt_blit_tile_list_device(root, image_w, image_h, tile_w, tile_h, root->d_multi_lists + (size_t)r * n_max, n_r,
                                          src, root->d_multi_frame, root->multi_stream)) != MT_OK) return rc2;
    }
    HIP_TRY(hipMemcpyAsync(out_rgb, root->d_multi_frame, (size_t)image_w * image_h * 3, hipMemcpyDeviceToHost, root->multi_stream));
    if (share_costs) {
      // all maps to the first replica's device, element-wise maximum, and back to every replica -- behind the frame's
      // copy on the same streams
      if ((rc2 = ensure_bytes((void **)&root->d_multi_maps, &root->multi_maps_bytes, (size_t)n * map_bytes)) != MT_OK) return rc2;
      if (!root->multi_comb_done) HIP_TRY(hipEventCreateWithFlags(&root->multi_comb_done, hipEventDisableTiming));
      for (int r = 0; r < n; r++) {
        mt_scene *s = scenes[r];
        unsigned *dst = root->d_multi_maps + (size_t)r * map_w * map_h;
        if (s->device != root->device || force_peer) HIP_TRY(hipMemcpyPeerAsync(dst, root->device, s->d_multi_map, s->device, map_bytes, root->multi_stream));
        else HIP_TRY(hipMemcpyAsync(dst, s->d_multi_map, map_bytes, hipMemcpyDeviceToDevice, root->multi_stream));
      }
      hipLaunchKernelGGL(max_maps_kernel, dim3(256), dim3(256), 0, root->multi_stream, root->d_multi_maps, n, (size_t)map_w * map_h);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(root->multi_comb_done, root->multi_stream));
      for (int r = 0; r < n; r++) {
        mt_scene *s = scenes[r];
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipStreamWaitEvent(s->multi_stream, root->multi_comb_done, 0));
        if (s->device != root->device || force_peer) HIP_TRY(hipMemcpyPeerAsync(s->d_multi_map, s->device, root->d_multi_maps, root->device, map_bytes, s->multi_stream));
        else HIP_TRY(hipMemcpyAsync(s->d_multi_map, root->d_multi_maps, map_bytes, hipMemcpyDeviceToDevice, s->multi_stream));
        if ((rc2 = mt_scene_import_costs_device(s, s->d_multi_map, map_w, map_h, s->multi_stream)) != MT_OK) return rc2;
      }
      HIP_TRY(hipSetDevice(root->device));
    }
    HIP_TRY(hipStreamSynchronize(root->multi_stream));
    return MT_OK;
  };
  if ((rc = phase2()) != MT_OK) {
    const std::string text = g_err;
    drain();
    forget();
    return fail(rc, "%s", text.c_str());
  }
  const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
  // every replica's launch has completed (the root's stream waited for their events); its device status and counters
  auto phase3 = [&]() -> int {
    int worst = MT_OK;
    std::string text;
    for (int r = 0; r < n; r++) {
      mt_scene *s = scenes[r];
      HIP_TRY(hipSetDevice(s->device));
      HIP_TRY(hipStreamSynchronize(s->multi_stream));
      unsigned long long c[ST_COUNT];
      HIP_TRY(hipMemcpy(c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
      const int st_rc = check_status(c);
      if (st_rc != MT_OK && worst == MT_OK) {  // (keep going: every replica is synchronised and read)
        worst = st_rc;
        text = g_err;
      }
      if (stats) {
        HIP_TRY(hipMemset(s->d_counters, 0, sizeof c));
        memset(&stats[r], 0, sizeof(mt_stats));
        fill_stats(c, &stats[r]);
        double a = 0.0, b = 0.0;
        if (tiles_of(r) > 0 && mt_scene_kernel_times(s, 1, &a, &b) == 1) stats[r].kernel_ms = a + b;
        stats[r].total_ms = total_ms;
      }
    }
    if (worst != MT_OK) return fail(worst, "%s", text.c_str());
    return MT_OK;
  };
  if ((rc = phase3()) != MT_OK) {
    const std::string text = g_err;
    drain();
    forget();
    return fail(rc, "%s", text.c_str());
  }
  for (int r = 0; r < n; r++) {
    scenes[r]->multi_geom = geom;
    scenes[r]->multi_rank = r;
    scenes[r]->multi_list_id = list_id;
  }
  root->multi_have_map = share_costs;
  root->multi_sensor = *sensor;
  root->multi_frames_at_rest = frames_at_rest;
  (void)hipSetDevice(root->device);  // (the calling thread's current device: the first replica's, as on entry to phase 2)
  return MT_OK;
}

int mt_intersect_rays(mt_scene *s, int n, const double *rays, int32_t *tri, int32_t *line_no,
                      double *t, double *point, mt_stats *stats) {
  if (!s || n < 0 || (n > 0 && !rays)) return fail(MT_ERR_ARG, "bad ray batch");
  if (stats) memset(stats, 0, sizeof *stats);
  if (n == 0) return MT_OK;
  HIP_TRY(hipSetDevice(s->device));
  double *d_rays = nullptr, *d_t = nullptr, *d_point = nullptr;
  int *d_tri = nullptr, *d_line = nullptr;
  int rc = MT_OK;
  auto cleanup = [&]() {
    (void)hipFree(d_rays); (void)hipFree(d_t); (void)hipFree(d_point);
    (void)hipFree(d_tri); (void)hipFree(d_line);
  };
#define TRY_OR_CLEAN(expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      cleanup();                                                                             \
      return fail(MT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                \
    }                                                                                        \
  } while (0)
  TRY_OR_CLEAN(hipMalloc((void **)&d_rays, (size_t)n * 48));
  TRY_OR_CLEAN(hipMalloc((void **)&d_t, (size_t)n * 8));
  TRY_OR_CLEAN(hipMalloc((void **)&d_point, (size_t)n * 24));
  TRY_OR_CLEAN(hipMalloc((void **)&d_tri, (size_t)n * 4));
  TRY_OR_CLEAN(hipMalloc((void **)&d_line, (size_t)n * 4));
  TRY_OR_CLEAN(hipMemcpy(d_rays, rays, (size_t)n * 48, hipMemcpyHostToDevice));
  TRY_OR_CLEAN(hipMemset(s->d_counters, 0, ST_COUNT * sizeof(unsigned long long)));
  const int block = s->waves_per_block * 64;
  // (DEEP layout: every wave of a launch has an area of global memory -- batches of at most 128 K rays per launch)
  const int per_launch = s->deep ? (n < (1 << 17) ? n : (1 << 17)) : n;
  if ((rc = ensure_deep(s, (size_t)((per_launch + block - 1) / block) * s->waves_per_block)) != MT_OK) {
    cleanup();
    return rc;
  }
  TRY_OR_CLEAN(hipMemcpyAsync(s->d_dev, &s->dev, sizeof(DevScene), hipMemcpyHostToDevice, nullptr));
  TRY_OR_CLEAN(hipEventRecord(s->ev0, nullptr));
  for (int at = 0; at < n; at += per_launch) {
    const int m = n - at < per_launch ? n - at : per_launch;
    MT_LAUNCH_D(intersect_kernel, dim3((m + block - 1) / block), dim3(block), s->lds_bytes, nullptr, s->dev, m,
                d_rays + (size_t)at * 6, d_tri + at, d_line + at, d_t + at, d_point + (size_t)at * 3, s->d_counters);
  }
  TRY_OR_CLEAN(hipGetLastError());
  TRY_OR_CLEAN(hipEventRecord(s->ev1, nullptr));
  TRY_OR_CLEAN(hipDeviceSynchronize());
  if (tri) TRY_OR_CLEAN(hipMemcpy(tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (line_no) TRY_OR_CLEAN(hipMemcpy(line_no, d_line, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (t) TRY_OR_CLEAN(hipMemcpy(t, d_t, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (point) TRY_OR_CLEAN(hipMemcpy(point, d_point, (size_t)n * 24, hipMemcpyDeviceToHost));
  cleanup();
#undef TRY_OR_CLEAN
  mt_stats local;
  if ((rc = mt_scene_read_stats(s, &local)) != MT_OK) return rc;
  if (stats) {
    *stats = local;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
  }
  return MT_OK;
}

}  // extern "C"
