// mt_device.h — device-side data layout shared by the kernels and the host
// side of libmythtracer_hip.so.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mythtracer_hip.h"

// Waves per SIMD the frame kernels are compiled for (register budget 512 / n per lane: 2 -> 256 VGPRs, nothing of
// the frame engines' context spilled).  The hit-set walk hides its latencies itself and wants the registers and the
// LDS: 3 (168 VGPRs) measured 6 % slower in rounds 2 and 3, 4 (128) slower still.
#ifndef MT_WAVES_PER_SIMD
#define MT_WAVES_PER_SIMD 2
#endif

namespace mt {

// One octree node, 96 bytes so that a wave fetches it with two scalar loads
// (s_load_dwordx16 + s_load_dwordx8).  lo/c/hi are OctTree::Node::aabb.min,
// ::center, ::aabb.max (octtree.h:47-53): the eight child boxes are
// combinations of these nine planes (octtree.cc:61-100), so the child slab
// tests need nothing else.
struct NodeRec {
  double lo[3];
  double c[3];
  double hi[3];
  int32_t first_child;  // 0 = leaf
  int32_t prim_begin;
  int32_t prim_count;
  int32_t child_mask;   // bit c: child c's subtree holds at least one triangle
  int32_t level;        // depth of the node, root = 0
  int32_t pad1;
};
static_assert(sizeof(NodeRec) == 96, "NodeRec must be 96 bytes");

// Nodes with at least kBigNode triangles are scanned wave-uniformly, smaller
// ones lane-parallel (mt_trace.h).  Swept on the 1080p room frame: 8 -> 11.6 ms,
// 16 -> 10.2, 32 -> 9.75, 48 -> 10.0, 64 -> 10.5.  The triangle stream carries one fp32 box
// per block of kGroupTris consecutive triangles (block b = stream positions
// [16 b, 16 b + 16), whatever nodes they belong to): a ray that provably
// misses the block box fails the reference's per-triangle AABB pre-filter
// (primitive_triangle.cc:73-76) for every member, so the block is skipped.
constexpr int kBigNode = 32;
constexpr int kGroupTris = 16;
#ifndef MT_SUPER_BLOCKS
#define MT_SUPER_BLOCKS 4
#endif
constexpr int kSuperBlocks = MT_SUPER_BLOCKS;   // blocks per second-level box
#ifndef MT_SUPER_MIN
#define MT_SUPER_MIN 12
#endif
constexpr int kSuperMin = MT_SUPER_MIN;  // lists that touch at least this many blocks are scanned through the second level

// Everything the hit-set traversal (mt_trace.h) needs to know about a
// node before it looks at triangles, in one 256-byte record that a wave
// stages in LDS with ONE load instruction (16 lanes x 16 bytes), one node ahead
// of its use: the list, the children, the fp32 union box of every child's
// subtree and of the node's own list (inverted when empty: every ray misses).
struct HsRec {
  int32_t first_child, prim_begin, prim_count, child_mask;
  // fp32 union boxes of the eight children's subtrees, one row of 24 floats per axis: the eight lower planes, the
  // eight upper planes, the lower planes AGAIN -- a lane reads [near x 8][far x 8] as 64 consecutive bytes that start
  // at the row (direction component >= 0) or 32 bytes into it (< 0): no per-plane select by the direction's sign
  float kid[3][24];
  float own[3][3];    // fp32 union box of the own list, per axis lo, hi, lo: [near][far] = two floats at the row or 4 bytes into it
  int32_t sl_begin;   // own list of 1..kHsShortList triangles: its first quad in DevScene::sl_box32, else -1
  int32_t ll_begin;   // own list longer than kHsShortList: its position in the spatially sorted copy (DevScene::ll_*), else -1
  int32_t pad0;
  double planes[9];   // lo xyz, centre xyz, hi xyz (NodeRec): the children's exact boxes
  int32_t pad1[2];
};
static_assert(sizeof(HsRec) == 432, "HsRec must be 432 bytes");
constexpr int kHsRecOwn = 304, kHsRecSl = 340, kHsRecLl = 344, kHsRecPlanes = 352;  // byte offsets the walk reads at (checked in mt_capi.hip)
constexpr int kHsRecLanes = 27;   // 16 bytes per lane
#ifndef MT_HS_SHORT
#define MT_HS_SHORT 32
#endif
constexpr int kHsShortList = MT_HS_SHORT;  // own lists up to this length are scanned from LDS (64 fp32 boxes staged behind the frames; one candidate bit each)
constexpr int kLlPad = 256;        // sorted long lists start at, and are padded to, multiples of this many entries (four supers = one quad of super boxes)
#ifndef MT_LL_DIRECT
#define MT_LL_DIRECT 2
#endif
#ifndef MT_LL_ROUND
#define MT_LL_ROUND 28
#endif
constexpr int kLlRound = MT_LL_ROUND;  // supers whose boxes one copy brings (a multiple of 4; 28 = 7 quads = 1 008 bytes of the 1 536 staged; one bit each in a 32-bit word)
constexpr int kLlDirect = MT_LL_DIRECT;  // lists of up to this many supers (x 64 entries): block quads without the super level (<= 6: one gathered copy; 0 / 1 / 2 / 3 / 4 / 6 measured: 2)
constexpr int kSlQuadFloats = 36;  // DevScene::sl_box32: four boxes = per axis [lo x 4][hi x 4][lo x 4] (144 bytes)

struct DevTexture {
  const void *texels;
  int32_t width, height, format, pad;
};

// Everything the kernels read.  Passed by value as a kernel argument.
struct DevScene {
  const NodeRec *nodes;
  const double *tri_aabb;    // 6 per triangle, node-stream order
  const float *tri_aabb32;   // the same boxes rounded to fp32 (conservative pre-filter)
  const float *grp_aabb32;   // fp32 union box of each block of kGroupTris consecutive stream triangles
  const float *sup_aabb32;   // fp32 union box of each run of kSuperBlocks consecutive blocks (long lists)
  const float *sub_aabb32;   // per node: fp32 union box of all triangles in its subtree
  const HsRec *hs_rec;       // per node (hit-set traversal)
  // Spatially SORTED copies of the long lists (more than kHsShortList triangles: the root's 1 550, its children's
  // 600-900 ...), used by the hit-set walk only.  The reference scans a node's list in list order and lets the later
  // of two equally distant hits win (octtree.cc:186-195); for rays whose Moeller-Trumbore distances cannot be NaN that
  // fold is the minimum under (distance ascending, list position descending), whatever the order of evaluation -- so
  // the CULLING structure over a list is free: here the list's triangles in the order of a median-split tree over
  // their boxes, 16 consecutive ones under one fp32 union box, 4 such blocks under another.  Every list starts at a
  // multiple of kLlPad entries and is padded to one (ll_tri = -1, inverted boxes).  ll_tri = the triangle's stream index
  // (= its list position: the tie rule's rank).
  const int32_t *ll_tri;
  const double *ll_exact;    // 15 per entry: the triangle's fp64 box (6) and vertices (9), for the candidates
  // the entries' fp32 boxes, a union box per 16 entries (block) and per 64 (super), as quads in the layout of sl_box32
  // (kSlQuadFloats floats per four boxes): entry quad e / 4, block quad e / 64 (the four blocks of a super), super
  // quad e / 256 -- every list starts at a multiple of kLlPad
  const float *ll_box_q, *ll_grp_q, *ll_sup_q;
  // The SHORT lists' fp32 boxes once more (1..kHsShortList triangles, list order), laid out for the walk's per-lane
  // reads: per quad of list positions kSlQuadFloats floats = per axis [lo x 4][hi x 4][lo x 4]; a list starts at quad
  // HsRec::sl_begin and is padded to whole quads with inverted boxes.
  const float *sl_box32;
  double bmax[3];            // max |coordinate| of any triangle box, per axis
  const double *tri_vertex;  // 9 per triangle
  const double *tri_normal;  // 9 per triangle
  const double *tri_uvw;     // 9 per triangle
  const int32_t *tri_mtl;
  const int32_t *tri_line;
  const mt_material *mtls;
  const DevTexture *texs;
  const mt_light *lights;
  int32_t n_lights;
  int32_t n_tris;
  int32_t n_nodes;
  int32_t tree_depth;
  int32_t force_mode;  // mt_scene_set_traversal_mode (include/mythtracer_hip.h)
  int32_t scene_regular;  // 1: all coordinates finite and boxes ordered
  int32_t pack_shift;     // 0: 20-byte stack frames; else bits of the (triangle index + 1) field of a packed frame
  // Debug heartbeat (normally NULL): host-visible words the kernel updates so
  // that a stuck launch can be diagnosed from the host (MT_DEBUG_HEARTBEAT=1).
  volatile unsigned long long *hb;
  // Phase profile (only in the -DMT_PROF build): cycle and event sums.
  unsigned long long *prof;
  // DEEP instantiations: per-wave areas of global memory for what does not stay in LDS (wave_deep_bytes)
  char *deep_base;
  unsigned long long deep_stride;
  // Copy of this struct in device memory (refreshed before every launch): the
  // non-inlined traversal takes this one pointer instead of a by-value struct.
  const DevScene *self;
};

constexpr int kProfTimeline = 16384;  // -DMT_PROF: stamps of one wave's walk (tag in the low byte)
enum { PROF_TRACE = 0, PROF_SCAN_RAYPAR, PROF_SCAN_TRANSPOSED, PROF_CHILDREN_UNWIND, PROF_N_RAYPAR,
       PROF_N_TRANSPOSED, PROF_N_CHUNKS, PROF_N_RAYPAR_TRIS, PROF_N_TRACES, PROF_SHADE,
       PROF_SCAN_M2F, PROF_SCAN_M2, PROF_SCAN_M1, PROF_SCAN_M0, PROF_N_M2F, PROF_N_M2, PROF_N_M1, PROF_N_M0,
       PROF_TRIS_M2F, PROF_TRIS_M1, PROF_TRIS_TRANSPOSED,
       PROF_G_GROUPS, PROF_G_LIVE, PROF_G_RANGES, PROF_G_RANGE_TRIS,
       PROF_NIN_SUM, PROF_NIN_LT8, PROF_NIN_LT24, PROF_WANT_SUM,
       PROF_TA_G, PROF_TB_G, PROF_TA_T, PROF_TB_T, PROF_TC_T, PROF_T_RAYS, PROF_M2F_CALL,
       PROF_HS_REC_T, PROF_HS_BIG_T, PROF_HS_SMALL_T, PROF_HS_TRANS_T, PROF_HS_KIDS_T, PROF_HS_RET_T, PROF_HS_CLOSE_T,
       PROF_HS_N_ENTER, PROF_HS_N_BIG, PROF_HS_N_SMALL, PROF_HS_N_EMPTY, PROF_HS_N_TRANS, PROF_HS_N_RET, PROF_HS_N_RETHIT,
       PROF_HS_LANES, PROF_HS_BIG_TRIS, PROF_HS_SMALL_TRIS, PROF_COUNT };

enum {
  ST_RAYS_PRIMARY = 0,
  ST_RAYS_SECONDARY,
  ST_RAYS_SHADOW,
  ST_BOX_TESTS,
  ST_NODE_VISITS,
  ST_TRI_TESTS,
  ST_MT_TESTS,
  ST_SHADED_HITS,
  ST_BYTES_VECTOR,   // bytes requested by per-lane (vector) loads and stores of the path, summed over lanes
  ST_WAVE_NODE_STEPS,
  ST_WAVE_TRI_STEPS,
  ST_BYTES_SCALAR,   // bytes requested by wave-uniform (scalar) loads: one box serves all rays of a wave
  ST_STATUS,  // 0 = ok, else a DEV_ERR_* code: a loop bound tripped (never expected)
  ST_COUNT
};

// Every data-dependent loop in the kernels carries an iteration bound derived
// from the scene size, so that a logic error ends the launch with a status code
// instead of hanging the GPU.
enum { DEV_OK = 0, DEV_ERR_TRAVERSAL_BOUND = 1, DEV_ERR_UNWIND_BOUND = 2, DEV_ERR_PIXEL_BOUND = 3,
       DEV_ERR_POOL = 4 };

// Tiling of one launch (see mt_render_tiles_device in the C ABI).
struct RenderParams {
  mt_sensor sensor;
  int32_t image_w, image_h;
  // region that is cut into tiles, in image coordinates
  int32_t region_x, region_y, region_w, region_h;
  int32_t tile_w, tile_h;
  int32_t tiles_x;         // tiles per row of the region
  int32_t first_tile, tile_stride, n_tiles;
  int32_t blocks_x, blocks_y;  // 8x8 pixel blocks per tile slot
  int32_t max_depth;
  uint32_t n_items;        // n_tiles * blocks_x * blocks_y
  uint8_t *out_rgb;
  mt_debug_px *out_debug;  // nullable; same slot layout as out_rgb
  unsigned long long *counters;  // ST_COUNT
  unsigned int *work_counter;     // [0] launch 1, [1] launch 2 / work units handed out so far
  // ---- throughput engine (mt_render.hip)
  int32_t *hit_prim;              // primary hit per pixel (slot layout of out_rgb)
  double *hit_t;
  unsigned int *class_count;      // [3] blocks per cost class
  unsigned int *class_list;       // [3][n_items] block ids per class
  double *frames;                 // recursion frames scratch
  int32_t from_primary;           // 1: launch 1 ran (hit_prim/hit_t, class lists); 0: order_* lists
  // ---- latency engine (mt_pool.h): per wave pool_cap records, the pool of
  // pending rays and the free list, pool_stride bytes apart
  char *pool_scratch;
  size_t pool_stride;
  int32_t pool_cap;
  uint32_t prio_units;            // the first prio_units units of the order run at raised wave priority
  // ---- both: cost feedback between frames of the same launch geometry (see the
  // schedule kernels): s_memtime ticks each block took in the previous frame
  // (or a forecast), and the work order derived from them.
  unsigned int *item_cost;        // [n_items] ticks (>> 6) of the last frame; pieces of a block add up
  unsigned int *item_forecast;    // [n_items] expected cost of the block as ONE unit in the coming frame
  unsigned int *item_whole;       // [n_items] the block's last cost measured as ONE unit, 0 none (state machine, camera at rest)
  unsigned int *item_qsum;        // [n_items] ... and the last sum over its four quarters: their ratio scales the one to the other
  unsigned int *order_item;       // [<= 16 n_items] block id of work unit w
  signed char *order_sub;         // [<= 16 n_items] -1 = whole block, 0..3 = quarter, 4..19 = 2x2 cell (pool only)
  unsigned int *n_work;           // number of work units in order_item/order_sub
  unsigned long long *item_cycles;  // debug (MT_DEBUG_ITEM_CYCLES): s_memtime ticks per work item
  // Frame-wide map of the previous frame's block costs (whole-block scale), one word per 8x8 block of the IMAGE, row
  // major, cost_map_w words per row -- all ranks' costs combined (mt_scene_import_costs_device).  A re-projected
  // forecast reads it instead of item_cost: the old-image position of a block mostly lies in another rank's tile.
  const unsigned int *cost_map;
  int32_t cost_map_w, cost_map_h;
  // Tile LIST launches (mt_render_tile_list_device: ownership of the frame's tiles balanced by cost): slot j holds tile
  // tile_list[j]; nullptr = the modular form, first_tile + j * tile_stride.  tile_slot[t] = slot of tile t or -1.
  const int32_t *tile_list;
  const int32_t *tile_slot;
  int32_t from_map;  // 1: forecast_kernel reads the costs from cost_map by image position even when the camera is at rest
  // One work order per XCD (MT_TUNE_XCD_QUEUES; nullptr = one order for the chip): queue q holds the units
  // [queues[kQueueStart + q], queues[kQueueStart + q + 1]) of order_item / order_sub, longest first, and hands them out
  // through the counter queues[q * kQueueStride]
  unsigned int *queues;
  unsigned short *item_cell;  // [n_items] the block's cell of the region grid (mt_order.h)
  unsigned int *order_ctl;    // the order kernels' forecast sums, histograms, region grids (kOrdWords; never reset by the host)
  unsigned int *item_unit;    // [n_items] what the block becomes in this launch, packed (order_count_kernel -> order_scatter_kernel)
  unsigned int *order_whist;  // [workgroups][kOrdKeysMax] every workgroup's histogram over the sort keys
};
constexpr int kQueues = 8, kQueueStride = 32, kQueueStart = kQueues * kQueueStride;  // (a counter per 128-byte line)
constexpr int kGridW = 64, kGridH = 32;            // the region grid: forecast cost per cell (order_kernel's area)
constexpr int kQueueWords = kQueueStart + 32;

// slot j of a launch -> tile of the region's grid, and back (-1: not this launch's)
__device__ __forceinline__ int tile_of_slot(const RenderParams &P, int j) {
  return P.tile_list != nullptr ? P.tile_list[j] : P.first_tile + j * P.tile_stride;
}
__device__ __forceinline__ int slot_of_tile(const RenderParams &P, int t) {
  if (P.tile_list != nullptr) return P.tile_slot[t];
  if (t < P.first_tile || (t - P.first_tile) % P.tile_stride != 0) return -1;
  const int j = (t - P.first_tile) / P.tile_stride;
  return j < P.n_tiles ? j : -1;
}

// Bytes of LDS one wave needs for its traversal stack.  Ordered descent: 20-byte frames, or 16-byte ones when a node
// index and a triangle index fit one word together (DevScene::pack_shift != 0).  Hit-set walk (mt_trace.h): trees of
// up to kHsMaxDepth levels use 24-byte frames for the levels that can hold a node with children, plus two
// wave-uniform words per level; the two traversals' frames share the same bytes.
// levels of an octree the hit-set walk takes: its per-level child masks hold 16 levels; LDS -- 24-byte frames per lane
// for all but the leaf level -- is 15 KB per wave at 9 levels (8 waves per CU) and 27 KB at 16 (4 waves per CU: the
// launch configuration halves the waves per workgroup until the budget holds).  Deeper trees take the ordered descent.
#ifndef MT_HS_MAX_DEPTH
#define MT_HS_MAX_DEPTH 16  // (-DMT_HS_MAX_DEPTH=15: what does the ordered descent cost on a tree the walk would take?  profiles/README.md)
#endif
constexpr int kHsMaxDepth = MT_HS_MAX_DEPTH;
// DEEP trees (round 4).  This kernel's throughput goes with the waves a CU holds (the room scene at 4 instead of 8 waves
// per CU: 1.8 times the frame time), and from 12 levels on the frames above leave room for 7, 6, at 16 levels for 5.
// The DEEP instantiations of the kernels (octrees of kDeepFromDepth .. kHsMaxDepth levels) keep only the walk's frames
// of the first kHsLdsLevels levels in LDS -- where nearly all of a walk's steps happen -- and put the deeper levels'
// frames and the whole stack of the ordered descent (irregular rays only, in such trees) into a per-wave area of
// global memory (DevScene::deep_base): 19 KB of LDS per wave whatever the depth, 8 waves per CU.
constexpr int kHsLdsLevels = 10;
constexpr int kDeepFromDepth = 12;
// ... and with the frames out of the LDS's way a third set of instantiations (DEEP = 2, octrees of 17 .. 24 levels)
// carries a third word of per-level child bytes: the walk takes octrees of up to 24 levels there (deeper ones, or
// DEEP_LAYOUT switched off above 16: ordered descent).  A set of its own: the third word in the 12 .. 16-level
// kernels costs the loft 6 % of its frame time.  0 no, 1 deep, 2 deep and wide.
constexpr int kHsMaxDepthDeep = MT_HS_MAX_DEPTH < 16 ? MT_HS_MAX_DEPTH : 24;
__host__ __device__ inline int deep_layout(int depth) {
  return depth < kDeepFromDepth || depth > kHsMaxDepthDeep ? 0 : (depth <= kHsMaxDepth ? 1 : 2);
}
__host__ __device__ inline size_t wave_frames_bytes(int depth, bool packed, bool deep = false) {
  if (deep) {
    const int lf = depth - 1 < kHsLdsLevels ? depth - 1 : kHsLdsLevels;
    return (((size_t)lf * 64 * 24 + (size_t)(depth - 1) * 8 + 15) & ~(size_t)15) + 2 * sizeof(HsRec) + (size_t)(depth - 1) * 80;
  }
  size_t n = (size_t)depth * 64 * (packed ? 16 : 20);
  if (depth <= kHsMaxDepth && depth > 1) {
    // frames, (node, first child) per level, room for two staged records (16-byte aligned), per level the nine
    // planes of the frame's node (80 bytes)
    size_t hs = (((size_t)(depth - 1) * (64 * 24 + 8) + 15) & ~(size_t)15) + 2 * sizeof(HsRec) + (size_t)(depth - 1) * 80;
    if (hs > n) n = hs;
  }
  return n;
}
__host__ __device__ inline size_t wave_stack_bytes(int depth, bool packed, bool deep = false) {
  size_t n = wave_frames_bytes(depth, packed, deep) + 64 * 24;  // the fp32 boxes of a short list: (kHsShortList / 4) quads of 144 bytes = 1 152 (the counters are in registers)
  return n;
}
// bytes of a wave's area in global memory (DEEP instantiations): the ordered descent's stack, 20 bytes per lane and
// level, then the walk's frames of the levels from kHsLdsLevels on, 24 bytes per lane and level
__host__ __device__ inline size_t wave_deep_bytes(int depth) {
  const int ld = depth - 1 > kHsLdsLevels ? depth - 1 - kHsLdsLevels : 0;
  return (size_t)depth * 64 * 20 + (size_t)ld * 64 * 24;
}

constexpr int kFrameSlots = 11;  // throughput engine: 10 doubles + 1 packed meta word per recursion frame

}  // namespace mt
