// mt_render.hip — the frame kernels (primary / schedule / render), the ray-batch
// kernel and the tile blit.
//
// Replaces the pixel loop of MythTracer::RayTrace(WorkChunk*)
// (mythtracer.cc:292-305): Sensor::GetRay (camera.cc:65-69), TraceRayWorker
// (mythtracer.cc:13-228) and V3DtoRGB (:235-241).
//
// Execution model: persistent waves.  Every wave owns a slice of the block's
// LDS (its traversal stack) and pulls work units (8x8-pixel blocks, or quarters
// of the longest ones) from one global counter until none are left; waves never
// synchronise with each other.  One lane renders one pixel -- or, in a quarter,
// four lanes render one pixel, each running the shadow loop of one light.  The
// reference's recursion (reflection, refraction) and its shadow loop are run as
// a per-lane state machine with ONE call site of the wave-synchronous traversal
// (mt_trace.h), so that whatever kind of ray each lane needs next, all 64 lanes
// traverse together.
#include "mt_shade.h"
#include "mt_pool.h"

namespace mt {

enum { MODE_RADIANCE = 0, MODE_SHADOW = 1 };
enum { STAGE_REFL = 0, STAGE_REFR = 1 };

// Recursion frames (one per level that has a child in flight) live in a
// global scratch buffer, [wave][level][slot][lane] doubles: coalesced, and
// touched only once per secondary ray.
// Behind the frames of a wave: `kParkSlots` more slots per lane for the parts of
// a lane's context that only a minority of passes needs -- the state of a shadow
// loop that crosses glass (start point, light power), the three terms of a light
// while a four-lane round waits for its slowest role, the ray direction of the
// call being shaded (needed again after its lights, for the child rays).  Kept
// in registers they would be spilled and reloaded around EVERY traversal call
// (that traffic costs about a tenth of the frame).
constexpr int kParkSlots = 18;
enum { PARK_START = 0, PARK_LP = 3, PARK_ADD1 = 6, PARK_ADD2 = 9, PARK_ADD3 = 12, PARK_DIR = 15 };
struct FrameIO {
  double *base;  // this wave's block
  double *park;  // this wave's parked values, [slot][lane]
  int lane;
  __device__ __forceinline__ void park3(int s, V3 v) const {
    park[(s + 0) * 64 + lane] = v.x; park[(s + 1) * 64 + lane] = v.y; park[(s + 2) * 64 + lane] = v.z;
  }
  __device__ __forceinline__ V3 unpark3(int s) const {
    return V3{park[(s + 0) * 64 + lane], park[(s + 1) * 64 + lane], park[(s + 2) * 64 + lane]};
  }
  __device__ __forceinline__ double *slot(int level, int s) const {
    return base + ((size_t)level * kFrameSlots + s) * 64 + lane;
  }
  __device__ __forceinline__ void put3(int level, int s, V3 v) const {
    *slot(level, s) = v.x; *slot(level, s + 1) = v.y; *slot(level, s + 2) = v.z;
  }
  __device__ __forceinline__ V3 get3(int level, int s) const {
    return V3{*slot(level, s), *slot(level, s + 1), *slot(level, s + 2)};
  }
};

// Work item -> pixel of the calling lane.  An item is an 8x8-pixel block of a
// tile slot.  sub < 0: one lane per pixel, all 64 pixels.  sub = 0..3: the 4x4
// quarter `sub` of the block with FOUR lanes per pixel (pixel = lane / 4,
// role = lane % 4), used to run the shadow loops of up to four lights of a
// pixel side by side.  sub = 4..19: one of the block's sixteen 2x2 cells, four lanes
// per pixel likewise -- blocks whose QUARTERS would be the launch's longest units
// (a pixel column of rays with a zero direction component: a pass costs with the
// number of such rays in it).
struct ItemGeom {
  int px, py;        // image coordinates of this lane's pixel
  bool inside;       // lane has a pixel
  size_t px_index;   // position in the output / hit buffers
};
__device__ __forceinline__ ItemGeom item_geometry(const RenderParams &P, unsigned item, int sub, int lane) {
  const int items_per_tile = P.blocks_x * P.blocks_y;
  const int j = (int)(item / (unsigned)items_per_tile);
  const int b = (int)(item % (unsigned)items_per_tile);
  const int tile = tile_of_slot(P, j);
  const int tx0 = P.region_x + (tile % P.tiles_x) * P.tile_w;
  const int ty0 = P.region_y + (tile / P.tiles_x) * P.tile_h;
  const int cw = min(P.tile_w, P.region_x + P.region_w - tx0);  // edge clipping as
  const int ch = min(P.tile_h, P.region_y + P.region_h - ty0);  // main_net_master.cc:205-206
  int ox, oy;
  bool lane_used = true;
  if (sub < 0) {
    ox = lane & 7;
    oy = lane >> 3;
  } else if (sub < 4) {
    const int q = lane >> 2;
    ox = (sub & 1) * 4 + (q & 3);
    oy = (sub >> 1) * 4 + (q >> 2);
  } else {  // a 2x2 cell (sub = 4 + cell), four lanes per pixel: lanes 0..15
    const int c = sub - 4, q = (lane >> 2) & 3;
    ox = (c & 3) * 2 + (q & 1);
    oy = (c >> 2) * 2 + (q >> 1);
    lane_used = lane < 16;
  }
  const int lx = (b % P.blocks_x) * 8 + ox;
  const int ly = (b / P.blocks_x) * 8 + oy;
  ItemGeom g;
  g.px = tx0 + lx;
  g.py = ty0 + ly;
  g.inside = lane_used && (lx < cw) && (ly < ch);
  g.px_index = (size_t)j * (size_t)P.tile_w * (size_t)P.tile_h + (size_t)ly * (size_t)cw + (size_t)lx;
  return g;
}

// Value of `v` held by role J of the caller's quad (four adjacent lanes), via
// DPP quad_perm.  Must be executed by all lanes of the quad together.
template <int J>
__device__ __forceinline__ int quad_get_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, J | (J << 2) | (J << 4) | (J << 6), 0xf, 0xf, false);
}
template <int J>
__device__ __forceinline__ double quad_get_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)quad_get_i32<J>((int)(unsigned)b);
  const unsigned hi = (unsigned)quad_get_i32<J>((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int J>
__device__ __forceinline__ V3 quad_get_v3(V3 v) {
  return V3{quad_get_f64<J>(v.x), quad_get_f64<J>(v.y), quad_get_f64<J>(v.z)};
}

// Work fetch.  Written WITHOUT a divergent branch: every lane issues the add
// (lane 0 adds 1, the others 0; hipcc merges them into one atomic per wave) and
// lane 0's return value is broadcast.  The obvious form
// `if (lane == 0) v = atomicAdd(..); v = readfirstlane(v);` was miscompiled by
// hipcc 7.2 (the broadcast was folded per control-flow path, so lanes 1..63
// kept looping on item 0 for ever).
__device__ __forceinline__ unsigned fetch_work(unsigned int *counter, int lane) {
  const unsigned v = atomicAdd(counter, lane == 0 ? 1u : 0u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

template <bool STATS>
__device__ __forceinline__ void flush_item_stats(LaneStats &st, unsigned long long *counters, int lane) {
  if (STATS) {  // one atomic per counter per work item
    for (int i = 0; i < ST_WAVE_NODE_STEPS; i++) {
      const unsigned s = wave_sum_u32(st.v[i]);
      if (lane == 0 && s) atomicAdd(counters + i, (unsigned long long)s);
    }
    if (lane == 0) {
      atomicAdd(counters + ST_WAVE_NODE_STEPS, (unsigned long long)st.wave_node_steps);
      atomicAdd(counters + ST_WAVE_TRI_STEPS, (unsigned long long)st.wave_tri_steps);
      atomicAdd(counters + ST_BYTES_SCALAR, (unsigned long long)st.bytes_scalar);
    }
    st.clear();
  }
}

// ---------------------------------------------------------------------------
// Launch 1 of a frame: the primary ray of every pixel (Sensor::GetRay +
// the level-0 OctTree::IntersectRay of TraceRayWorker, mythtracer.cc:18-36).
// Stores the hit (primitive, distance) per pixel, fills the optional debug
// buffer, and files every 8x8 block under a cost class judged from the
// materials it sees, so that launch 2 can start with the expensive blocks:
//   class 2: some pixel hit a transparent material (deep refraction trees),
//   class 1: some pixel hit a reflective one, class 0: everything else.
template <bool STATS, int DEEP>
__global__ __launch_bounds__(256, MT_WAVES_PER_SIMD) void primary_kernel(DevScene S, RenderParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  WaveStack stk;
  stk.bind(smem, wave_in_block, S.tree_depth, S.pack_shift, DEEP != 0);
  const MT_CONST mt_material *mtls = as_const(S.mtls);
  LaneStats st;
  st.clear();
  const V3 cam_origin = v3_load(P.sensor.origin);
  const V3 s_start = v3_load(P.sensor.start_point);
  const V3 s_ds = v3_load(P.sensor.delta_scanline);
  const V3 s_dp = v3_load(P.sensor.delta_pixel);
  for (;;) {
    const unsigned item = fetch_work(P.work_counter, lane);
    if (item >= P.n_items) break;
    const ItemGeom g = item_geometry(P, item, -1, lane);
    V3 rd = v3(0, 0, 1);
    if (g.inside) {  // Sensor::GetRay, camera.cc:65-69
      const V3 d = s_start + (s_ds * (double)g.py) + (s_dp * (double)g.px);
      rd = normalized(d);
    }
    int prim;
    double t;
    const TraceOut to = trace_wave<STATS, DEEP>(S.self, stk.base, lane, g.inside, cam_origin.x, cam_origin.y,
                                          cam_origin.z, rd.x, rd.y, rd.z);
    add_trace_stats<STATS>(st, to);
    prim = to.prim;
    t = to.t;
    const int trc = to.status;
    if (trc != DEV_OK) {
      if (lane == 0) atomicMax(P.counters + ST_STATUS, (unsigned long long)trc);
      break;
    }
    int cls = 0;
    if (g.inside) {
      if (STATS) st.v[ST_RAYS_PRIMARY]++;
      P.hit_prim[g.px_index] = prim;
      P.hit_t[g.px_index] = t;
      if (prim >= 0) {
        const int m = S.tri_mtl[prim];
        if (m >= 0) {
          const MT_CONST mt_material *mm = mtls + m;
          cls = mm->transparency > 0.0 ? 2 : (mm->reflectance > 0.0 ? 1 : 0);
        }
      }
      if (P.out_debug != nullptr) {  // mythtracer.cc:23-36
        mt_debug_px *dbg = P.out_debug + g.px_index;
        dbg->reserved = 0;
        if (prim < 0) {
          dbg->line_no = -1;
          dbg->point[0] = dbg->point[1] = dbg->point[2] = __builtin_nan("");
        } else {
          dbg->line_no = S.tri_line[prim];
          dbg->point[0] = cam_origin.x + rd.x * t;  // primitive_triangle.cc:141
          dbg->point[1] = cam_origin.y + rd.y * t;
          dbg->point[2] = cam_origin.z + rd.z * t;
        }
      }
    }
    const int item_cls = __builtin_amdgcn_readfirstlane(
        (__builtin_amdgcn_ballot_w64(cls == 2) != 0ull) ? 2 : ((__builtin_amdgcn_ballot_w64(cls == 1) != 0ull) ? 1 : 0));
    const unsigned slot = atomicAdd(P.class_count + item_cls, lane == 0 ? 1u : 0u);
    if (lane == 0) {
      P.class_list[(size_t)item_cls * P.n_items + slot] = item;
      // launch 2 adds the block's measured cost; bit 31 = "rendered as quarters"
      P.item_cost[item] = item_cls == 2 ? 0x80000000u : 0u;
    }
    flush_item_stats<STATS>(st, P.counters, lane);
  }
}

// ---------------------------------------------------------------------------
// The work order of a launch with a cost history -- forecast per block, units longest first, the longest blocks in
// pieces -- is made by order_kernel (mt_order.h, included below).  A frame cannot finish before its slowest work
// item, and the per-pixel ray chains cannot be split, so for the state machine:
//   * blocks are handed out longest first (bucket sort on log2 of the cost, eight buckets per octave);
//   * blocks that took longer than quad_share (0.8) of an even share of the
//     frame's work are cut into four quarters with FOUR lanes per pixel (the
//     shadow loops of a pixel's lights run side by side: a shorter chain for
//     about 1.7x the work, which is why only the few longest blocks get it;
//     measured on the 1080p room frame: 0.25 -> 13.3 ms, 0.45 -> 11.7, 0.6 ->
//     10.8, 0.8 -> 10.0, 0.9 -> 10.8, 1.0 -> 12.0, never -> 25.0).
// The order changes nothing about what is computed for a pixel.  Costs are in
// units of 64 s_memtime ticks; a block rendered as quarters reports their sum,
// which is scaled back (kQuadWork) before it is compared again.
// The block costs of the launch described by P, on the whole-block scale of the state machine, into a frame-wide
// map (see RenderParams::cost_map): block (bx, by) of the IMAGE -> map[by * map_w + bx]; blocks of tiles that are not
// this launch's keep what the map holds (the caller zeroes it).  engine = the engine that measured the costs (how
// to read the words: forecast_kernel), wq1 / wq2 = the work factors of its pieces.
__global__ void export_costs_kernel(RenderParams P, int engine, float wq1, float wq2, float w_cells, const unsigned char *form,
                                    unsigned *map, int map_w, int map_h) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n_items) return;
  const unsigned word = P.item_cost[i];
  float c;
  if (engine == 2 || (engine == 3 && form[i] >= 2)) {
    const unsigned lvl = engine == 2 ? (word >> 30) : (unsigned)(form[i] - 1);
    c = (float)(word & 0x3fffffffu);
    c = lvl == 1u ? c / wq1 : (lvl >= 2u ? c / wq2 : c);
  } else {
    // (state machine: bit 31 = measured as quarters, bits 31 + 30 = as cells)
    c = (float)(word & ((word >> 31) ? 0x3fffffffu : 0x7fffffffu));
    if (word >> 31) c /= ((word >> 30) & 1u) ? w_cells : 1.7f;
  }
  const int per_tile = P.blocks_x * P.blocks_y;
  const int j = (int)(i / (unsigned)per_tile), b = (int)(i % (unsigned)per_tile);
  const int tile = tile_of_slot(P, j);
  const int px = P.region_x + (tile % P.tiles_x) * P.tile_w + (b % P.blocks_x) * 8;
  const int py = P.region_y + (tile / P.tiles_x) * P.tile_h + (b / P.blocks_x) * 8;
  const int bx = px >> 3, by = py >> 3;
  if (px >= P.region_x + P.region_w || py >= P.region_y + P.region_h || bx >= map_w || by >= map_h) return;  // (clipped edge tiles)
  const unsigned v = (unsigned)c;
  map[(size_t)by * map_w + bx] = v > 0u ? v : 1u;
}

// maps[0][i] = max over the n maps of maps[r][i] (mt_render_frame_multi: the replicas' cost maps combined)
__global__ void max_maps_kernel(unsigned *maps, int n, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
    unsigned m = maps[i];
    for (int r = 1; r < n; r++) m = max(m, maps[(size_t)r * words + i]);
    maps[i] = m;
  }
}

// ---------------------------------------------------------------------------
// Ownership of a multi-GPU frame's tiles, balanced by cost.  The reference's master hands its WorkChunks out
// dynamically: a worker asks for the next one when it is done (main_net_master.cc:62-80), so no worker idles while
// another has chunks queued.  Ranks that render at the same time cannot pull from one queue without a collective per
// tile; what they can do is compute THE SAME balanced assignment, each by itself, from the frame-wide cost map every
// rank holds after the all-reduce (RenderParams::cost_map): tiles sorted by the cost of their blocks, most expensive
// first (ties: lower tile number first -- the order is a pure function of the map), and dealt out in rounds that
// change direction (0 1 .. N-1, N-1 .. 1 0, 0 1 ..), so that every rank holds one tile of every round: equal tile
// counts -- buffer sizes and the gather stay what they were -- and cost sums that differ by less than one tile of a round.
// tile_cost_kernel: cost[t] = sum over the tile's 8x8 blocks of the map (an all-zero map -- no frame measured yet --
// orders the tiles by number).
__global__ void tile_cost_kernel(const unsigned *map, int map_w, int map_h, int image_w, int image_h, int tile_w,
                                 int tile_h, int tiles_x, int n_tiles, unsigned long long *cost) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const int x0 = (t % tiles_x) * tile_w, y0 = (t / tiles_x) * tile_h;
  const int x1 = min(x0 + tile_w, image_w), y1 = min(y0 + tile_h, image_h);
  unsigned long long sum = 0ull;
  for (int by = y0 >> 3; by <= ((y1 - 1) >> 3) && by < map_h; by++) {
    for (int bx = x0 >> 3; bx <= ((x1 - 1) >> 3) && bx < map_w; bx++) sum += map[(size_t)by * map_w + bx];
  }
  cost[t] = sum;
}
// order[p] = the tile at position p of the descending order (rank by counting: n <= a few thousand tiles).
__global__ void tile_order_kernel(const unsigned long long *cost, int n_tiles, int32_t *order) {
  __shared__ unsigned long long s_c[256];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long mine = t < n_tiles ? cost[t] : 0ull;
  int rank = 0;
  for (int base = 0; base < n_tiles; base += 256) {
    __syncthreads();
    if (base + (int)threadIdx.x < n_tiles) s_c[threadIdx.x] = cost[base + threadIdx.x];
    __syncthreads();
    const int m = min(256, n_tiles - base);
    for (int k = 0; k < m; k++) {
      const unsigned long long c = s_c[k];
      rank += (c > mine || (c == mine && base + k < t)) ? 1 : 0;
    }
  }
  if (t < n_tiles) order[rank] = t;
}
// The tiles of `rank` in slot order: slot q = the rank's tile of round q.  n = dealt_tile_count(n_tiles, world, rank).
__host__ __device__ inline int dealt_position(int q, int world, int rank) { return q * world + ((q & 1) ? world - 1 - rank : rank); }
__host__ inline int dealt_tile_count(int n_tiles, int world, int rank) {
  int n = 0;
  for (int q = n_tiles / world - 1; q <= n_tiles / world; q++) {  // full rounds hold every rank; the last, partial one may
    if (q >= 0 && dealt_position(q, world, rank) < n_tiles) n = q + 1;
  }
  return n;
}
__global__ void deal_tiles_kernel(const int32_t *order, int n_tiles, int world, int rank, int n, int32_t *list) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int p = dealt_position(q, world, rank);
  list[q] = order != nullptr ? order[p] : p;  // (no order yet: positions = tile numbers)
}
// tile_slot[t] = -1 for all t, then slot j for the tiles of the list
__global__ void tile_slot_kernel(const int32_t *list, int n, int32_t *tile_slot, int n_tiles_total, int phase) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (phase == 0) {
    if (i < n_tiles_total) tile_slot[i] = -1;
  } else if (i < n) {
    const int t = list[i];
    if (t >= 0 && t < n_tiles_total) tile_slot[t] = i;  // (a tile outside the grid has no pixels: its slot stays unwritten)
  }
}

constexpr int kSchedThreads = 1024;
constexpr int kSchedBuckets = 8 * 32;
__device__ __forceinline__ int cost_bucket(unsigned c) {  // descending cost = ascending bucket
  if (c == 0u) return kSchedBuckets - 1;
  const int e = 31 - __builtin_clz(c);                        // octave
  const int f = e >= 3 ? (int)((c >> (e - 3)) & 7u) : 0;      // eighth within it
  return kSchedBuckets - 1 - (e * 8 + f);
}
// Work order of a HYBRID launch (engine 3, hybrid_kernel below): blocks whose forecast lies above pool_share of an
// even share of the frame's work go to the ray pool in pieces (4x4 quarters; 2x2 cells when a quarter would still be
// cell_factor times above that; order_sub = kHybridPoolSub + piece); the others go to the state machine (whole, or
// above quad_share as quarters with four lanes per pixel).  The pool's units come first, longest expected unit first
// within either kind.  form[b] records how block b is rendered (the next forecast needs it to read the costs).
}  // namespace mt
#include "mt_order.h"
namespace mt {

// ---------------------------------------------------------------------------
// The frame kernel: TraceRayWorker for every pixel of the launch's tiles and
// the pixel store.  With cost history (P.from_primary == 0) it traces the
// primary rays itself and takes its work units in schedule_kernel's order;
// without, it continues from primary_kernel's hits: reflective blocks first,
// then the transparent ones as quarters, then the rest.  Either way the
// expensive units start first and run at raised wave priority: the frame cannot
// finish before its slowest unit and a pixel's ray chain cannot be split.
// (Plain 4x4 quarters with one lane per pixel do not help -- a pass over 16 lanes
// costs about 80 % of a pass over 64; quarters pay off only with four lanes per
// pixel, i.e. with the pixel's shadow loops running side by side.)
// sm_engine is the body shared by render_kernel and hybrid_kernel (below); `carry` and the return value as for
// pool_engine (mt_pool.h).
template <bool STATS, bool MIXED, int DEEP>
__device__ __forceinline__ unsigned sm_engine(const DevScene &S, const RenderParams &P, unsigned carry) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int wave_id = blockIdx.x * waves_per_block + wave_in_block;
  WaveStack stk;
  stk.bind(smem, wave_in_block, S.tree_depth, S.pack_shift, DEEP != 0);
  FrameIO fio;
  fio.base = P.frames + (size_t)wave_id * (size_t)(P.max_depth > 0 ? P.max_depth : 1) * kFrameSlots * 64;
  fio.park = P.frames + (size_t)gridDim.x * waves_per_block * (size_t)(P.max_depth > 0 ? P.max_depth : 1) * kFrameSlots * 64 +
             (size_t)wave_id * kParkSlots * 64;
  fio.lane = lane;

  const MT_CONST mt_material *mtls = as_const(S.mtls);
  const MT_CONST mt_light *lights = as_const(S.lights);

  LaneStats st;
  st.clear();

  const V3 cam_origin = v3_load(P.sensor.origin);
  const V3 s_start = v3_load(P.sensor.start_point);
  const V3 s_ds = v3_load(P.sensor.delta_scanline);
  const V3 s_dp = v3_load(P.sensor.delta_pixel);
  unsigned result = kCarryDone;
  const bool from_primary = P.from_primary != 0;
  const unsigned n2 = P.class_count[2], n1 = P.class_count[1], n0 = P.class_count[0];
  const unsigned n_work = from_primary ? 4u * n2 + n1 + n0 : *P.n_work;
  // (hybrid launches: the state machine's units lie behind the pool's n_work[1] and have a counter of their own)
  const unsigned w_base = MIXED ? P.n_work[1] : 0u;

  // one work order per XCD (mt_queues.h)
  const bool xcd_queues = !MIXED && !from_primary && P.queues != nullptr;
  QueueFetch qf;
  qf.init();

  for (;;) {
    unsigned w = carry;
    if (xcd_queues) w = qf.next(P, lane, n_work);
    else if (carry == kCarryNone) w = w_base + fetch_work(P.work_counter + (MIXED ? 2 : 1), lane);
    carry = kCarryNone;
    if (S.hb) {
      const unsigned long long ex = __builtin_amdgcn_read_exec();
      if (lane == 0) { S.hb[wave_id * 4 + 0] = 1 | ((unsigned long long)w << 8); S.hb[wave_id * 4 + 1] = ex; }
    }
    if (w >= n_work) break;
    unsigned item;
    int sub = -1;
    if (!from_primary) {
      // Order of the previous frame's measured block costs, longest first
      // (schedule_kernel); the longest blocks come as four quarters.
      item = P.order_item[w];
      sub = (int)P.order_sub[w];
      const unsigned rank = xcd_queues ? qf.rank : w, of = xcd_queues ? qf.len : n_work;
      if (rank < (of >> 4)) __builtin_amdgcn_s_setprio(3);
      else if (rank < (of >> 2)) __builtin_amdgcn_s_setprio(2);
      else __builtin_amdgcn_s_setprio(0);
    } else if (w < n1) {
      // No history: reflective blocks (class 1) first — the longest of them (a
      // floor that mirrors glass: 30+ sequential rays per pixel) cannot be told
      // apart beforehand, so all of them start early; then the class-2 blocks,
      // each as four quarters with four lanes per pixel (a pixel's shadow loops
      // run side by side, which halves its ray chain); then everything else.
      item = P.class_list[(size_t)1 * P.n_items + w];
      __builtin_amdgcn_s_setprio(3);
    } else if (w < n1 + 4u * n2) {
      item = P.class_list[(size_t)2 * P.n_items + ((w - n1) >> 2)];
      sub = (int)((w - n1) & 3u);
      __builtin_amdgcn_s_setprio(2);
    } else {
      item = P.class_list[w - 4u * n2 - n1];
      __builtin_amdgcn_s_setprio(0);
    }
    sub = __builtin_amdgcn_readfirstlane(sub);
    item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);

    const unsigned long long item_t0 = __builtin_amdgcn_s_memtime();
#ifdef MT_DIAG
    unsigned long long diag_trace_ticks = 0;
#endif
    const ItemGeom g = item_geometry(P, item, sub, lane);
    bool alive = g.inside;
    const size_t px_index = g.px_index;

    // ---- per-lane state of TraceRayWorker.  R lanes serve one pixel (R = 4
    // in quad mode, else 1): role 0 owns the pixel and runs the recursion;
    // during a "light round" role j runs the shadow loop of light round_base+j.
    const bool quad = sub >= 0;
    const int R = quad ? 4 : 1;
    const int role = quad ? (lane & 3) : 0;
    const bool owner = role == 0;
    const unsigned long long my_group = quad ? (0xFull << (lane & ~3)) : (1ull << lane);
    enum { MODE_IDLE = 2 };
    int mode = owner ? MODE_RADIANCE : MODE_IDLE;
    int level = 0;
    bool in_object = false;
    double coef = 1.0;
    V3 ro = cam_origin, rd = v3(0, 0, 1);
    if (alive) {  // Sensor::GetRay, camera.cc:65-69 (same arithmetic as launch 1)
      const V3 d = s_start + (s_ds * (double)g.py) + (s_dp * (double)g.px);
      rd = normalized(d);
    }
    // (Not kept across passes, because every live value is spilled around the
    // traversal call: the light direction -- it IS the shadow ray's direction rd --
    // and the reflected direction, recomputed from dir and Nn when it is needed.)
    // Parked instead (FrameIO::park): dir; start point and light power of a shadow
    // loop once it has crossed glass (`crossed`); a role's three terms in a
    // four-lane round.
    V3 Pt = v3(0, 0, 0), Nn = Pt, surf = Pt, color = Pt;
    bool has_light = false, has_add3 = false, crossed = false;
    double refl_dot = 0.0;
    int mtl = -1, li = 0, round_base = 0;
    bool traversing = false;
    bool want_round = false, waiting_round = false;

    // Bound on traversals per item: every ray of a pixel is at most one pass; a
    // pixel needs at most 2^(max_depth+1) radiance rays, each with one shadow
    // loop per light whose iterations each cross a different surface.
    const long long pass_bound =
        (2ll << P.max_depth) * (1 + (long long)S.n_lights * ((long long)S.n_tris + 2)) + 16;
    long long passes = 0;
    while (__ballot(alive) != 0ull) {
      int prim = -1;
      double t = 0.0;
      if (S.hb) {
        const unsigned long long am = __ballot(alive);
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        if (lane == 0) {
          S.hb[wave_id * 4 + 0] = 2 | ((unsigned long long)passes << 8);
          S.hb[wave_id * 4 + 2] = am;
          S.hb[wave_id * 4 + 3] = ex;
        }
      }
      int trc = DEV_OK;
      const bool tracing = alive && mode != MODE_IDLE;
      if (passes == 0 && from_primary) {  // the primary hit was found by launch 1
        prim = tracing ? P.hit_prim[px_index] : -1;
        t = tracing ? P.hit_t[px_index] : 0.0;
      } else {
#ifdef MT_DIAG
        const unsigned long long diag_tt0 = __builtin_amdgcn_s_memtime();
#endif
        const TraceOut to = trace_wave<STATS, DEEP>(S.self, stk.base, lane, tracing, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
#ifdef MT_DIAG
        asm volatile("" :: "v"(to.prim));
        diag_trace_ticks += __builtin_amdgcn_s_memtime() - diag_tt0;
#endif
        add_trace_stats<STATS>(st, to);
        prim = to.prim;
        t = to.t;
        trc = to.status;

      }
      if (S.hb) {
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        if (lane == 0) { S.hb[wave_id * 4 + 0] = 3 | ((unsigned long long)passes << 8); S.hb[wave_id * 4 + 3] = ex; }
      }
      if (trc != DEV_OK || ++passes > pass_bound) {
        if (lane == 0) atomicMax(P.counters + ST_STATUS, (unsigned long long)(trc != DEV_OK ? trc : DEV_ERR_PIXEL_BOUND));
        alive = false;
        break;
      }

      // ---- stage 1 (per lane): consume the traversal result
      bool after_lights = false, do_return = false;
      V3 retval = v3(0, 0, 0);
      V3 add1 = retval, add2 = retval, add3 = retval;  // this role's contributions to `color`, in the order they are added
      if (tracing) {
        if (mode == MODE_RADIANCE) {
          if (STATS) {  // with launch 1, level 0 was counted there
            if (level > 0) st.v[ST_RAYS_SECONDARY]++;
            else if (!from_primary) st.v[ST_RAYS_PRIMARY]++;
          }
          if (prim < 0) {  // mythtracer.cc:23-31
            do_return = true;
          } else {
            if (STATS) {
              st.v[ST_SHADED_HITS]++;
              st.v[ST_BYTES_VECTOR] += 72u + 72u + 4u + 64u;  // vertices, normals, material index, material
            }
            Pt = ro + rd * t;  // primitive_triangle.cc:141
            const V3 dir = rd;
            const double *vtx = S.tri_vertex + (size_t)prim * 9;
            const Bary w = barycentric(vtx, Pt);
            Nn = interpolate(S.tri_normal + (size_t)prim * 9, w);  // :38
            const V3 towards_camera = -dir;
            double normal_ray_dot = dot(Nn, towards_camera);
            if (normal_ray_dot < 0.0) {  // :42-45
              Nn = -Nn;
              normal_ray_dot = dot(Nn, towards_camera);
            }
            mtl = S.tri_mtl[prim];
            if (mtl < 0) {  // :49-52
              normal_ray_dot = (normal_ray_dot + 1.0) * 0.5;
              retval = v3(normal_ray_dot, normal_ray_dot, normal_ray_dot);
              do_return = true;
            } else {
              const MT_CONST mt_material *m = mtls + mtl;
              surf = v3(m->ambient[0], m->ambient[1], m->ambient[2]);  // :58
              if (m->tex >= 0) {  // :59-64
                const V3 uvw = interpolate(S.tri_uvw + (size_t)prim * 9, w);
                surf = surf * texture_color_at(S.texs[m->tex], uvw.x, uvw.y);
              }
              const V3 Rd = dir - Nn * (2 * dot(dir, Nn));  // :68-69
              refl_dot = dot(Rd, towards_camera);  // :170, the same for every light
              fio.park3(PARK_DIR, dir);
              color = v3(0, 0, 0);
              round_base = 0;
              if (S.n_lights > 0) want_round = true;
              else after_lights = true;
              mode = MODE_IDLE;
            }
          }
        } else {  // ---- one iteration of the shadow loop, mythtracer.cc:94-156
          if (STATS) {
            st.v[ST_RAYS_SHADOW]++;
            st.v[ST_BYTES_VECTOR] += 96u + 4u + 32u;  // light, occluder's material index and transparency
          }
          const MT_CONST mt_light *lt = lights + li;
          const V3 lpos = v3(lt->position[0], lt->position[1], lt->position[2]);
          bool light_done = false, in_shadow = false;
          V3 start = Pt, lp = v3(1.0, 1.0, 1.0);  // :90, :94 -- until the loop has crossed glass
          if (crossed) {
            start = fio.unpark3(PARK_START);
            lp = fio.unpark3(PARK_LP);
          }
          if (prim < 0) {
            light_done = true;  // :109-112
          } else {
            const double light_distance = distance(start, lpos);  // :101-102
            if (t > light_distance) {
              light_done = true;  // :115-118
            } else {
              // :121 dereferences shadow_primitive->mtl unconditionally (a
              // crash for material-less occluders); defined here as opaque.
              const int sm = S.tri_mtl[prim];
              const double s_tr = sm >= 0 ? mtls[sm].transparency : 0.0;
              if (s_tr == 0.0) {
                lp = v3(0, 0, 0);
                in_shadow = true;
                light_done = true;
              } else {
                if (!traversing) {  // :129-132
                  const MT_CONST mt_material *smm = mtls + sm;
                  const V3 tf = v3(smm->transmission_filter[0], smm->transmission_filter[1],
                                   smm->transmission_filter[2]);
                  lp = lp * (tf * s_tr);
                }
                traversing = !traversing;
                const V3 sp = ro + rd * t;
                start = sp + (rd * 0.0000001);  // :137 (rd is light_direction)
                if (sqr_distance(Pt, start) > sqr_distance(Pt, lpos)) {
                  light_done = true;  // :141-145
                } else if (lp.x <= 0.001 && lp.y <= 0.001 && lp.z <= 0.001) {
                  lp = v3(0, 0, 0);  // :149-155
                  in_shadow = true;
                  light_done = true;
                } else {
                  ro = start + (rd * 0.00001);  // next iteration, :95-99; rd stays light_direction
                  fio.park3(PARK_START, start);
                  fio.park3(PARK_LP, lp);
                  crossed = true;
                }
              }
            }
          }
          if (light_done) {
            // The three terms this light adds to `color` (:83-84, :163-167,
            // :169-177), computed exactly as written there; the pixel's owner
            // adds them in light order.
            const MT_CONST mt_material *m = mtls + mtl;
            const V3 amb = v3(lt->ambient[0], lt->ambient[1], lt->ambient[2]);
            add1 = amb * surf;
            lp.x = std_max(lp.x, amb.x);  // :159-161
            lp.y = std_max(lp.y, amb.y);
            lp.z = std_max(lp.z, amb.z);
            const V3 kd = v3(m->diffuse[0], m->diffuse[1], m->diffuse[2]);
            const V3 ld = v3(lt->diffuse[0], lt->diffuse[1], lt->diffuse[2]);
            add2 = kd * surf * dot(rd, Nn) * ld * lp;
            has_add3 = false;
            if (!in_shadow && refl_dot > 0) {
              const V3 ks = v3(m->specular[0], m->specular[1], m->specular[2]);
              const V3 ls = v3(lt->specular[0], lt->specular[1], lt->specular[2]);
              add3 = ks * surf * ::pow(refl_dot, m->specular_exp) * ls;
              has_add3 = true;
            }
            if (quad) {  // the round may have to wait for a slower role
              fio.park3(PARK_ADD1, add1);
              fio.park3(PARK_ADD2, add2);
              fio.park3(PARK_ADD3, add3);
            }
            mode = MODE_IDLE;
          }
        }
      }

      // ---- stage 2 (all lanes of the wave together: quad broadcasts)
      // (a) a round is complete when no role of the pixel is still in a shadow
      //     loop: the owner adds the roles' terms in light order.
      {
        const unsigned long long busy = __ballot(alive && mode == MODE_SHADOW);
        const bool round_done = owner && waiting_round && ((busy & my_group) == 0ull);
        const int hl0 = has_light ? 1 : 0, h30 = has_add3 ? 1 : 0;
        if (quad && has_light) {  // possibly computed in an earlier pass
          add1 = fio.unpark3(PARK_ADD1);
          add2 = fio.unpark3(PARK_ADD2);
          add3 = fio.unpark3(PARK_ADD3);
        }
        // role 0 is the owner itself
        if (round_done && has_light) {
          color = color + add1;
          color = color + add2;
          if (has_add3) color = color + add3;
        }
        if (quad) {
          const int hl1 = quad_get_i32<1>(hl0), hl2 = quad_get_i32<2>(hl0), hl3 = quad_get_i32<3>(hl0);
          const int h31 = quad_get_i32<1>(h30), h32 = quad_get_i32<2>(h30), h33 = quad_get_i32<3>(h30);
          const V3 a11 = quad_get_v3<1>(add1), a21 = quad_get_v3<1>(add2), a31 = quad_get_v3<1>(add3);
          const V3 a12 = quad_get_v3<2>(add1), a22 = quad_get_v3<2>(add2), a32 = quad_get_v3<2>(add3);
          const V3 a13 = quad_get_v3<3>(add1), a23 = quad_get_v3<3>(add2), a33 = quad_get_v3<3>(add3);
          if (round_done) {
            if (hl1) { color = color + a11; color = color + a21; if (h31) color = color + a31; }
            if (hl2) { color = color + a12; color = color + a22; if (h32) color = color + a32; }
            if (hl3) { color = color + a13; color = color + a23; if (h33) color = color + a33; }
          }
        }
        if (round_done) {
          waiting_round = false;
          round_base += R;
          if (round_base < S.n_lights) want_round = true;
          else after_lights = true;
        }
      }
      // (b) start of a round: the owner's shading point goes to every role, each
      //     role with a light sets up its shadow loop (head of the light loop,
      //     mythtracer.cc:78-99).
      {
        int wr = (owner && want_round) ? 1 : 0;
        int rb = round_base, mt_ = mtl;
        V3 bP = Pt, bN = Nn, bS = surf;
        double bR = refl_dot;
        if (quad) {
          wr = quad_get_i32<0>(wr);
          rb = quad_get_i32<0>(rb);
          mt_ = quad_get_i32<0>(mt_);
          bP = quad_get_v3<0>(bP);
          bN = quad_get_v3<0>(bN);
          bS = quad_get_v3<0>(bS);
          bR = quad_get_f64<0>(bR);
        }
        if (alive && wr) {
          Pt = bP; Nn = bN; surf = bS; refl_dot = bR; mtl = mt_;
          li = rb + role;
          has_light = li < S.n_lights;
          has_add3 = false;
          if (has_light) {
            const MT_CONST mt_light *lt = lights + li;
            const V3 lpos = v3(lt->position[0], lt->position[1], lt->position[2]);
            rd = normalized(lpos - Pt);  // light_direction, :79-80
            traversing = false;
            crossed = false;
            ro = Pt + (rd * 0.00001);  // start_point = intersection_point, :94-97
            mode = MODE_SHADOW;
          } else {
            mode = MODE_IDLE;
          }
          if (owner) {
            want_round = false;
            waiting_round = true;
          }
        }
      }

      // ---- stage 3 (owner, per lane): recursion decisions and returns
      if (alive && owner) {
        if (after_lights) {
          const MT_CONST mt_material *m = mtls + mtl;
          const double refl = m->reflectance, tr = m->transparency;
          const V3 dir = fio.unpark3(PARK_DIR);
          if (level < P.max_depth && refl > 0.0 && coef > 0.01 && !in_object) {  // :181-189
            if (STATS) st.v[ST_BYTES_VECTOR] += 2u * 88u;  // recursion frame, written now and read at the return
            fio.put3(level, 0, color);
            fio.put3(level, 3, Pt);
            fio.put3(level, 6, dir);
            *fio.slot(level, 9) = coef;
            *(long long *)fio.slot(level, 10) =
                (long long)mtl | ((long long)(in_object ? 1 : 0) << 32) | ((long long)STAGE_REFL << 33);
            const V3 Rd = dir - Nn * (2 * dot(dir, Nn));  // :68-69 again: same operands, same value
            ro = Pt + (Rd * 0.0001);  // :70-74
            rd = Rd;
            coef = coef * refl;
            level++;
            mode = MODE_RADIANCE;
          } else if (level < P.max_depth && tr > 0.0) {  // :192-225
            fio.put3(level, 0, color);
            *(long long *)fio.slot(level, 10) =
                (long long)mtl | ((long long)(in_object ? 1 : 0) << 32) | ((long long)STAGE_REFR << 33);
            const V3 rdir = normalized(dir);  // :208-212 (direction unchanged, re-normalised)
            ro = Pt + rdir * 0.00001;
            rd = rdir;
            in_object = !in_object;
            level++;
            mode = MODE_RADIANCE;
          } else {
            retval = color;
            do_return = true;
          }
        }

        while (do_return) {  // unwinding TraceRayWorker returns
          if (level == 0) {
            if (STATS) st.v[ST_BYTES_VECTOR] += 3u;
            uint8_t *o = P.out_rgb + px_index * 3;  // V3DtoRGB + chunk-local store, :301
            o[0] = channel_to_u8(retval.x);
            o[1] = channel_to_u8(retval.y);
            o[2] = channel_to_u8(retval.z);
            alive = false;
            break;
          }
          level--;
          const long long meta = *(long long *)fio.slot(level, 10);
          const int fm = (int)(meta & 0xffffffffll);
          const bool f_in = ((meta >> 32) & 1) != 0;
          const int stage = (int)((meta >> 33) & 1);
          const MT_CONST mt_material *m = mtls + fm;
          const V3 fcolor = fio.get3(level, 0);
          if (stage == STAGE_REFL) {
            color = fcolor + retval * m->reflectance;  // :185-188
            const double tr = m->transparency;
            if (tr > 0.0) {  // level < max_depth holds: this frame pushed a child
              Pt = fio.get3(level, 3);
              const V3 dir = fio.get3(level, 6);
              coef = *fio.slot(level, 9);
              fio.put3(level, 0, color);
              *(long long *)fio.slot(level, 10) =
                  (long long)fm | ((long long)(f_in ? 1 : 0) << 32) | ((long long)STAGE_REFR << 33);
              const V3 rdir = normalized(dir);
              ro = Pt + rdir * 0.00001;
              rd = rdir;
              in_object = !f_in;
              level++;
              mode = MODE_RADIANCE;
              do_return = false;
            } else {
              retval = color;
            }
          } else {
            const V3 tf = v3(m->transmission_filter[0], m->transmission_filter[1],
                             m->transmission_filter[2]);
            retval = fcolor + retval * tf * m->transparency;  // :220-224
          }
        }
      }
      // helpers leave with their owner
      if (quad) {
        const int oa = quad_get_i32<0>(alive ? 1 : 0);
        if (!owner) alive = alive && (oa != 0);
      }
    }

    if (S.hb && lane == 0) S.hb[wave_id * 4 + 0] = 4;
    {
      const unsigned long long ticks = __builtin_amdgcn_s_memtime() - item_t0;
      if (lane == 0) {
        const unsigned long long c = ticks >> 6;  // quarters of a block add up
        atomicAdd(P.item_cost + item, c > 0x0fffffffull ? 0x0fffffffu : (unsigned)c);
        if (P.item_cycles) {
          P.item_cycles[(size_t)w * 2] = ticks;
          P.item_cycles[(size_t)(P.n_items * 48u + w) * 2] = item_t0;  // start stamp (scripts/unit_timeline.py)
          P.item_cycles[(size_t)(P.n_items * 48u + w) * 2 + 1] = (unsigned long long)wave_id;
          P.item_cycles[(size_t)w * 2 + 1] =
              ((unsigned long long)passes << 40) | ((unsigned long long)item << 8) | (unsigned)(sub + 1);
#ifdef MT_DIAG
          P.item_cycles[(size_t)(P.n_items * 32u + w) * 2] = diag_trace_ticks;
#endif
        }
      }
    }
    flush_item_stats<STATS>(st, P.counters, lane);
  }
  if (S.hb && lane == 0) S.hb[wave_id * 4 + 0] = 5;
  __builtin_amdgcn_s_setprio(0);
  return result;
}

template <bool STATS, int DEEP>
__global__ __launch_bounds__(256, MT_WAVES_PER_SIMD) void render_kernel(DevScene S, RenderParams P) {
  (void)sm_engine<STATS, false, DEEP>(S, P, kCarryNone);
}

// The HYBRID frame kernel (engine 3): the work order starts with the longest blocks, cut into pieces for the ray
// pool (order_sub = kHybridPoolSub + piece: their chains of dependent passes are the recursion depth, not the number of
// rays of the recursion tree), and goes on with everything else for the state machine (lowest cost per ray).  Every
// wave works through the pool's units first -- they are the ones a frame must not end on -- and becomes a state-machine
// wave with the first unit it fetches that is not the pool's.  (One merged order with waves changing over in both
// directions was measured too: 1.5 % slower on a frame without pool units, no better at N = 8.)  Same arithmetic per
// pixel in either part, as in the two kernels above.
// STARTERS (round 4): the two parts have a counter each, and n_work[2] waves -- one per workgroup, as many as the state
// machine has units expected to take more than a third of an even share -- skip the pool's part: the state machine's
// longest blocks start with the launch instead of behind the pool's part (which lasts a quarter of a rank's share of
// the 4K frame at N = 8; a block of 26-32 passes that starts there ends 20-30 % after everybody else).
template <bool STATS, int DEEP>
__global__ __launch_bounds__(256, MT_WAVES_PER_SIMD) void hybrid_kernel(DevScene S, RenderParams P) {
  const bool starter = (threadIdx.x >> 6) == (blockDim.x >> 6) - 1 && blockIdx.x < P.n_work[2];  // (the last wave of a workgroup)
  if (!starter) {
    if (pool_engine<STATS, true, DEEP>(S, P, kCarryNone) == kCarryFail) return;  // a device-side bound tripped (status is set)
  }
  (void)sm_engine<STATS, true, DEEP>(S, P, kCarryNone);
}

// OctTree::IntersectRay for a batch of arbitrary rays: lane i of the grid
// traces ray i.
template <int DEEP>
__global__ __launch_bounds__(256, MT_WAVES_PER_SIMD) void intersect_kernel(DevScene S, int n, const double *rays,
                                                        int *out_tri, int *out_line,
                                                        double *out_t, double *out_point,
                                                        unsigned long long *counters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  WaveStack stk;
  stk.bind(smem, wave_in_block, S.tree_depth, S.pack_shift, DEEP != 0);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool want = i < n;
  double o[3] = {0, 0, 0}, d[3] = {0, 0, 1};
  if (want) {
    for (int k = 0; k < 3; k++) {
      o[k] = rays[(size_t)i * 6 + k];
      d[k] = rays[(size_t)i * 6 + 3 + k];
    }
  }
  LaneStats st;
  st.clear();
  int prim;
  double t;
  const TraceOut to = trace_wave<true, DEEP>(S.self, stk.base, lane, want, o[0], o[1], o[2], d[0], d[1], d[2]);
  add_trace_stats<true>(st, to);
  prim = to.prim;
  t = to.t;
  const int trc = to.status;
  if (trc != DEV_OK && counters && lane == 0) atomicMax(counters + ST_STATUS, (unsigned long long)trc);
  if (want) {
    if (out_tri) out_tri[i] = prim;
    if (out_line) out_line[i] = prim >= 0 ? S.tri_line[prim] : -1;
    if (out_t) out_t[i] = prim >= 0 ? t : __builtin_nan("");
    if (out_point) {
      for (int k = 0; k < 3; k++) {
        out_point[(size_t)i * 3 + k] = prim >= 0 ? o[k] + d[k] * t : __builtin_nan("");
      }
    }
  }
  if (counters) {
    for (int k = 0; k < ST_WAVE_NODE_STEPS; k++) {
      const unsigned s = wave_sum_u32(st.v[k]);
      if (lane == 0 && s) atomicAdd(counters + k, (unsigned long long)s);
    }
    if (lane == 0) {
      atomicAdd(counters + ST_WAVE_NODE_STEPS, (unsigned long long)st.wave_node_steps);
      atomicAdd(counters + ST_WAVE_TRI_STEPS, (unsigned long long)st.wave_tri_steps);
      atomicAdd(counters + ST_BYTES_SCALAR, (unsigned long long)st.bytes_scalar);
    }
  }
}

// BlitWorkChunk (main_net_master.cc:223-236) for a buffer of tile slots.
__global__ void blit_tiles_kernel(int image_w, int image_h, int tile_w, int tile_h, int tiles_x,
                                  int first_tile, int tile_stride, int n_tiles, const int32_t *tile_list,
                                  const uint8_t *tiles, uint8_t *image) {
  const size_t slot_bytes = (size_t)tile_w * tile_h * 3;
  const size_t total = (size_t)n_tiles * tile_w * tile_h;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(p / ((size_t)tile_w * tile_h));
    const int q = (int)(p % ((size_t)tile_w * tile_h));
    const int tile = tile_list != nullptr ? tile_list[j] : first_tile + j * tile_stride;
    const int x0 = (tile % tiles_x) * tile_w, y0 = (tile / tiles_x) * tile_h;
    const int cw = min(tile_w, image_w - x0), ch = min(tile_h, image_h - y0);
    if (q >= cw * ch) continue;
    const int lx = q % cw, ly = q / cw;
    const uint8_t *src = tiles + (size_t)j * slot_bytes + (size_t)q * 3;
    uint8_t *dst = image + ((size_t)(y0 + ly) * image_w + (x0 + lx)) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
  }
}

#define MT_INSTANTIATE(DEEP_)                                                                    \
  template __global__ void render_kernel<true, DEEP_>(DevScene, RenderParams);                   \
  template __global__ void render_kernel<false, DEEP_>(DevScene, RenderParams);                  \
  template __global__ void primary_kernel<true, DEEP_>(DevScene, RenderParams);                  \
  template __global__ void primary_kernel<false, DEEP_>(DevScene, RenderParams);                 \
  template __global__ void pool_kernel<true, DEEP_>(DevScene, RenderParams);                     \
  template __global__ void pool_kernel<false, DEEP_>(DevScene, RenderParams);                    \
  template __global__ void hybrid_kernel<true, DEEP_>(DevScene, RenderParams);                   \
  template __global__ void hybrid_kernel<false, DEEP_>(DevScene, RenderParams);                  \
  template __global__ void probe_kernel<DEEP_>(DevScene, RenderParams);                          \
  template __global__ void intersect_kernel<DEEP_>(DevScene, int, const double *, int *, int *, double *, double *, unsigned long long *);
MT_INSTANTIATE(0)
MT_INSTANTIATE(1)
MT_INSTANTIATE(2)
#undef MT_INSTANTIATE

}  // namespace mt
