// mt_pool.h -- the LATENCY engine of the frame kernels: a per-wave ray pool.
// (The throughput engine is the per-lane state machine in mt_render.hip; both
// compute every pixel with the same operations in the same order.)
#pragma once
#include "mt_shade.h"
#include "mt_queues.h"

namespace mt {

// ---------------------------------------------------------------------------
// Ray pool of one wave.
//
// TraceRayWorker (mythtracer.cc:13-228) is a recursion: one radiance ray, then
// per light a shadow loop, then up to two child calls whose results are added
// after everything else.  What a call RETURNS is a fixed expression of its
// parts,
//     color = ((((0 + a1_0) + a2_0) [+ a3_0]) + a1_1 ... ) [+ refl * Refl] [+ refr * Tf * Tr]
// but none of the parts depends on another one: every light's shadow loop, the
// reflected ray and the refracted ray are all known the moment the hit is
// shaded (the child rays depend on the hit, the material, level, in_object and
// the reflection coefficient only, :181-189, :192-225).  So a call is kept as a
// RECORD in a per-wave scratch area, its rays are put into a pool, and every
// pass of the wave traces up to 64 rays of the pool -- whatever pixels, levels
// and lights they belong to.  A finished part is stored in its record; the ray
// that completes a record evaluates the expression above in the reference's
// order and hands the value to the parent record.  Same operations, same
// operands, same order => same bits; but a pixel's chain of dependent passes
// is its recursion DEPTH (plus glass crossings of a shadow loop), not the
// number of rays of its recursion tree, and nothing of a ray's context stays
// in registers across the traversal.
//
// Record layout, in doubles (kRecFixed + kLightSlot * n_lights per record):
constexpr int R_RO = 0;        // [3] ray origin; after the hit: intersection point
constexpr int R_RD = 3;        // [3] ray direction; after the hit: shading normal
constexpr int R_SURF = 6;      // [3] surface colour (:58-64)
constexpr int R_REFLDOT = 9;   // reflected_direction . towards_camera (:170)
constexpr int R_COEF = 10;     // current_reflection_coef of this call
constexpr int R_META = 11;     // u64: parent record | level << 32 | flags (M_*)
constexpr int R_PM = 12;       // i32 parts still missing | i32 material index
constexpr int R_RETREFL = 13;  // [3] value returned by the reflection child
constexpr int R_RETREFR = 16;  // [3] value returned by the refraction child
constexpr int R_PX = 19;       // u64: output position of the pixel (root records)
constexpr int R_LIGHTS = 20;    // n_lights slots of kLightSlot doubles
constexpr int kRecFixed = R_LIGHTS;
// Per light: while its shadow loop is under way [0..2] start_point, [3..5]
// light_power, [6] traversing_through_object; afterwards the three terms the
// light adds to the colour: [0..2] (:83-84), [3..5] (:163-167), [6..8]
// (:169-177) and [9] != 0 when the third one exists.
constexpr int kLightSlot = 10;
constexpr unsigned long long M_NOPARENT = 0xffffffffull;
constexpr unsigned long long M_IN_OBJECT = 1ull << 40, M_KIND_REFR = 1ull << 41,
                             M_HAS_REFL = 1ull << 42, M_HAS_REFR = 1ull << 43;
// Pool entry: record | flags | (light + 1) << 24; light field 0 = the record's radiance ray.
constexpr unsigned E_REC_MASK = 0xfffffu, E_ALLOC = 1u << 20, E_CONT = 1u << 21, E_NONE = 0xffffffffu;
constexpr int kRootRecords = 64;

__device__ __forceinline__ V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(double *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ int lanes_below(unsigned long long mask) {  // set bits of mask below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// Memory written by one lane and read by another lane of the SAME wave (pool
// entries, free list, values handed to a parent record): the vector L1 is
// shared by the CU, so workgroup scope is enough to order them.
__device__ __forceinline__ void wave_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void wave_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// Work item -> pixel of the calling lane.  An item is an 8x8-pixel block of a
// tile slot; a work UNIT is the whole block (sub < 0, lane = pixel), one of its
// four 4x4 quarters (sub = 0..3, lanes 0..15) or one of its sixteen 2x2 cells
// (sub = 4..19, lanes 0..3): the longest blocks are handed out in pieces.
struct PoolGeom {
  int px, py;        // image coordinates of this lane's pixel
  bool inside;       // lane has a pixel
  size_t px_index;   // position in the output buffers
};
__device__ __forceinline__ PoolGeom pool_geometry(const RenderParams &P, unsigned item, int sub, int lane) {
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
    ox = (sub & 1) * 4 + (lane & 3);
    oy = (sub >> 1) * 4 + ((lane >> 2) & 3);
    lane_used = lane < 16;
  } else {
    const int c = sub - 4;
    ox = (c & 3) * 2 + (lane & 1);
    oy = (c >> 2) * 2 + ((lane >> 1) & 1);
    lane_used = lane < 4;
  }
  const int lx = (b % P.blocks_x) * 8 + ox;
  const int ly = (b / P.blocks_x) * 8 + oy;
  PoolGeom g;
  g.px = tx0 + lx;
  g.py = ty0 + ly;
  g.inside = lane_used && (lx < cw) && (ly < ch);
  g.px_index = (size_t)j * (size_t)P.tile_w * (size_t)P.tile_h + (size_t)ly * (size_t)cw + (size_t)lx;
  return g;
}

// Does the 8x8 block at (x0, y0) hold a pixel whose primary ray (Sensor::GetRay, camera.cc:65-69) has an exactly zero
// direction component?  (The normalisation cannot create or remove a zero.)
__device__ inline bool block_has_zero_component_ray(const mt_sensor &S, int x0, int y0) {
  bool any = false;
  for (int y = 0; y < 8; y++) {
    double r[3];
    for (int k = 0; k < 3; k++) r[k] = S.start_point[k] + S.delta_scanline[k] * (double)(y0 + y);
    for (int x = 0; x < 8; x++) {
      for (int k = 0; k < 3; k++) any = any || (r[k] + S.delta_pixel[k] * (double)(x0 + x)) == 0.0;
    }
  }
  return any;
}

// Work fetch.  Written WITHOUT a divergent branch: every lane issues the add
// (lane 0 adds 1, the others 0; hipcc merges them into one atomic per wave) and
// lane 0's return value is broadcast.  The obvious form
// `if (lane == 0) v = atomicAdd(..); v = readfirstlane(v);` was miscompiled by
// hipcc 7.2 (the broadcast was folded per control-flow path, so lanes 1..63
// kept looping on item 0 for ever).
__device__ __forceinline__ unsigned pool_fetch_work(unsigned int *counter, int lane) {
  const unsigned v = atomicAdd(counter, lane == 0 ? 1u : 0u);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

template <bool STATS>
__device__ __forceinline__ void pool_flush_item_stats(LaneStats &st, unsigned long long *counters, int lane) {
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
// Cost forecast for a launch without history (first frame of a geometry): the
// primary rays of FOUR pixels per 8x8 block are traced (1/16 of the primary
// rays; a lane per sample, so a wave probes 16 blocks) and the block is
// weighted by the most expensive material they see -- reflective and
// transparent surfaces start recursions -- so that order_kernel can hand
// out the blocks that are probably long first and in pieces.  (One sample per
// block misses the rims of the glass spheres: those blocks then run late and
// whole, and the first frame took 21 ms instead of 14.)  The forecast only
// orders the work; nothing computed for a pixel depends on it.
template <int DEEP>
__global__ __launch_bounds__(256, MT_WAVES_PER_SIMD) void probe_kernel(DevScene S, RenderParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  WaveStack stk;
  stk.bind(smem, wave_in_block, S.tree_depth, S.pack_shift, DEEP != 0);
  const unsigned sample = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned item = sample >> 2;
  const int which = (int)(sample & 3u);
  const bool have = item < P.n_items;
  // pixels (1,1), (5,1), (1,5), (5,5) of the block; (0,0) where those fall off a clipped tile
  PoolGeom g = pool_geometry(P, have ? item : 0u, -1, (1 + 4 * (which & 1)) + 8 * (1 + 4 * (which >> 1)));
  if (!g.inside) g = pool_geometry(P, have ? item : 0u, -1, 0);
  const bool want = have && g.inside;
  const V3 cam_origin = v3_load(P.sensor.origin);
  V3 rd = v3(0, 0, 1);
  if (want) {  // Sensor::GetRay, camera.cc:65-69
    const V3 d = v3_load(P.sensor.start_point) + (v3_load(P.sensor.delta_scanline) * (double)g.py) +
                 (v3_load(P.sensor.delta_pixel) * (double)g.px);
    rd = normalized(d);
  }
  const TraceOut to = trace_wave<false, DEEP>(S.self, stk.base, lane, want, cam_origin.x, cam_origin.y, cam_origin.z,
                                        rd.x, rd.y, rd.z);
  if (to.status != DEV_OK && lane == 0) atomicMax(P.counters + ST_STATUS, (unsigned long long)to.status);
  // Forecast in the unit of the measured costs (64 s_memtime ticks): passes
  // the block will need x what such a pass costs on this kind of surface
  // (coherent shadow rays ~0.25 M ticks a pass; rays mirrored or refracted by
  // curved surfaces are incoherent, ~0.4 M and ~0.8 M).
  unsigned cost = 0u;
  if (want) {
    cost = 4000u;  // one pass: the primary rays
    if (to.prim >= 0) {
      const int m = S.tri_mtl[to.prim];
      if (m >= 0) {
        const mt_material *mm = S.mtls + m;
        const unsigned per_level = 1u + (unsigned)S.n_lights;
        if (mm->transparency > 0.0) cost = 25000u * per_level * (unsigned)(P.max_depth + 1) * 2u;
        else if (mm->reflectance > 0.0) cost = 25000u * per_level * (unsigned)(P.max_depth + 1);
        else cost = 4000u * per_level;
      }
    }
  }
  // the block's forecast = the largest of its four samples (they sit in four adjacent lanes)
  cost = max(cost, (unsigned)__builtin_amdgcn_update_dpp(0, (int)cost, 0xb1, 0xf, 0xf, false));  // quad_perm 1,0,3,2
  cost = max(cost, (unsigned)__builtin_amdgcn_update_dpp(0, (int)cost, 0x4e, 0xf, 0xf, false));  // quad_perm 2,3,0,1
  if (have && which == 0) {
    // (a block on the pixel column or row whose primary rays have a zero direction component: among the longest)
    const PoolGeom g0 = pool_geometry(P, item, -1, 0);
    if (block_has_zero_component_ray(P.sensor, g0.px, g0.py)) cost = max(cost, 16000u * 30u);
    P.item_cost[item] = cost;
  }
}

// ---------------------------------------------------------------------------
// Work order from the block costs of the previous frame (or from
// probe_kernel's forecast), made by order_kernel<2> (mt_order.h):
//   * units are handed out longest first (bucket sort on log2 of the expected
//     cost, eight buckets per octave);
//   * a unit is traced by ONE wave, pass after pass, and the frame cannot end
//     before its longest unit: a block expected to take more than `cut_share`
//     of an even share of the frame's work is handed out as its four 4x4
//     quarters, and if a quarter would still be above that, as its sixteen 2x2
//     cells.  Pieces cost little extra where it matters: the long blocks are
//     the ones whose rays are incoherent, and an incoherent pass costs about in
//     proportion to its rays.  (Hysteresis: a block that was cut stays cut until
//     its cost falls below 0.7 of the threshold.)
// The order changes nothing about what is computed for a pixel.  Costs are in
// units of 64 s_memtime ticks; bits 30-31 of a block's cost word hold the
// granularity it was measured at (its pieces add up in the low bits).
constexpr int kPoolSchedThreads = 1024;
constexpr int kPoolSchedBuckets = 8 * 32;
__device__ __forceinline__ int pool_cost_bucket(unsigned c) {  // descending cost = ascending bucket
  if (c == 0u) return kPoolSchedBuckets - 1;
  const int e = 31 - __builtin_clz(c);                        // octave
  const int f = e >= 3 ? (int)((c >> (e - 3)) & 7u) : 0;      // eighth within it
  return kPoolSchedBuckets - 1 - (e * 8 + f);
}
struct SchedParams {
  float cut_share;      // threshold, as a share of (frame work / waves)
  float piece_time[3];  // expected time of ONE unit relative to the whole block, per granularity
  float piece_work[3];  // measured cost of all pieces of a block relative to the whole block
  float cell_factor;    // quarters above cell_factor x the threshold are cut again
  int own_costs;        // 1: the cost words were written by this engine (granularity in bits 30-31)
};
__device__ __forceinline__ void sched_decide(unsigned word, unsigned forecast, float cut_above, const SchedParams &sp,
                                             int &level, unsigned &unit_cost) {
  const int was = sp.own_costs ? (int)(word >> 30) : 0;
  const float c = (float)forecast;  // as a whole block (forecast_kernel)
  const float thr = was > 0 ? 0.7f * cut_above : cut_above;
  // 2x2 cells (level 2) only for blocks whose QUARTERS would each exceed the
  // threshold several times over: a pass costs about the same whether it
  // carries 64 rays or 4, so small pieces multiply the work.
  level = 0;
  if (c > thr) level = (c * sp.piece_time[1] > sp.cell_factor * thr) ? 2 : 1;
  unit_cost = (unsigned)(c * sp.piece_time[level]);
}
// ---------------------------------------------------------------------------
// The frame kernel: TraceRayWorker for every pixel of the launch's tiles and
// the pixel store.  Persistent waves pull work units in order_kernel's
// order; the first ones (the longest) run at raised wave priority.
// pool_engine is the body shared by pool_kernel (all units of the launch) and hybrid_kernel (mt_render.hip; MIXED: the
// order holds units of both engines -- the pool's first, n_work[1] of them, marked by kHybridPoolSub in order_sub and
// handed out through work_counter[1]; the state machine's behind them through work_counter[2]).  `carry`: a unit index
// this wave has fetched already, or kCarryNone.  Returns kCarryDone when its units are exhausted, kCarryFail after a
// failure.
constexpr unsigned kCarryNone = 0xffffffffu, kCarryDone = 0xfffffffeu, kCarryFail = 0xfffffffdu;
constexpr int kHybridPoolSub = 32;
template <bool STATS, bool MIXED, int DEEP>
__device__ __forceinline__ unsigned pool_engine(const DevScene &S, const RenderParams &P, unsigned carry) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  const int wave_id = blockIdx.x * waves_per_block + wave_in_block;
  WaveStack stk;
  stk.bind(smem, wave_in_block, S.tree_depth, S.pack_shift, DEEP != 0);

  // this wave's scratch: records, pool (a stack of entries), free list (a stack of record numbers)
  const int n_lights = S.n_lights;
  const int RS = kRecFixed + kLightSlot * n_lights;  // even: records stay 16-byte aligned
  const int cap = P.pool_cap;
  char *const scratch = P.pool_scratch + (size_t)wave_id * P.pool_stride;
  double *const recs = (double *)scratch;
  unsigned *const pool = (unsigned *)(scratch + (size_t)cap * RS * sizeof(double));
  unsigned *const freel = pool + (size_t)cap * (n_lights > 0 ? n_lights : 1);
  auto rec_ptr = [&](unsigned r) -> double * {
    return (double *)__builtin_assume_aligned(recs + (size_t)r * RS, 16);
  };

  const mt_material *mtls = S.mtls;
  const mt_light *lights = S.lights;

  LaneStats st;
  st.clear();

  const V3 cam_origin = v3_load(P.sensor.origin);
  const V3 s_start = v3_load(P.sensor.start_point);
  const V3 s_ds = v3_load(P.sensor.delta_scanline);
  const V3 s_dp = v3_load(P.sensor.delta_pixel);
  const unsigned n_work = MIXED ? P.n_work[1] : *P.n_work;
  unsigned result = kCarryDone;
  // Records a pass may need on top of those its rays already have, kept back
  // while the free ones run low (see the throttle below).
  const int reserve = 4 * (P.max_depth + 1);

  for (;;) {
    unsigned w = carry;
    if (carry == kCarryNone) w = pool_fetch_work(P.work_counter + 1, lane);
    carry = kCarryNone;
    if (S.hb) {
      const unsigned long long ex = __builtin_amdgcn_read_exec();
      if (lane == 0) { S.hb[wave_id * 4 + 0] = 1 | ((unsigned long long)w << 8); S.hb[wave_id * 4 + 1] = ex; }
    }
    if (w >= n_work) break;
    const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)P.order_item[w]);
    int sub = __builtin_amdgcn_readfirstlane((int)P.order_sub[w]);
    if (MIXED) sub -= kHybridPoolSub;
    // The longest units run at raised priority: one per SIMD at most (more
    // would only compete with each other), so that they get a SIMD's issue
    // slots ahead of the two short-unit waves that share it.
    if (w < P.prio_units) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);

    const unsigned long long item_t0 = __builtin_amdgcn_s_memtime();
    const PoolGeom g = pool_geometry(P, item, sub, lane);

    // ---- the unit's root records (one per pixel, record number = lane) and their rays
    unsigned n_pool = 0, n_free = 0;
    int hw = kRootRecords;  // records >= hw have not been used in this unit
    {
      const unsigned long long in_mask = __ballot(g.inside);
      if (g.inside) {  // Sensor::GetRay, camera.cc:65-69
        const V3 d = s_start + (s_ds * (double)g.py) + (s_dp * (double)g.px);
        double *R = rec_ptr((unsigned)lane);
        st3(R + R_RO, cam_origin);
        st3(R + R_RD, normalized(d));
        R[R_COEF] = 1.0;  // TraceRay: level 0, not in an object, coefficient 1 (:230-233)
        *(unsigned long long *)(R + R_META) = M_NOPARENT;
        *(unsigned long long *)(R + R_PX) = (unsigned long long)g.px_index;
        pool[lanes_below(in_mask)] = (unsigned)lane | (P.max_depth > 0 ? E_ALLOC : 0u);
      }
      n_pool = (unsigned)__builtin_popcountll(in_mask);
    }
    wave_release();

    // Bound on passes per unit: every pass traces at least one ray; a pixel has
    // at most 2^(max_depth+1) radiance rays, each with one shadow loop per light
    // whose iterations each cross a different surface.
    const long long pass_bound =
        64ll * ((2ll << P.max_depth) * (1 + (long long)n_lights * ((long long)S.n_tris + 2)) + 16);
    long long passes = 0;
    bool failed = false;
#ifdef MT_DIAG
    unsigned diag_nodes = 0, diag_a = 0, diag_t = 0, diag_v = 0, diag_rays = 0;
    unsigned long long diag_trace_ticks = 0;
#endif
    while (n_pool != 0u) {
      wave_acquire();
      // ---- take up to 64 rays from the top of the pool.  A radiance ray below
      // the last level may need two new records when it is shaded; while free
      // records are short only as many of those are taken as can be served with
      // `reserve` records left over, at least one.  Entries are taken from the
      // top (last in, first out), so the pool then works through the recursion
      // depth first, which needs at most 2 new records per level and returns
      // them before the next branch is entered: progress is guaranteed.
      const unsigned take0 = n_pool < 64u ? n_pool : 64u;
      unsigned e = E_NONE;
      if ((unsigned)lane < take0) e = pool[n_pool - 1u - (unsigned)lane];
      unsigned take = take0;
      {
        const int avail = (int)n_free + (cap - hw);
        const unsigned long long am = __ballot(e != E_NONE && (e & E_ALLOC) != 0u);
        const int kmax = avail >= reserve + 2 ? (avail - reserve) >> 1 : (avail >= 2 ? 1 : 0);
        if (__builtin_popcountll(am) > kmax) {
          unsigned long long r = am;
          for (int i = 0; i < kmax; i++) r &= r - 1ull;
          take = (unsigned)__builtin_ctzll(r);  // lanes below the (kmax+1)-th such ray
        }
      }
      if (S.hb) {
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        if (lane == 0) {
          S.hb[wave_id * 4 + 0] = 2 | ((unsigned long long)passes << 8);
          S.hb[wave_id * 4 + 2] = ((unsigned long long)n_pool << 32) | take;
          S.hb[wave_id * 4 + 3] = ex;
        }
      }
      if (take == 0u || ++passes > pass_bound) {
        if (lane == 0) atomicMax(P.counters + ST_STATUS, (unsigned long long)(take == 0u ? DEV_ERR_POOL : DEV_ERR_PIXEL_BOUND));
        failed = true;
        break;
      }
      if ((unsigned)lane >= take) e = E_NONE;
      n_pool -= take;

      // ---- the ray of each entry
      const bool active = e != E_NONE;
      const unsigned rec = e & E_REC_MASK;
      const int li = (int)(e >> 24) - 1;  // -1: radiance ray
      const bool is_shadow = active && li >= 0;
      const bool is_rad = active && li < 0;
      V3 ro = cam_origin, rd = v3(0, 0, 1);
      if (active) {
        const double *R = rec_ptr(rec);
        ro = ld3(R + R_RO);
        if (is_rad) {
          rd = ld3(R + R_RD);
        } else {  // head of the shadow loop, mythtracer.cc:79-99
          const mt_light *lt = lights + li;
          const V3 lpos = v3(lt->position[0], lt->position[1], lt->position[2]);
          const V3 L = normalized(lpos - ro);  // ro holds the intersection point here
          const V3 start = (e & E_CONT) ? ld3(R + R_LIGHTS + li * kLightSlot) : ro;
          ro = start + (L * 0.00001);
          rd = L;
        }
      }
#ifdef MT_DIAG
      const unsigned long long diag_tt0 = __builtin_amdgcn_s_memtime();
#endif
      const TraceOut to = trace_wave<STATS, DEEP>(S.self, stk.base, lane, active, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z);
      add_trace_stats<STATS>(st, to);
#ifdef MT_DIAG
      diag_trace_ticks += __builtin_amdgcn_s_memtime() - diag_tt0;
      {
        const unsigned d = (unsigned)__builtin_amdgcn_readfirstlane((int)to.wave_tri_steps);
        diag_nodes += (unsigned)__builtin_amdgcn_readfirstlane((int)to.wave_node_steps);
        diag_a += d & 0xfffu; diag_t += (d >> 12) & 0x3ffu; diag_v += (d >> 22) & 0x3ffu;
        diag_rays += (unsigned)__builtin_popcountll(__ballot(active));
      }
#endif
      const int prim = to.prim;
      const double t = to.t;

      if (S.hb) {
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        if (lane == 0) { S.hb[wave_id * 4 + 0] = 3 | ((unsigned long long)passes << 8); S.hb[wave_id * 4 + 3] = ex; }
      }
      if (to.status != DEV_OK) {
        if (lane == 0) atomicMax(P.counters + ST_STATUS, (unsigned long long)to.status);
        failed = true;
        break;
      }

      // ---- consume the traversal results (per lane)
      bool finish = false;      // this lane hands `retval` of record `rec_c` to its parent
      bool part_done = false;   // this lane completed one part (a light) of record `rec`
      unsigned rec_c = rec;
      V3 retval = v3(0, 0, 0);
      bool fresh = false;       // a radiance hit that was shaded now: its shadow loops start
      bool again = false;       // shadow loop goes on: same light, next iteration
      bool need_refl = false, need_refr = false;
      V3 c_refl_o = retval, c_refl_d = retval, c_refr_o = retval, c_refr_d = retval;
      double c_refl_coef = 0.0, c_refr_coef = 0.0;
      unsigned long long c_meta = 0ull;  // level and in_object of the children (kind/parent added below)
      if (is_rad) {
        double *R = rec_ptr(rec);
        const unsigned long long meta = *(const unsigned long long *)(R + R_META);
        const int level = (int)((meta >> 32) & 0xffull);
        const bool in_object = (meta & M_IN_OBJECT) != 0ull;
        if (STATS) {
          if (level > 0) st.v[ST_RAYS_SECONDARY]++;
          else st.v[ST_RAYS_PRIMARY]++;
          st.v[ST_BYTES_VECTOR] += 4u + 48u + 16u;  // pool entry, ray, meta
        }
        if (level == 0 && P.out_debug != nullptr) {  // mythtracer.cc:23-36
          mt_debug_px *dbg = P.out_debug + *(const unsigned long long *)(R + R_PX);
          dbg->reserved = 0;
          if (prim < 0) {
            dbg->line_no = -1;
            dbg->point[0] = dbg->point[1] = dbg->point[2] = __builtin_nan("");
          } else {
            dbg->line_no = S.tri_line[prim];
            dbg->point[0] = ro.x + rd.x * t;  // primitive_triangle.cc:141
            dbg->point[1] = ro.y + rd.y * t;
            dbg->point[2] = ro.z + rd.z * t;
          }
        }
        if (prim < 0) {  // mythtracer.cc:23-31
          finish = true;
        } else {
          if (STATS) {
            st.v[ST_SHADED_HITS]++;
            st.v[ST_BYTES_VECTOR] += 72u + 72u + 4u + 64u + 120u;  // vertices, normals, material (index), record fields written
          }
          const V3 Pt = ro + rd * t;  // primitive_triangle.cc:141
          const V3 dir = rd;
          const double *vtx = S.tri_vertex + (size_t)prim * 9;
          const Bary bw = barycentric(vtx, Pt);
          V3 Nn = interpolate(S.tri_normal + (size_t)prim * 9, bw);  // :38
          const V3 towards_camera = -dir;
          double normal_ray_dot = dot(Nn, towards_camera);
          if (normal_ray_dot < 0.0) {  // :42-45
            Nn = -Nn;
            normal_ray_dot = dot(Nn, towards_camera);
          }
          const int mtl = S.tri_mtl[prim];
          if (mtl < 0) {  // :49-52
            normal_ray_dot = (normal_ray_dot + 1.0) * 0.5;
            retval = v3(normal_ray_dot, normal_ray_dot, normal_ray_dot);
            finish = true;
          } else {
            const mt_material *m = mtls + mtl;
            V3 surf = v3(m->ambient[0], m->ambient[1], m->ambient[2]);  // :58
            if (m->tex >= 0) {  // :59-64
              const V3 uvw = interpolate(S.tri_uvw + (size_t)prim * 9, bw);
              surf = surf * texture_color_at(S.texs[m->tex], uvw.x, uvw.y);
            }
            const V3 Rd = dir - Nn * (2 * dot(dir, Nn));  // :68-69
            const double refl_dot = dot(Rd, towards_camera);  // :170, the same for every light
            const double coef = R[R_COEF];
            const double refl = m->reflectance, tr = m->transparency;
            need_refl = level < P.max_depth && refl > 0.0 && coef > 0.01 && !in_object;  // :181-184
            need_refr = level < P.max_depth && tr > 0.0;                                  // :193
            if (need_refl) {  // :70-74, :185-188
              c_refl_o = Pt + (Rd * 0.0001);
              c_refl_d = Rd;
              c_refl_coef = coef * refl;
            }
            if (need_refr) {  // :208-224 (direction unchanged, re-normalised)
              const V3 rdir = normalized(dir);
              c_refr_o = Pt + rdir * 0.00001;
              c_refr_d = rdir;
              c_refr_coef = coef;
            }
            c_meta = ((unsigned long long)(level + 1) << 32) | (in_object ? M_IN_OBJECT : 0ull);
            st3(R + R_RO, Pt);
            st3(R + R_RD, Nn);
            st3(R + R_SURF, surf);
            R[R_REFLDOT] = refl_dot;
            *(unsigned long long *)(R + R_META) =
                meta | (need_refl ? M_HAS_REFL : 0ull) | (need_refr ? M_HAS_REFR : 0ull);
            const int parts = n_lights + (need_refl ? 1 : 0) + (need_refr ? 1 : 0);
            *(unsigned long long *)(R + R_PM) =
                (unsigned long long)(unsigned)parts | ((unsigned long long)(unsigned)mtl << 32);
            if (parts == 0) finish = true;  // no lights, no children: colour (0, 0, 0)
            else fresh = n_lights > 0;
          }
        }
      } else if (is_shadow) {  // ---- one iteration of the shadow loop, mythtracer.cc:94-156
        if (STATS) {
          st.v[ST_RAYS_SHADOW]++;
          st.v[ST_BYTES_VECTOR] += 96u + 4u + 32u + 2u * 80u + 80u;  // light, occluder material, record fields read twice, slot written
        }
        double *R = rec_ptr(rec);
        double *slot = R + R_LIGHTS + li * kLightSlot;
        const mt_light *lt = lights + li;
        const V3 lpos = v3(lt->position[0], lt->position[1], lt->position[2]);
        const V3 Pt = ld3(R + R_RO);
        const V3 L = rd;  // light_direction, as set up before the traversal
        const bool cont = (e & E_CONT) != 0u;
        V3 start = cont ? ld3(slot) : Pt;
        V3 lp = cont ? ld3(slot + 3) : v3(1.0, 1.0, 1.0);
        bool traversing = cont ? (slot[6] != 0.0) : false;
        bool light_done = false, in_shadow = false;
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
                const mt_material *smm = mtls + sm;
                const V3 tf = v3(smm->transmission_filter[0], smm->transmission_filter[1],
                                 smm->transmission_filter[2]);
                lp = lp * (tf * s_tr);
              }
              traversing = !traversing;
              const V3 sp = ro + rd * t;
              start = sp + (L * 0.0000001);  // :137
              if (sqr_distance(Pt, start) > sqr_distance(Pt, lpos)) {
                light_done = true;  // :141-145
              } else if (lp.x <= 0.001 && lp.y <= 0.001 && lp.z <= 0.001) {
                lp = v3(0, 0, 0);  // :149-155
                in_shadow = true;
                light_done = true;
              } else {  // next iteration, :95-99
                st3(slot, start);
                st3(slot + 3, lp);
                slot[6] = traversing ? 1.0 : 0.0;
                again = true;
              }
            }
          }
        }
        if (light_done) {
          // The three terms this light adds to `color` (:83-84, :163-167,
          // :169-177), computed exactly as written there; they are added in
          // light order when the record is complete.
          const int mtl = *((const int *)(R + R_PM) + 1);
          const mt_material *m = mtls + mtl;
          const V3 surf = ld3(R + R_SURF);
          const V3 Nn = ld3(R + R_RD);
          const double refl_dot = R[R_REFLDOT];
          const V3 amb = v3(lt->ambient[0], lt->ambient[1], lt->ambient[2]);
          st3(slot, amb * surf);
          lp.x = std_max(lp.x, amb.x);  // :159-161
          lp.y = std_max(lp.y, amb.y);
          lp.z = std_max(lp.z, amb.z);
          const V3 kd = v3(m->diffuse[0], m->diffuse[1], m->diffuse[2]);
          const V3 ld = v3(lt->diffuse[0], lt->diffuse[1], lt->diffuse[2]);
          st3(slot + 3, kd * surf * dot(L, Nn) * ld * lp);
          double has3 = 0.0;
          if (!in_shadow && refl_dot > 0) {
            const V3 ks = v3(m->specular[0], m->specular[1], m->specular[2]);
            const V3 ls = v3(lt->specular[0], lt->specular[1], lt->specular[2]);
            st3(slot + 6, ks * surf * ::pow(refl_dot, m->specular_exp) * ls);
            has3 = 1.0;
          }
          slot[9] = has3;
          part_done = true;
        }
      }

      // ---- records for the child calls (free list first, then unused ones)
      unsigned e_refl = E_NONE, e_refr = E_NONE;
      {
        const unsigned long long m1 = __ballot(need_refl), m2 = __ballot(need_refr);
        const unsigned n1 = (unsigned)__builtin_popcountll(m1), total = n1 + (unsigned)__builtin_popcountll(m2);
        if (total != 0u) {
          const unsigned j1 = (unsigned)lanes_below(m1), j2 = n1 + (unsigned)lanes_below(m2);
          const unsigned child_alloc = ((int)((c_meta >> 32) & 0xffull) < P.max_depth) ? E_ALLOC : 0u;
          if (need_refl) {
            const unsigned r = j1 < n_free ? freel[n_free - 1u - j1] : (unsigned)hw + (j1 - n_free);
            double *C = rec_ptr(r);
            st3(C + R_RO, c_refl_o);
            st3(C + R_RD, c_refl_d);
            C[R_COEF] = c_refl_coef;
            *(unsigned long long *)(C + R_META) = (unsigned long long)rec | c_meta;
            e_refl = r | child_alloc;
          }
          if (need_refr) {
            const unsigned r = j2 < n_free ? freel[n_free - 1u - j2] : (unsigned)hw + (j2 - n_free);
            double *C = rec_ptr(r);
            st3(C + R_RO, c_refr_o);
            st3(C + R_RD, c_refr_d);
            C[R_COEF] = c_refr_coef;
            *(unsigned long long *)(C + R_META) = ((unsigned long long)rec | c_meta | M_KIND_REFR) ^ M_IN_OBJECT;
            e_refr = r | child_alloc;
          }
          const unsigned from_free = total < n_free ? total : n_free;
          n_free -= from_free;
          hw += (int)(total - from_free);
        }
      }

      // ---- completed records hand their value upwards (at most one level per round)
      wave_release();
      bool complete = false;  // every part of record rec_c is there: evaluate it
      if (part_done) {
        const int old = atomicSub((int *)(rec_ptr(rec) + R_PM), 1);
        complete = old == 1;
      }
      for (int round = 0; round <= P.max_depth + 1; round++) {
        if (__ballot(complete || finish) == 0ull) break;
        wave_acquire();
        if (complete) {
          // TraceRayWorker's colour, :76-226: the lights in order, then the
          // reflection, then the refraction.
          const double *R = rec_ptr(rec_c);
          V3 color = v3(0, 0, 0);
          for (int l = 0; l < n_lights; l++) {
            const double *slot = R + R_LIGHTS + l * kLightSlot;
            color = color + ld3(slot);
            color = color + ld3(slot + 3);
            if (slot[9] != 0.0) color = color + ld3(slot + 6);
          }
          const unsigned long long meta = *(const unsigned long long *)(R + R_META);
          const mt_material *m = mtls + *((const int *)(R + R_PM) + 1);
          if (meta & M_HAS_REFL) color = color + ld3(R + R_RETREFL) * m->reflectance;  // :185-188
          if (meta & M_HAS_REFR) {                                                      // :220-224
            const V3 tf = v3(m->transmission_filter[0], m->transmission_filter[1], m->transmission_filter[2]);
            color = color + ld3(R + R_RETREFR) * tf * m->transparency;
          }
          retval = color;
          finish = true;
          complete = false;
        }
        unsigned parent = 0u;
        bool to_parent = false;
        if (finish) {
          const double *R = rec_ptr(rec_c);
          const unsigned long long meta = *(const unsigned long long *)(R + R_META);
          if ((meta & 0xffffffffull) == M_NOPARENT) {
            uint8_t *o = P.out_rgb + *(const unsigned long long *)(R + R_PX) * 3;  // V3DtoRGB + chunk-local store, :301
            o[0] = channel_to_u8(retval.x);
            o[1] = channel_to_u8(retval.y);
            o[2] = channel_to_u8(retval.z);
          } else {
            parent = (unsigned)(meta & 0xffffffffull);
            st3(rec_ptr(parent) + ((meta & M_KIND_REFR) ? R_RETREFR : R_RETREFL), retval);
            to_parent = true;
          }
          finish = false;
        }
        {  // the finished child records go back to the free list
          const unsigned long long fm = __ballot(to_parent);
          if (to_parent) freel[n_free + (unsigned)lanes_below(fm)] = rec_c;
          n_free += (unsigned)__builtin_popcountll(fm);
        }
        wave_release();
        if (to_parent) {
          const int old = atomicSub((int *)(rec_ptr(parent) + R_PM), 1);
          complete = old == 1;
          rec_c = parent;
        }
      }

      // ---- new rays, bottom to top: refraction children, reflection children,
      // then the shadow rays light by light, light 0 on top -- the next pass
      // starts with rays of one light from neighbouring pixels.
      {
        const unsigned long long m2 = __ballot(e_refr != E_NONE);
        if (e_refr != E_NONE) pool[n_pool + (unsigned)lanes_below(m2)] = e_refr;
        n_pool += (unsigned)__builtin_popcountll(m2);
        const unsigned long long m1 = __ballot(e_refl != E_NONE);
        if (e_refl != E_NONE) pool[n_pool + (unsigned)lanes_below(m1)] = e_refl;
        n_pool += (unsigned)__builtin_popcountll(m1);
        for (int l = n_lights - 1; l >= 0; l--) {
          const bool mine = fresh || (again && li == l);
          const unsigned long long ml = __ballot(mine);
          if (mine) pool[n_pool + (unsigned)lanes_below(ml)] = rec | ((unsigned)(l + 1) << 24) | (again ? E_CONT : 0u);
          n_pool += (unsigned)__builtin_popcountll(ml);
        }
      }
      wave_release();
    }

    if (S.hb && lane == 0) S.hb[wave_id * 4 + 0] = 4;
    {
      const unsigned long long ticks = __builtin_amdgcn_s_memtime() - item_t0;
      if (lane == 0) {
        const unsigned long long c = ticks >> 6;  // the pieces of a block add up
        atomicAdd(P.item_cost + item, c > 0x03ffffffull ? 0x03ffffffu : (unsigned)c);
        if (P.item_cycles) {
          P.item_cycles[(size_t)w * 2] = ticks;
          P.item_cycles[(size_t)(P.n_items * 48u + w) * 2] = item_t0;  // start stamp (scripts/unit_timeline.py)
          P.item_cycles[(size_t)(P.n_items * 48u + w) * 2 + 1] = (unsigned long long)wave_id;
          P.item_cycles[(size_t)w * 2 + 1] =
              ((unsigned long long)passes << 40) | ((unsigned long long)item << 8) | (unsigned)(sub + 1);
#ifdef MT_DIAG
          P.item_cycles[(size_t)(P.n_items * 32u + w) * 2] = diag_trace_ticks;
          P.item_cycles[(size_t)(P.n_items * 16u + w) * 2] = ((unsigned long long)diag_nodes << 32) | diag_a;
          P.item_cycles[(size_t)(P.n_items * 16u + w) * 2 + 1] =
              ((unsigned long long)diag_rays << 40) | ((unsigned long long)diag_t << 20) | diag_v;
#endif
        }
      }
    }
    pool_flush_item_stats<STATS>(st, P.counters, lane);
    if (failed) {
      result = kCarryFail;
      break;
    }
  }
  __builtin_amdgcn_s_setprio(0);
  return result;
}

template <bool STATS, int DEEP>
__global__ __launch_bounds__(256, MT_WAVES_PER_SIMD) void pool_kernel(DevScene S, RenderParams P) {
  (void)pool_engine<STATS, false, DEEP>(S, P, kCarryNone);
  if (S.hb && (threadIdx.x & 63) == 0) S.hb[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4 + 0] = 5;
}

}  // namespace mt
