// mt_order.h -- the work order of a launch with a cost history (or probe_kernel's guess): three small kernels on 64
// workgroups of 1 024 threads.
//
// Rounds 2-4 made the order in two kernels behind two memsets and a copy: forecast_kernel (a thread per block) and a
// schedule kernel that sorted the units on ONE workgroup (45 us: a bucket sort through same-address LDS atomics, the
// region table of the per-XCD work orders built by a thread per stripe) -- 105 us in front of a 4.7 ms frame kernel.
// Now every thread keeps its own blocks from the first step to the last, and a parallel counting sort does the rest:
//   1  order_forecast_kernel: forecast of the thread's blocks (cost history, re-projected when the camera moved: see
//      forecast_item); their sum and -- state-machine launches with one work order per XCD (mt_queues.h) -- the region
//      grid; the counters of the coming frame kernel and the OTHER launch's area are zeroed on the way (no memsets);
//   2a order_count_kernel: what every block becomes (whole / quarters / pool pieces: the engines' rules, unchanged;
//      kept as one packed word per block) and the workgroup's histogram over (region or engine part, cost bucket),
//      written to global memory;
//   2b order_scatter_kernel: every workgroup sums the histograms -- all of them: the keys' totals, whose prefix sums put
//      the units longest first within a region / part; the workgroups before it: its offset within each key -- and
//      scatters its blocks' units.
// Within a key the units come workgroup by workgroup, i.e. in the order of the picture's blocks (inside a workgroup's
// range in LDS-atomic order); nothing computed for a pixel depends on any of it.
// 53 us for the three steps as ONE kernel with two grid barriers (sc1 hand-offs) -- but a grid barrier needs all its
// workgroups resident, and two such kernels of different scenes or processes that each hold half of the CUs wait for each
// other until their spins time out; the kernel boundaries cost a few microseconds more and assume nothing.
#pragma once
#include "mt_pool.h"

namespace mt {

constexpr int kOrdGroups = 64;       // workgroups of each of the three kernels (MT_TUNE_ORDER_GROUPS; swept 16 .. 256: 64)
constexpr int kOrdGroupsMax = 256;
constexpr int kOrdThreads = 1024;
constexpr int kOrdKeysMax = kQueues * 256;
// RenderParams::order_ctl, unsigned words (zero at allocation, never reset by the host):
//   two areas used by alternate launches (a launch zeroes the other one), each: [0..1] u64 sum of the forecasts,
//   [2] starters, [32 ..) the region grid
constexpr int kOrdArea = 32, kOrdAreaGrid = 32, kOrdAreaWords = kOrdAreaGrid + kGridW * kGridH;
constexpr int kOrdWords = kOrdArea + 2 * kOrdAreaWords;

struct ForecastArgs {
  mt_sensor old;       // the camera the costs were measured under
  int reproject, radius, pool;
  float w1, w2;
  unsigned unseen;
  float blend;
  const unsigned char *form;
  float wq, wc, step_px;
  float w_cells;  // state machine: summed cost of a block's sixteen cells over its cost as one unit
  int old_irr;   // the old camera's frame had pixels whose primary rays have a zero direction component
  int new_irr;   // ... and so has this one's (found on the host, exactly: most frames have none, and looking for them is half of a forecast's instructions)
};
struct OrderArgs {
  int n_waves;
  unsigned epoch;          // this scene's order launches so far (which area)
  int queue_mode;          // state machine: MT_TUNE_XCD_QUEUES (0: one order)
  float quad_share, quad_keep;                                         // state machine
  float cell_share, cell_time;   // ... blocks with zero-component rays whose quarters are expected above cell_share x the cutting threshold go out as sixteen cells
  int new_irr;                   // this frame has pixels whose primary rays have a zero direction component (ForecastArgs::new_irr)
  float pool_share, piece_time1, piece_time2, cell_factor, starter_share;  // hybrid
  unsigned max_starters;
  unsigned char *form_out;
  SchedParams sp;          // ray pool
};

// a / b and a % b for 0 <= a < 2^22, b >= 1 through the divisor's reciprocal (eight instructions instead of the thirty
// of an integer division: a re-projected forecast makes sixty of them per block)
struct FastDiv {
  int b;
  float inv;
  __device__ __forceinline__ void set(int b_) { b = b_ > 0 ? b_ : 1; inv = 1.0f / (float)b; }
  __device__ __forceinline__ int div(int a, int &rem) const {
    int q = (int)((float)a * inv);
    int r = a - q * b;
    if (r < 0) { q--; r += b; }
    else if (r >= b) { q++; r -= b; }
    rem = r;
    return q;
  }
};

// Cost forecast of block i from the previous frame's measured block costs.  When the
// camera has not moved a block's forecast is its own last cost.  When it has
// (an animation: main_local.cc:51-76 turns it 2 degrees per frame, 25..70
// pixels depending on where in the picture), what was expensive in block b is
// now somewhere else: the centre ray of every block of the NEW frame is
// projected into the OLD camera's image (exact for a rotation; a translation
// adds parallax, hence `radius`), and the forecast is the maximum over the old
// blocks within `radius` of that position -- or, for a direction the old frame
// did not see, `unseen`, the mean cost of a block.  The forecast only orders
// the work and picks the blocks handed out in pieces.
// pool != 0: cost words of the latency engine (granularity in bits 30-31).
// form != nullptr: the previous launch was a HYBRID one -- form[b] = how block b was rendered: 0 / 1 by the state
// machine (whole / as quarters: bit 31 of its cost word, as ever), 2 / 3 by the ray pool as quarters / cells, whose
// summed costs are scaled to the state machine's whole-block scale by wq / wc.
// Writes item_forecast[i] (and the two forms' memory, item_whole / item_qsum); bx, by = the block's position in the
// launch's region, in blocks (the region grid's coordinates).
__device__ __forceinline__ unsigned forecast_item(const RenderParams &P, const ForecastArgs &A, unsigned i, int &gx, int &gy) {
  const mt_sensor &old = A.old;
  const unsigned char *form = A.form;
  auto cost_from = [&](unsigned word, unsigned fmv) -> unsigned {  // fmv: the block's form in a hybrid launch (else 0)
    if (fmv >= 2u) {
      return (unsigned)((float)(word & 0x3fffffffu) / (fmv == 2u ? A.wq : A.wc));
    }
    if (A.pool) {
      const unsigned lvl = word >> 30;
      const float c = (float)(word & 0x3fffffffu);
      return (unsigned)(lvl == 1u ? c / A.w1 : (lvl >= 2u ? c / A.w2 : c));
    }
    // state machine: bit 31 = measured as quarters (their sum), bits 31 + 30 = as sixteen cells
    const float c = (float)(word & ((word >> 31) ? 0x3fffffffu : 0x7fffffffu));
    return (unsigned)((word >> 31) ? c / (((word >> 30) & 1u) ? A.w_cells : A.w1) : c);
  };
  auto cost_of = [&](unsigned word, unsigned idx) -> unsigned { return cost_from(word, form != nullptr ? (unsigned)form[idx] : 0u); };
  const int per_tile = P.blocks_x * P.blocks_y;
  FastDiv d_per_tile, d_blocks_x, d_tiles_x, d_tile_w, d_tile_h;
  d_per_tile.set(per_tile); d_blocks_x.set(P.blocks_x); d_tiles_x.set(P.tiles_x); d_tile_w.set(P.tile_w); d_tile_h.set(P.tile_h);
  int b;
  const int j = i < (1u << 22) ? d_per_tile.div((int)i, b) : (int)(i / (unsigned)per_tile);
  if (!(i < (1u << 22))) b = (int)(i % (unsigned)per_tile);
  const int tile = tile_of_slot(P, j);
  int b_col, tile_col;
  const int b_row = d_blocks_x.div(b, b_col), tile_row = tile < (1 << 22) && tile >= 0 ? d_tiles_x.div(tile, tile_col) : tile / P.tiles_x;
  if (!(tile < (1 << 22) && tile >= 0)) tile_col = tile % P.tiles_x;
  if (!A.reproject && P.from_map) {
    // A tile-list launch whose list differs from the previous launch's (the ownership of the tiles was re-balanced):
    // the slots' own cost words belong to other tiles; the block's cost comes from the frame-wide map (all ranks' costs
    // of the previous frame, mt_scene_import_costs_device) at its own position.  0 = nobody reported it.
    const int bx_ = (P.region_x + tile_col * P.tile_w + b_col * 8) >> 3;
    const int by_ = (P.region_y + tile_row * P.tile_h + b_row * 8) >> 3;
    unsigned f = A.unseen;
    if (bx_ < P.cost_map_w && by_ < P.cost_map_h) {
      const unsigned c = P.cost_map[(size_t)by_ * P.cost_map_w + bx_];
      if (c != 0u) f = c;
    }
    P.item_forecast[i] = f;
    gx = bx_ - (P.region_x >> 3);
    gy = by_ - (P.region_y >> 3);
    return f;
  }
  if (!A.reproject) {
    // (bit 31, state machine only: the block was rendered as quarters -- the scheduler's hysteresis)
    // blend > 0 (the camera stands still and the previous launch made a forecast too): the new forecast is a mix
    // of the old one and the measurement -- a block near a cutting threshold is otherwise measured whole in one
    // frame and in pieces in the next, and the schedule alternates between two states
    // (everything this block's forecast reads is requested before anything is looked at: one round trip, not four)
    const unsigned word = P.item_cost[i];
    const bool forms = !A.pool && P.item_whole != nullptr;
    const unsigned w_old = forms ? P.item_whole[i] : 0u, qs_old = forms ? P.item_qsum[i] : 0u;
    const unsigned f_old = A.blend > 0.0f ? P.item_forecast[i] : 0u;
    unsigned f = cost_of(word, i);
    if (forms && !(form != nullptr && form[i] >= 2) && (word >> 30) != 3u) {  // (a block measured in cells keeps the fixed factor)
      // A block near the cutting threshold: measured whole it costs c, in four pieces s, and s / w1 is only a guess
      // of c -- when the guess is below the threshold and c above it, the block changes its form every few frames
      // and every frame that renders it whole ends late.  Once both have been measured, THEIR ratio scales the one to
      // the other, and the forecast of the block no longer depends on the form it was rendered in.
      const unsigned c = word & ((word >> 31) ? 0x3fffffffu : 0x7fffffffu);
      if (word >> 31) P.item_qsum[i] = c; else P.item_whole[i] = c;
      const unsigned w = (word >> 31) ? w_old : c, qs = (word >> 31) ? c : qs_old;
      if ((word >> 31) && w > 0u && qs > 0u) {
        const float ratio = fminf(fmaxf((float)qs / (float)w, 1.0f), 4.0f);
        f = (unsigned)((float)c / ratio);
      }
    }
    if (A.blend > 0.0f) f = (unsigned)(A.blend * (float)(f_old & 0x7fffffffu) + (1.0f - A.blend) * (float)f);
    P.item_forecast[i] = f | (A.pool ? 0u : (word & 0x80000000u));
    gx = ((tile_col * P.tile_w) >> 3) + b_col;
    gy = ((tile_row * P.tile_h) >> 3) + b_row;
    return f;
  }
  const int tx0 = P.region_x + tile_col * P.tile_w, ty0 = P.region_y + tile_row * P.tile_h;
  const double px = tx0 + b_col * 8 + 4.0, py = ty0 + b_row * 8 + 4.0;  // block centre
  double d[3], m[3][3];
  for (int k = 0; k < 3; k++) {
    d[k] = P.sensor.start_point[k] + P.sensor.delta_scanline[k] * py + P.sensor.delta_pixel[k] * px;
    m[k][0] = old.delta_pixel[k];
    m[k][1] = old.delta_scanline[k];
    m[k][2] = -d[k];
  }
  // old.start + old.dp * x + old.ds * y = lambda * d   (Cramer's rule)
  auto det3 = [](const double a[3][3]) {
    return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
  };
  const double D = det3(m);
  unsigned best = A.unseen;
  if (D != 0.0) {
    double mx[3][3], my[3][3], ml[3][3];
    for (int k = 0; k < 3; k++) {
      for (int c = 0; c < 3; c++) mx[k][c] = my[k][c] = ml[k][c] = m[k][c];
      mx[k][0] = -old.start_point[k];
      my[k][1] = -old.start_point[k];
      ml[k][2] = -old.start_point[k];
    }
    double ox = det3(mx) / D, oy = det3(my) / D;
    const double lambda = det3(ml) / D;
    // A direction the old frame did not see, but not by much (the strip that a turning camera brings into view: up to
    // two tiles wide at 3840 pixels): what lies just inside the old frame's edge is the better guess than the mean
    // cost of a block -- objects continue across the edge.
    if (lambda > 0.0) {
      const double slack = 160.0;
      if (ox < P.region_x && ox >= P.region_x - slack) ox = P.region_x + 0.5;
      if (oy < P.region_y && oy >= P.region_y - slack) oy = P.region_y + 0.5;
      if (ox >= P.region_x + P.region_w && ox < P.region_x + P.region_w + slack) ox = P.region_x + P.region_w - 0.5;
      if (oy >= P.region_y + P.region_h && oy < P.region_y + P.region_h + slack) oy = P.region_y + P.region_h - 0.5;
    }
    // old pixel -> old block of the SAME launch geometry (single-chunk launches and tiles alike)
    if (lambda > 0.0 && ox >= P.region_x && oy >= P.region_y && ox < P.region_x + P.region_w && oy < P.region_y + P.region_h) {
      // The old blocks within `radius`, nine at a time: first where each one's cost word lies, then the nine loads
      // together, then the maximum (one position after the other, every one waiting for its own load, this loop was
      // most of the kernel's 26 us).
      bool any = false;
      unsigned mxc = 0u;
      const int side = 2 * A.radius + 1, n_pos = side * side;
      constexpr unsigned kNone = 0xffffffffu;
      for (int p0 = 0; p0 < n_pos; p0 += 9) {
        unsigned at[9];
#pragma unroll
        for (int k = 0; k < 9; k++) {
          at[k] = kNone;
          const int p = p0 + k;
          if (p >= n_pos) continue;
          const int dx = p % side - A.radius, dy = p / side - A.radius;
          const double qx = ox + (double)A.step_px * dx, qy = oy + (double)A.step_px * dy;
          if (qx < P.region_x || qy < P.region_y || qx >= P.region_x + P.region_w || qy >= P.region_y + P.region_h) continue;
          // (an old block whose rays had a zero direction component says nothing about the turned camera's rays)
          if (A.old_irr && block_has_zero_component_ray(old, (int)qx & ~7, (int)qy & ~7)) continue;
          if (P.cost_map != nullptr) {  // every rank's costs of the old frame (0 = nobody reported that block)
            const int mx_ = (int)qx >> 3, my_ = (int)qy >> 3;
            if (mx_ < P.cost_map_w && my_ < P.cost_map_h) at[k] = (unsigned)((size_t)my_ * P.cost_map_w + mx_);
            continue;
          }
          int lx, ly;
          const int tx = d_tile_w.div((int)qx - P.region_x, lx), ty = d_tile_h.div((int)qy - P.region_y, ly);
          // (a launch of ONE tile -- a chunk, the whole frame: slot 0; else the tile's slot, if it is this launch's at all)
          const int jj = (P.n_tiles == 1 && P.tile_list == nullptr) ? (ty * P.tiles_x + tx == P.first_tile ? 0 : -1)
                                                                                            : slot_of_tile(P, ty * P.tiles_x + tx);
          if (jj < 0) continue;  // another rank's tile
          at[k] = (unsigned)((size_t)jj * per_tile + (size_t)(ly / 8) * P.blocks_x + lx / 8);
        }
        unsigned word[9], fm[9];
        const unsigned *src = P.cost_map != nullptr ? P.cost_map : P.item_cost;
#pragma unroll
        for (int k = 0; k < 9; k++) {
          word[k] = at[k] != kNone ? src[at[k]] : 0u;
          fm[k] = (at[k] != kNone && P.cost_map == nullptr && form != nullptr) ? (unsigned)form[at[k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
          if (at[k] == kNone) continue;
          if (P.cost_map != nullptr) {
            if (word[k] != 0u) {
              mxc = max(mxc, word[k]);
              any = true;
            }
          } else {
            mxc = max(mxc, cost_from(word[k], fm[k]));
            any = true;
          }
        }
      }
      if (any) best = mxc;
    }
  }
  // A block with a primary ray that has a ZERO direction component (a camera on an axis: one pixel column or row of
  // the frame) is among the frame's longest, whatever the old frame measured where it projects to -- the turned
  // camera's rays there were ordinary ones.  Such rays leave the hit-set walk for the exact, lane-serial descent and
  // pass the reference's box tests through NaN (mt_trace.h): about thirty times a block's mean cost.  Telling the
  // scheduler so lets those blocks start first and in pieces.
  if (A.new_irr && block_has_zero_component_ray(P.sensor, (int)px - 4, (int)py - 4)) best = max(best, A.unseen * 30u);
  P.item_forecast[i] = best;
  gx = ((int)px - P.region_x) >> 3;
  gy = ((int)py - P.region_y) >> 3;
  return best;
}

// What block i becomes in the coming launch: `n` units under sort key `key`, order_sub of unit q = sub0 + q (sub0 = -1:
// the whole block), the word its cost starts from, and (hybrid) its form.  Packed into one word between the count and
// the scatter kernel: key 0..10, n code 11..12 (1 / 4 / 16), sub0 code 13..15, cost word code 16..18, form 19..20, starter 21.
struct OrderUnit {
  unsigned key, n, cost_word;
  int sub0, form;
  bool starter;
  __device__ __forceinline__ unsigned pack() const {
    const unsigned nc = n == 1u ? 0u : (n == 4u ? 1u : 2u);
    const unsigned sc = sub0 < 0 ? 0u : (sub0 == 0 ? 1u : (sub0 == 4 ? 2u : (sub0 == kHybridPoolSub ? 3u : 4u)));
    const unsigned cc = cost_word == 0u ? 0u : (cost_word == 0x80000000u ? 1u : (cost_word == (1u << 30) ? 2u : (cost_word == (2u << 30) ? 3u : 4u)));
    return key | (nc << 11) | (sc << 13) | (cc << 16) | ((unsigned)form << 19) | (starter ? 1u << 21 : 0u);
  }
  __device__ __forceinline__ void unpack(unsigned w) {
    key = w & 0x7ffu;
    const unsigned nc = (w >> 11) & 3u, sc = (w >> 13) & 7u, cc = (w >> 16) & 7u;
    n = nc == 0u ? 1u : (nc == 1u ? 4u : 16u);
    sub0 = sc == 0u ? -1 : (sc == 1u ? 0 : (sc == 2u ? 4 : (sc == 3u ? kHybridPoolSub : kHybridPoolSub + 4)));
    cost_word = cc == 0u ? 0u : (cc == 1u ? 0x80000000u : (cc == 2u ? (1u << 30) : (cc == 3u ? (2u << 30) : 0xc0000000u)));
    form = (int)((w >> 19) & 3u);
    starter = ((w >> 21) & 1u) != 0u;
  }
};
static_assert(kOrdKeysMax <= 2048, "an OrderUnit packs its key into 11 bits");

// Workgroup wg takes the blocks [wg x chunk, (wg + 1) x chunk), chunk = ceil(n / G) -- the SAME blocks in all three kernels,
// every workgroup busy however few blocks the launch has, and its blocks NEIGHBOURS in the picture: the units of one cost
// bucket then come in runs of neighbouring blocks (dealt out block i -> workgroup i mod G, the first frames of the ray
// pool -- few distinct forecasts, so a bucket is most of the picture -- took 8.5 instead of 7.8 ms: consecutive units
// were 64 blocks apart).
#define MT_ORD_THREAD()                                                                                              \
  const int tid = threadIdx.x, wg = blockIdx.x;                                                                      \
  const unsigned G = gridDim.x, chunk = (P.n_items + G - 1u) / G;                                                    \
  const unsigned i_begin = (unsigned)wg * chunk, i_end = i_begin + chunk < P.n_items ? i_begin + chunk : P.n_items;  \
  unsigned *ctl = P.order_ctl;                                                                                       \
  unsigned *area = ctl + kOrdArea + (O.epoch & 1u) * kOrdAreaWords;                                                  \
  (void)wg; (void)area; (void)i_begin; (void)i_end
#define MT_ORD_FOR_ITEMS(i) for (unsigned i = i_begin + (unsigned)tid; i < i_end; i += (unsigned)kOrdThreads)

// ---- 1: forecasts, their sum, the region grid; the next launch's area and this launch's counters zeroed
template <int KIND>
__global__ __launch_bounds__(kOrdThreads) void order_forecast_kernel(RenderParams P, ForecastArgs A, OrderArgs O) {
  __shared__ unsigned long long s_sum;
  MT_ORD_THREAD();
  unsigned *area_next = ctl + kOrdArea + ((O.epoch + 1u) & 1u) * kOrdAreaWords;
  const bool queues = KIND == 0 && P.queues != nullptr && O.queue_mode != 0;
  for (unsigned c = (unsigned)wg * kOrdThreads + (unsigned)tid; c < (unsigned)kOrdAreaWords; c += G * kOrdThreads) area_next[c] = 0u;
  if (wg == 0) {
    if (tid < 16) P.work_counter[tid] = 0u;  // (the frame kernel's counters, the class counts, n_work)
    if (queues) {
      for (int c = tid; c < kQueueWords; c += kOrdThreads) P.queues[c] = 0u;
    }
  }
  if (tid == 0) s_sum = 0ull;
  __syncthreads();
  unsigned long long part = 0ull;
  const int n_cols = (P.region_w + 7) >> 3, n_rows = (P.region_h + 7) >> 3;
  MT_ORD_FOR_ITEMS(i) {
    int gx, gy;
    const unsigned f = forecast_item(P, A, i, gx, gy);
    part += f & 0x7fffffffu;
    if (queues) {  // the block's cell of the region grid (mt_queues.h)
      int cx = gx * kGridW / (n_cols > 0 ? n_cols : 1), cy = gy * kGridH / (n_rows > 0 ? n_rows : 1);
      cx = cx < 0 ? 0 : (cx < kGridW ? cx : kGridW - 1);
      cy = cy < 0 ? 0 : (cy < kGridH ? cy : kGridH - 1);
      const unsigned cell = (unsigned)(cy * kGridW + cx);
      P.item_cell[i] = (unsigned short)cell;
      atomicAdd(area + kOrdAreaGrid + cell, ((f & 0x7fffffffu) >> 6) + 1u);  // (+ 1: a block costs something whatever its forecast says)
    }
  }
  if (part != 0ull) atomicAdd(&s_sum, part);
  __syncthreads();
  if (tid == 0 && s_sum != 0ull) atomicAdd((unsigned long long *)area, s_sum);
}

// ---- 2a: what every block becomes; the workgroup's histogram
template <int KIND>
__global__ __launch_bounds__(kOrdThreads) void order_count_kernel(RenderParams P, OrderArgs O) {
  __shared__ unsigned s_starters;
  __shared__ unsigned s_hist[kOrdKeysMax];
  __shared__ RegionShared s_reg;
  MT_ORD_THREAD();
  const bool queues = KIND == 0 && P.queues != nullptr && O.queue_mode != 0;
  const int n_keys = KIND == 0 ? (queues ? kQueues * 256 : 256) : (KIND == 1 ? 2 * 256 : 256);
  if (tid == 0) s_starters = 0u;
  for (int k = tid; k < n_keys; k += kOrdThreads) s_hist[k] = 0u;
  const unsigned long long total_forecast = *(const unsigned long long *)area;
  const float share = (float)total_forecast / (float)(O.n_waves > 0 ? O.n_waves : 1);
  if (queues) build_region_table(area + kOrdAreaGrid, O.queue_mode, s_reg, tid, kOrdThreads);
  else __syncthreads();
  const float kQuarterTime = 0.45f;
  unsigned my_starters = 0u;
  MT_ORD_FOR_ITEMS(i) {
    OrderUnit u;
    u.starter = false;
    u.form = 0;
    if constexpr (KIND == 0) {
      // blocks above quad_share of an even share of the frame's work are cut into four quarters with FOUR lanes per
      // pixel; a block that was rendered as quarters stays so until its forecast falls well below the threshold
      const unsigned f = P.item_forecast[i], c = f & 0x7fffffffu;
      const float quad_above = share * O.quad_share, quad_keep = quad_above * O.quad_keep;
      const bool quad = (float)c > ((f >> 31) ? quad_keep : quad_above) && c > 0u;
      // ... and as sixteen 2x2 cells when a QUARTER would still be among the launch's longest units AND the block holds
      // rays with a zero direction component (a pixel column or row of a camera on an axis): their passes cost with the
      // number of such rays in them (1.14 M cycles with one, 1.66 M with four), so smaller pieces shorten the chain --
      // the loft's repeated frame was ONE such quarter's 16 passes, 13.1 ms, with every other wave done at 9.2.  (Any
      // other block in cells is only more work: room panning 4.79 -> 6.9 ms with every long block cut so.)
      bool cells = false;
      if (quad && O.new_irr && (float)c * kQuarterTime > O.cell_share * quad_above) {
        const int per_tile = P.blocks_x * P.blocks_y;
        const int j = (int)(i / (unsigned)per_tile), b = (int)(i % (unsigned)per_tile);
        const int tile = tile_of_slot(P, j);
        const int x0 = P.region_x + (tile % P.tiles_x) * P.tile_w + (b % P.blocks_x) * 8;
        const int y0 = P.region_y + (tile / P.tiles_x) * P.tile_h + (b / P.blocks_x) * 8;
        cells = block_has_zero_component_ray(P.sensor, x0, y0);
      }
      const unsigned unit = cells ? (unsigned)((float)c * O.cell_time) : (quad ? (unsigned)((float)c * kQuarterTime) : c);
      const unsigned reg = queues ? (unsigned)s_reg.cellreg[P.item_cell[i]] : 0u;
      u.key = reg * 256u + (unsigned)cost_bucket(unit);
      u.n = cells ? 16u : (quad ? 4u : 1u);
      u.sub0 = cells ? 4 : (quad ? 0 : -1);
      u.cost_word = cells ? 0xc0000000u : (quad ? 0x80000000u : 0u);
    } else if constexpr (KIND == 1) {
      // blocks whose forecast lies above pool_share of an even share go to the ray pool in pieces (4x4 quarters; 2x2
      // cells when a quarter would still be cell_factor times above that); the others to the state machine
      const unsigned c = P.item_forecast[i] & 0x7fffffffu;
      const float quad_above = share * O.quad_share, pool_above = share * O.pool_share;
      unsigned unit;
      if ((float)c > pool_above && c > 0u) {
        u.form = ((float)c * O.piece_time1 > O.cell_factor * pool_above) ? 3 : 2;
        unit = (unsigned)((float)c * (u.form == 3 ? O.piece_time2 : O.piece_time1));
        u.n = u.form == 3 ? 16u : 4u;
        u.sub0 = kHybridPoolSub + (u.form == 3 ? 4 : 0);
      } else if ((float)c > quad_above && c > 0u) {
        u.form = 1; unit = (unsigned)((float)c * kQuarterTime); u.n = 4u; u.sub0 = 0;
      } else {
        u.form = 0; unit = c; u.n = 1u; u.sub0 = -1;
      }
      u.key = (u.form >= 2 ? 0u : 256u) + (unsigned)cost_bucket(unit);  // the pool's units come first
      u.cost_word = u.form == 1 ? 0x80000000u : 0u;
      u.starter = u.form < 2 && (float)unit > share * O.starter_share;
    } else {
      int level;
      unsigned unit;
      sched_decide(P.item_cost[i], P.item_forecast[i] & 0x7fffffffu, share * O.sp.cut_share, O.sp, level, unit);
      u.key = (unsigned)pool_cost_bucket(unit);
      u.n = level == 0 ? 1u : (level == 1 ? 4u : 16u);
      u.sub0 = level == 0 ? -1 : (level == 1 ? 0 : 4);
      u.cost_word = (unsigned)level << 30;
    }
    P.item_unit[i] = u.pack();
    atomicAdd(&s_hist[u.key], u.n);
    if (u.starter) my_starters += u.n;
  }
  if (my_starters) atomicAdd(&s_starters, my_starters);
  __syncthreads();
  unsigned *whist = P.order_whist + (size_t)wg * kOrdKeysMax;
  for (int k = tid; k < n_keys; k += kOrdThreads) whist[k] = s_hist[k];
  if (tid == 0 && s_starters != 0u) atomicAdd(area + 2, s_starters);
}

// ---- 2b: prefix sums over the keys (every workgroup for itself: units longest first within a region / part), the scatter
template <int KIND>
__global__ __launch_bounds__(kOrdThreads) void order_scatter_kernel(RenderParams P, OrderArgs O) {
  __shared__ unsigned s_pos[kOrdKeysMax];  // next position of this workgroup per key
  __shared__ unsigned s_wave[kOrdThreads / 64];
  MT_ORD_THREAD();
  const bool queues = KIND == 0 && P.queues != nullptr && O.queue_mode != 0;
  const int n_keys = KIND == 0 ? (queues ? kQueues * 256 : 256) : (KIND == 1 ? 2 * 256 : 256);
  // thread t: keys 2t, 2t + 1 (n_keys <= 2 x kOrdThreads)
  unsigned tot[2] = {0u, 0u}, before[2] = {0u, 0u};
  if (2 * tid < n_keys) {  // (n_keys is even; the two keys of a thread are one 8-byte load per workgroup)
    const uint2 *col = (const uint2 *)(P.order_whist + 2 * tid);
    for (unsigned w0 = 0; w0 < G; w0 += 8u) {  // (eight loads in flight)
      uint2 v[8];
#pragma unroll
      for (unsigned k = 0; k < 8u; k++) v[k] = w0 + k < G ? col[(size_t)(w0 + k) * (kOrdKeysMax / 2)] : make_uint2(0u, 0u);
#pragma unroll
      for (unsigned k = 0; k < 8u; k++) {
        tot[0] += v[k].x; tot[1] += v[k].y;
        if (w0 + k < (unsigned)wg) { before[0] += v[k].x; before[1] += v[k].y; }
      }
    }
  }
  // exclusive scan of tot[0] + tot[1] over the threads: inside each wave by shuffles, then over the 16 waves
  const unsigned mine = tot[0] + tot[1];
  unsigned incl = mine;
  const int lane = tid & 63, wave = tid >> 6;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned v = (unsigned)__shfl_up((int)incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned wave_base = 0u, all = 0u;
  for (int w = 0; w < kOrdThreads / 64; w++) {
    const unsigned v = s_wave[w];
    if (w < wave) wave_base += v;
    all += v;
  }
  const unsigned excl = wave_base + incl - mine;
  if (2 * tid < n_keys) s_pos[2 * tid] = excl + before[0];
  if (2 * tid + 1 < n_keys) s_pos[2 * tid + 1] = excl + tot[0] + before[1];
  if (wg == 0) {
    // bounds of the parts: queue q = keys [256 q, 256 q + 256) (mt_queues.h); hybrid: the pool's part = keys [0, 256)
    if (queues && (2 * tid) % 256 == 0 && 2 * tid < n_keys) P.queues[kQueueStart + (2 * tid) / 256] = excl;
    if (KIND == 1 && 2 * tid == 256) {
      const unsigned st = area[2];
      P.n_work[1] = excl;
      P.n_work[2] = excl == 0u ? 0u : (st < O.max_starters ? st : O.max_starters);  // (no pool part: nobody needs to skip it)
    }
    if (tid == 0) {
      if (queues) P.queues[kQueueStart + kQueues] = all;
      P.n_work[0] = all;
    }
  }
  __syncthreads();
  MT_ORD_FOR_ITEMS(i) {
    OrderUnit u;
    u.unpack(P.item_unit[i]);
    const unsigned at = atomicAdd(&s_pos[u.key], u.n);
    for (unsigned q = 0; q < u.n; q++) {
      P.order_item[at + q] = i;
      P.order_sub[at + q] = (signed char)(u.sub0 < 0 ? -1 : u.sub0 + (int)q);
    }
    P.item_cost[i] = u.cost_word;  // reset for the coming frame (bit 31 / bits 30-31: the form it will be measured in)
    if (KIND == 1) O.form_out[i] = (unsigned char)u.form;
  }
}
#undef MT_ORD_THREAD
#undef MT_ORD_FOR_ITEMS

}  // namespace mt
