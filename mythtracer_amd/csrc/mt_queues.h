// One work order per XCD (MT_TUNE_XCD_QUEUES, RenderParams::queues).
//
// MI355X has eight XCDs with an L2 of their own each; a wave that takes the next unit of ONE chip-wide work order
// traces rays anywhere in the picture, and every L2 ends up holding the whole tree (hit rate 0.82 on the room scene,
// 0.66 on the loft).  Here the picture is cut into eight regions of equal forecast cost -- mode 1: eight stripes of
// block columns; mode 2 (default): four stripes, each cut into an upper and a lower half -- and every region gets a work
// order of its own (longest unit first, as ever).  A wave drains the queue of its XCD (HW_REG_XCC_ID) first and then
// helps with the fullest other queue: an XCD's L2 holds its region's part of the tree and of the triangle streams
// (0.92 / 0.82).  The order changes nothing about what is computed for a pixel.
//
// The cuts run along the cells of a kGridW x kGridH grid over the launch's region: forecast_kernel adds every block's
// forecast to its cell (queues[kQueueGrid ...]) and notes the cell (item_cell); the schedule kernels turn the grid's
// marginals into cuts and a table cell -> region (build_region_table), and sort by (region, cost bucket).
#pragma once
#include "mt_device.h"

namespace mt {

// forecast_kernel: block i (position bx, by in the launch's region, in blocks) with forecast f
__device__ __forceinline__ void note_cell(const RenderParams &P, unsigned i, unsigned f, int bx, int by) {
  const int n_cols = (P.region_w + 7) >> 3, n_rows = (P.region_h + 7) >> 3;
  int cx = bx * kGridW / (n_cols > 0 ? n_cols : 1), cy = by * kGridH / (n_rows > 0 ? n_rows : 1);
  cx = cx < 0 ? 0 : (cx < kGridW ? cx : kGridW - 1);
  cy = cy < 0 ? 0 : (cy < kGridH ? cy : kGridH - 1);
  const unsigned cell = (unsigned)(cy * kGridW + cx);
  P.item_cell[i] = (unsigned short)cell;
  atomicAdd(P.queues + kQueueGrid + cell, ((f & 0x7fffffffu) >> 6) + 1u);  // (+ 1: a block costs something whatever its forecast says)
}

struct RegionShared {
  unsigned grid[kGridW * kGridH];
  unsigned char cellreg[kGridW * kGridH];  // cell -> region
  unsigned long long colsum[kGridW];
  unsigned long long rowsum[4 * kGridH];
  int cut[kQueues + 1];   // stripe s = cell columns [cut[s], cut[s + 1])
  int rowcut[kQueues];    // mode 2: stripe s is cut in front of this cell row
};

// All threads of the (single) workgroup call this; R.cellreg is valid after it returns.
__device__ __forceinline__ void build_region_table(const RenderParams &P, int mode, RegionShared &R, int tid, int n_threads) {
  const int n_stripes = mode == 2 ? 4 : kQueues;
  for (int c = tid; c < kGridW * kGridH; c += n_threads) R.grid[c] = P.queues[kQueueGrid + c];
  __syncthreads();
  if (tid < kGridW) {
    unsigned long long t = 0ull;
    for (int y = 0; y < kGridH; y++) t += R.grid[y * kGridW + tid];
    R.colsum[tid] = t;
  }
  __syncthreads();
  // stripes of equal cost: cut s lies behind the first cell column at which the running cost reaches s / n_stripes of the total
  if (tid <= n_stripes) {
    unsigned long long total = 0ull;
    for (int x = 0; x < kGridW; x++) total += R.colsum[x];
    int cut = tid == 0 ? 0 : kGridW;
    if (tid > 0 && tid < n_stripes && total != 0ull) {
      unsigned long long acc = 0ull;
      for (int x = 0; x < kGridW; x++) {
        acc += R.colsum[x];
        if (acc * (unsigned long long)n_stripes >= total * (unsigned long long)tid) {
          cut = x + 1;
          break;
        }
      }
    }
    R.cut[tid] = cut;
  }
  __syncthreads();
  if (mode == 2) {  // per stripe the cell row that halves its cost
    if (tid < 4 * kGridH) {
      const int st = tid / kGridH, y = tid % kGridH;
      unsigned long long t = 0ull;
      for (int x = R.cut[st]; x < R.cut[st + 1] && x < kGridW; x++) t += R.grid[y * kGridW + x];
      R.rowsum[tid] = t;
    }
    __syncthreads();
    if (tid < 4) {
      unsigned long long total = 0ull, acc = 0ull;
      for (int y = 0; y < kGridH; y++) total += R.rowsum[tid * kGridH + y];
      int cut = kGridH;
      for (int y = 0; y < kGridH; y++) {
        acc += R.rowsum[tid * kGridH + y];
        if (acc * 2ull >= total) {
          cut = y + 1;
          break;
        }
      }
      R.rowcut[tid] = cut;
    }
    __syncthreads();
  }
  for (int c = tid; c < kGridW * kGridH; c += n_threads) {
    const int cx = c % kGridW, cy = c / kGridW;
    int st = 0;
    while (st + 1 < n_stripes && cx >= R.cut[st + 1]) st++;
    R.cellreg[c] = (unsigned char)(mode == 2 ? st * 2 + (cy >= R.rowcut[st] ? 1 : 0) : st);
  }
  __syncthreads();
}

// A wave's view of the queues.  set = 0: the launch's only set of queues; hybrid launches have two (the pool's units,
// the state machine's), kQueues counters and kQueues + 1 bounds each.
struct QueueFetch {
  unsigned my_queue, dry, rank, len;
  __device__ __forceinline__ void init() {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    my_queue = xcc & (unsigned)(kQueues - 1);
    dry = 0u;
    rank = 0u;
    len = 1u;
  }
  // the next unit of the order (its index in order_item / order_sub), or `none` when every queue is dry
  __device__ __forceinline__ unsigned next(const RenderParams &P, int lane, unsigned none) {
    while (dry != (1u << kQueues) - 1u) {
      if ((dry >> my_queue) & 1u) {
        // the own queue is dry: on to the one with the most units left (a glance, not a reservation)
        unsigned left = 0u;
        if (lane < kQueues) {
          const unsigned s0 = P.queues[kQueueStart + lane], s1 = P.queues[kQueueStart + lane + 1];
          const unsigned taken = __hip_atomic_load(P.queues + lane * kQueueStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          left = (taken < s1 - s0 && !((dry >> lane) & 1u)) ? s1 - s0 - taken : 0u;
        }
        unsigned best_left = 0u, best_q = 0u;
        for (int q = 0; q < kQueues; q++) {
          const unsigned l = (unsigned)__builtin_amdgcn_readlane((int)left, q);
          if (l > best_left) {
            best_left = l;
            best_q = (unsigned)q;
          }
        }
        if (best_left == 0u) break;
        my_queue = best_q;
      }
      const unsigned s0 = P.queues[kQueueStart + my_queue], s1 = P.queues[kQueueStart + my_queue + 1];
      const unsigned v = atomicAdd(P.queues + my_queue * kQueueStride, lane == 0 ? 1u : 0u);
      const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
      if (k < s1 - s0) {
        rank = k;
        len = s1 - s0;
        return s0 + k;
      }
      dry |= 1u << my_queue;
    }
    return none;
  }
};

}  // namespace mt
