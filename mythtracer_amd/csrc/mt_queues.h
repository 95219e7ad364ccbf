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
// The cuts run along the cells of a kGridW x kGridH (64 x 32) grid over the launch's region: order_kernel (mt_order.h) adds every
// block's forecast to its cell and notes the cell (item_cell), turns the grid's marginals into cuts and a table
// cell -> region (build_region_table), and sorts by (region, cost bucket).
#pragma once
#include "mt_device.h"

namespace mt {

static_assert(kGridW == 64 && kGridH == 32, "build_region_table deals its steps out to 1 024 threads by these sizes");
struct RegionShared {
  unsigned grid[kGridW * kGridH];
  unsigned char cellreg[kGridW * kGridH];  // cell -> region
  unsigned long long colsum[kGridW];       // cost per cell column, then its running sum
  unsigned long long rowsum[4 * kGridH];
  int cut[kQueues + 1];   // stripe s = cell columns [cut[s], cut[s + 1])
  int rowcut[kQueues];    // mode 2: stripe s is cut in front of this cell row
};

// All threads of the workgroup call this (n_threads = 1 024); R.cellreg is valid after it returns.
// `grid`: filled by order_forecast_kernel's atomics (mt_order.h).
// Every step is spread over the workgroup's threads (the serial form on a grid of 128 x 64 cells -- a thread per stripe
// walking the columns -- took 19 us of dependent LDS reads in front of every frame).
__device__ __forceinline__ void build_region_table(const unsigned *grid, int mode, RegionShared &R, int tid, int n_threads) {
  const int n_stripes = mode == 2 ? 4 : kQueues;
  {  // (both loads of a thread in flight together)
    const int c0 = tid, c1 = tid + n_threads;
    const unsigned v0 = c0 < kGridW * kGridH ? grid[c0] : 0u;
    const unsigned v1 = c1 < kGridW * kGridH ? grid[c1] : 0u;
    if (c0 < kGridW * kGridH) R.grid[c0] = v0;
    if (c1 < kGridW * kGridH) R.grid[c1] = v1;
  }
  if (tid < kGridW) R.colsum[tid] = 0ull;
  if (tid < 4 * kGridH) R.rowsum[tid] = 0ull;
  if (tid <= kQueues) R.cut[tid] = tid == 0 ? 0 : kGridW;
  if (tid < kQueues) R.rowcut[tid] = kGridH;
  __syncthreads();
  {  // cost per cell column: sixteen threads per column, two rows each
    const int col = tid & (kGridW - 1), part = tid / kGridW;
    if (part < kGridH / 2) {
      const unsigned long long t = (unsigned long long)R.grid[(2 * part) * kGridW + col] + R.grid[(2 * part + 1) * kGridW + col];
      if (t != 0ull) atomicAdd(&R.colsum[col], t);
    }
  }
  __syncthreads();
  // running sum over the columns (one wave), then the cuts: cut s lies behind the first cell column at which the running
  // cost reaches s / n_stripes of the total
  if (tid < kGridW) {
    unsigned long long run = R.colsum[tid];
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long v = __shfl_up(run, d, 64);
      if (tid >= d) run += v;
    }
    const unsigned long long total = __shfl(run, 63, 64), before = __shfl_up(run, 1, 64);
    if (total != 0ull) {
      for (int st = 1; st < n_stripes; st++) {
        const unsigned long long want = total * (unsigned long long)st;
        if (run * (unsigned long long)n_stripes >= want && !(tid > 0 && before * (unsigned long long)n_stripes >= want)) R.cut[st] = tid + 1;
      }
    }
  }
  __syncthreads();
  if (mode == 2) {  // per stripe the cell row that halves its cost: eight threads per (stripe, row), every eighth column each
    {
      const int y = tid & (kGridH - 1), st = (tid / kGridH) & 3, part = tid / (4 * kGridH);
      if (part < 8) {
        unsigned long long t = 0ull;
        for (int x = R.cut[st] + part; x < R.cut[st + 1] && x < kGridW; x += 8) t += R.grid[y * kGridW + x];
        if (t != 0ull) atomicAdd(&R.rowsum[st * kGridH + y], t);
      }
    }
    __syncthreads();
    if (tid < 4 * kGridH) {  // half a wave per stripe
      const int st = tid / kGridH, y = tid & (kGridH - 1);
      unsigned long long acc = R.rowsum[tid];
      for (int d = 1; d < kGridH; d <<= 1) {
        const unsigned long long v = __shfl_up(acc, d, kGridH);
        if (y >= d) acc += v;
      }
      const unsigned long long total = __shfl(acc, kGridH - 1, kGridH);
      const unsigned long long reached64 = __ballot(acc * 2ull >= total);
      const unsigned reached = (unsigned)(reached64 >> (32 * (st & 1)));
      if (y == 0) R.rowcut[st] = reached != 0u ? __builtin_ctz(reached) + 1 : kGridH;
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
