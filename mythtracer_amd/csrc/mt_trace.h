// mt_trace.h — wave-synchronous octree traversal for gfx950 (wave64).
//
// Replaces OctTree::IntersectRay (octtree.cc:26-40),
// Node::NodeIntersectRay (:138-167), Node::PrimitiveIntersectRay (:169-257)
// and Triangle::IntersectRay (primitive_triangle.cc:81-143) of the reference,
// with bit-identical results.
//
// Two traversals live here (one ray per lane in both).  Regular rays in the
// automatic mode take the HIT-SET WALK (the MT_HS block of trace_wave: the
// wave walks the tree depth-first, every node once for all its lanes, children
// in any order; the reference's choice among the children that hold a hit is
// reproduced by its own rule, "the entered, acceptable child with the smallest
// (entry distance, index)" -- see the comment there and DESIGN.md section 3.1).
// Irregular rays, the diagnostic modes and trees deeper than kHsMaxDepth (16 levels) take
// the ORDERED DESCENT:
//   * every lane keeps its own recursion state (the reference's call stack of
//     PrimitiveIntersectRay) in an LDS-resident per-lane stack;
//   * small nodes (< kBigNode triangles) are scanned lane-parallel, every lane
//     its own node -- unless only a handful of lanes is left (what the hit-set
//     walk leaves behind: rays with a zero direction component), whose lists are
//     then scanned together, packed over the wave's lanes, one (ray, triangle)
//     pair per lane (scan_small_packed_call); for the big ones the wave repeatedly picks the lowest-
//     numbered node any lane still has to scan (nodes are numbered breadth-first,
//     so the big top-of-tree lists are scanned once for all lanes that need
//     them) and scans it for those lanes only.  Its boxes are then WAVE-UNIFORM:
//     fetched with scalar loads (one box serves 64 rays) and fed to the VALU as
//     SGPR operands;
//   * only work whose outcome is provably "the reference's pre-filter rejects
//     it" is skipped: an fp32 conservative filter before the fp64 box test,
//     fp32 union boxes over blocks of 16 triangles and over subtrees (see
//     Filter32, scan_grouped_call, subtree_may_hit, degenerate_axis);
//   * the rare triangles that pass the box pre-filter are parked per lane and
//     resolved (Möller–Trumbore) in batches, in the reference's order.
//
// Exactness.  fp64 wherever a result is decided, no FMA contraction (-ffp-contract=off), IEEE
// division and square root.  Three evaluation modes of the slab test produce
// the same booleans as the reference's std::min/std::max formulation:
//   mode 0 (exact)   literal std::min/std::max compare+select; used whenever
//                    a ray could produce NaN (a zero direction component);
//   mode 1 (regular) v_min_f64/v_max_f64; identical when no operand is NaN
//                    (only the sign of a zero may differ, which no comparison
//                    here can see);
//   mode 2 (octant)  when all lanes scanning a node share the direction sign
//                    octant, near/far planes are picked per axis by the sign
//                    (max(t1,t2) is t2 when 1/d > 0 because rounding is
//                    monotonic), which removes six min/max per box.
// No distance-based pruning is done: the reference tests every triangle of a
// visited node and its tie-breaking (later equal-distance hit wins,
// octtree.cc:186-195) is reproduced literally.
#pragma once
#include <type_traits>
#include "mt_device.h"

namespace mt {

#define MT_CONST __attribute__((address_space(4)))
// Per-lane work counters of one traversal (STATS instantiations only): in registers (the LDS behind the frames
// stages node records; the timed kernels are the ones without counters).
#define MT_CNT_ADD(i, v) (cntr[i] += (v))
#define MT_CNT_SET(i, v) (cntr[i] = (v))
#define MT_CNT_GET(i) (cntr[i])

#ifndef MT_PACK_SMALL
#define MT_PACK_SMALL 8  // at most this many lanes with exact scans: their small nodes packed over the wave (0: off; 16: a pass of 16 such rays 3.2 -> 4.2 M cycles)
#endif

// Phase profiling (diagnostic build only: python -m mythtracer_amd.build --prof).
#ifdef MT_PROF
#define MT_PROF_DECL unsigned long long prof_acc[PROF_COUNT] = {0}; unsigned long long prof_t0 = 0, prof_t1 = 0
#define MT_PROF_BEGIN(var) var = __builtin_amdgcn_s_memtime()
#define MT_PROF_END(slot, var) prof_acc[slot] += __builtin_amdgcn_s_memtime() - var
#define MT_PROF_COUNT(slot, n) prof_acc[slot] += (unsigned long long)(n)
#define MT_TL(tag) do { if (tl_on && tl_i < 512u && tl_base + tl_i < (unsigned long long)kProfTimeline) { \
    S.prof[PROF_COUNT + 1 + tl_base + tl_i] = (__builtin_amdgcn_s_memtime() << 8) | (unsigned)(tag); tl_i++; } } while (0)
#define MT_PROF_FLUSH(ptr, lane) do { if ((ptr) && (lane) == 0) { for (int pi_ = 0; pi_ < PROF_COUNT; pi_++) if (prof_acc[pi_]) atomicAdd((ptr) + pi_, prof_acc[pi_]); } } while (0)
#else
#define MT_PROF_DECL
#define MT_PROF_BEGIN(var)
#define MT_PROF_END(slot, var)
#define MT_PROF_COUNT(slot, n)
#define MT_PROF_FLUSH(ptr, lane)
#define MT_TL(tag)
#endif

template <typename T>
__device__ __forceinline__ const MT_CONST T *as_const(const T *p) {
  return (const MT_CONST T *)(uintptr_t)p;
}

// std::min / std::max exactly as libstdc++ (NaN and operand order).
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double std_min3(double a, double b, double c) {
  double r = a;
  if (b < r) r = b;
  if (c < r) r = c;
  return r;
}
__device__ __forceinline__ double std_max3(double a, double b, double c) {
  double r = a;
  if (r < b) r = b;
  if (r < c) r = c;
  return r;
}

template <bool EXACT>
__device__ __forceinline__ double mn(double a, double b) {
  if constexpr (EXACT) return std_min(a, b);
  else return __builtin_fmin(a, b);
}
template <bool EXACT>
__device__ __forceinline__ double mx(double a, double b) {
  if constexpr (EXACT) return std_max(a, b);
  else return __builtin_fmax(a, b);
}
template <bool EXACT>
__device__ __forceinline__ double mn3(double a, double b, double c) {
  if constexpr (EXACT) return std_min3(a, b, c);
  else return __builtin_fmin(__builtin_fmin(a, b), c);
}
template <bool EXACT>
__device__ __forceinline__ double mx3(double a, double b, double c) {
  if constexpr (EXACT) return std_max3(a, b, c);
  else return __builtin_fmax(__builtin_fmax(a, b), c);
}

// Arguments of a non-inlined device function arrive in VGPRs; these put a
// wave-uniform value back into SGPRs so that it can feed scalar loads.
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane_id) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, lane_id);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), lane_id);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float readlane_f32(float v, int lane_id) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane_id));
}

// Wave-wide minimum of a 32-bit value; every lane must be active.
__device__ __forceinline__ int wave_min_i32(int v) {
  // all-reduce inside each row of 16 lanes with row rotations ...
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));  // row_ror:8
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));  // row_ror:4
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));  // row_ror:2
  v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));  // row_ror:1
  // ... then combine the four rows on the scalar unit.
  int a = __builtin_amdgcn_readlane(v, 0);
  int b = __builtin_amdgcn_readlane(v, 16);
  int c = __builtin_amdgcn_readlane(v, 32);
  int d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);
  unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
  unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
  unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return a + b + c + d;
}

// Per-lane work counters of one wave (see mt_stats).
struct LaneStats {
  unsigned v[ST_WAVE_NODE_STEPS];  // the per-lane ones
  unsigned wave_node_steps, wave_tri_steps, bytes_scalar;  // wave-uniform
  __device__ void clear() {
    for (int i = 0; i < ST_WAVE_NODE_STEPS; i++) v[i] = 0;
    wave_node_steps = wave_tri_steps = bytes_scalar = 0;
  }
};

// Result of scanning one node's triangle list for a lane.
struct ScanOut {
  int best;           // stream index of the closest hit in the list, -1 none
  double best_t;
  unsigned mt_tests;  // Möller–Trumbore evaluations performed for this lane
  // bytes this scan requested: per lane (vector loads) and once for the wave
  // (scalar loads; wave-uniform).  Counted in the STATS instantiations only.
  unsigned bytes_v = 0, bytes_s = 0;
#ifdef MT_PROF
  unsigned n_groups = 0, n_live = 0, n_ranges = 0, n_range_tris = 0;
  unsigned t_a = 0, t_b = 0, t_c = 0;  // section times (s_memtime ticks)
#endif
};

// The per-wave traversal stack in LDS, structure-of-arrays over [depth][lane]
// so that every access is bank-conflict free.  Held as a byte offset into the
// block's LDS allocation (not as generic pointers, which would turn every
// access into a flat_* instruction inside the non-inlined traversal).
#define MT_LDS __attribute__((address_space(3)))
struct WaveStack {
  unsigned base;   // byte offset of this wave's region in LDS
  int depth;       // frames per lane
  int pack_shift;  // see DevScene::pack_shift
  // bt: best distance so far in that node; fc: first child of that node;
  // bp: best primitive so far (-1 none); ord: bits 0-23 child order (3 bits
  // each), 24-27 count, 28-31 position.  Packed frames hold fc and bp + 1 in
  // one word (fb).
  __device__ __forceinline__ unsigned frame_bytes() const { return pack_shift ? 16u : 20u; }
  __device__ __forceinline__ MT_LDS double *bt() const { return (MT_LDS double *)(uintptr_t)base; }
  __device__ __forceinline__ MT_LDS int *fc() const { return (MT_LDS int *)(uintptr_t)(base + (unsigned)depth * 64u * 8u); }
  __device__ __forceinline__ MT_LDS int *bp() const { return (MT_LDS int *)(uintptr_t)(base + (unsigned)depth * 64u * 12u); }
  __device__ __forceinline__ MT_LDS unsigned *ord() const {
    return (MT_LDS unsigned *)(uintptr_t)(base + (unsigned)depth * 64u * (pack_shift ? 12u : 16u));
  }
  __device__ __forceinline__ void bind(char *smem_base, int wave_in_block, int tree_depth, int shift, bool deep = false) {
    (void)smem_base;  // dynamic LDS starts at offset 0 of the block's allocation (no static LDS is declared)
    depth = tree_depth;
    pack_shift = shift;
    base = (unsigned)wave_in_block * (unsigned)wave_stack_bytes(tree_depth, shift != 0, deep);
  }
};

// Möller–Trumbore, primitive_triangle.cc:110-142, for one lane's triangle.
__device__ __forceinline__ bool moller_trumbore_v(const double *v, double ox, double oy,
                                                  double oz, double dx, double dy, double dz,
                                                  double *t_out);
__device__ __forceinline__ bool moller_trumbore(const double *vtx, double ox, double oy,
                                                double oz, double dx, double dy, double dz,
                                                double *t_out) {
  const MT_CONST double *p = as_const(vtx);
  const double v[9] = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]};
  return moller_trumbore_v(v, ox, oy, oz, dx, dy, dz, t_out);
}
// The same on vertices that are already in registers.
// The same test without its early exits (primitive_triangle.cc:110-142 evaluated to the end, the four verdicts ANDed):
// where a wave resolves candidates of many lanes at once some lane nearly always goes the whole way, and the
// skipped arithmetic is worth less than the branches.  A rejected determinant may make the later values infinite or NaN;
// they are never looked at.
__device__ __forceinline__ bool moller_trumbore_flat(const double *v, double ox, double oy,
                                                     double oz, double dx, double dy, double dz,
                                                     double *t_out) {
  const double v0x = v[0], v0y = v[1], v0z = v[2];
  const double e1x = v[3] - v0x, e1y = v[4] - v0y, e1z = v[5] - v0z;
  const double e2x = v[6] - v0x, e2y = v[7] - v0y, e2z = v[8] - v0z;
  const double px = dy * e2z - dz * e2y;
  const double py = dz * e2x - dx * e2z;
  const double pz = dx * e2y - dy * e2x;
  const double det = px * e1x + py * e1y + pz * e1z;
  const bool det_ok = !(det >= -0.00000001 && det < 0.00000001);
  const double inv_det = 1.0 / det;
  const double tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
  const double u = (px * tx + py * ty + pz * tz) * inv_det;
  const bool u_ok = !(u < 0.0 || u > 1.0);
  const double qx = ty * e1z - tz * e1y;
  const double qy = tz * e1x - tx * e1z;
  const double qz = tx * e1y - ty * e1x;
  const double vv = (qx * dx + qy * dy + qz * dz) * inv_det;
  const bool v_ok = !(vv < 0.0 || u + vv > 1.0);
  const double dist = (qx * e2x + qy * e2y + qz * e2z) * inv_det;
  const bool t_ok = !(dist < 0.0);
  *t_out = dist;
  return det_ok & u_ok & v_ok & t_ok;
}
__device__ __forceinline__ bool moller_trumbore_v(const double *v, double ox, double oy,
                                                  double oz, double dx, double dy, double dz,
                                                  double *t_out) {
  const double v0x = v[0], v0y = v[1], v0z = v[2];
  const double e1x = v[3] - v0x, e1y = v[4] - v0y, e1z = v[5] - v0z;
  const double e2x = v[6] - v0x, e2y = v[7] - v0y, e2z = v[8] - v0z;
  // pvec = direction x e2
  const double px = dy * e2z - dz * e2y;
  const double py = dz * e2x - dx * e2z;
  const double pz = dx * e2y - dy * e2x;
  const double det = px * e1x + py * e1y + pz * e1z;  // e1.Dot(pvec)
  if (det >= -0.00000001 && det < 0.00000001) return false;
  const double inv_det = 1.0 / det;
  const double tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
  const double u = (px * tx + py * ty + pz * tz) * inv_det;  // tvec.Dot(pvec)
  if (u < 0.0 || u > 1.0) return false;
  // qvec = tvec x e1
  const double qx = ty * e1z - tz * e1y;
  const double qy = tz * e1x - tx * e1z;
  const double qz = tx * e1y - ty * e1x;
  const double vv = (qx * dx + qy * dy + qz * dz) * inv_det;  // direction.Dot(qvec)
  if (vv < 0.0 || u + vv > 1.0) return false;
  const double dist = (qx * e2x + qy * e2y + qz * e2z) * inv_det;  // e2.Dot(qvec)
  if (dist < 0.0) return false;
  *t_out = dist;
  return true;
}

struct RayRegs {
  double ox, oy, oz;
  double dx, dy, dz;
  double ix, iy, iz;  // Ray::inv_direction, octtree.cc:30-33
};
// A ray as nine scalar parameters: a 72-byte struct would be passed to a
// non-inlined function through scratch memory (by reference), scalars travel
// in VGPRs.
#define MT_RAY_PARAMS double ox_, double oy_, double oz_, double dx_, double dy_, double dz_, \
                      double ix_, double iy_, double iz_
#define MT_RAY_ARGS(r) (r).ox, (r).oy, (r).oz, (r).dx, (r).dy, (r).dz, (r).ix, (r).iy, (r).iz
#define MT_F32_PARAMS float fix_, float fiy_, float fiz_, float fcnx_, float fcny_, float fcnz_, \
                      float fcfx_, float fcfy_, float fcfz_
#define MT_F32_ARGS(f) (f).ix, (f).iy, (f).iz, (f).cnx, (f).cny, (f).cnz, (f).cfx, (f).cfy, (f).cfz
#define MT_F32_FROM_PARAMS(f) \
  Filter32 f;                 \
  f.ix = fix_; f.iy = fiy_; f.iz = fiz_; f.cnx = fcnx_; f.cny = fcny_; f.cnz = fcnz_; \
  f.cfx = fcfx_; f.cfy = fcfy_; f.cfz = fcfz_
#define MT_RAY_FROM_PARAMS(r) \
  RayRegs r;                  \
  r.ox = ox_; r.oy = oy_; r.oz = oz_; r.dx = dx_; r.dy = dy_; r.dz = dz_; r.ix = ix_; r.iy = iy_; r.iz = iz_

// Resolves the parked candidate of every lane that has one.
template <bool STATS>
__device__ __forceinline__ void flush_candidates(const DevScene &S, const RayRegs &r,
                                                 int &pend, int &best, double &best_t,
                                                 LaneStats &st) {
  if (pend >= 0) {
    double t;
    if (STATS) {
      st.v[ST_MT_TESTS]++;
      st.v[ST_BYTES_VECTOR] += 72u;
    }
    if (moller_trumbore(S.tri_vertex + (size_t)pend * 9, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, &t)) {
      // octtree.cc:186-195: keep the old one only if it exists and is
      // strictly closer.
      if (!(best >= 0 && t > best_t)) {
        best = pend;
        best_t = t;
      }
    }
    pend = -1;
  }
}

// Slab pre-filter of Triangle::IntersectRay (primitive_triangle.cc:83-108) for
// one wave-uniform box a[0..5] (min xyz, max xyz) against each lane's ray.
// MODE 2: OCT = compile-time sign octant (bit0 x, bit1 y, bit2 z negative).
// Returns the WAVE MASK of lanes whose ray passes (v_cmp straight into an
// SGPR pair; inactive lanes read 0), so that the common "nobody passed" case
// costs no vector instruction beyond the two compares.
template <int MODE, int OCT>
__device__ __forceinline__ unsigned long long slab_pass(const double *a, const RayRegs &r) {
  constexpr int kOGE = 3, kOLE = 5, kUGE = 11, kULE = 13;  // llvm FCmp predicates
  constexpr bool EX = (MODE == 0);
  if constexpr (MODE == 2) {
    // near/far plane per axis picked by the (wave-uniform) direction sign
    constexpr int NX = (OCT & 1) ? 3 : 0, FX = (OCT & 1) ? 0 : 3;
    constexpr int NY = (OCT & 2) ? 4 : 1, FY = (OCT & 2) ? 1 : 4;
    constexpr int NZ = (OCT & 4) ? 5 : 2, FZ = (OCT & 4) ? 2 : 5;
    const double tnx = (a[NX] - r.ox) * r.ix, tfx = (a[FX] - r.ox) * r.ix;
    const double tny = (a[NY] - r.oy) * r.iy, tfy = (a[FY] - r.oy) * r.iy;
    const double tnz = (a[NZ] - r.oz) * r.iz, tfz = (a[FZ] - r.oz) * r.iz;
    const double tmax = __builtin_fmin(__builtin_fmin(tfx, tfy), tfz);
    const double tmin = __builtin_fmax(__builtin_fmax(tnx, tny), tnz);
    // no NaN can occur in this mode, so this equals !(tmax < 0) && !(tmin > tmax)
    return __builtin_amdgcn_fcmp(tmax, 0.0, kOGE) & __builtin_amdgcn_fcmp(tmin, tmax, kOLE);
  } else {
    const double t1 = (a[0] - r.ox) * r.ix, t2 = (a[3] - r.ox) * r.ix;
    const double t3 = (a[1] - r.oy) * r.iy, t4 = (a[4] - r.oy) * r.iy;
    const double t5 = (a[2] - r.oz) * r.iz, t6 = (a[5] - r.oz) * r.iz;
    const double tmax = mn3<EX>(mx<EX>(t1, t2), mx<EX>(t3, t4), mx<EX>(t5, t6));
    const double tmin = mx3<EX>(mn<EX>(t1, t2), mn<EX>(t3, t4), mn<EX>(t5, t6));
    if constexpr (EX) {  // !(tmax < 0) && !(tmin > tmax), NaN-aware: "unordered or ..."
      return __builtin_amdgcn_fcmp(tmax, 0.0, kUGE) & __builtin_amdgcn_fcmp(tmin, tmax, kULE);
    } else {
      return __builtin_amdgcn_fcmp(tmax, 0.0, kOGE) & __builtin_amdgcn_fcmp(tmin, tmax, kOLE);
    }
  }
}

// ---- box fetch ------------------------------------------------------------
// Two boxes (96 bytes) travel as one s_load_dwordx16 + one s_load_dwordx8 into
// 24 SGPRs.  The loads are issued by hand (inline asm) because the fetch of the
// NEXT pair has to be in flight while the current pair is evaluated, and hipcc
// sinks its own scalar loads down to their first use (no overlap at all).
// Rules that make this safe (see cdna_hip_programming.md 5.7):
//   * scalar loads return out of order, so the only usable wait is
//     lgkmcnt(0); exactly ONE pair is in flight at any time and it is awaited
//     (await_pair) before the next one is issued;
//   * nothing reads a PairRegs between issue_pair and await_pair: the data is
//     only reachable through await_pair's "+s" outputs;
//   * no fetch is left in flight when the loop exits (the last iteration does
//     not issue one), so the registers are free to be reused afterwards.
// tools/check_asm_prefetch.py verifies on the generated ISA that the
// destination registers are untouched between each load and its wait.
typedef double d8v __attribute__((ext_vector_type(8)));
typedef double d4v __attribute__((ext_vector_type(4)));
struct PairRegs {
  d8v lo;  // box0: min xyz, max xyz ; box1: min x, min y
  d4v hi;  // box1: min z, max xyz
  __device__ __forceinline__ double at(int i) const { return i < 8 ? lo[i] : hi[i - 8]; }
};

__device__ __forceinline__ void issue_pair(PairRegs &q, const MT_CONST double *p) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40"
               : "=&s"(q.lo), "=&s"(q.hi)
               : "s"(p));
}
__device__ __forceinline__ void await_pair(PairRegs &q) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.lo), "+s"(q.hi));
}

// Evaluates the two boxes of `q` (stream positions k, k+1 of the node) for the
// calling lanes and parks / resolves the candidates in stream order.
template <int MODE, int OCT, bool STATS>
__device__ __forceinline__ void scan_pair(const DevScene &S, const RayRegs &r, const PairRegs &q,
                                          int pb, int k, int pc, int &pend,
                                          unsigned long long &pmask, int &best, double &best_t,
                                          LaneStats &st) {
  const double b0[6] = {q.lo[0], q.lo[1], q.lo[2], q.lo[3], q.lo[4], q.lo[5]};
  const double b1[6] = {q.lo[6], q.lo[7], q.hi[0], q.hi[1], q.hi[2], q.hi[3]};
  const unsigned long long pm0 = slab_pass<MODE, OCT>(b0, r);
  const unsigned long long pm1 = (k + 1 < pc) ? slab_pass<MODE, OCT>(b1, r) : 0ull;
  if (pm0 | pm1) {
    const unsigned long long me = 1ull << (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
    const bool pass0 = (pm0 & me) != 0, pass1 = (pm1 & me) != 0;
    if (pm0 & pmask) {  // some lane would need a second slot: resolve first
      flush_candidates<STATS>(S, r, pend, best, best_t, st);
      pmask = 0;
    }
    if (pass0) pend = pb + k;
    pmask |= pm0;
    if (pm1 & pmask) {
      flush_candidates<STATS>(S, r, pend, best, best_t, st);
      pmask = 0;
    }
    if (pass1) pend = pb + k + 1;
    pmask |= pm1;
  }
}

// Scans the triangle list [pb, pb+pc) of one node for the calling lanes.
// MODE: 0 exact, 1 regular, 2 octant-uniform (OCT = the shared sign octant).
template <int MODE, int OCT, bool STATS>
__device__ __forceinline__ void scan_node_prims(const DevScene &S, const RayRegs &r, int pb,
                                                int pc, int &best, double &best_t,
                                                LaneStats &st) {
  if (pc <= 0) return;
  const MT_CONST double *p = as_const(S.tri_aabb) + (size_t)pb * 6;
  int pend = -1;
  unsigned long long pmask = 0;  // lanes holding a parked candidate
  // Two buffers; the look-ahead load is issued unconditionally (the stream is
  // padded), so every issue meets its wait on every path.
  PairRegs A, B;
  issue_pair(A, p);
  await_pair(A);
  if (STATS) st.bytes_scalar += 96u * (unsigned)((pc + 1) / 2 + 1);  // two fp64 boxes per fetch, one look-ahead
  for (int k = 0;;) {
    issue_pair(B, p + 12);
    scan_pair<MODE, OCT, STATS>(S, r, A, pb, k, pc, pend, pmask, best, best_t, st);
    await_pair(B);
    k += 2;
    if (k >= pc) break;
    p += 24;
    issue_pair(A, p);
    scan_pair<MODE, OCT, STATS>(S, r, B, pb, k, pc, pend, pmask, best, best_t, st);
    await_pair(A);
    k += 2;
    if (k >= pc) break;
  }
  if (pmask) flush_candidates<STATS>(S, r, pend, best, best_t, st);
}

// ---- fp32 conservative pre-filter ------------------------------------------
// The exact test needs 18 fp64 instructions per box and rejects 99.8 % of the
// boxes, nearly all of them by a wide margin.  Filter32 rejects most of those
// with 10 fp32 instructions and NEVER rejects a box the exact test accepts;
// survivors go through the exact fp64 test, so results cannot change.
//
// Exact quantities (per axis, NaN-free mode): t = fl64(fl64(x - o) * i) for a
// box plane x, the ray's o and i = 1/d; the box passes iff max(near t) <=
// min(far t) and min(far t) >= 0.
// Filter: X = fl32(x) (host), I = fl32(i), F = fma32(X, I, C) with a per-ray
// constant C.  With u = 2^-24, |x| <= bmax and M = (bmax + |o|) * |i|:
//   |X*I - x*i| <= 2.01 u |x||i|,   |F - (X*I + C)| <= u (|X*I| + |C|),
//   |t - (x - o) i| <= 2^-51 M,     |C| <= M + E
// so |F - C - (-o*i) - t| <= 5 u M whenever E <= 8 u M.  Choosing
//   E = 2^-21 M + 2^-100,  Cn <= -o*i - E (rounded down),  Cf >= -o*i + E (up)
// gives  fma(X, I, Cn) <= t <= fma(X, I, Cf)  for every plane of the scene:
// lower bounds of the near values, upper bounds of the far values.  Hence
//   max3(lower near) > min3(upper far)  or  min3(upper far) < 0   =>  exact test fails.
// Preconditions (else the filter is off for the wave): M <= 2^120 on every
// axis (no fp32 overflow) and the NaN-free mode.  2^-100 covers fp32 underflow.
struct Filter32 {
  float ix, iy, iz;
  float cnx, cny, cnz;
  float cfx, cfy, cfz;
};

// next representable fp32 below / above a finite value
__device__ __forceinline__ float f32_pred(float f) {
  unsigned b = __builtin_bit_cast(unsigned, f);
  if ((b & 0x7fffffffu) == 0u) return __builtin_bit_cast(float, 0x80000001u);  // below +-0: -denorm_min
  b = (b & 0x80000000u) ? b + 1u : b - 1u;
  return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float f32_succ(float f) {
  unsigned b = __builtin_bit_cast(unsigned, f);
  if ((b & 0x7fffffffu) == 0u) return __builtin_bit_cast(float, 0x00000001u);
  b = (b & 0x80000000u) ? b - 1u : b + 1u;
  return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float f32_not_above(double c) {
  float f = (float)c;
  if ((double)f > c) f = f32_pred(f);
  return f;
}
__device__ __forceinline__ float f32_not_below(double c) {
  float f = (float)c;
  if ((double)f < c) f = f32_succ(f);
  return f;
}

// ---- rays with ONE zero direction component --------------------------------
// 1/d = +-inf on exactly one axis a, the other two reciprocals finite and
// non-zero, finite origin.  For a box whose a-range does not contain o[a] even
// after widening it by one fp32 ulp (the fp32 copies are rounded to nearest),
// the reference's slab test (aabb.cc, via primitive_triangle.cc:73-76) gives
// t1 = t2 = +inf or t1 = t2 = -inf on that axis, no NaN; with finite far
// distances on the other two axes that is tmin = +inf > tmax, or tmax = -inf
// < 0: a miss.  The same holds for every box inside such a box, so union boxes
// (blocks, subtrees) may be skipped for these rays too.  Nothing is concluded
// when o[a] lies inside the range: a member box whose min plane equals o[a]
// passes through NaN whatever the other axes say.  (All magnitudes involved
// are bounded by 2^400, so the finite products stay finite.)
__device__ __forceinline__ int degenerate_axis(double ox, double oy, double oz, double ix, double iy,
                                               double iz) {
  // magnitudes bounded so that (plane - o) * (1/d) cannot overflow on the two
  // regular axes (the caller checks the scene's coordinates against 2^400 too)
  const bool fo = __builtin_fabs(ox) <= 0x1p400 && __builtin_fabs(oy) <= 0x1p400 && __builtin_fabs(oz) <= 0x1p400;
  const bool rx = __builtin_fabs(ix) <= 0x1p400 && ix != 0.0, ry = __builtin_fabs(iy) <= 0x1p400 && iy != 0.0,
             rz = __builtin_fabs(iz) <= 0x1p400 && iz != 0.0;
  if (!fo) return -1;
  if (__builtin_isinf(ix) && ry && rz) return 0;
  if (__builtin_isinf(iy) && rx && rz) return 1;
  if (__builtin_isinf(iz) && rx && ry) return 2;
  return -1;
}
__device__ __forceinline__ bool outside_on_axis(const float *b, int axis, double o) {
  return o < (double)f32_pred(b[axis]) || o > (double)f32_succ(b[3 + axis]);
}
// subtree boxes of the eight children -> keep mask for such a ray
__device__ __forceinline__ unsigned degenerate_keep_mask(const float *sub, int axis, double o) {
  unsigned keep = 0u;
#pragma unroll
  for (int c = 0; c < 8; c++) {
    if (!outside_on_axis(sub + c * 6, axis, o)) keep |= 1u << c;
  }
  return keep;
}

// Returns false when the filter must not be used for this ray.
__device__ __forceinline__ bool make_filter32(const DevScene &S, const RayRegs &r, Filter32 &f) {
  const double o[3] = {r.ox, r.oy, r.oz}, iv[3] = {r.ix, r.iy, r.iz};
  float I[3], Cn[3], Cf[3];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const double M = (S.bmax[k] + __builtin_fabs(o[k])) * __builtin_fabs(iv[k]);
    ok = ok && (M <= 0x1p120);  // false for NaN/inf as well
    const double E = M * 0x1p-21 + 0x1p-100;
    const double oi = o[k] * iv[k];
    I[k] = (float)iv[k];
    Cn[k] = f32_not_above(-oi - E);
    Cf[k] = f32_not_below(-oi + E);
  }
  f.ix = I[0]; f.iy = I[1]; f.iz = I[2];
  f.cnx = Cn[0]; f.cny = Cn[1]; f.cnz = Cn[2];
  f.cfx = Cf[0]; f.cfy = Cf[1]; f.cfz = Cf[2];
  return ok;
}

// fp32 verdict for one wave-uniform box b[0..5] (min xyz, max xyz): mask of
// lanes for which the exact test MAY pass.  NaN compares as "may pass".
// OCT = 0..7: the lanes' common sign octant.  OCT = 8: lanes of any octants --
// near = min and far = max of the two products per axis; fma is monotonic in the
// plane coordinate, so these ARE the octant form's values for every lane.
template <int OCT>
__device__ __forceinline__ unsigned long long filter32_pass(const float *b, const Filter32 &f) {
  constexpr int kUGE = 11, kULE = 13;
  float tnx, tny, tnz, tfx, tfy, tfz;
  if constexpr (OCT == 8) {
    tnx = __builtin_fminf(__builtin_fmaf(b[0], f.ix, f.cnx), __builtin_fmaf(b[3], f.ix, f.cnx));
    tfx = __builtin_fmaxf(__builtin_fmaf(b[0], f.ix, f.cfx), __builtin_fmaf(b[3], f.ix, f.cfx));
    tny = __builtin_fminf(__builtin_fmaf(b[1], f.iy, f.cny), __builtin_fmaf(b[4], f.iy, f.cny));
    tfy = __builtin_fmaxf(__builtin_fmaf(b[1], f.iy, f.cfy), __builtin_fmaf(b[4], f.iy, f.cfy));
    tnz = __builtin_fminf(__builtin_fmaf(b[2], f.iz, f.cnz), __builtin_fmaf(b[5], f.iz, f.cnz));
    tfz = __builtin_fmaxf(__builtin_fmaf(b[2], f.iz, f.cfz), __builtin_fmaf(b[5], f.iz, f.cfz));
  } else {
    constexpr int NX = (OCT & 1) ? 3 : 0, FX = (OCT & 1) ? 0 : 3;
    constexpr int NY = (OCT & 2) ? 4 : 1, FY = (OCT & 2) ? 1 : 4;
    constexpr int NZ = (OCT & 4) ? 5 : 2, FZ = (OCT & 4) ? 2 : 5;
    tnx = __builtin_fmaf(b[NX], f.ix, f.cnx); tfx = __builtin_fmaf(b[FX], f.ix, f.cfx);
    tny = __builtin_fmaf(b[NY], f.iy, f.cny); tfy = __builtin_fmaf(b[FY], f.iy, f.cfy);
    tnz = __builtin_fmaf(b[NZ], f.iz, f.cnz); tfz = __builtin_fmaf(b[FZ], f.iz, f.cfz);
  }
  const float lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
  const float hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
  return __builtin_amdgcn_fcmpf(hi, 0.0f, kUGE) & __builtin_amdgcn_fcmpf(lo, hi, kULE);
}


// One per-lane box (a child's subtree box) against the lane's own filter.
__device__ __forceinline__ bool subtree_may_hit(const float *b, const Filter32 &f, bool sx, bool sy, bool sz) {
  const float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4], b5 = b[5];
  const float tnx = __builtin_fmaf(sx ? b3 : b0, f.ix, f.cnx), tfx = __builtin_fmaf(sx ? b0 : b3, f.ix, f.cfx);
  const float tny = __builtin_fmaf(sy ? b4 : b1, f.iy, f.cny), tfy = __builtin_fmaf(sy ? b1 : b4, f.iy, f.cfy);
  const float tnz = __builtin_fmaf(sz ? b5 : b2, f.iz, f.cnz), tfz = __builtin_fmaf(sz ? b2 : b5, f.iz, f.cfz);
  const float lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
  const float hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
  return !(hi < 0.0f) && !(lo > hi);  // NaN: keep
}

// The same test with the planes already picked by the direction's signs (the hit-set walk reads them that way).
__device__ __forceinline__ bool near_far_may_hit(float nx, float fx, float ny, float fy, float nz, float fz, const Filter32 &f) {
  const float tnx = __builtin_fmaf(nx, f.ix, f.cnx), tfx = __builtin_fmaf(fx, f.ix, f.cfx);
  const float tny = __builtin_fmaf(ny, f.iy, f.cny), tfy = __builtin_fmaf(fy, f.iy, f.cfy);
  const float tnz = __builtin_fmaf(nz, f.iz, f.cnz), tfz = __builtin_fmaf(fz, f.iz, f.cfz);
  // !(hi < 0) && !(lo > hi) with the zero folded into the entry distance: one maximum fewer
  const float lo0 = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, 0.0f));
  const float hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
  return !(lo0 > hi);  // NaN: keep
}

// keep mask of the eight subtree boxes `sub` (children fc .. fc + 7) for a ray with one zero direction component:
// the range rule.
__device__ __forceinline__ unsigned degenerate_children(const DevScene *self, const float *sub, int fc, double ox,
                                                        double oy, double oz, double ix, double iy, double iz) {
  const int axis = degenerate_axis(ox, oy, oz, ix, iy, iz);
  if (axis < 0) return 0xffu;
  const double o = axis == 0 ? ox : (axis == 1 ? oy : oz);
  (void)self;
  (void)fc;
  return degenerate_keep_mask(sub, axis, o);
}

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
struct QuadRegs {  // four fp32 boxes = 96 bytes
  f16v lo;
  f8v hi;
};
__device__ __forceinline__ void issue_quad(QuadRegs &q, const MT_CONST float *p) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40"
               : "=&s"(q.lo), "=&s"(q.hi)
               : "s"(p));
}
__device__ __forceinline__ void await_quad(QuadRegs &q) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.lo), "+s"(q.hi));
}

// Four boxes (stream positions k..k+3): fp32 verdicts, then the exact fp64
// test for the boxes some lane survived, in stream order.
template <int OCT, bool STATS>
__device__ __forceinline__ void scan_quad(const DevScene &S, const RayRegs &r, const Filter32 &f,
                                          const QuadRegs &q, QuadRegs &next, int pb, int k, int pc, int &pend,
                                          unsigned long long &pmask, int &best, double &best_t,
                                          LaneStats &st) {
  const float b0[6] = {q.lo[0], q.lo[1], q.lo[2], q.lo[3], q.lo[4], q.lo[5]};
  const float b1[6] = {q.lo[6], q.lo[7], q.lo[8], q.lo[9], q.lo[10], q.lo[11]};
  const float b2[6] = {q.lo[12], q.lo[13], q.lo[14], q.lo[15], q.hi[0], q.hi[1]};
  const float b3[6] = {q.hi[2], q.hi[3], q.hi[4], q.hi[5], q.hi[6], q.hi[7]};
  unsigned long long m[4];
  m[0] = filter32_pass<OCT>(b0, f);
  m[1] = (k + 1 < pc) ? filter32_pass<OCT>(b1, f) : 0ull;
  m[2] = (k + 2 < pc) ? filter32_pass<OCT>(b2, f) : 0ull;
  m[3] = (k + 3 < pc) ? filter32_pass<OCT>(b3, f) : 0ull;
  if ((m[0] | m[1] | m[2] | m[3]) == 0ull) return;
  // Slow path (a few per cent of the quads).  It needs many more registers
  // (fp64 box, Möller–Trumbore), so first let the look-ahead fetch land: from
  // here on nothing is in flight and the compiler may spill what it likes.
  await_quad(next);
  const unsigned long long me = 1ull << (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  const MT_CONST double *boxes = as_const(S.tri_aabb) + (size_t)(pb + k) * 6;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (m[j] == 0ull) continue;  // wave-uniform
    const MT_CONST double *a = boxes + j * 6;
    const double b[6] = {a[0], a[1], a[2], a[3], a[4], a[5]};
    if (STATS) st.bytes_scalar += 48u;
    const unsigned long long pm = slab_pass<(OCT == 8 ? 1 : 2), (OCT == 8 ? 0 : OCT)>(b, r) & m[j];
    if (pm == 0ull) continue;
    if (pm & pmask) {  // some lane would need a second slot: resolve first
      flush_candidates<STATS>(S, r, pend, best, best_t, st);
      pmask = 0;
    }
    if ((pm & me) != 0) pend = pb + k + j;
    pmask |= pm;
  }
}

template <bool EX>
__device__ __forceinline__ bool slab_pass_lane(const double *b, const RayRegs &r);
// ---- "fused" variant of the filtered scan (hit-set traversal) ------------------
// The fp32 verdicts of a quad only mark per-lane CANDIDATES (bit = stream position
// relative to `base`, at most 64 at a time); resolve_candidates then lets every
// lane work through ITS candidates in stream order -- exact fp64 box and the
// vertices fetched together, one round trip per candidate, all lanes and boxes
// in flight at once -- instead of one wave-uniform scalar fetch per surviving box
// followed by a parked Möller–Trumbore pass.  Same tests on the same triangles in
// the same order per lane (octtree.cc:177-196), so the same result.
template <int OCT>
__device__ __forceinline__ void mark_quad(const Filter32 &f, const QuadRegs &q, int k, int pc, int rel, int lane,
                                          unsigned long long &cand) {
  const float b0[6] = {q.lo[0], q.lo[1], q.lo[2], q.lo[3], q.lo[4], q.lo[5]};
  const float b1[6] = {q.lo[6], q.lo[7], q.lo[8], q.lo[9], q.lo[10], q.lo[11]};
  const float b2[6] = {q.lo[12], q.lo[13], q.lo[14], q.lo[15], q.hi[0], q.hi[1]};
  const float b3[6] = {q.hi[2], q.hi[3], q.hi[4], q.hi[5], q.hi[6], q.hi[7]};
  unsigned long long m[4];
  m[0] = filter32_pass<OCT>(b0, f);
  m[1] = (k + 1 < pc) ? filter32_pass<OCT>(b1, f) : 0ull;
  m[2] = (k + 2 < pc) ? filter32_pass<OCT>(b2, f) : 0ull;
  m[3] = (k + 3 < pc) ? filter32_pass<OCT>(b3, f) : 0ull;
  if ((m[0] | m[1] | m[2] | m[3]) == 0ull) return;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    if (m[j] == 0ull) continue;  // wave-uniform
    cand |= ((m[j] >> lane) & 1ull) << (rel + j);
  }
}

template <bool STATS>
__device__ __forceinline__ void resolve_candidates(const DevScene &S, const RayRegs &r, int base,
                                                   unsigned long long &cand, int &best, double &best_t,
                                                   LaneStats &st) {
  for (int guard = 0; guard < 64 && __ballot(cand != 0ull) != 0ull; guard++) {
    if (cand != 0ull) {
      const int t = base + __builtin_ctzll(cand);
      cand &= cand - 1ull;
      const double *ep = S.tri_aabb + (size_t)t * 6;
      const double *vp = S.tri_vertex + (size_t)t * 9;
      const double e[6] = {ep[0], ep[1], ep[2], ep[3], ep[4], ep[5]};
      const double v[9] = {vp[0], vp[1], vp[2], vp[3], vp[4], vp[5], vp[6], vp[7], vp[8]};
      if (STATS) st.v[ST_BYTES_VECTOR] += 120u;
      if (slab_pass_lane<false>(e, r)) {
        if (STATS) st.v[ST_MT_TESTS]++;
        double tt;
        if (moller_trumbore_v(v, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, &tt)) {
          if (!(best >= 0 && tt > best_t)) {
            best = t;
            best_t = tt;
          }
        }
      }
    }
  }
  cand = 0ull;
}

template <int OCT, bool STATS>
__device__ __forceinline__ void scan_node_fused(const DevScene &S, const RayRegs &r, const Filter32 &f, int pb,
                                                int pc, int &best, double &best_t, LaneStats &st) {
  if (pc <= 0) return;
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const MT_CONST float *p = as_const(S.tri_aabb32) + (size_t)pb * 6;
  unsigned long long cand = 0ull;
  int base = pb;
  QuadRegs A, B;
  issue_quad(A, p);
  await_quad(A);
  if (STATS) st.bytes_scalar += 96u * (unsigned)((pc + 3) / 4 + 1);
  for (int k = 0;;) {
    issue_quad(B, p + 24);  // unconditional look-ahead: the stream is padded
    if (pb + k - base > 60) {
      resolve_candidates<STATS>(S, r, base, cand, best, best_t, st);
      base = pb + k;
    }
    mark_quad<OCT>(f, A, k, pc, pb + k - base, lane, cand);
    await_quad(B);
    k += 4;
    if (k >= pc) break;
    p += 48;
    issue_quad(A, p);
    if (pb + k - base > 60) {
      resolve_candidates<STATS>(S, r, base, cand, best, best_t, st);
      base = pb + k;
    }
    mark_quad<OCT>(f, B, k, pc, pb + k - base, lane, cand);
    await_quad(A);
    k += 4;
    if (k >= pc) break;
  }
  if (__ballot(cand != 0ull) != 0ull) resolve_candidates<STATS>(S, r, base, cand, best, best_t, st);
}

// Ray-parallel scan with the fp32 pre-filter (octant-uniform mode only).
template <int OCT, bool STATS>
__device__ __forceinline__ void scan_node_filtered(const DevScene &S, const RayRegs &r,
                                                   const Filter32 &f, int pb, int pc, int &best,
                                                   double &best_t, LaneStats &st) {
  if (pc <= 0) return;
  const MT_CONST float *p = as_const(S.tri_aabb32) + (size_t)pb * 6;
  int pend = -1;
  unsigned long long pmask = 0;
  QuadRegs A, B;
  issue_quad(A, p);
  await_quad(A);
  if (STATS) st.bytes_scalar += 96u * (unsigned)((pc + 3) / 4 + 1);  // four fp32 boxes per fetch, one look-ahead
  for (int k = 0;;) {
    issue_quad(B, p + 24);  // unconditional look-ahead: the stream is padded
    scan_quad<OCT, STATS>(S, r, f, A, B, pb, k, pc, pend, pmask, best, best_t, st);
    await_quad(B);
    k += 4;
    if (k >= pc) break;
    p += 48;
    issue_quad(A, p);
    scan_quad<OCT, STATS>(S, r, f, B, A, pb, k, pc, pend, pmask, best, best_t, st);
    await_quad(A);
    k += 4;
    if (k >= pc) break;
  }
  if (pmask) flush_candidates<STATS>(S, r, pend, best, best_t, st);
}

// fp32 verdicts for four group boxes: bit j set when some lane may hit group j.
template <int OCT>
__device__ __forceinline__ unsigned group_quad(const QuadRegs &q, const Filter32 &f, int left) {
  const float b0[6] = {q.lo[0], q.lo[1], q.lo[2], q.lo[3], q.lo[4], q.lo[5]};
  const float b1[6] = {q.lo[6], q.lo[7], q.lo[8], q.lo[9], q.lo[10], q.lo[11]};
  const float b2[6] = {q.lo[12], q.lo[13], q.lo[14], q.lo[15], q.hi[0], q.hi[1]};
  const float b3[6] = {q.hi[2], q.hi[3], q.hi[4], q.hi[5], q.hi[6], q.hi[7]};
  // branch-free (the list is padded; boxes past its end are masked off)
  const unsigned m = (filter32_pass<OCT>(b0, f) != 0ull ? 1u : 0u) |
                     (filter32_pass<OCT>(b1, f) != 0ull ? 2u : 0u) |
                     (filter32_pass<OCT>(b2, f) != 0ull ? 4u : 0u) |
                     (filter32_pass<OCT>(b3, f) != 0ull ? 8u : 0u);
  return left >= 4 ? m : (m & ((1u << left) - 1u));
}

// The group boxes of a big node (one per kGroupTris triangles, wave-uniform),
// at most 64 of them: bit g of the result = some lane may hit group g.
template <int OCT>
__device__ __forceinline__ unsigned long long group_live_mask(const MT_CONST float *p, int n,
                                                              const Filter32 &f) {
  unsigned long long live = 0ull;
  QuadRegs A, B;
  issue_quad(A, p);
  await_quad(A);
  for (int k = 0;;) {
    issue_quad(B, p + 24);  // unconditional look-ahead: the array is padded
    live |= (unsigned long long)group_quad<OCT>(A, f, n - k) << k;
    await_quad(B);
    k += 4;
    if (k >= n) break;
    p += 48;
    issue_quad(A, p);
    live |= (unsigned long long)group_quad<OCT>(B, f, n - k) << k;
    await_quad(A);
    k += 4;
    if (k >= n) break;
  }
  return live;
}

// Transposed scan: ONE ray at a time, 64 TRIANGLES per step (lane = triangle).
// Used when only a few lanes want a node: scanning a 90-triangle list for 2
// rays costs 4 of these steps instead of 90 ray-parallel ones.  All 64 lanes
// of the wave take part, whatever node they themselves are waiting for.
// Equivalence with the reference's sequential loop (octtree.cc:177-196): the
// hits of a chunk are folded in ascending triangle order with the same
// "keep the old one only if it is strictly closer" rule, chunk after chunk.
template <bool EX, bool STATS>
__device__ __forceinline__ void scan_node_transposed(const DevScene &S, const RayRegs &r, int lane,
                                                     unsigned long long inmask, int pb, int pc,
                                                     int &best, double &best_t, LaneStats &st) {
  const double *boxes = S.tri_aabb + (size_t)pb * 6;
  unsigned long long todo = inmask;
  while (todo != 0ull) {
    const int L = __builtin_ctzll(todo);
    todo &= todo - 1;
    RayRegs u;  // lane L's ray, wave-uniform
    u.ox = readlane_f64(r.ox, L); u.oy = readlane_f64(r.oy, L); u.oz = readlane_f64(r.oz, L);
    u.dx = readlane_f64(r.dx, L); u.dy = readlane_f64(r.dy, L); u.dz = readlane_f64(r.dz, L);
    u.ix = readlane_f64(r.ix, L); u.iy = readlane_f64(r.iy, L); u.iz = readlane_f64(r.iz, L);
    int ubest = -1;
    double ubest_t = 0.0;
    unsigned mt_count = 0;
    for (int base = 0; base < pc; base += 64) {
      const int tri = base + lane;
      const int tri_c = tri < pc ? tri : pc - 1;
      const double *bp = boxes + (size_t)tri_c * 6;
      const double b[6] = {bp[0], bp[1], bp[2], bp[3], bp[4], bp[5]};
      if (STATS) st.v[ST_BYTES_VECTOR] += 48u;
      const unsigned long long pm =
          slab_pass<EX ? 0 : 1, 0>(b, u) & __builtin_amdgcn_ballot_w64(tri < pc);
      if (pm != 0ull) {
        if (STATS) mt_count += (unsigned)__builtin_popcountll(pm);
        const bool mine = ((pm >> lane) & 1ull) != 0;
        double t = 0.0;
        bool hit = false;
        if (mine) {
          if (STATS) st.v[ST_BYTES_VECTOR] += 72u;
          hit = moller_trumbore(S.tri_vertex + (size_t)(pb + tri_c) * 9, u.ox, u.oy, u.oz, u.dx, u.dy, u.dz, &t);
        }
        unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
        while (hm != 0ull) {  // ascending triangle order
          const int i = __builtin_ctzll(hm);
          hm &= hm - 1;
          const double ti = readlane_f64(t, i);
          if (!(ubest >= 0 && ti > ubest_t)) {
            ubest = pb + base + i;
            ubest_t = ti;
          }
        }
      }
    }
    if (lane == L) {
      best = ubest;
      best_t = ubest_t;
      if (STATS) st.v[ST_MT_TESTS] += mt_count;
    }
  }
}

// Transposed scan with block boxes (regular rays only): per ray, 64 BLOCK boxes
// per step (lane = block, fp32 conservative test with the ray's own filter
// constants), then the triangles of the blocks that may be hit, four blocks
// (64 triangles) per step, exactly as above: exact fp64 box test, Möller–
// Trumbore, hits folded in ascending stream order.
static_assert(64 % kGroupTris == 0, "a wave must hold whole blocks");
template <bool EX, bool STATS>
__device__ __forceinline__ void scan_node_transposed_blocks(const DevScene &S, const RayRegs &r,
                                                            const Filter32 &f, int lane,
                                                            unsigned long long inmask, int pb, int pc,
                                                            int &best, double &best_t, LaneStats &st,
                                                            unsigned *tp = nullptr) {
  const int b0 = pb / kGroupTris, nb = (pb + pc - 1) / kGroupTris - b0 + 1;
  const float *blk = S.grp_aabb32 + (size_t)b0 * 6;
  unsigned long long todo = inmask;
  while (todo != 0ull) {
    const int L = __builtin_ctzll(todo);
    todo &= todo - 1;
#ifdef MT_PROF
    const unsigned long long tt0 = __builtin_amdgcn_s_memtime();
    unsigned long long tt_tri = 0;
#endif
    RayRegs u;  // lane L's ray, wave-uniform
    u.ox = readlane_f64(r.ox, L); u.oy = readlane_f64(r.oy, L); u.oz = readlane_f64(r.oz, L);
    u.dx = readlane_f64(r.dx, L); u.dy = readlane_f64(r.dy, L); u.dz = readlane_f64(r.dz, L);
    u.ix = readlane_f64(r.ix, L); u.iy = readlane_f64(r.iy, L); u.iz = readlane_f64(r.iz, L);
    const float fix = readlane_f32(f.ix, L), fiy = readlane_f32(f.iy, L), fiz = readlane_f32(f.iz, L);
    const float cnx = readlane_f32(f.cnx, L), cny = readlane_f32(f.cny, L), cnz = readlane_f32(f.cnz, L);
    const float cfx = readlane_f32(f.cfx, L), cfy = readlane_f32(f.cfy, L), cfz = readlane_f32(f.cfz, L);
    // near / far plane of each axis by the ray's direction sign (wave-uniform)
    const int nx = __builtin_signbit(u.ix) ? 3 : 0, ny = __builtin_signbit(u.iy) ? 4 : 1,
              nz = __builtin_signbit(u.iz) ? 5 : 2;
    const int fx = 3 - nx, fy = 5 - ny, fz = 7 - nz;
    const int deg_axis = EX ? degenerate_axis(u.ox, u.oy, u.oz, u.ix, u.iy, u.iz) : -1;
    const double deg_o = deg_axis == 0 ? u.ox : (deg_axis == 1 ? u.oy : u.oz);
    int ubest = -1;
    double ubest_t = 0.0;
    unsigned mt_count = 0;
#ifdef MT_PROF
    const unsigned long long tt1 = __builtin_amdgcn_s_memtime();
#endif
    for (int g0 = 0; g0 < nb; g0 += 64) {
      const int g = g0 + lane;
      const float *bp = blk + (size_t)(g < nb ? g : nb - 1) * 6;
      if (STATS) st.v[ST_BYTES_VECTOR] += 24u;
      const float tnx = __builtin_fmaf(bp[nx], fix, cnx), tfx = __builtin_fmaf(bp[fx], fix, cfx);
      const float tny = __builtin_fmaf(bp[ny], fiy, cny), tfy = __builtin_fmaf(bp[fy], fiy, cfy);
      const float tnz = __builtin_fmaf(bp[nz], fiz, cnz), tfz = __builtin_fmaf(bp[fz], fiz, cfz);
      const float lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
      const float hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
      // "may pass" unless provably not (NaN compares as may pass, like filter32_pass);
      // EX: an irregular ray -- only the one-zero-component test above applies
      bool may = !(hi < 0.0f) && !(lo > hi);
      if (EX) {
        // the range rule
        const bool range_ok = !(deg_axis >= 0 && outside_on_axis(bp, deg_axis, deg_o));
        may = range_ok;
      }
      unsigned long long live = __builtin_amdgcn_ballot_w64(may && g < nb);
#ifdef MT_PROF
      const unsigned long long tt2 = __builtin_amdgcn_s_memtime();
#endif
      while (live != 0ull) {
        // the next 64 / kGroupTris live blocks, one per slice of the wave
        constexpr int kSlices = 64 / kGroupTris;
        const int slice = lane / kGroupTris;
        int myq = -1;
#pragma unroll
        for (int q = 0; q < kSlices; q++) {
          if (live != 0ull) {
            const int blk = g0 + __builtin_ctzll(live);
            live &= live - 1;
            if (slice == q) myq = blk;
          }
        }
        const int tri = (b0 + myq) * kGroupTris + (lane % kGroupTris);  // stream position
        const bool ok = myq >= 0 && tri >= pb && tri < pb + pc;
        const int tri_c = ok ? tri : pb;
        const double *bx = S.tri_aabb + (size_t)tri_c * 6;
        const double b[6] = {bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]};
        if (STATS) st.v[ST_BYTES_VECTOR] += 48u;
        const unsigned long long pm = slab_pass<EX ? 0 : 1, 0>(b, u) & __builtin_amdgcn_ballot_w64(ok);
        if (pm == 0ull) continue;
        if (STATS) mt_count += (unsigned)__builtin_popcountll(pm);
        const bool mine = ((pm >> lane) & 1ull) != 0;
        double t = 0.0;
        bool hit = false;
        if (STATS && mine) st.v[ST_BYTES_VECTOR] += 72u;
        if (mine) hit = moller_trumbore(S.tri_vertex + (size_t)tri_c * 9, u.ox, u.oy, u.oz, u.dx, u.dy, u.dz, &t);
        unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
        while (hm != 0ull) {  // ascending lane = ascending stream position
          const int i = __builtin_ctzll(hm);
          hm &= hm - 1;
          const double ti = readlane_f64(t, i);
          if (!(ubest >= 0 && ti > ubest_t)) {
            ubest = __builtin_amdgcn_readlane(tri_c, i);
            ubest_t = ti;
          }
        }
      }
#ifdef MT_PROF
      tt_tri += __builtin_amdgcn_s_memtime() - tt2;
#endif
    }
#ifdef MT_PROF
    if (tp) {
      const unsigned long long tt3 = __builtin_amdgcn_s_memtime();
      tp[0] += (unsigned)(tt1 - tt0);                 // broadcast of the ray
      tp[1] += (unsigned)(tt3 - tt1 - tt_tri);        // block stage
      tp[2] += (unsigned)tt_tri;                      // triangle stage
    }
#endif
    if (lane == L) {
      best = ubest;
      best_t = ubest_t;
      if (STATS) st.v[ST_MT_TESTS] += mt_count;
    }
  }
}


// Child slab tests + ordering of the hit children, octtree.cc:204-216.
// Returns ord (3 bits per entry) | count << 24.
// keep: bit c clear = child c's subtree provably holds no triangle this ray's
// pre-filter accepts; such a child is left out of the list.
// The sort is stable and a child without a hit only makes the parent's loop
// move on (octtree.cc:222-251), so leaving it out changes nothing.
template <int MODE>
// um (wave-uniform): children outside it are not looked at at all (regular
// mode only; the caller passes the union of the lanes' non-empty children when
// subtree skipping is on, so these are children `keep` would drop anyway).
// sub/f (regular mode): the children's subtree boxes are tested here, and only
// for the children some lane enters.
__device__ __forceinline__ unsigned order_children(const MT_CONST NodeRec *N, const RayRegs &r,
                                                   unsigned keep = 0xffu, unsigned um = 0xffu,
                                                   const float *sub = nullptr, const Filter32 *f = nullptr) {
  constexpr bool EX = (MODE == 0);
  double xmin[2], xmax[2], ymin[2], ymax[2], zmin[2], zmax[2];
  {
    const double t0 = (N->lo[0] - r.ox) * r.ix, tc = (N->c[0] - r.ox) * r.ix,
                 t1 = (N->hi[0] - r.ox) * r.ix;
    // NodeIntersectRay's t1/t2 are (min - o)*inv, (max - o)*inv in that order
    xmax[0] = mx<EX>(t0, tc); xmin[0] = mn<EX>(t0, tc);
    xmax[1] = mx<EX>(tc, t1); xmin[1] = mn<EX>(tc, t1);
  }
  {
    const double t0 = (N->lo[1] - r.oy) * r.iy, tc = (N->c[1] - r.oy) * r.iy,
                 t1 = (N->hi[1] - r.oy) * r.iy;
    ymax[0] = mx<EX>(t0, tc); ymin[0] = mn<EX>(t0, tc);
    ymax[1] = mx<EX>(tc, t1); ymin[1] = mn<EX>(tc, t1);
  }
  {
    const double t0 = (N->lo[2] - r.oz) * r.iz, tc = (N->c[2] - r.oz) * r.iz,
                 t1 = (N->hi[2] - r.oz) * r.iz;
    zmax[0] = mx<EX>(t0, tc); zmin[0] = mn<EX>(t0, tc);
    zmax[1] = mx<EX>(tc, t1); zmin[1] = mn<EX>(tc, t1);
  }
  double tm[8];
  bool valid[8];
#pragma unroll
  for (int c = 0; c < 8; c++) {
    if constexpr (!EX) {
      if (sub != nullptr && ((um >> c) & 1u) == 0u) {  // empty subtree for every lane: wave-uniform skip
        valid[c] = false;
        tm[c] = 0.0;
        continue;
      }
    }
    const int xh = c & 1, zh = (c >> 1) & 1, yh = (c >> 2) & 1;  // octtree.cc:61-100
    const double tmax = mn3<EX>(xmax[xh], ymax[yh], zmax[zh]);
    const double tmin = mx3<EX>(xmin[xh], ymin[yh], zmin[zh]);
    if constexpr (EX) valid[c] = !(tmax < 0.0) && !(tmin > tmax);
    else valid[c] = (tmax >= 0.0) & (tmin <= tmax) & (((keep >> c) & 1u) != 0u);
    if constexpr (!EX) {
      if (sub != nullptr) {
        if (__ballot(valid[c]) != 0ull) {
          valid[c] = valid[c] && subtree_may_hit(sub + c * 6, *f, __builtin_signbit(r.ix),
                                                 __builtin_signbit(r.iy), __builtin_signbit(r.iz));
        }
      }
    }
    tm[c] = tmin;
  }
  unsigned ord = 0, cnt = 0;
  if constexpr (!EX) {
    // At most one child left for any lane of the wave: nothing to sort.
    unsigned any = 0u;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (__ballot(valid[c]) != 0ull) any |= 1u << c;
    }
    if ((any & (any - 1u)) == 0u) {
      if (any == 0u) return 0u;
      const unsigned c0 = (unsigned)__builtin_ctz(any);
      bool v0 = false;
#pragma unroll
      for (int c = 0; c < 8; c++) v0 = v0 || (valid[c] && (unsigned)c == c0);
      return v0 ? (c0 | (1u << 24)) : 0u;
    }
    // No NaN keys: the stable sort by tmin is the order by (tmin, index).
    // Only the children that are left for SOME lane of the wave (`any`, 2..4 of
    // them for coherent rays) take part in the pair comparisons.
    unsigned rank[8];
#pragma unroll
    for (int c = 0; c < 8; c++) rank[c] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (((any >> i) & 1u) == 0u) continue;  // wave-uniform
#pragma unroll
      for (int j = i + 1; j < 8; j++) {
        if (((any >> j) & 1u) == 0u) continue;  // wave-uniform
        const bool both = valid[i] && valid[j];
        const bool j_first = tm[j] < tm[i];
        rank[i] += (both && j_first) ? 1u : 0u;
        rank[j] += (both && !j_first) ? 1u : 0u;
      }
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (((any >> c) & 1u) == 0u) continue;  // wave-uniform
      if (valid[c]) {
        ord |= (unsigned)c << (3 * rank[c]);
        cnt++;
      }
    }
  } else {
    // Two shortcuts that the literal sort below would arrive at too (round 4: these rays' units are the longest of
    // the frame, and the literal sort was half of their node steps):
    //   * no key of an entered child is NaN: the insertion sort is stable and `<` is a strict weak order on the keys
    //     (+-inf included, -0 == +0), so the result is the order by (tmin, index) -- rank counting;
    //   * EVERY entered child's key is NaN (a ray with one zero direction component whose origin lies ON a plane of
    //     the node: the children on the NaN side pass, the others fail outright): `vd < x` and `x < vd` are false for
    //     every pair, nothing ever moves: the children stay in index order.
    // Keys of both kinds together (two zero components) take the literal sort.
    bool any_nan = false, all_nan = true;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const bool isn = tm[c] != tm[c];
      any_nan = any_nan || (valid[c] && isn);
      all_nan = all_nan && (!valid[c] || isn);
    }
    if (!any_nan || all_nan) {
      unsigned rank[8];
#pragma unroll
      for (int c = 0; c < 8; c++) rank[c] = 0;
      if (!any_nan) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
          for (int j = i + 1; j < 8; j++) {
            const bool both = valid[i] && valid[j];
            const bool j_first = tm[j] < tm[i];
            rank[i] += (both && j_first) ? 1u : 0u;
            rank[j] += (both && !j_first) ? 1u : 0u;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
          for (int j = i + 1; j < 8; j++) rank[j] += (valid[i] && valid[j]) ? 1u : 0u;
        }
      }
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (valid[c]) {
          ord |= (unsigned)c << (3 * rank[c]);
          cnt++;
        }
      }
    } else {
    // libstdc++ std::sort on <= 16 elements is __insertion_sort; restated
    // literally (front-rotate branch and unguarded linear insert) so that NaN
    // keys land where the reference puts them.
    double sd[8];
    unsigned si[8];
#pragma unroll
    for (int c = 0; c < 8; c++) { sd[c] = 0.0; si[c] = 0; }
    unsigned m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (valid[c]) {
        const double vd = tm[c];
        if (m == 0) {
          sd[0] = vd; si[0] = c;
        } else if (vd < sd[0]) {
#pragma unroll
          for (int j = 7; j >= 1; j--) {
            if ((unsigned)j <= m) { sd[j] = sd[j - 1]; si[j] = si[j - 1]; }
          }
          sd[0] = vd; si[0] = c;
        } else {
          bool moving = true;
#pragma unroll
          for (int j = 7; j >= 1; j--) {
            if (moving && (unsigned)j <= m) {
              if (vd < sd[j - 1]) {
                sd[j] = sd[j - 1]; si[j] = si[j - 1];
              } else {
                sd[j] = vd; si[j] = c;
                moving = false;
              }
            }
          }
        }
        m++;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) ord |= (si[j] & 7u) << (3 * j);
    cnt = m;
    }
    // entries >= cnt are zero-filled garbage; they are never read (pos < cnt)
    if (keep != 0xffu) {
      // With NaN keys the insertion sort's outcome depends on every element
      // that takes part, so skipped children are taken out AFTER the sort.
      unsigned o2 = 0, c2 = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const unsigned c = (ord >> (3 * j)) & 7u;
        if ((unsigned)j < cnt && ((keep >> c) & 1u) != 0u) {
          o2 |= c << (3 * c2);
          c2++;
        }
      }
      ord = o2;
      cnt = c2;
    }
  }
  return (ord & 0x00ffffffu) | (cnt << 24);
}

// ---- lane-parallel scan of a SMALL node ---------------------------------------
// A leaf of the reference's octree holds fewer than SPLIT_BOUNDARY = 16
// triangles and deep split nodes keep only a few straddlers; streaming such a
// list wave-uniformly costs a whole wave step (node record, scalar fetch
// latency, call) for a handful of tests.  Instead every lane scans ITS OWN small
// node here, all lanes at once, with per-lane loads: the same tests in the same
// order (octtree.cc:177-196), so the same result.
template <bool EX>
__device__ __forceinline__ bool slab_pass_lane(const double *b, const RayRegs &r) {
  const double t1 = (b[0] - r.ox) * r.ix, t2 = (b[3] - r.ox) * r.ix;
  const double t3 = (b[1] - r.oy) * r.iy, t4 = (b[4] - r.oy) * r.iy;
  const double t5 = (b[2] - r.oz) * r.iz, t6 = (b[5] - r.oz) * r.iz;
  const double tmax = mn3<EX>(mx<EX>(t1, t2), mx<EX>(t3, t4), mx<EX>(t5, t6));
  const double tmin = mx3<EX>(mn<EX>(t1, t2), mn<EX>(t3, t4), mn<EX>(t5, t6));
  return !(tmax < 0.0) && !(tmin > tmax);
}

// Scalar parameters only (they travel in registers; a by-value struct would go
// through the stack).  Boxes are fetched three at a time so that their load
// latencies overlap; the tests then run in list order.
template <bool EX, bool STATS>
__device__ __attribute__((noinline)) ScanOut scan_small_lane_call(const double *b64, const double *vtx,
                                                                  int pb, int pc, double ox, double oy,
                                                                  double oz, double dx, double dy,
                                                                  double dz, double ix, double iy,
                                                                  double iz) {
  RayRegs r;
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.dx = dx; r.dy = dy; r.dz = dz;
  r.ix = ix; r.iy = iy; r.iz = iz;
  ScanOut o{-1, 0.0, 0u};
  for (int k = 0; k < pc; k += 3) {
    if (STATS) o.bytes_v += 3u * 48u;
    double b[3][6];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int kk = (k + j < pc) ? (k + j) : (pc - 1);  // clamped: the load is always valid
      const double *bp = b64 + (size_t)(pb + kk) * 6;
#pragma unroll
      for (int i = 0; i < 6; i++) b[j][i] = bp[i];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (k + j >= pc) break;
      if (!slab_pass_lane<EX>(b[j], r)) continue;
      if (STATS) {
        o.mt_tests++;
        o.bytes_v += 72u;
      }
      double t;
      if (moller_trumbore(vtx + (size_t)(pb + k + j) * 9, ox, oy, oz, dx, dy, dz, &t)) {
        if (!(o.best >= 0 && t > o.best_t)) {
          o.best = pb + k + j;
          o.best_t = t;
        }
      }
    }
  }
  return o;
}

// Subtree boxes: sub[c*6 .. c*6+5] = fp32 union box of every triangle stored in
// child c or below it.  A ray that misses it (conservatively, in fp32; same
// argument as for the block boxes) fails the reference's AABB pre-filter for
// every triangle down there, so visiting that child could only return "no hit"
// (subtree_may_hit, applied inside order_children to the children some lane
// enters).

// Regular-mode ordering, inlined into the traversal; the exact-mode variant
// below stays a function of its own.
__device__ __forceinline__ unsigned order_children_regular(const NodeRec *N, const RayRegs &r, const float *sub,
                                                           const Filter32 &f, bool uniform_node) {
  unsigned keep = 0xffu, um = 0xffu;
  if (sub != nullptr) {
    // children with an empty subtree (NodeRec::child_mask) are dropped by the
    // subtree test anyway: leave them out of all the arithmetic, wave-wide
    if (uniform_node) {
      um = (unsigned)as_const(uniform_ptr(N))->child_mask & 0xffu;
    } else {
      const unsigned m = (unsigned)N->child_mask;
      um = 0u;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (__ballot(((m >> c) & 1u) != 0u) != 0ull) um |= 1u << c;
      }
    }
  }
  return order_children<1>(as_const(uniform_node ? uniform_ptr(N) : N), r, keep, um,
                           sub == nullptr ? nullptr : (uniform_node ? uniform_ptr(sub) : sub), &f);
}

// Exact-mode ordering (NaN-capable rays) as functions of their own: the literal
// insertion sort needs ~60 registers and is rare.  sub = the eight subtree boxes
// of the node's children for rays with one zero direction component
// (degenerate_axis), or nullptr.  _lane: per-lane node; the other: wave-uniform.
__device__ __attribute__((noinline)) unsigned order_children_exact_lane_call(const NodeRec *N, double ox,
                                                                             double oy, double oz, double ix,
                                                                             double iy, double iz,
                                                                             const float *sub,
                                                                             const DevScene *self, int fc) {
  RayRegs r;
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.dx = 0.0; r.dy = 0.0; r.dz = 0.0;
  r.ix = ix; r.iy = iy; r.iz = iz;
  unsigned keep = 0xffu;
  if (sub != nullptr) keep = degenerate_children(self, sub, fc, ox, oy, oz, ix, iy, iz);
  return order_children<0>(as_const(N), r, keep);
}
__device__ __attribute__((noinline)) unsigned order_children_exact_call(const NodeRec *N, double ox, double oy,
                                                                        double oz, double ix, double iy,
                                                                        double iz, const float *sub,
                                                                        const DevScene *self, int fc) {
  RayRegs r;
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.dx = 0.0; r.dy = 0.0; r.dz = 0.0;
  r.ix = ix; r.iy = iy; r.iz = iz;
  unsigned keep = 0xffu;
  if (sub != nullptr) keep = degenerate_children(uniform_ptr(self), uniform_ptr(sub), uniform_i32(fc), ox, oy, oz, ix, iy, iz);
  return order_children<0>(as_const(uniform_ptr(N)), r, keep);
}

// Small nodes of a FEW lanes, exact tests, triangle-parallel and PACKED: the lists of all lanes in `need` (each
// lane's own node: pb / pc per lane, pc < kBigNode) are laid end to end over the wave's lanes -- lane = one (ray,
// triangle) pair, 64 pairs per trip -- every lane fetches its pair's box and tests it against the OWNER's ray
// (ds_bpermute), Moeller-Trumbore for the boxes that pass, and the hits are folded into their owners in ascending
// slot order = list order per owner (octtree.cc:177-196: a later hit wins unless it is strictly farther).  What a
// lane scanning its list by itself does in pc trips of one triangle, sixty lanes idle, takes ONE trip here for six
// lists of ten triangles.  Must be called by all 64 lanes; the result is the owner lane's.
template <bool STATS>
__device__ __attribute__((noinline)) ScanOut scan_small_packed_call(const double *b64, const double *vtx, int pb, int pc,
                                                                    unsigned need_lo, unsigned need_hi, MT_RAY_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  b64 = uniform_ptr(b64);
  vtx = uniform_ptr(vtx);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const unsigned long long need =
      ((unsigned long long)(unsigned)uniform_i32((int)need_hi) << 32) | (unsigned)uniform_i32((int)need_lo);
  ScanOut o{-1, 0.0, 0u};
  int total = 0;
  for (unsigned long long w = need; w != 0ull; w &= w - 1ull) total += __builtin_amdgcn_readlane(pc, __builtin_ctzll(w));
  auto from_owner = [&](double v, int owner) -> double {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(owner * 4, (int)(unsigned)u);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(owner * 4, (int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  };
  for (int base = 0; base < total; base += 64) {
    const int slot = base + lane;
    // whose list does this slot belong to, and which triangle of it?
    int owner = 0, tri = 0, off = 0;
    bool valid = false;
    for (unsigned long long w = need; w != 0ull; w &= w - 1ull) {
      const int L = __builtin_ctzll(w);
      const int pcl = __builtin_amdgcn_readlane(pc, L), pbl = __builtin_amdgcn_readlane(pb, L);
      if (slot >= off && slot < off + pcl) {
        owner = L;
        tri = pbl + (slot - off);
        valid = true;
      }
      off += pcl;
    }
    RayRegs u;
    u.ox = from_owner(r.ox, owner); u.oy = from_owner(r.oy, owner); u.oz = from_owner(r.oz, owner);
    u.dx = from_owner(r.dx, owner); u.dy = from_owner(r.dy, owner); u.dz = from_owner(r.dz, owner);
    u.ix = from_owner(r.ix, owner); u.iy = from_owner(r.iy, owner); u.iz = from_owner(r.iz, owner);
    bool pass = false;
    if (valid) {
      const double *bp = b64 + (size_t)tri * 6;
      const double b[6] = {bp[0], bp[1], bp[2], bp[3], bp[4], bp[5]};
      pass = slab_pass_lane<true>(b, u);
    }
    const unsigned long long pm = __ballot(pass);
    if (STATS) {
      // (the owner counts what was fetched and tested for its list)
      unsigned long long mine_slots = 0ull;
      {
        int off2 = 0;
        for (unsigned long long w = need; w != 0ull; w &= w - 1ull) {
          const int L = __builtin_ctzll(w);
          const int pcl = __builtin_amdgcn_readlane(pc, L);
          const int a = off2 - base, e = off2 + pcl - base;  // this owner's slots within the trip: [a, e)
          if (lane == L && e > 0 && a < 64) {
            const int a0 = a < 0 ? 0 : a, e0 = e > 64 ? 64 : e;
            mine_slots = (e0 - a0 >= 64 ? ~0ull : ((1ull << (e0 - a0)) - 1ull)) << a0;
          }
          off2 += pcl;
        }
      }
      o.bytes_v += 48u * (unsigned)__builtin_popcountll(mine_slots) + 72u * (unsigned)__builtin_popcountll(mine_slots & pm);
      o.mt_tests += (unsigned)__builtin_popcountll(mine_slots & pm);
    }
    if (pm == 0ull) continue;
    double t = 0.0;
    bool hit = false;
    if (pass) hit = moller_trumbore(vtx + (size_t)tri * 9, u.ox, u.oy, u.oz, u.dx, u.dy, u.dz, &t);
    unsigned long long hm = __ballot(hit);
    while (hm != 0ull) {  // ascending slot = list order within every owner's list
      const int i = __builtin_ctzll(hm);
      hm &= hm - 1ull;
      const double ti = readlane_f64(t, i);
      const int ow = __builtin_amdgcn_readlane(owner, i), tr = __builtin_amdgcn_readlane(tri, i);
      if (lane == ow && !(o.best >= 0 && ti > o.best_t)) {
        o.best = tr;
        o.best_t = ti;
      }
    }
  }
  return o;
}

// ---- non-inlined entry points of the node scans ------------------------------
// Each scan loop is a function of its own: own register allocation, so that the
// SGPR box buffers (48 registers, partly in flight) never compete with the
// traversal's own state.  Arguments arrive in VGPRs and are made uniform again.

// The filtered scans take the scene through its device copy (DevScene::self,
// scalar loads) instead of four pointers: with the ray (18 dwords) and the
// filter constants (9) the argument list must stay within the 32 dwords the
// calling convention passes in registers -- more would go through scratch.
__device__ __forceinline__ DevScene scan_ctx_self(const DevScene *self) {
  const MT_CONST DevScene *G = as_const(uniform_ptr(self));
  DevScene S;
  S.tri_aabb = G->tri_aabb;
  S.tri_aabb32 = G->tri_aabb32;
  S.grp_aabb32 = G->grp_aabb32;
  S.sup_aabb32 = G->sup_aabb32;
  S.tri_vertex = G->tri_vertex;
  S.ll_tri = G->ll_tri;
  S.ll_exact = G->ll_exact;
  S.self = uniform_ptr(self);
  return S;
}

__device__ __forceinline__ DevScene scan_ctx(const float *b32, const double *b64, const double *vtx) {
  DevScene S;
  S.tri_aabb32 = uniform_ptr(b32);
  S.tri_aabb = uniform_ptr(b64);
  S.tri_vertex = uniform_ptr(vtx);
  S.self = nullptr;
  return S;
}

// Lane-parallel scan of a small node with the fp32 pre-filter (regular rays):
// every lane tests the triangles of ITS node -- fp32 box first (planes picked by
// the lane's own direction signs; half the bytes and a third of the VALU time of
// the fp64 test), exact fp64 test and Möller–Trumbore only for the survivors.
// Inlined into the traversal (no scalar box buffers here, so nothing to protect
// from its register allocation; the call cost 2-3 % of the frame).
template <bool STATS>
__device__ __forceinline__ ScanOut scan_small_lane_f32_call(const DevScene *self, int pb, int pc,
                                                                      MT_RAY_PARAMS, MT_F32_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  MT_F32_FROM_PARAMS(f);
  const DevScene S = scan_ctx_self(self);
  const bool sx = __builtin_signbit(r.ix), sy = __builtin_signbit(r.iy), sz = __builtin_signbit(r.iz);
  ScanOut o{-1, 0.0, 0u};
  for (int k = 0; k < pc; k += 4) {
    if (STATS) o.bytes_v += 4u * 24u;
    float b[4][6];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int kk = (k + j < pc) ? (k + j) : (pc - 1);  // clamped: the load is always valid
      const float *bp = S.tri_aabb32 + (size_t)(pb + kk) * 6;
#pragma unroll
      for (int i = 0; i < 6; i++) b[j][i] = bp[i];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (k + j >= pc) break;
      if (!subtree_may_hit(b[j], f, sx, sy, sz)) continue;  // the same conservative test, per lane
      const double *bx = S.tri_aabb + (size_t)(pb + k + j) * 6;
      const double e[6] = {bx[0], bx[1], bx[2], bx[3], bx[4], bx[5]};
      if (STATS) o.bytes_v += 48u;
      if (!slab_pass_lane<false>(e, r)) continue;
      if (STATS) {
        o.mt_tests++;
        o.bytes_v += 72u;
      }
      double t;
      if (moller_trumbore(S.tri_vertex + (size_t)(pb + k + j) * 9, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, &t)) {
        if (!(o.best >= 0 && t > o.best_t)) {
          o.best = pb + k + j;
          o.best_t = t;
        }
      }
    }
  }
  return o;
}

template <int OCT, bool STATS, bool FUSED = false>
__device__ __attribute__((noinline)) ScanOut scan_filtered_call(const DevScene *self, int pb, int pc,
                                                                MT_RAY_PARAMS, MT_F32_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  MT_F32_FROM_PARAMS(f);
  const DevScene S = scan_ctx_self(self);
  LaneStats st;
  st.clear();
  ScanOut o{-1, 0.0, 0u};
  if constexpr (FUSED) scan_node_fused<OCT, STATS>(S, r, f, uniform_i32(pb), uniform_i32(pc), o.best, o.best_t, st);
  else scan_node_filtered<OCT, STATS>(S, r, f, uniform_i32(pb), uniform_i32(pc), o.best, o.best_t, st);
  o.mt_tests = st.v[ST_MT_TESTS];
  o.bytes_v = st.v[ST_BYTES_VECTOR];
  o.bytes_s = st.bytes_scalar;
  return o;
}

// Big node, ray-parallel, with block boxes: first the boxes of the stream
// blocks the node's list [pb, pb+pc) touches (64 at a time), then the filtered
// scan over every run of blocks that some lane may hit.  A ray that misses a
// block box (conservatively, in fp32) fails the reference's AABB pre-filter
// (primitive_triangle.cc:73-76) for each member: its exact slab interval on a
// member box lies inside the one on the union box (fp64 rounding is
// monotonic), so skipping the block changes nothing.  Runs are visited in list
// order and folded with the reference's rule (octtree.cc:186-194: a later hit
// wins unless it is strictly farther).
template <int OCT, bool STATS, bool FUSED = false>
__device__ __attribute__((noinline)) ScanOut scan_grouped_call(const DevScene *self, int pb, int pc,
                                                               MT_RAY_PARAMS, MT_F32_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  MT_F32_FROM_PARAMS(f);
  self = uniform_ptr(self);
  const DevScene S = scan_ctx_self(self);
  const MT_CONST float *gp = as_const(S.grp_aabb32);
  pb = uniform_i32(pb);
  pc = uniform_i32(pc);
  ScanOut o{-1, 0.0, 0u};
  LaneStats st;
  st.clear();
  const int b0 = pb / kGroupTris, nb = (pb + pc - 1) / kGroupTris - b0 + 1;
  for (int g0 = 0; g0 < nb; g0 += 64) {
    const int n = (nb - g0) < 64 ? (nb - g0) : 64;
#ifdef MT_PROF
    o.n_groups += (unsigned)n;
#endif
#ifdef MT_PROF
    const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
#endif
    unsigned long long live;
    if (FUSED && nb >= kSuperMin) {
      // two levels: the boxes of the runs of 8 blocks this chunk touches first, block boxes only inside the live runs
      // (a ray that misses a run's union box misses every block box in it: the same monotonicity argument)
      const int s0 = (b0 + g0) / kSuperBlocks, s1 = (b0 + g0 + n - 1) / kSuperBlocks;
      unsigned long long sl = group_live_mask<OCT>(as_const(S.sup_aabb32) + (size_t)s0 * 6, s1 - s0 + 1, f);
      if (STATS) st.bytes_scalar += 96u * (unsigned)((s1 - s0 + 4) / 4 + 1);
      live = 0ull;
      while (sl != 0ull) {
        const int si = s0 + __builtin_ctzll(sl);
        sl &= sl - 1ull;
        int first = si * kSuperBlocks, last = first + kSuperBlocks;
        if (first < b0 + g0) first = b0 + g0;
        if (last > b0 + g0 + n) last = b0 + g0 + n;
        live |= group_live_mask<OCT>(gp + (size_t)first * 6, last - first, f) << (first - (b0 + g0));
        if (STATS) st.bytes_scalar += 96u * 3u;
      }
    } else {
      live = group_live_mask<OCT>(gp + (size_t)(b0 + g0) * 6, n, f);
    }
    if (STATS && !(FUSED && nb >= kSuperMin)) st.bytes_scalar += 96u * (unsigned)((n + 3) / 4 + 1);  // block boxes, four per fetch
#ifdef MT_PROF
    const unsigned long long tg1 = __builtin_amdgcn_s_memtime();
    o.t_a += (unsigned)(tg1 - tg0);
#endif
    while (live != 0ull) {
      const int a = __builtin_ctzll(live);
      const unsigned long long rest = ~(live >> a);  // zero bit = end of the run
      const int len = rest != 0ull ? __builtin_ctzll(rest) : 64 - a;
      int first = (b0 + g0 + a) * kGroupTris;
      int last = first + len * kGroupTris;
      if (first < pb) first = pb;
      if (last > pb + pc) last = pb + pc;
      // in list order, best / best_t running through (octtree.cc:186-194)
      if constexpr (FUSED) scan_node_fused<OCT, STATS>(S, r, f, first, last - first, o.best, o.best_t, st);
      else scan_node_filtered<OCT, STATS>(S, r, f, first, last - first, o.best, o.best_t, st);
#ifdef MT_PROF
      o.n_ranges++;
      o.n_live += (unsigned)len;
      o.n_range_tris += (unsigned)(last - first);
#endif
      live = (a + len >= 64) ? 0ull : (live >> (a + len)) << (a + len);
    }
#ifdef MT_PROF
    o.t_b += (unsigned)(__builtin_amdgcn_s_memtime() - tg1);
#endif
  }
  o.mt_tests = st.v[ST_MT_TESTS];
  o.bytes_v = st.v[ST_BYTES_VECTOR];
  o.bytes_s = st.bytes_scalar;
  return o;
}


// ---- long list through its spatially sorted copy (DevScene::ll_*; hit-set walk, NaN-free rays only) -----------------
// The walk's scan_long marks per-lane candidates among a super's 64 entries (super boxes -> block boxes of the live
// supers -> the fp32 boxes of the live blocks' entries, all staged through LDS); here every lane resolves its
// candidates (exact fp64 box, Moeller-Trumbore) in ANY order and keeps the minimum under (distance, -stream index):
// the reference's fold (octtree.cc:177-196) for distances that cannot be NaN.
template <bool STATS>
__device__ __forceinline__ void resolve_sorted(const int32_t *ll_tri, const double *ll_exact, const RayRegs &r, int base,
                                               unsigned long long &cand, int &best, double &best_t, LaneStats &st) {
  const MT_CONST int32_t *lt = as_const(ll_tri);
  for (int guard = 0; guard < 64 && __ballot(cand != 0ull) != 0ull; guard++) {
    if (cand != 0ull) {
      const int pos = base + __builtin_ctzll(cand);
      cand &= cand - 1ull;
      // exact box and vertices come from the sorted copy too (15 doubles per entry): one round trip per candidate,
      // the triangle's index travelling with them (it is only needed when the candidate is a hit; padding entries
      // hold inverted fp32 boxes and are never marked, their exact entry is all zeros and fails the determinant test)
      const int t = lt[pos];
      const __attribute__((address_space(1))) double *ep = (const __attribute__((address_space(1))) double *)ll_exact + (size_t)pos * 15;
      const double e[6] = {ep[0], ep[1], ep[2], ep[3], ep[4], ep[5]};
      const double v[9] = {ep[6], ep[7], ep[8], ep[9], ep[10], ep[11], ep[12], ep[13], ep[14]};
      if (STATS) st.v[ST_BYTES_VECTOR] += 124u;
      if (slab_pass_lane<false>(e, r)) {
        if (STATS) st.v[ST_MT_TESTS]++;
        double tt;
        if (moller_trumbore_flat(v, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, &tt)) {
          if (t >= 0 && (best < 0 || tt < best_t || (tt == best_t && t > best))) {
            best = t;
            best_t = tt;
          }
        }
      }
    }
  }
  cand = 0ull;
}


template <int MODE, int OCT, bool STATS>
__device__ __attribute__((noinline)) ScanOut scan_exact_call(const double *b64, const double *vtx, int pb,
                                                             int pc, MT_RAY_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  const DevScene S = scan_ctx(nullptr, b64, vtx);
  LaneStats st;
  st.clear();
  ScanOut o{-1, 0.0, 0u};
  scan_node_prims<MODE, OCT, STATS>(S, r, uniform_i32(pb), uniform_i32(pc), o.best, o.best_t, st);
  o.mt_tests = st.v[ST_MT_TESTS];
  o.bytes_v = st.v[ST_BYTES_VECTOR];
  o.bytes_s = st.bytes_scalar;
  return o;
}

template <bool EX, bool STATS>
__device__ __attribute__((noinline)) ScanOut scan_transposed_call(const double *b64, const double *vtx,
                                                                  int pb, int pc, int lane,
                                                                  unsigned inmask_lo, unsigned inmask_hi,
                                                                  MT_RAY_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  const DevScene S = scan_ctx(nullptr, b64, vtx);
  const unsigned long long inmask =
      ((unsigned long long)(unsigned)uniform_i32((int)inmask_hi) << 32) | (unsigned)uniform_i32((int)inmask_lo);
  LaneStats st;
  st.clear();
  ScanOut o{-1, 0.0, 0u};
  scan_node_transposed<EX, STATS>(S, r, lane, inmask, uniform_i32(pb), uniform_i32(pc), o.best, o.best_t, st);
  o.mt_tests = st.v[ST_MT_TESTS];
  o.bytes_v = st.v[ST_BYTES_VECTOR];
  o.bytes_s = st.bytes_scalar;
  return o;
}

template <bool EX, bool STATS>
__device__ __attribute__((noinline)) ScanOut scan_transposed_blocks_call(const DevScene *self, int pb, int pc,
                                                                         bool in, MT_RAY_PARAMS, MT_F32_PARAMS) {
  MT_RAY_FROM_PARAMS(r);
  MT_F32_FROM_PARAMS(f);
  const DevScene S = scan_ctx_self(self);
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const unsigned long long inmask = __builtin_amdgcn_ballot_w64(in);
  LaneStats st;
  st.clear();
  ScanOut o{-1, 0.0, 0u};
#ifdef MT_PROF
  unsigned tp[3] = {0, 0, 0};
  scan_node_transposed_blocks<EX, STATS>(S, r, f, lane, inmask, uniform_i32(pb), uniform_i32(pc), o.best, o.best_t, st, tp);
  o.t_c = tp[0]; o.t_a = tp[1]; o.t_b = tp[2];
#else
  scan_node_transposed_blocks<EX, STATS>(S, r, f, lane, inmask, uniform_i32(pb), uniform_i32(pc), o.best, o.best_t, st);
#endif
  o.mt_tests = st.v[ST_MT_TESTS];
  o.bytes_v = st.v[ST_BYTES_VECTOR];
  o.bytes_s = st.bytes_scalar;
  return o;
}

template <bool STATS, bool FUSED = false>
__device__ __forceinline__ ScanOut scan_filtered_dispatch(const DevScene &S, int oct, const float *g32,
                                                          int pb, int pc, const RayRegs &r,
                                                          const Filter32 &f) {
  switch (oct) {
    case 0: return (g32 ? scan_grouped_call<0, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<0, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 1: return (g32 ? scan_grouped_call<1, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<1, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 2: return (g32 ? scan_grouped_call<2, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<2, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 3: return (g32 ? scan_grouped_call<3, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<3, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 4: return (g32 ? scan_grouped_call<4, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<4, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 5: return (g32 ? scan_grouped_call<5, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<5, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 6: return (g32 ? scan_grouped_call<6, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<6, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    case 7: return (g32 ? scan_grouped_call<7, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<7, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
    default: return (g32 ? scan_grouped_call<8, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f))
                        : scan_filtered_call<8, STATS, FUSED>(S.self, pb, pc, MT_RAY_ARGS(r), MT_F32_ARGS(f)));
  }
}

template <bool STATS>
__device__ __forceinline__ ScanOut scan_octant_dispatch(const DevScene &S, int oct, int pb, int pc,
                                                        const RayRegs &r) {
  switch (oct) {
    case 0: return scan_exact_call<2, 0, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    case 1: return scan_exact_call<2, 1, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    case 2: return scan_exact_call<2, 2, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    case 3: return scan_exact_call<2, 3, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    case 4: return scan_exact_call<2, 4, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    case 5: return scan_exact_call<2, 5, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    case 6: return scan_exact_call<2, 6, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
    default: return scan_exact_call<2, 7, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
  }
}

// Asynchronous copy global -> LDS, 16 bytes per active lane: lane i's bytes land at lds_base + 16 i
// (wave-uniform base in M0).  Written as asm because hipcc 7.2 handles the builtin inconsistently: it
// either waits vmcnt(0) before the NEXT LDS read of any address (which serialises the copy with the
// work it was meant to overlap) or, across a loop back-edge, not at all.  The waits are explicit at the
// readers (s_waitcnt vmcnt).  M0 is declared clobbered: LLVM uses it for its own LDS-DMA / indirect-index / lane-select
// sequences and merges or hoists its initialisations across blocks -- an undeclared write could be miscompiled.
__device__ __forceinline__ void lds_dma16(const char *src, unsigned lds_base) {
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_base);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(base), "v"(src) : "memory", "m0");
}

// One closest-hit query per lane.  Must be called by all 64 lanes of the wave
// (want = false for lanes without a ray).  out_prim = stream index or -1.
// Result of one traversal, returned by value (in registers).
struct TraceOut {
  int status;   // DEV_OK or DEV_ERR_*
  int prim;     // stream index of the closest hit, -1 none
  double t;
  unsigned box_tests, node_visits, tri_tests, mt_tests, bytes_vector;  // per lane
  unsigned wave_node_steps, wave_tri_steps, bytes_scalar;              // wave-uniform

};

// It takes ONE pointer to the scene description in device memory (read with scalar loads) and returns its result in
// registers.  Must be called by all 64 lanes (want = false for lanes without a ray).
// Inlined into every engine since round 4 (-DMT_TRACE_CALL builds it as a function again): as a call it had its own
// register allocation, but every pass saved and restored 39 callee-saved registers through scratch -- 1.7 GB written
// and read per 1080p frame, nearly all of the frame kernel's memory-side traffic -- for a frame 1 % SLOWER (4.91
// against 4.86 ms warm, 4.96 against 4.90 panning).
#ifdef MT_TRACE_CALL
#define MT_TRACE_ATTR __attribute__((noinline))
#else
#define MT_TRACE_ATTR __forceinline__
#endif
template <bool STATS, int DEEP = 0>
__device__ MT_TRACE_ATTR TraceOut trace_wave(const DevScene *scene, unsigned stack_base, int lane,
                                                         bool want_all, double ox, double oy, double oz,
                                                         double dx, double dy, double dz) {
  constexpr bool WIDE = DEEP >= 2;  // DEEP 2: child bytes for 24 levels (mt_device.h, deep_layout)
  const MT_CONST DevScene *G = as_const(uniform_ptr(scene));
  DevScene S;
  S.nodes = G->nodes;
  S.tri_aabb = G->tri_aabb;
  S.tri_aabb32 = G->tri_aabb32;
  S.grp_aabb32 = G->grp_aabb32;
  S.sub_aabb32 = G->sub_aabb32;
  S.hs_rec = G->hs_rec;
  S.sl_box32 = G->sl_box32;
  S.ll_box_q = G->ll_box_q;
  S.ll_grp_q = G->ll_grp_q;
  S.ll_sup_q = G->ll_sup_q;
  S.ll_tri = G->ll_tri;
  S.ll_exact = G->ll_exact;
  S.self = uniform_ptr(scene);
  S.tri_vertex = G->tri_vertex;
  S.bmax[0] = G->bmax[0]; S.bmax[1] = G->bmax[1]; S.bmax[2] = G->bmax[2];
  S.n_tris = G->n_tris;
  S.n_nodes = G->n_nodes;
  S.tree_depth = G->tree_depth;
  S.force_mode = G->force_mode;
  S.scene_regular = G->scene_regular;
  S.pack_shift = G->pack_shift;
  S.hb = nullptr;
  S.prof = G->prof;
  LaneStats st;
  st.clear();
  int out_prim;
  double out_t;
  MT_PROF_DECL;
  MT_PROF_BEGIN(prof_t0);
  MT_PROF_COUNT(PROF_N_TRACES, 1);
  WaveStack stk;
  stk.base = (unsigned)uniform_i32((int)stack_base);
  stk.depth = S.tree_depth;
  stk.pack_shift = S.pack_shift;
  // the ordered descent's stack: in LDS, or (DEEP) in this wave's area of global memory
  char *deep_area = nullptr;
  if constexpr (DEEP) {
    const unsigned wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    deep_area = uniform_ptr(G->deep_base + (size_t)wave_global * G->deep_stride);
  }
  using BtPtr = typename std::conditional<(DEEP != 0), double *, MT_LDS double *>::type;
  using IPtr = typename std::conditional<(DEEP != 0), int *, MT_LDS int *>::type;
  using UPtr = typename std::conditional<(DEEP != 0), unsigned *, MT_LDS unsigned *>::type;
  BtPtr stk_bt;
  IPtr stk_fc, stk_bp;
  UPtr stk_ord;
  if constexpr (DEEP) {
    stk_bt = (double *)deep_area;
    stk_fc = (int *)(deep_area + (size_t)stk.depth * 64 * 8);
    stk_bp = (int *)(deep_area + (size_t)stk.depth * 64 * 12);
    stk_ord = (unsigned *)(deep_area + (size_t)stk.depth * 64 * 16);
  } else {
    stk_bt = stk.bt();
    stk_fc = stk.fc();
    stk_bp = stk.bp();
    stk_ord = stk.ord();
  }
  const unsigned frames_end = stk.base + (unsigned)wave_frames_bytes(stk.depth, stk.pack_shift != 0, DEEP != 0);
  unsigned cntr[5] = {0u, 0u, 0u, 0u, 0u};  // per-lane work counters of this traversal (STATS only)
  (void)cntr;
  const int pack_shift = stk.pack_shift;  // wave-uniform
  if (STATS) {
    MT_CNT_SET(0, 0u); MT_CNT_SET(1, 0u); MT_CNT_SET(2, 0u); MT_CNT_SET(3, 0u);
    MT_CNT_SET(4, 0u);
  }
  RayRegs r;
  r.ox = ox; r.oy = oy; r.oz = oz;
  r.dx = dx; r.dy = dy; r.dz = dz;
  r.ix = 1.0 / dx; r.iy = 1.0 / dy; r.iz = 1.0 / dz;
  out_prim = -1;
  out_t = 0.0;

  // A lane is "regular" when no slab product can be NaN: finite origin and a
  // finite, non-zero reciprocal direction on every axis (see header comment).
  const bool fin = __builtin_isfinite(ox) && __builtin_isfinite(oy) && __builtin_isfinite(oz) &&
                   __builtin_isfinite(r.ix) && __builtin_isfinite(r.iy) && __builtin_isfinite(r.iz) &&
                   r.ix != 0.0 && r.iy != 0.0 && r.iz != 0.0;
  // A wave that holds both kinds is traversed twice, the regular lanes first:
  // one ray with a zero direction component (a pixel row level with the
  // camera, say) must not push the other 63 onto the exact path, which has
  // neither the fp32 filter nor the block / subtree boxes.
  const int rounds = (S.scene_regular != 0 && S.force_mode != 1 && __ballot(want_all && !fin) != 0ull &&
                      __ballot(want_all && fin) != 0ull) ? 2 : 1;
  int status = DEV_OK;
  for (int round = 0; round < rounds; round++) {
  const bool want = rounds == 1 ? want_all : (want_all && (fin == (round == 0)));
  const bool all_regular = (S.scene_regular != 0) && (S.force_mode != 1) &&
                           (__ballot(want && !fin) == 0ull);
  Filter32 f32;
  const bool f32_ok = make_filter32(S, r, f32);
  const bool use_filter = all_regular && (S.force_mode != 4) && (S.force_mode != 2) &&
                          (__ballot(want && !f32_ok) == 0ull);

  // Subtree culling (see subtree_may_hit); off in the modes without the fp32
  // filter and in mode 7, where the counters then match the reference's
  // un-pruned traversal exactly.
  const bool cull = use_filter && (S.force_mode != 7);
  // The same two kinds of boxes for irregular rays with one zero direction
  // component (degenerate_axis): automatic mode only.
  const bool irr_boxes = !all_regular && (S.scene_regular != 0) && (S.force_mode == 0) &&
                         S.bmax[0] <= 0x1p400 && S.bmax[1] <= 0x1p400 && S.bmax[2] <= 0x1p400;

  const MT_CONST NodeRec *nodes = as_const(S.nodes);
  int cur = -1;
  if (want) {
    // root box test, octtree.cc:35-37 (exact form; once per ray)
    const MT_CONST NodeRec *R = nodes;
    if (STATS) {
      MT_CNT_SET(0, 1u);
      st.bytes_scalar += 96u;  // the root record
    }
    const double t1 = (R->lo[0] - ox) * r.ix, t2 = (R->hi[0] - ox) * r.ix;
    const double t3 = (R->lo[1] - oy) * r.iy, t4 = (R->hi[1] - oy) * r.iy;
    const double t5 = (R->lo[2] - oz) * r.iz, t6 = (R->hi[2] - oz) * r.iz;
    const double tmax = std_min3(std_max(t1, t2), std_max(t3, t4), std_max(t5, t6));
    const double tmin = std_max3(std_min(t1, t2), std_min(t3, t4), std_min(t5, t6));
    if (!(tmax < 0.0) && !(tmin > tmax)) cur = 0;
  }
  int depth = 0;
  const int sxl = __builtin_signbit(r.ix) ? 1 : 0;
  const int syl = __builtin_signbit(r.iy) ? 1 : 0;
  const int szl = __builtin_signbit(r.iz) ? 1 : 0;

  // Record of the node a lane has to process next (valid while cur >= 0).
  int cur_fc = 0, cur_pb = 0, cur_pc = 0;
  auto load_record = [&](int node) {
    const int *q = (const int *)(S.nodes + node) + 18;  // NodeRec: first_child, prim_begin, prim_count, (mask), level
    cur_fc = q[0];
    cur_pb = q[1];
    cur_pc = q[2];
  };
  if (cur >= 0) load_record(cur);

  // The tail of PrimitiveIntersectRay for one node (octtree.cc:199-256): order
  // the children the ray enters, then descend into the next one or hand the
  // node's result up, as often as that completes parents.  Ends with the lane's
  // next node in `cur` (+ its record) or with cur = -1 and the final result.
  auto finish_node = [&](int fc, unsigned ordw, int best, double best_t) {
    int my_fc = fc;
    unsigned pos = 0;
    for (int guard = 0;; guard++) {
      if (guard > S.tree_depth + 1) {  // cannot happen: one pop per level at most
        cur = -2;
        break;
      }
      const unsigned n_ord = (ordw >> 24) & 15u;
      if (pos < n_ord) {
        const int child = my_fc + (int)((ordw >> (3 * pos)) & 7u);
        pos++;
        if (depth >= S.tree_depth) {  // cannot happen: stack sized for the validated depth
          cur = -2;
          break;
        }
        const int at = depth * 64 + lane;
        if (pack_shift) {
          stk_fc[at] = (int)(((unsigned)my_fc << pack_shift) | (unsigned)(best + 1));
        } else {
          stk_fc[at] = my_fc;
          stk_bp[at] = best;
        }
        stk_bt[at] = best_t;
        stk_ord[at] = (ordw & 0x0fffffffu) | (pos << 28);
        depth++;
        cur = child;
        load_record(child);
        if (STATS) MT_CNT_ADD(4, 16u);
        break;
      }
      if (depth == 0) {
        out_prim = best;
        out_t = best_t;
        cur = -1;
        break;
      }
      depth--;
      const int at = depth * 64 + lane;
      const unsigned po = stk_ord[at];
      int pbp;
      double pbt = stk_bt[at];
      if (pack_shift) {
        const unsigned fb = (unsigned)stk_fc[at];
        my_fc = (int)(fb >> pack_shift);
        pbp = (int)(fb & ((1u << pack_shift) - 1u)) - 1;
      } else {
        pbp = stk_bp[at];
        my_fc = stk_fc[at];
      }
      pos = po >> 28;
      ordw = po & 0x0fffffffu;
      if (best >= 0 && !(pbp >= 0 && best_t > pbt)) {  // :233-246 take it and break
        pbp = best;
        pbt = best_t;
        pos = (po >> 24) & 15u;
      }
      best = pbp;
      best_t = pbt;
    }
  };

  // Nodes with fewer than kBigNode triangles are scanned lane-parallel, larger
  // ones wave-uniformly.  (Every leaf is small: the reference splits at 16.)
  const bool lane_phase = (S.force_mode != 5);
  // Each wave step retires at least one (lane, node) visit and a lane visits a
  // node at most once per query, so 64 * n_nodes steps can never be exceeded.
  const long long step_bound = 64ll * (long long)S.n_nodes + 64;
  long long steps = 0;
#ifdef MT_DIAG
  unsigned diag_a_trips = 0, diag_transposed = 0;
#endif
  // ---- hit-set traversal (regular rays, automatic mode) ---------------------
  // What PrimitiveIntersectRay returns for a node is a function of three things
  // only (octtree.cc:169-257): the best hit of the node's own list; for every
  // child, whether the ray enters its box and at what distance (the sort key);
  // and what the same function returns for the children that hold a hit --
  // children without one `continue` and leave no trace.  The loop over the
  // sorted children takes the FIRST one whose hit is not farther than the own
  // one and stops: with NaN-free keys and a stable sort that is the entered,
  // acceptable child with the smallest (entry distance, index).  So the order
  // in which the children are LOOKED AT is free: here the whole wave walks the
  // tree depth-first in index order, every node once for all lanes whose
  // filter does not rule its subtree out (records and boxes staged in LDS by
  // LDS-DMA, wave-uniform control flow, no per-lane stack walk, no sort), each child's result is offered to its
  // parent's frame when the wave comes back from it (exact entry test of that
  // one child, comparison with the frame's best candidate so far), and a frame
  // is closed when its last child is done.  Nodes the reference would not have
  // reached (behind its early exit) may be looked at; that changes the work,
  // not the result.
  // (tame: magnitudes for which no Moeller-Trumbore distance can be NaN or infinite -- |d| <= 2^100, |o| and every
  // coordinate of the scene <= 2^200: all intermediate products stay below 2^700 -- so that a list's hits may be folded
  // in any order, scan_long / resolve_sorted)
  const bool tame = __builtin_fabs(dx) <= 0x1p100 && __builtin_fabs(dy) <= 0x1p100 && __builtin_fabs(dz) <= 0x1p100 &&
                    __builtin_fabs(ox) <= 0x1p200 && __builtin_fabs(oy) <= 0x1p200 && __builtin_fabs(oz) <= 0x1p200;
  bool hs_done = false;
  if (cull && S.force_mode == 0 && S.tree_depth <= (WIDE ? kHsMaxDepthDeep : kHsMaxDepth) && S.n_tris < (1 << 28) &&
      S.bmax[0] <= 0x1p200 && S.bmax[1] <= 0x1p200 && S.bmax[2] <= 0x1p200 && __ballot(want && !tame) == 0ull) {
    hs_done = true;
    const int L = S.tree_depth > 1 ? S.tree_depth - 1 : 0;  // levels that can hold a node with children
    const int Lf = (DEEP && L > kHsLdsLevels) ? kHsLdsLevels : L;  // ... whose frames are in LDS (DEEP: the others in deep_area)
    MT_LDS double *const h_own_t = (MT_LDS double *)(uintptr_t)stk.base;  // [Lf][64] own list's best distance
    MT_LDS double *const h_win_t = h_own_t + Lf * 64;                      // [Lf][64] best child candidate so far
    MT_LDS int *const h_own_p = (MT_LDS int *)(h_win_t + Lf * 64);         // [Lf][64] own best triangle, -1 none
    MT_LDS int *const h_win_p = h_own_p + Lf * 64;                         // [Lf][64] candidate triangle | child slot << 28, -1 none
    MT_LDS int *const h_node = h_win_p + Lf * 64;                          // [L][2] (layout only; the values live in lane_node / lane_fc)
    // the frames of the levels from kHsLdsLevels on (DEEP): behind the descent's stack in the wave's global area
    double *g_own_t = nullptr, *g_win_t = nullptr;
    int *g_own_p = nullptr, *g_win_p = nullptr;
    if constexpr (DEEP) {
      const int Ld = L > kHsLdsLevels ? L - kHsLdsLevels : 0;
      g_own_t = (double *)(deep_area + (size_t)stk.depth * 64 * 20);
      g_win_t = g_own_t + Ld * 64;
      g_own_p = (int *)(g_win_t + Ld * 64);
      g_win_p = g_own_p + Ld * 64;
    }
    (void)g_own_t; (void)g_win_t; (void)g_own_p; (void)g_win_p;
#define HS_LD(aL, aG, l) ((DEEP && (l) >= kHsLdsLevels) ? (aG)[((l) - kHsLdsLevels) * 64 + lane] : (aL)[(l) * 64 + lane])
#define HS_ST(aL, aG, l, v) do { if (DEEP && (l) >= kHsLdsLevels) (aG)[((l) - kHsLdsLevels) * 64 + lane] = (v); else (aL)[(l) * 64 + lane] = (v); } while (0)
    // wave-uniform per level: the frame's node and its first child, level l in LANE l of a register pair
    // (v_readlane / v_writelane with the level as lane select: no LDS round trip in the walk's bookkeeping)
    int lane_node = 0, lane_fc = 0;
    const unsigned stage = ((unsigned)(uintptr_t)(h_node + L * 2) + 15u) & ~15u;  // the staged HsRec (room for two)
    MT_LDS double *const h_planes = (MT_LDS double *)(uintptr_t)(stage + 2u * (unsigned)sizeof(HsRec));  // [L][10] wave-uniform
    const unsigned tstage = frames_end;                                    // a short list's staged quads of fp32 boxes (<= 1 152 B)
    // The arrays the walk reads, as values of their own: read out of the scene structure where they are used, a
    // pointer comes as part of a sixteen-register tuple that is spilled and reloaded whole.
    const char *hs_bytes = (const char *)S.hs_rec;
    const char *sl_bytes = (const char *)S.sl_box32;
    const char *supq = (const char *)S.ll_sup_q, *grpq = (const char *)S.ll_grp_q, *boxq = (const char *)S.ll_box_q;
    const int32_t *ll_tri_ = S.ll_tri;
    const double *ll_exact_ = S.ll_exact;
    const double *tri_aabb_ = S.tri_aabb, *tri_vertex_ = S.tri_vertex;
    asm volatile("" : "+s"(hs_bytes), "+s"(sl_bytes), "+s"(supq), "+s"(grpq), "+s"(boxq), "+s"(ll_tri_), "+s"(ll_exact_),
                      "+s"(tri_aabb_), "+s"(tri_vertex_));
#ifdef MT_PROF
    const bool tl_on = S.prof != nullptr && lane == 0 && stk.base == 0u && __builtin_amdgcn_workgroup_id_x() == 0;
    unsigned long long tl_base = 0ull;  // 512 slots per traversal, reserved with ONE atomic; the stamps are plain stores
    unsigned tl_i = 0u;
    if (tl_on) tl_base = atomicAdd(S.prof + PROF_COUNT, 512ull);
    MT_TL(1);  // walk begins
#endif
    int lev = -1;                        // frame on top of the stack, -1 none
    // wave-uniform; byte l of the pair: children of frame l still to look at
    unsigned long long pendA = 0ull, pendB = 0ull, pendC = 0ull;  // (pendC / wantC: levels 16 .. 23, the WIDE instantiations only)
    // per lane; byte l of the pair: children of frame l this lane's filter lets through
    unsigned long long wantA = 0ull, wantB = 0ull, wantC = 0ull;
    auto get8 = [&](unsigned long long a, unsigned long long b, unsigned long long c, int l) -> unsigned {
      if constexpr (WIDE) return (unsigned)((l < 8 ? a : (l < 16 ? b : c)) >> (8 * (l & 7))) & 0xffu;
      else return (unsigned)((l < 8 ? a : b) >> (8 * (l & 7))) & 0xffu;
    };
    auto set8 = [&](unsigned long long &a, unsigned long long &b, unsigned long long &c, int l, unsigned v) {
      const int sh_ = 8 * (l & 7);
      if (l < 8) a = (a & ~(0xffull << sh_)) | ((unsigned long long)v << sh_);
      else if (!WIDE || l < 16) b = (b & ~(0xffull << sh_)) | ((unsigned long long)v << sh_);
      else c = (c & ~(0xffull << sh_)) | ((unsigned long long)v << sh_);
    };
    int node = 0;
    unsigned long long m = __ballot(cur == 0);
    int ret_p = -1;
    double ret_t = 0.0;
    int slot = 0;  // child slot the result in ret_* comes from
    bool entering = true;
    // A node's record is copied to LDS when the node is entered (fetching it one node ahead was measured 2 %
    // slower: the other wave of the SIMD covers the latency, and finding the next node is not free).
    auto hs_fetch = [&](int nd) {
      if (lane < kHsRecLanes) lds_dma16(hs_bytes + (size_t)nd * sizeof(HsRec) + (size_t)lane * 16, stage);
    };
    // Children are looked at near to far for the octant of the wave's first ray (child index
    // bits: 0 = x high, 1 = z high, 2 = y high): a candidate found early lets the lanes drop the
    // children that sort behind it, which is the reference's early exit (octtree.cc:246).
    unsigned flip = 0u;
    if (m != 0ull) {
      // (the octant of the MAJORITY of the rays instead: 1 % slower)
      const int fl = __builtin_ctzll(m);
      flip = (unsigned)(__builtin_amdgcn_readlane(sxl, fl) | (__builtin_amdgcn_readlane(szl, fl) << 1) |
                        (__builtin_amdgcn_readlane(syl, fl) << 2));
    }
    auto pick = [&](unsigned td) -> int {  // td != 0: the child to look at next
      unsigned t = td;
      if (flip & 1u) t = ((t & 0x55u) << 1) | ((t >> 1) & 0x55u);
      if (flip & 2u) t = ((t & 0x33u) << 2) | ((t >> 2) & 0x33u);
      if (flip & 4u) t = ((t & 0x0fu) << 4) | ((t >> 4) & 0x0fu);
      return (int)((unsigned)__builtin_ctz(t) ^ flip);
    };
    // copies `bytes` (<= 2048, wave-uniform) from src to LDS: one or two LDS-DMA instructions; returns how many
    auto dma_range = [&](const char *src, unsigned lds, int bytes) -> int {
      if (lane * 16 < bytes) lds_dma16(src + (size_t)lane * 16, lds);
      if (bytes > 1024) {
        if (lane * 16 < bytes - 1024) lds_dma16(src + 1024 + (size_t)lane * 16, lds + 1024u);
        return 2;
      }
      return 1;
    };
    // waits until at most n of the copies / loads issued so far are still in flight (they complete in order)
    auto wait_vm = [&](int n) {
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      }
    };
    // fp32 verdicts on a staged short list (its quads at lds in the layout of DevScene::sl_box32, n boxes): bit k = this
    // lane's ray may hit box k.  Per quad and axis the lane reads [near x 4][far x 4] as 32 consecutive bytes.
    auto list_bits = [&](unsigned lds, int n) -> unsigned long long {
      typedef float f4v_ __attribute__((ext_vector_type(4)));
      const unsigned bx = lds + (sxl != 0 ? 16u : 0u), by = lds + 48u + (syl != 0 ? 16u : 0u), bz = lds + 96u + (szl != 0 ? 16u : 0u);
      unsigned cand = 0u;  // (kHsShortList <= 32 positions)
      for (int k = 0; k < n; k += 4) {  // boxes past the list are padding (inverted) and masked off below
        const unsigned q = (unsigned)(k >> 2) * (unsigned)(kSlQuadFloats * 4);
        const MT_LDS f4v_ *px = (const MT_LDS f4v_ *)(uintptr_t)(bx + q);
        const MT_LDS f4v_ *py = (const MT_LDS f4v_ *)(uintptr_t)(by + q);
        const MT_LDS f4v_ *pz = (const MT_LDS f4v_ *)(uintptr_t)(bz + q);
        const f4v_ nx = px[0], fx = px[1], ny = py[0], fy = py[1], nz = pz[0], fz = pz[1];
        // the quad's four verdicts as one nibble (select between inline constants), shifted into place once
        const unsigned q4 = (near_far_may_hit(nx.x, fx.x, ny.x, fy.x, nz.x, fz.x, f32) ? 1u : 0u) |
                            (near_far_may_hit(nx.y, fx.y, ny.y, fy.y, nz.y, fz.y, f32) ? 2u : 0u) |
                            (near_far_may_hit(nx.z, fx.z, ny.z, fy.z, nz.z, fz.z, f32) ? 4u : 0u) |
                            (near_far_may_hit(nx.w, fx.w, ny.w, fy.w, nz.w, fz.w, f32) ? 8u : 0u);
        cand |= q4 << k;
      }
      return (unsigned long long)(cand & (n >= 32 ? ~0u : ((1u << n) - 1u)));
    };
    // fp32 verdicts on nq staged quads (layout of DevScene::sl_box32) at lds: nibble q of the result = this lane's ray
    // may hit boxes 4 q .. 4 q + 3; bit 4 q + j of `any` = some lane of amask may hit box 4 q + j (read off the
    // compares' lane masks)
    auto quads_verdicts = [&](unsigned lds, int nq, unsigned long long amask, unsigned &any) -> unsigned {
      typedef float f4v_ __attribute__((ext_vector_type(4)));
      const unsigned bx = lds + (sxl != 0 ? 16u : 0u), by = lds + 48u + (syl != 0 ? 16u : 0u), bz = lds + 96u + (szl != 0 ? 16u : 0u);
      unsigned w = 0u;
      for (int q = 0; q < nq; q++) {
        const unsigned o = (unsigned)q * (unsigned)(kSlQuadFloats * 4);
        const MT_LDS f4v_ *px = (const MT_LDS f4v_ *)(uintptr_t)(bx + o);
        const MT_LDS f4v_ *py = (const MT_LDS f4v_ *)(uintptr_t)(by + o);
        const MT_LDS f4v_ *pz = (const MT_LDS f4v_ *)(uintptr_t)(bz + o);
        const f4v_ nx = px[0], fx = px[1], ny = py[0], fy = py[1], nz = pz[0], fz = pz[1];
        const bool p0 = near_far_may_hit(nx.x, fx.x, ny.x, fy.x, nz.x, fz.x, f32);
        const bool p1 = near_far_may_hit(nx.y, fx.y, ny.y, fy.y, nz.y, fz.y, f32);
        const bool p2 = near_far_may_hit(nx.z, fx.z, ny.z, fy.z, nz.z, fz.z, f32);
        const bool p3 = near_far_may_hit(nx.w, fx.w, ny.w, fy.w, nz.w, fz.w, f32);
        const unsigned q4 = (p0 ? 1u : 0u) | (p1 ? 2u : 0u) | (p2 ? 4u : 0u) | (p3 ? 8u : 0u);
        w |= q4 << (4 * q);
        unsigned a4 = 0u;
        if ((__ballot(p0) & amask) != 0ull) a4 |= 1u;
        if ((__ballot(p1) & amask) != 0ull) a4 |= 2u;
        if ((__ballot(p2) & amask) != 0ull) a4 |= 4u;
        if ((__ballot(p3) & amask) != 0ull) a4 |= 8u;
        any |= a4 << (4 * q);
      }
      return w;
    };
    // Own list longer than kHsShortList through its spatially sorted copy (DevScene::ll_*), level by level through LDS:
    // the super boxes of up to kLlRound supers (1 792 entries) by one copy; the block quads of up to six live supers GATHERED by one copy
    // (lane 9 g + r reads piece r of the g-th live super's quad); the entry quads of two live blocks per copy; every
    // lane marks its candidates of a super's 64 entries and resolves them in any order (resolve_sorted).  Which
    // supers / blocks are live is wave-uniform (some lane of `act` may hit the box).
    // (a list of up to kLlDirect supers skips the super boxes: the block quads of all its supers are one copy)
    // (n = the list's length: (n + 63) / 64 supers hold entries, the copy is padded to a multiple of kLlPad)
    auto long_first_copy = [&](int lb, int n) {
      const unsigned qb = (unsigned)(kSlQuadFloats * 4);
      const int n_sup = (n + 63) >> 6;
      if (n_sup <= kLlDirect) dma_range(grpq + (size_t)(lb >> 6) * qb, tstage, n_sup * (int)qb);
      else dma_range(supq + (size_t)(lb >> 8) * qb, tstage, (n_sup >= kLlRound ? kLlRound / 4 : ((n_sup + 3) >> 2)) * (int)qb);
    };
    auto scan_long = [&](int lb, int n, bool act, int &b_, double &bt_, LaneStats &ls) {
      const unsigned long long amask = __ballot(act);
      const int n_sup = (n + 63) >> 6;
      const unsigned g9 = (unsigned)lane / 9u, r9 = (unsigned)lane - g9 * 9u;
      const unsigned qb = (unsigned)(kSlQuadFloats * 4);  // bytes of a quad
      const bool direct = n_sup <= kLlDirect;
      for (int s0 = 0; s0 < n_sup; s0 += kLlRound) {
        unsigned sup_any = 0u;
        if (direct) {
          sup_any = (1u << n_sup) - 1u;  // (all of them "live": their block quads are already on their way)
        } else {
          const int nq = (n_sup - s0) >= kLlRound ? kLlRound / 4 : ((n_sup - s0 + 3) >> 2);
          if (s0 != 0) dma_range(supq + (size_t)((lb >> 8) + (s0 >> 2)) * qb, tstage, nq * (int)qb);  // (round 0: long_first_copy)
          wait_vm(0);
          (void)quads_verdicts(tstage, nq, amask, sup_any);
          sup_any = (unsigned)uniform_i32((int)sup_any);
          if (STATS) ls.bytes_scalar += (unsigned)nq * qb;
        }
        while (sup_any != 0u) {
          // up to six live supers of this round: chunk = their bits, in ascending order
          unsigned chunk = 0u;
          int cnt = 0, mine = -1;
          while (sup_any != 0u && cnt < 6) {
            const int sj = __builtin_ctz(sup_any);
            sup_any &= sup_any - 1u;
            chunk |= 1u << sj;
            if ((int)g9 == cnt) mine = sj;
            cnt++;
          }
          if (!direct && mine >= 0) lds_dma16(grpq + (size_t)((lb >> 6) + s0 + mine) * qb + (size_t)r9 * 16, tstage);
          wait_vm(0);
          unsigned blk_any = 0u;
          (void)quads_verdicts(tstage, cnt, amask, blk_any);
          blk_any = (unsigned)uniform_i32((int)blk_any);
          if (STATS) ls.bytes_scalar += (unsigned)cnt * qb;
          for (int g = 0; chunk != 0u; g++) {
            const int sj = __builtin_ctz(chunk);
            chunk &= chunk - 1u;
            unsigned ba = (blk_any >> (4 * g)) & 0xfu;
            if (ba == 0u) continue;
            const int base = lb + (s0 + sj) * 64;
            unsigned long long cand = 0ull;
            // the live blocks' entries, four quads each: the next block's copy runs while this one's boxes are tested
            int b0 = __builtin_ctz(ba);
            ba &= ba - 1u;
            unsigned cur = tstage, nxt = tstage + 4u * qb;
            dma_range(boxq + (size_t)((base + b0 * 16) >> 2) * qb, cur, 4 * (int)qb);
            for (;;) {
              int b1 = -1;
              if (ba != 0u) {
                b1 = __builtin_ctz(ba);
                ba &= ba - 1u;
                dma_range(boxq + (size_t)((base + b1 * 16) >> 2) * qb, nxt, 4 * (int)qb);
                wait_vm(1);
              } else {
                wait_vm(0);
              }
              unsigned unused = 0u;
              cand |= (unsigned long long)quads_verdicts(cur, 4, amask, unused) << (b0 * 16);
              if (STATS) ls.bytes_scalar += 4u * qb;
              if (b1 < 0) break;
              b0 = b1;
              const unsigned t_ = cur;
              cur = nxt;
              nxt = t_;
            }
            if (!act) cand = 0ull;
            if (__ballot(cand != 0ull) != 0ull) resolve_sorted<STATS>(ll_tri_, ll_exact_, r, base, cand, b_, bt_, ls);
          }
        }
      }
    };
    // every lane resolves ITS candidates of the list starting at stream position pb_ in list order
    // (octtree.cc:177-196): exact box and vertices are fetched together, one round trip per candidate
    auto resolve_list = [&](int pb_, unsigned long long cand, int &b_, double &bt_, unsigned &mt_, unsigned &bv_) {
      typedef const __attribute__((address_space(1))) double *gdp;
      for (int guard = 0; guard <= 64 && __ballot(cand != 0ull) != 0ull; guard++) {
        if (cand != 0ull) {
          const int t = pb_ + __builtin_ctzll(cand);
          cand &= cand - 1ull;
          const gdp ep = (gdp)tri_aabb_ + (size_t)t * 6;
          const gdp vp = (gdp)tri_vertex_ + (size_t)t * 9;
          const double e[6] = {ep[0], ep[1], ep[2], ep[3], ep[4], ep[5]};
          const double v[9] = {vp[0], vp[1], vp[2], vp[3], vp[4], vp[5], vp[6], vp[7], vp[8]};
          bv_ += 120u;
          if (slab_pass_lane<false>(e, r)) {
            mt_++;
            double tt;
            if (moller_trumbore_flat(v, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, &tt)) {
              if (!(b_ >= 0 && tt > bt_)) {
                b_ = t;
                bt_ = tt;
              }
            }
          }
        }
      }
    };
    // Offers the result (rp_, rt_; rp_ < 0 = none) that the lanes brought back from child slot_ to the frame
    // on top, and lets the lanes that hold a candidate drop the children that sort behind it.
    auto offer = [&](int slot_, int rp_, double rt_) {
      // offer child `slot_`'s result to frame `lev`: octtree.cc:204-211 (does the
      // ray enter that child's box, at what distance) and :226-246 (not farther
      // than the own hit; first in sorted order = smallest (distance, index))
      struct { double lo[3], c[3], hi[3]; } Pv;
      {
        const MT_LDS double *pl = h_planes + lev * 10;
        Pv.lo[0] = pl[0]; Pv.lo[1] = pl[1]; Pv.lo[2] = pl[2];
        Pv.c[0] = pl[3]; Pv.c[1] = pl[4]; Pv.c[2] = pl[5];
        Pv.hi[0] = pl[6]; Pv.hi[1] = pl[7]; Pv.hi[2] = pl[8];
      }
      const auto *P = &Pv;
      double amin[3][2], amax[3][2];
      {
        const double t0 = (P->lo[0] - r.ox) * r.ix, tc = (P->c[0] - r.ox) * r.ix, t1 = (P->hi[0] - r.ox) * r.ix;
        amax[0][0] = mx<false>(t0, tc); amin[0][0] = mn<false>(t0, tc);
        amax[0][1] = mx<false>(tc, t1); amin[0][1] = mn<false>(tc, t1);
      }
      {
        const double t0 = (P->lo[1] - r.oy) * r.iy, tc = (P->c[1] - r.oy) * r.iy, t1 = (P->hi[1] - r.oy) * r.iy;
        amax[1][0] = mx<false>(t0, tc); amin[1][0] = mn<false>(t0, tc);
        amax[1][1] = mx<false>(tc, t1); amin[1][1] = mn<false>(tc, t1);
      }
      {
        const double t0 = (P->lo[2] - r.oz) * r.iz, tc = (P->c[2] - r.oz) * r.iz, t1 = (P->hi[2] - r.oz) * r.iz;
        amax[2][0] = mx<false>(t0, tc); amin[2][0] = mn<false>(t0, tc);
        amax[2][1] = mx<false>(tc, t1); amin[2][1] = mn<false>(tc, t1);
      }
      // child index bits: 0 = x high, 1 = z high, 2 = y high (octtree.cc:61-100)
      const bool xh = (slot_ & 1) != 0, zh = (slot_ & 2) != 0, yh = (slot_ & 4) != 0;  // wave-uniform
      const double tmax = mn3<false>(xh ? amax[0][1] : amax[0][0], yh ? amax[1][1] : amax[1][0], zh ? amax[2][1] : amax[2][0]);
      const double tmin = mx3<false>(xh ? amin[0][1] : amin[0][0], yh ? amin[1][1] : amin[1][0], zh ? amin[2][1] : amin[2][0]);
      const bool entered = (tmax >= 0.0) & (tmin <= tmax);
      const int own_p = HS_LD(h_own_p, g_own_p, lev);
      const double own_t = HS_LD(h_own_t, g_own_t, lev);
      if (rp_ >= 0 && entered && !(own_p >= 0 && rt_ > own_t)) {
        const int wp = HS_LD(h_win_p, g_win_p, lev);
        bool take = wp < 0;
        if (!take) {
          const int kw = (int)((unsigned)wp >> 28);
          const bool wxh = (kw & 1) != 0, wzh = (kw & 2) != 0, wyh = (kw & 4) != 0;
          const double wmin = mx3<false>(wxh ? amin[0][1] : amin[0][0], wyh ? amin[1][1] : amin[1][0], wzh ? amin[2][1] : amin[2][0]);
          take = (tmin < wmin) || (tmin == wmin && slot_ < kw);
        }
        if (take) {
          HS_ST(h_win_p, g_win_p, lev, rp_ | (slot_ << 28));
          HS_ST(h_win_t, g_win_t, lev, rt_);
        }
      }
      // Lanes that hold a candidate drop the children that sort behind it: the reference's
      // loop would have stopped before them (they could only be looked at, never taken).
      {
        unsigned rest = get8(pendA, pendB, pendC, lev);
        if (rest != 0u) {
          const int wp = HS_LD(h_win_p, g_win_p, lev);
          const int kw = (int)((unsigned)wp >> 28) & 7;
          const bool wxh = (kw & 1) != 0, wzh = (kw & 2) != 0, wyh = (kw & 4) != 0;
          const double wmin = mx3<false>(wxh ? amin[0][1] : amin[0][0], wyh ? amin[1][1] : amin[1][0], wzh ? amin[2][1] : amin[2][0]);
          unsigned my = get8(wantA, wantB, wantC, lev);
          unsigned still = 0u;
          while (rest != 0u) {
            const int c2 = __builtin_ctz(rest);
            rest &= rest - 1u;
            const bool cxh = (c2 & 1) != 0, czh = (c2 & 2) != 0, cyh = (c2 & 4) != 0;  // wave-uniform
            const double cmin = mx3<false>(cxh ? amin[0][1] : amin[0][0], cyh ? amin[1][1] : amin[1][0], czh ? amin[2][1] : amin[2][0]);
            const bool behind = wp >= 0 && !((cmin < wmin) || (cmin == wmin && c2 < kw));
            if (behind) my &= ~(1u << c2);
            if (__ballot(((my >> c2) & 1u) != 0u) != 0ull) still |= 1u << c2;
          }
          set8(wantA, wantB, wantC, lev, my);
          set8(pendA, pendB, pendC, lev, still);
        }
      }
    };
    if (m != 0ull) {
      hs_fetch(0);
      for (;;) {
      if (++steps > step_bound) {
        status = DEV_ERR_TRAVERSAL_BOUND;
        break;
      }
      // The walk's state is the same in every lane; say so (hipcc's divergence analysis loses it
      // across the loop and would run the bookkeeping below as per-lane vector code).
      lev = uniform_i32(lev);
      node = uniform_i32(node);
      slot = uniform_i32(slot);
      entering = uniform_i32(entering ? 1 : 0) != 0;
      pendA = uniform_u64(pendA);
      pendB = uniform_u64(pendB);
      if constexpr (WIDE) pendC = uniform_u64(pendC);
      m = ((unsigned long long)(unsigned)uniform_i32((int)(m >> 32)) << 32) | (unsigned)uniform_i32((int)m);
      if (entering) {
        MT_TL(2);  // ENTER
        const bool in = ((m >> lane) & 1ull) != 0ull;
        MT_PROF_BEGIN(prof_t1);
        if (node != 0) hs_fetch(node);  // (the root's record was requested before the loop)
        // hipcc 7.2 does not wait for an LDS-DMA before LDS reads through a pointer it cannot
        // trace back to the DMA's destination: the wait is explicit
#ifdef MT_PROF
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MT_PROF
        MT_PROF_COUNT(PROF_HS_CLOSE_T, __builtin_amdgcn_s_memtime() - tw0);
#endif
        MT_TL(3);  // record landed
        const unsigned rec = stage;
        const MT_LDS int *ri = (const MT_LDS int *)(uintptr_t)rec;
        const int fc = uniform_i32(ri[0]), pb = uniform_i32(ri[1]), pc = uniform_i32(ri[2]);
        const unsigned cm = (unsigned)uniform_i32(ri[3]);
#ifdef MT_PROF
        asm volatile("" :: "s"(fc), "s"(pb), "s"(pc), "s"(cm));
        MT_PROF_END(PROF_HS_REC_T, prof_t1);
        MT_PROF_COUNT(PROF_HS_N_ENTER, 1);
        MT_PROF_COUNT(PROF_HS_LANES, __builtin_popcountll(m));
        MT_PROF_BEGIN(prof_t1);
#endif
        // a short list's fp32 boxes (DevScene::sl_box32) are copied to LDS while the children are tested
        const bool small_list = pc > 0 && pc <= kHsShortList;
        if (small_list) {
          dma_range(sl_bytes + (size_t)uniform_i32(ri[kHsRecSl / 4]) * (size_t)(kSlQuadFloats * 4), tstage,
                    ((pc + 3) >> 2) * (kSlQuadFloats * 4));
        }
        // ... and the first level of a long list's sorted copy (scan_long finds it on its way)
        const bool long_list = pc > kHsShortList;
        if (long_list) long_first_copy(uniform_i32(ri[kHsRecLl / 4]), pc);
        if (STATS) {
          st.wave_node_steps++;
          st.wave_tri_steps += (unsigned)pc;
          st.bytes_scalar += (unsigned)sizeof(HsRec);  // the staged record, once for the wave
          if (in) {
            MT_CNT_ADD(1, 1u);
            MT_CNT_ADD(2, (unsigned)pc);
            if (fc != 0) MT_CNT_ADD(0, 8u);
          }
        }
        // which children does some lane's filter let through?  All nine boxes (eight subtrees, the
        // own list) are read in one batch and tested without branches.
        typedef float f4v __attribute__((ext_vector_type(4)));
        unsigned bits = 0u, any = 0u;
        if (cm != 0u) {  // (most nodes a wave enters are leaves)
          // per axis the lane's [near x 8][far x 8] planes: 64 consecutive bytes of the record's row (HsRec::kid)
          const MT_LDS f4v *kx = (const MT_LDS f4v *)(uintptr_t)(rec + 16u + (sxl != 0 ? 32u : 0u));
          const MT_LDS f4v *ky = (const MT_LDS f4v *)(uintptr_t)(rec + 112u + (syl != 0 ? 32u : 0u));
          const MT_LDS f4v *kz = (const MT_LDS f4v *)(uintptr_t)(rec + 208u + (szl != 0 ? 32u : 0u));
          float pnx[8], pfx[8], pny[8], pfy[8], pnz[8], pfz[8];
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const f4v a = kx[i], b = kx[2 + i], c = ky[i], d = ky[2 + i], e = kz[i], g = kz[2 + i];
            pnx[i * 4 + 0] = a.x; pnx[i * 4 + 1] = a.y; pnx[i * 4 + 2] = a.z; pnx[i * 4 + 3] = a.w;
            pfx[i * 4 + 0] = b.x; pfx[i * 4 + 1] = b.y; pfx[i * 4 + 2] = b.z; pfx[i * 4 + 3] = b.w;
            pny[i * 4 + 0] = c.x; pny[i * 4 + 1] = c.y; pny[i * 4 + 2] = c.z; pny[i * 4 + 3] = c.w;
            pfy[i * 4 + 0] = d.x; pfy[i * 4 + 1] = d.y; pfy[i * 4 + 2] = d.z; pfy[i * 4 + 3] = d.w;
            pnz[i * 4 + 0] = e.x; pnz[i * 4 + 1] = e.y; pnz[i * 4 + 2] = e.z; pnz[i * 4 + 3] = e.w;
            pfz[i * 4 + 0] = g.x; pfz[i * 4 + 1] = g.y; pfz[i * 4 + 2] = g.z; pfz[i * 4 + 3] = g.w;
          }
#pragma unroll
          for (int c = 0; c < 8; c++) {
            const bool pass = near_far_may_hit(pnx[c], pfx[c], pny[c], pfy[c], pnz[c], pfz[c], f32);
            if (pass) bits |= 1u << c;
            // (the compare's lane mask IS the ballot: which children some entering lane lets through costs no vector code)
            if ((__ballot(pass) & m) != 0ull) any |= 1u << c;
          }
          bits = in ? (bits & cm) : 0u;  // (cm: children with an empty subtree hold an inverted box anyway)
          any &= cm;
        }
        MT_TL(4);  // child tests done
        // the own list's union box decides who scans it
        bool in_list = false;
        if (pc > 0) {
          const MT_LDS float *wx = (const MT_LDS float *)(uintptr_t)(rec + (unsigned)kHsRecOwn + (sxl != 0 ? 4u : 0u));
          const MT_LDS float *wy = (const MT_LDS float *)(uintptr_t)(rec + (unsigned)kHsRecOwn + 12u + (syl != 0 ? 4u : 0u));
          const MT_LDS float *wz = (const MT_LDS float *)(uintptr_t)(rec + (unsigned)kHsRecOwn + 24u + (szl != 0 ? 4u : 0u));
          in_list = in && near_far_may_hit(wx[0], wx[1], wy[0], wy[1], wz[0], wz[1], f32);
        }
        const unsigned long long lm = __ballot(in_list);
#ifdef MT_PROF
        asm volatile("" :: "v"(bits), "s"(any), "s"(lm));
        MT_PROF_END(PROF_HS_KIDS_T, prof_t1);
        MT_PROF_BEGIN(prof_t1);
#endif
        MT_TL(5);  // copies issued
        int best = -1;
        double best_t = 0.0;
        if (small_list && lm != 0ull) {
          wait_vm(0);  // the list's boxes have landed
          unsigned long long cand = in_list ? list_bits(tstage, pc) : 0ull;  // per lane: list positions whose fp32 box the ray may hit
          if (STATS) st.bytes_scalar += 24u * (unsigned)pc;
          unsigned mt = 0u, bv = 0u;
          resolve_list(pb, cand, best, best_t, mt, bv);
          if (status != DEV_OK) break;
          if (STATS) {
            if (mt) MT_CNT_ADD(3, mt);
            if (bv) MT_CNT_ADD(4, bv);
          }
#ifdef MT_PROF
          asm volatile("" :: "v"(best), "v"(best_t));
          MT_PROF_END(PROF_HS_SMALL_T, prof_t1); MT_PROF_COUNT(PROF_HS_N_SMALL, 1); MT_PROF_COUNT(PROF_HS_SMALL_TRIS, pc);
#endif
        } else if (lm != 0ull) {
          const bool blocks_ok = pc > kHsShortList;
          (void)blocks_ok;
          {
            // (every list that comes here is longer than kHsShortList: it has a spatially sorted copy)
            LaneStats ls;
            ls.clear();
            scan_long(uniform_i32(ri[kHsRecLl / 4]), pc, in_list, best, best_t, ls);
            if (STATS) {
              st.bytes_scalar += ls.bytes_scalar;
              if (ls.v[ST_MT_TESTS]) MT_CNT_ADD(3, ls.v[ST_MT_TESTS]);
              if (ls.v[ST_BYTES_VECTOR]) MT_CNT_ADD(4, ls.v[ST_BYTES_VECTOR]);
            }
          }
#ifdef MT_PROF
          asm volatile("" :: "v"(best), "v"(best_t));
          if (blocks_ok) { MT_PROF_END(PROF_HS_BIG_T, prof_t1); MT_PROF_COUNT(PROF_HS_N_BIG, 1); MT_PROF_COUNT(PROF_HS_BIG_TRIS, pc); }
          else { MT_PROF_END(PROF_HS_SMALL_T, prof_t1); MT_PROF_COUNT(PROF_HS_N_SMALL, 1); MT_PROF_COUNT(PROF_HS_SMALL_TRIS, pc); }
#endif
        } else {
          MT_PROF_COUNT(PROF_HS_N_EMPTY, 1);
        }
        MT_TL(6);  // own list done
        entering = false;
        if (any == 0u) {  // a leaf, or nothing to look at below: the node's result is its own list's
          ret_p = best;
          ret_t = best_t;
          slot = lev >= 0 ? node - __builtin_amdgcn_readlane(lane_fc, lev) : 0;
          continue;
        }
        // open a frame (for ALL lanes: the ones outside m hold "nothing" in it)
        lev++;
        if (lev >= L) {  // cannot happen: a node with children is above the deepest level
          status = DEV_ERR_UNWIND_BOUND;
          break;
        }
        HS_ST(h_own_t, g_own_t, lev, best_t);
        HS_ST(h_own_p, g_own_p, lev, best);
        HS_ST(h_win_p, g_win_p, lev, -1);
        lane_node = lane == lev ? node : lane_node;
        lane_fc = lane == lev ? fc : lane_fc;
        if (lane < 9) h_planes[lev * 10 + lane] = ((const MT_LDS double *)(uintptr_t)(rec + (unsigned)kHsRecPlanes))[lane];
        set8(wantA, wantB, wantC, lev, bits);
        set8(pendA, pendB, pendC, lev, any);
        ret_p = -1;  // nothing comes back yet
        MT_TL(7);  // frame open
      } else {
        MT_TL(8);  // back at a frame
        if (lev < 0) break;  // ret_* is the root's result
        MT_PROF_BEGIN(prof_t1);
        MT_PROF_COUNT(PROF_HS_N_RET, 1);
        if (__ballot(ret_p >= 0) != 0ull) {
          MT_PROF_COUNT(PROF_HS_N_RETHIT, 1);
          offer(slot, ret_p, ret_t);
          pendA = uniform_u64(pendA);  // (wave-uniform: built from ballots; said so, or picking the next child runs as vector code)
          pendB = uniform_u64(pendB);
          if constexpr (WIDE) pendC = uniform_u64(pendC);
        }
        const unsigned todo = get8(pendA, pendB, pendC, lev);
        MT_PROF_END(PROF_HS_RET_T, prof_t1);
        if (todo == 0u) {  // close the frame: octtree.cc:248-256
          const int wp = HS_LD(h_win_p, g_win_p, lev);
          if (wp >= 0) {
            ret_p = wp & 0x0fffffff;
            ret_t = HS_LD(h_win_t, g_win_t, lev);
          } else {
            ret_p = HS_LD(h_own_p, g_own_p, lev);
            ret_t = HS_LD(h_own_t, g_own_t, lev);
          }
          const int closed = __builtin_amdgcn_readlane(lane_node, lev);
          lev--;
          slot = lev >= 0 ? closed - __builtin_amdgcn_readlane(lane_fc, lev) : 0;
          continue;
        }
        MT_TL(9);  // offer done, next child picked
        const int c = pick(todo);
        set8(pendA, pendB, pendC, lev, todo & ~(1u << c));
        m = __ballot(((get8(wantA, wantB, wantC, lev) >> c) & 1u) != 0u);
        node = __builtin_amdgcn_readlane(lane_fc, lev) + c;
        entering = true;
      }
      }
    }
    if (want && status == DEV_OK) {
      out_prim = ret_p;
      out_t = ret_t;
    }
    cur = -1;
  }
  if (!hs_done)
  // Main loop of the ordered descent.  (A) every lane works through the small nodes on its path by itself.
  // (B) When all lanes wait at big nodes, the wave takes the LOWEST-NUMBERED one (breadth-first numbering: the top
  // of the tree first, so that lanes above catch up with lanes waiting deeper) and scans it for the lanes that wait
  // at it; its children are ordered and those lanes step on (finish_node).
  {
  for (;;) {
    // ---- phase A: every lane works through its own small nodes
    if (lane_phase) {
      MT_PROF_BEGIN(prof_t1);
#if MT_PACK_SMALL > 0
      // A handful of lanes with exact scans (what the hit-set walk leaves behind: rays with a zero direction
      // component, four to eight lanes of a unit on such a pixel column): their small nodes' lists are scanned
      // together, packed over the wave's lanes (scan_small_packed_call), instead of every lane by itself.
      if (!all_regular) {
        for (int guard = 0;; guard++) {
          const bool mine = cur >= 0 && cur_pc < kBigNode;
          const unsigned long long need = __ballot(mine);
          if (need == 0ull || __builtin_popcountll(need) > MT_PACK_SMALL) break;
          if (guard > S.n_nodes) {
            if (mine) cur = -2;
            break;
          }
          const ScanOut o = scan_small_packed_call<STATS>(S.tri_aabb, S.tri_vertex, mine ? cur_pb : 0, mine ? cur_pc : 0,
                                                          (unsigned)need, (unsigned)(need >> 32), MT_RAY_ARGS(r));
          if (mine) {
            if (STATS) {
              MT_CNT_ADD(1, 1u);
              MT_CNT_ADD(2, (unsigned)cur_pc);
              if (o.mt_tests) MT_CNT_ADD(3, o.mt_tests);
              if (o.bytes_v) MT_CNT_ADD(4, o.bytes_v);
            }
            unsigned ordw = 0;
            const int fc = cur_fc;
            if (fc != 0) {
              if (STATS) MT_CNT_ADD(0, 8u);
              ordw = order_children_exact_lane_call(S.nodes + cur, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                    irr_boxes ? S.sub_aabb32 + (size_t)fc * 6 : nullptr, S.self, fc);
              if (STATS) MT_CNT_ADD(4, 88u + 24u * (((ordw >> 24) & 15u) + 1u));
            }
            finish_node(fc, ordw, o.best, o.best_t);
          }
        }
      }
#endif
      for (int guard = 0; cur >= 0 && cur_pc < kBigNode; guard++) {
        if (guard > S.n_nodes) {
          cur = -2;
          break;
        }
#ifdef MT_DIAG
        diag_a_trips++;
#endif
        const ScanOut o = (use_filter && S.force_mode != 4)
            ? scan_small_lane_f32_call<STATS>(S.self, cur_pb, cur_pc, MT_RAY_ARGS(r), MT_F32_ARGS(f32))
            : all_regular
            ? scan_small_lane_call<false, STATS>(S.tri_aabb, S.tri_vertex, cur_pb, cur_pc, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.ix, r.iy, r.iz)
            : scan_small_lane_call<true, STATS>(S.tri_aabb, S.tri_vertex, cur_pb, cur_pc, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.ix, r.iy, r.iz);
        if (STATS) {
          MT_CNT_ADD(1, 1u);
          MT_CNT_ADD(2, (unsigned)cur_pc);
          if (o.mt_tests) MT_CNT_ADD(3, o.mt_tests);
          if (o.bytes_v) MT_CNT_ADD(4, o.bytes_v);
        }
        unsigned ordw = 0;
        const int fc = cur_fc;
        if (fc != 0) {
          if (STATS) MT_CNT_ADD(0, 8u);
          const NodeRec *Np = S.nodes + cur;
          const float *sub = cull ? S.sub_aabb32 + (size_t)fc * 6 : nullptr;
          ordw = all_regular ? order_children_regular(Np, r, sub, f32, false)
                             : order_children_exact_lane_call(Np, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                              irr_boxes ? S.sub_aabb32 + (size_t)fc * 6 : nullptr, S.self, fc);
        }
        if (STATS && fc != 0) {  // node planes + record (88 B) and the subtree boxes of the children entered
          MT_CNT_ADD(4, 88u + 24u * (((ordw >> 24) & 15u) + 1u));
        }
        finish_node(fc, ordw, o.best, o.best_t);
      }
      MT_PROF_END(PROF_SHADE, prof_t1);
      if (__ballot(cur == -2) != 0ull) {
        status = DEV_ERR_UNWIND_BOUND;
        break;
      }
    }

    // ---- phase B: one BIG node, wave-uniform
    const int n = wave_min_i32(cur >= 0 ? cur : 0x7fffffff);
    if (n == 0x7fffffff) break;
    if (++steps > step_bound || n < 0 || n >= S.n_nodes) {
      status = DEV_ERR_TRAVERSAL_BOUND;
      break;
    }
    const bool in = (cur == n);
    const unsigned long long inmask = __ballot(in);
    // the node's record, from a lane that holds it (no memory round trip)
    const int src = __builtin_ctzll(inmask);
    const int fc = __builtin_amdgcn_readlane(cur_fc, src);
    const int pb = __builtin_amdgcn_readlane(cur_pb, src);
    const int pc = __builtin_amdgcn_readlane(cur_pc, src);
    if (STATS) {
      st.wave_node_steps++;
      st.wave_tri_steps += (unsigned)pc;
    }
    // wave-uniform mode choice for this node
    int mode = 0;
    int sx = 0, sy = 0, sz = 0;
    if (all_regular) {
      mode = 1;
      if (S.force_mode != 2) {
        const unsigned long long mxs = __ballot(in && sxl), mys = __ballot(in && syl),
                                 mzs = __ballot(in && szl);
        if ((mxs == 0 || mxs == inmask) && (mys == 0 || mys == inmask) &&
            (mzs == 0 || mzs == inmask)) {
          mode = 2;
          sx = mxs != 0; sy = mys != 0; sz = mzs != 0;
        }
      }
    }
    // Few lanes on this node?  Then go triangle-parallel (cost model in
    // instruction counts: a ray-parallel step ~20, a transposed chunk ~45 plus
    // ~30 to broadcast each ray).
    const int n_in = __builtin_popcountll(inmask);
    MT_PROF_COUNT(PROF_NIN_SUM, n_in);
    MT_PROF_COUNT(PROF_NIN_LT8, n_in < 8 ? 1 : 0);
    MT_PROF_COUNT(PROF_NIN_LT24, n_in < 24 ? 1 : 0);
    MT_PROF_COUNT(PROF_WANT_SUM, __builtin_popcountll(__ballot(cur >= 0)));
    const int chunks = (pc + 63) >> 6;
    // With block boxes (regular rays, big node) both forms get much cheaper:
    // ray-parallel ~14 per block + 200, transposed ~160 + 40 per 64 blocks per ray.
    const bool blocks_ok = all_regular && use_filter && pc >= kBigNode && S.force_mode != 6;
    const int nblk = (pb + pc - 1) / kGroupTris - pb / kGroupTris + 1;
    const bool blocks_irr = irr_boxes && pc >= kBigNode && n_in <= 16;
    const bool transposed =
        (S.force_mode != 3) && pc > 0 &&
        (blocks_irr ? true
         : blocks_ok ? (n_in * (160 + 40 * ((nblk + 63) >> 6)) < 200 + 14 * nblk)
                     : (n_in * (30 + 45 * chunks) < 20 * pc));
    int best = -1;
    double best_t = 0.0;
    MT_PROF_BEGIN(prof_t1);
#ifdef MT_DIAG
    if (transposed) diag_transposed++;
#endif
    if (transposed) {
      MT_PROF_COUNT(PROF_N_TRANSPOSED, 1);
      MT_PROF_COUNT(PROF_N_CHUNKS, n_in * chunks);
      if (STATS) st.wave_tri_steps += (unsigned)(n_in * chunks) - (unsigned)pc;  // replaces the pc counted above
      const ScanOut o = blocks_ok
          ? scan_transposed_blocks_call<false, STATS>(S.self, pb, pc, in, MT_RAY_ARGS(r), MT_F32_ARGS(f32))
          : blocks_irr
          ? scan_transposed_blocks_call<true, STATS>(S.self, pb, pc, in, MT_RAY_ARGS(r), MT_F32_ARGS(f32))
          : (mode == 0)
          ? scan_transposed_call<true, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, lane, (unsigned)inmask, (unsigned)(inmask >> 32), MT_RAY_ARGS(r))
          : scan_transposed_call<false, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, lane, (unsigned)inmask, (unsigned)(inmask >> 32), MT_RAY_ARGS(r));
      best = o.best;
      best_t = o.best_t;
      if (STATS && o.mt_tests) MT_CNT_ADD(3, o.mt_tests);
        if (STATS && o.bytes_v) MT_CNT_ADD(4, o.bytes_v);
      MT_PROF_END(PROF_SCAN_TRANSPOSED, prof_t1);
#ifdef MT_PROF
      MT_PROF_COUNT(PROF_TA_T, __builtin_amdgcn_readfirstlane(o.t_a));
      MT_PROF_COUNT(PROF_TB_T, __builtin_amdgcn_readfirstlane(o.t_b));
      MT_PROF_COUNT(PROF_TC_T, __builtin_amdgcn_readfirstlane(o.t_c));
      MT_PROF_COUNT(PROF_T_RAYS, n_in);
#endif
    } else {
      MT_PROF_COUNT(PROF_N_RAYPAR, 1);
      MT_PROF_COUNT(PROF_N_RAYPAR_TRIS, pc);
    }
    if (in) {
      if (STATS) {
        MT_CNT_ADD(1, 1u);
        MT_CNT_ADD(2, (unsigned)pc);
      }
      if (!transposed) {
        const int oct = sx | (sy << 1) | (sz << 2);
        ScanOut o;
        const float *g32 = nullptr;
        if (pc >= kBigNode && S.force_mode != 6) g32 = S.grp_aabb32;
        if (mode == 2 && use_filter) {
#ifdef MT_PROF
          const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
          o = scan_filtered_dispatch<STATS>(S, oct, g32, pb, pc, r, f32);
#ifdef MT_PROF
          MT_PROF_COUNT(PROF_M2F_CALL, __builtin_amdgcn_s_memtime() - tc0);
          MT_PROF_COUNT(PROF_G_GROUPS, __builtin_amdgcn_readfirstlane(o.n_groups));
          MT_PROF_COUNT(PROF_G_LIVE, __builtin_amdgcn_readfirstlane(o.n_live));
          MT_PROF_COUNT(PROF_G_RANGES, __builtin_amdgcn_readfirstlane(o.n_ranges));
          MT_PROF_COUNT(PROF_G_RANGE_TRIS, __builtin_amdgcn_readfirstlane(o.n_range_tris));
          MT_PROF_COUNT(PROF_TA_G, __builtin_amdgcn_readfirstlane(o.t_a));
          MT_PROF_COUNT(PROF_TB_G, __builtin_amdgcn_readfirstlane(o.t_b));
#endif
        } else if (mode == 1 && use_filter && g32 != nullptr) {
          // Mixed sign octants: the octant-free form of the filter (OCT = 8).
          o = scan_filtered_dispatch<STATS>(S, 8, g32, pb, pc, r, f32);
        }
        else if (mode == 2) o = scan_octant_dispatch<STATS>(S, oct, pb, pc, r);
        else if (mode == 1) o = scan_exact_call<1, 0, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
        else o = scan_exact_call<0, 0, STATS>(S.tri_aabb, S.tri_vertex, pb, pc, MT_RAY_ARGS(r));
        best = o.best;
        best_t = o.best_t;
        if (STATS && o.mt_tests) MT_CNT_ADD(3, o.mt_tests);
        if (STATS && o.bytes_v) MT_CNT_ADD(4, o.bytes_v);
        if (STATS) st.bytes_scalar += (unsigned)__builtin_amdgcn_readfirstlane((int)o.bytes_s);  // wave-uniform
      }
#ifdef MT_PROF
    }
    if (!transposed) {
      MT_PROF_END(PROF_SCAN_RAYPAR, prof_t1);
      const int slot_ = (mode == 2 && use_filter) ? 0 : (mode == 2 ? 1 : (mode == 1 ? 2 : 3));
      MT_PROF_END(PROF_SCAN_M2F + slot_, prof_t1);
      MT_PROF_COUNT(PROF_N_M2F + slot_, 1);
      if (slot_ == 0) MT_PROF_COUNT(PROF_TRIS_M2F, pc);
      if (slot_ == 2) MT_PROF_COUNT(PROF_TRIS_M1, pc);
    } else {
      MT_PROF_COUNT(PROF_TRIS_TRANSPOSED, pc);
    }
    MT_PROF_BEGIN(prof_t1);
    if (in) {
#endif
      unsigned ordw = 0;
      if (fc != 0) {
        if (STATS) MT_CNT_ADD(0, 8u);
        const NodeRec *Np = S.nodes + n;
        const float *sub = cull ? S.sub_aabb32 + (size_t)fc * 6 : nullptr;
        ordw = (mode == 0) ? order_children_exact_call(Np, r.ox, r.oy, r.oz, r.ix, r.iy, r.iz,
                                                       irr_boxes ? S.sub_aabb32 + (size_t)fc * 6 : nullptr, S.self, fc)
                           : order_children_regular(Np, r, sub, f32, true);
      }
      if (STATS && fc != 0) st.bytes_scalar += 96u + 24u * 8u;  // node record + the children's subtree boxes, once for the wave
      finish_node(fc, ordw, best, best_t);
    }
    MT_PROF_END(PROF_CHILDREN_UNWIND, prof_t1);
    if (__ballot(cur == -2) != 0ull) {
      status = DEV_ERR_UNWIND_BOUND;
      break;
    }
  }
  }
#ifdef MT_DIAG
  {  // wave-uniform diagnostics packed into wave_tri_steps: A-phase trips (max over lanes) | transposed << 12
    unsigned mx = diag_a_trips;
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
    st.wave_tri_steps = (mx & 0xfffu) | ((diag_transposed & 0x3ffu) << 12);
  }
#endif
  if (status != DEV_OK) break;
  }  // rounds
  MT_PROF_END(PROF_TRACE, prof_t0);
  MT_PROF_FLUSH(S.prof, lane);
  if (status != DEV_OK) {
    out_prim = -1;
    out_t = 0.0;
  }
  TraceOut o;
  o.status = status;
  o.prim = out_prim;
  o.t = out_t;
  o.box_tests = STATS ? MT_CNT_GET(0) : 0u;
  o.node_visits = STATS ? MT_CNT_GET(1) : 0u;
  o.tri_tests = STATS ? MT_CNT_GET(2) : 0u;
  o.mt_tests = STATS ? MT_CNT_GET(3) : 0u;
  o.bytes_vector = STATS ? MT_CNT_GET(4) : 0u;
  o.wave_node_steps = st.wave_node_steps;
  o.wave_tri_steps = st.wave_tri_steps;
  o.bytes_scalar = st.bytes_scalar;

  return o;
}

// Folds a traversal's counters into the caller's.
template <bool STATS>
__device__ __forceinline__ void add_trace_stats(LaneStats &st, const TraceOut &o) {
  if (STATS) {
    st.v[ST_BOX_TESTS] += o.box_tests;
    st.v[ST_NODE_VISITS] += o.node_visits;
    st.v[ST_TRI_TESTS] += o.tri_tests;
    st.v[ST_MT_TESTS] += o.mt_tests;
    st.v[ST_BYTES_VECTOR] += o.bytes_vector;
    st.bytes_scalar += (unsigned)__builtin_amdgcn_readfirstlane((int)o.bytes_scalar);
    st.wave_node_steps += (unsigned)__builtin_amdgcn_readfirstlane((int)o.wave_node_steps);
    st.wave_tri_steps += (unsigned)__builtin_amdgcn_readfirstlane((int)o.wave_tri_steps);
  }
}

}  // namespace mt
