// Micro-benchmark: how fast can a wave stream wave-uniform 24-byte fp32 boxes
// and run the 10-instruction conservative filter on them, per fetch scheme?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define MT_CONST __attribute__((address_space(4)))
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
struct Quad { f16v lo; f8v hi; };
__device__ __forceinline__ void issue_quad(Quad &q, const MT_CONST float *p) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40" : "=&s"(q.lo), "=&s"(q.hi) : "s"(p));
}
__device__ __forceinline__ void await_quad(Quad &q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.lo), "+s"(q.hi)); }
struct F { float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz; };
__device__ __forceinline__ unsigned long long filt(const float *b, const F &f) {
  const float tnx = __builtin_fmaf(b[0], f.ix, f.cnx), tfx = __builtin_fmaf(b[3], f.ix, f.cfx);
  const float tny = __builtin_fmaf(b[1], f.iy, f.cny), tfy = __builtin_fmaf(b[4], f.iy, f.cfy);
  const float tnz = __builtin_fmaf(b[2], f.iz, f.cnz), tfz = __builtin_fmaf(b[5], f.iz, f.cfz);
  const float lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
  const float hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
  return __builtin_amdgcn_fcmpf(hi, 0.0f, 11) & __builtin_amdgcn_fcmpf(lo, hi, 13);
}
__device__ __forceinline__ unsigned long long quad_eval(const Quad &q, const F &f) {
  const float b0[6] = {q.lo[0], q.lo[1], q.lo[2], q.lo[3], q.lo[4], q.lo[5]};
  const float b1[6] = {q.lo[6], q.lo[7], q.lo[8], q.lo[9], q.lo[10], q.lo[11]};
  const float b2[6] = {q.lo[12], q.lo[13], q.lo[14], q.lo[15], q.hi[0], q.hi[1]};
  const float b3[6] = {q.hi[2], q.hi[3], q.hi[4], q.hi[5], q.hi[6], q.hi[7]};
  return filt(b0, f) | filt(b1, f) | filt(b2, f) | filt(b3, f);
}
typedef float f4v __attribute__((ext_vector_type(4)));
struct Hex { f16v a; f16v b; f4v c; };   // six boxes = 144 bytes
__device__ __forceinline__ void issue_hex(Hex &q, const MT_CONST float *p) {
  asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x80" : "=&s"(q.a), "=&s"(q.b), "=&s"(q.c) : "s"(p));
}
__device__ __forceinline__ void await_hex(Hex &q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q.a), "+s"(q.b), "+s"(q.c)); }
__device__ __forceinline__ unsigned long long hex_eval(const Hex &q, const F &f) {
  float v[36];
#pragma unroll
  for (int i = 0; i < 16; i++) { v[i] = q.a[i]; v[16 + i] = q.b[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) v[32 + i] = q.c[i];
  unsigned long long m = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) m |= filt(&v[j * 6], f);
  return m;
}
// variant 0: double-buffered quads (what mt_trace.h does)
// variant 1: single quad, no overlap
// variant 2: compiler-managed scalar loads of 8 boxes per batch, no overlap
// variant 3: LDS-staged: 1 KiB per global_load_lds, ds_read broadcast
template <int V>
__global__ __launch_bounds__(256) void k(const float *boxes, int n_boxes, int list_len, int reps, unsigned long long *out, unsigned long long *cyc, int cold) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  F f; f.ix = 1.0f + lane * 1e-3f; f.iy = 0.9f; f.iz = 1.1f; f.cnx = -1e9f; f.cny = -1e9f; f.cnz = -1e9f; f.cfx = -2e9f; f.cfy = -2e9f; f.cfz = -2e9f;
  unsigned long long acc = 0;
  // every wave scans a different list (start offset pseudo-random) like different nodes
  unsigned start = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)wave * 2654435761u) % (unsigned)(n_boxes - list_len - 64)));
  start &= ~3u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; rep++) {
    if (cold) start = ((unsigned)__builtin_amdgcn_readfirstlane((int)((((unsigned)wave * 2654435761u) ^ ((unsigned)rep * 40503u * 65537u)) % (unsigned)(n_boxes - list_len - 64)))) & ~3u;
    const MT_CONST float *p = (const MT_CONST float *)(uintptr_t)(boxes + (size_t)start * 6);
    if (V == 0) {
      Quad A, B;
      issue_quad(A, p); await_quad(A);
      for (int k = 0;;) {
        const bool mb = k + 4 < list_len;
        if (mb) issue_quad(B, p + 24);
        acc |= quad_eval(A, f);
        if (!mb) break;
        await_quad(B); k += 4; p += 48;
        const bool ma = k + 4 < list_len;
        if (ma) issue_quad(A, p);
        acc |= quad_eval(B, f);
        if (!ma) break;
        await_quad(A); k += 4;
      }
    } else if (V == 4) {
      Hex A, B;
      issue_hex(A, p); await_hex(A);
      for (int k = 0;;) {
        const bool mb = k + 6 < list_len;
        if (mb) issue_hex(B, p + 36);
        acc |= hex_eval(A, f);
        if (!mb) break;
        await_hex(B); k += 6; p += 72;
        const bool ma = k + 6 < list_len;
        if (ma) issue_hex(A, p);
        acc |= hex_eval(B, f);
        if (!ma) break;
        await_hex(A); k += 6;
      }
    } else if (V == 1) {
      Quad A;
      for (int k = 0; k < list_len; k += 4) { issue_quad(A, p + (size_t)k * 6); await_quad(A); acc |= quad_eval(A, f); }
    } else if (V == 2) {
      for (int k = 0; k < list_len; k += 8) {
        float b[48];
#pragma unroll
        for (int i = 0; i < 48; i++) b[i] = p[(size_t)k * 6 + i];
#pragma unroll
        for (int j = 0; j < 8; j++) acc |= filt(&b[j * 6], f);
      }
    } else {
      // LDS staging: slot = 1 KiB per wave-instruction (42 boxes = 1008 B used); 2 slots ring
      char *my = smem + (threadIdx.x >> 6) * 2048;
      const float *g = boxes + (size_t)start * 6;
      const int slots = (list_len + 41) / 42;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + lane * 4), (__attribute__((address_space(3))) void *)my, 16, 0, 0);
      for (int s = 0; s < slots; s++) {
        if (s + 1 < slots)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)(s + 1) * 252 + lane * 4), (__attribute__((address_space(3))) void *)(my + ((s + 1) & 1) * 1024), 16, 0, 0);
        // wait for slot s (one newer load may stay in flight)
        if (s + 1 < slots) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const __attribute__((address_space(3))) float *lb = (const __attribute__((address_space(3))) float *)(my + (s & 1) * 1024);
        const int nb = min(42, list_len - s * 42);
        for (int j = 0; j < nb; j += 2) {
          float b[12];
#pragma unroll
          for (int i = 0; i < 12; i++) b[i] = lb[j * 6 + i];
          acc |= filt(&b[0], f) | filt(&b[6], f);
        }
        asm volatile("" ::: "memory");
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[wave] = acc; cyc[wave] = t1 - t0; }
}
int main(int argc, char **argv) {
  const int n_boxes = 101152, list_len = argc > 1 ? atoi(argv[1]) : 400, reps = 200;
  std::vector<float> h((size_t)n_boxes * 6 + 1024);
  for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) * 0.4f;
  float *d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  unsigned long long *out, *cyc; hipMalloc(&out, 8 * 65536); hipMalloc(&cyc, 8 * 65536);
  const int cold = argc > 2 ? atoi(argv[2]) : 0;
  for (int v = 0; v < 5; v++) for (int wpb_blocks = 2; wpb_blocks <= 3; wpb_blocks++) {
    if (v == 1 || v == 3) continue;
    const int blocks = 256 * wpb_blocks, waves = blocks * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; it++) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 8192, 0, d, n_boxes, list_len, reps, out, cyc, cold);
      if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 8192, 0, d, n_boxes, list_len, reps, out, cyc, cold);
      if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 8192, 0, d, n_boxes, list_len, reps, out, cyc, cold);
      if (v == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 8192, 0, d, n_boxes, list_len, reps, out, cyc, cold);
      if (v == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 8192, 0, d, n_boxes, list_len, reps, out, cyc, cold);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(waves); hipMemcpy(c.data(), cyc, 8 * waves, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : c) sum += (double)x;
    const double boxes_total = (double)waves * reps * list_len;
    printf("variant %d waves/SIMD %d: %.3f ms, %.1f cycles/box/wave, %.2f Gbox/s, SIMD-cycles/box %.1f\n", v, wpb_blocks, ms,
           sum / boxes_total, boxes_total / ms / 1e6, sum / boxes_total / wpb_blocks);
  }
  return 0;
}
