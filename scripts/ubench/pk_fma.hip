// Issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950: 12 fused multiply-adds per loop trip as 12 v_fma_f32 or as
// 6 v_pk_fma_f32 (independent accumulators), one and two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o pk_fma pk_fma.hip && ./pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_fma(float *out, int n, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7, x8 = x0 + 8,
        x9 = x0 + 9, x10 = x0 + 10, x11 = x0 + 11;
  for (int i = 0; i < n; i++) {
    asm volatile("v_fma_f32 %0, %0, %12, %13\n\tv_fma_f32 %1, %1, %12, %13\n\tv_fma_f32 %2, %2, %12, %13\n\tv_fma_f32 %3, %3, %12, %13\n\t"
                 "v_fma_f32 %4, %4, %12, %13\n\tv_fma_f32 %5, %5, %12, %13\n\tv_fma_f32 %6, %6, %12, %13\n\tv_fma_f32 %7, %7, %12, %13\n\t"
                 "v_fma_f32 %8, %8, %12, %13\n\tv_fma_f32 %9, %9, %12, %13\n\tv_fma_f32 %10, %10, %12, %13\n\tv_fma_f32 %11, %11, %12, %13"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9), "+v"(x10), "+v"(x11)
                 : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + x8 + x9 + x10 + x11;
}
__global__ void k_pk(float *out, int n, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1.0f}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f;
  f2 A = {a, a}, B = {b, b};
  for (int i = 0; i < n; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %6, %7\n\tv_pk_fma_f32 %1, %1, %6, %7\n\tv_pk_fma_f32 %2, %2, %6, %7\n\t"
                 "v_pk_fma_f32 %3, %3, %6, %7\n\tv_pk_fma_f32 %4, %4, %6, %7\n\tv_pk_fma_f32 %5, %5, %6, %7"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5)
                 : "v"(A), "v"(B));
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// the box test's use: op_sel broadcasts and a negated half
__global__ void k_pk_sel(float *out, int n, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1.0f}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f;
  f2 A = {a, a * 0.5f}, B = {b, b};
  for (int i = 0; i < n; i++) {
    asm volatile("v_pk_fma_f32 %0, %6, %6, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
                 "v_pk_fma_f32 %1, %6, %6, %1 op_sel:[0,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
                 "v_pk_fma_f32 %2, %6, %6, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
                 "v_pk_fma_f32 %3, %6, %6, %3 op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 %4, %6, %6, %4 op_sel:[1,0,0] op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 %5, %6, %6, %5 op_sel:[1,0,0] op_sel_hi:[1,0,1]"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5)
                 : "v"(A), "v"(B));
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <typename K>
static float run(K k, int blocks, float *d, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, n, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, n, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  float *d; hipMalloc(&d, 4096 * 256 * 4);
  const int n = 200000;
  for (int blocks : {256, 512, 1024}) {  // 1, 2, 4 waves per SIMD
    const float a = run(k_fma, blocks, d, n), b = run(k_pk, blocks, d, n), c = run(k_pk_sel, blocks, d, n);
    const double fl = 12.0 * n;  // multiply-adds per lane
    printf("waves/SIMD %d: 12 v_fma_f32 %.3f ms (%.2f cyc/instr/wave @2.4GHz)  6 v_pk_fma_f32 %.3f ms (%.2f)  with op_sel/neg %.3f ms\n",
           blocks / 256, a, a * 2.4e6 / fl / (blocks / 256.0) , b, b * 2.4e6 / (6.0 * n) / (blocks / 256.0), c);
  }
  return 0;
}
