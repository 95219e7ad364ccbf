// How should a frame get from HBM into a caller's std::vector?  (mt_render_chunk, include/mythtracer_hip.h)
//   hipcc -O2 -o d2h_paths d2h_paths.hip && ./d2h_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  for (size_t bytes : {(size_t)1920 * 1080 * 3, (size_t)3840 * 2160 * 3}) {
    void *d; hipMalloc(&d, bytes); hipMemset(d, 7, bytes);
    std::vector<unsigned char> v(bytes, 1);
    void *pinned; hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
    memset(pinned, 0, bytes);
    hipDeviceSynchronize();
    const int N = 20;
    double t0, a = 0, b = 0, c = 0, e = 0, r1 = 0, r2 = 0;
    for (int i = 0; i < N; i++) {
      t0 = now(); hipMemcpy(v.data(), d, bytes, hipMemcpyDeviceToHost); a += now() - t0;
      t0 = now(); hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost); double x = now(); memcpy(v.data(), pinned, bytes); b += now() - t0; e += now() - x;
      t0 = now(); hipHostRegister(v.data(), bytes, hipHostRegisterDefault); double y = now(); r1 += y - t0;
      hipMemcpy(v.data(), d, bytes, hipMemcpyDeviceToHost); double z = now(); hipHostUnregister(v.data()); c += now() - t0; r2 += now() - z;
    }
    hipHostRegister(v.data(), bytes, hipHostRegisterDefault);
    double g = 0;
    for (int i = 0; i < N; i++) { t0 = now(); hipMemcpy(v.data(), d, bytes, hipMemcpyDeviceToHost); g += now() - t0; }
    hipHostUnregister(v.data());
    printf("%zu bytes: pageable hipMemcpy %.3f ms | pinned staging + memcpy %.3f (memcpy alone %.3f) | register + copy + unregister %.3f (register %.3f, unregister %.3f) | copy into registered memory %.3f\n",
           bytes, a / N, b / N, e / N, c / N, r1 / N, r2 / N, g / N);
    hipFree(d); hipHostFree(pinned);
  }
  return 0;
}
