// What do s_memtime (clock64) and s_memrealtime (wall_clock64) count on this part, and what shader clock do
// fp32 MFMAs actually run at -- for one wave alone and with every SIMD of the chip loaded?
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/clock_probe tools/probe/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(long long* out, int n, int chains) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f;
  const long long w0 = wall_clock64(), t0 = clock64();
  for (int i = 0; i < n; ++i) {
    for (int c = 0; c < 4; ++c)
      if (c < chains) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
  if (s == 12345.678f) out[2] = 1;
}
int main() {
  long long* d; hipMalloc(&d, 64);
  long long h[3];
  const int n = 20000;
  struct { int grid, block, chains; const char* what; } cfg[] = {
      {1, 64, 1, "1 wave, 1 dependent chain"}, {1, 64, 4, "1 wave, 4 independent chains"},
      {512, 256, 4, "512 WG x 4 waves (2 waves/SIMD on all 256 CUs), 4 chains"}, {1024, 256, 4, "1024 WG x 4 waves"}};
  for (auto& c : cfg) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(c.grid), dim3(c.block), 0, 0, d, n, c.chains);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      const double wall_s = h[1] * 1e-8, mfmas = (double)n * c.chains;
      if (rep == 1)
        printf("%-62s kernel %.3f ms | s_memtime %lld = %.3f GHz vs realtime | per MFMA: %.1f memtime ticks, %.2f ns -> %.3f GHz if 64 clk/MFMA | %.1f TFLOP/s chip\n",
               c.what, ms, h[0], h[0] / wall_s / 1e9, h[0] / mfmas, wall_s / mfmas * 1e9, 64.0 / (wall_s / mfmas) / 1e9,
               mfmas * 4096.0 * (c.block / 64) * c.grid / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
