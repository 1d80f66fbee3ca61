// Which fp32 matrix shape sustains more FLOP/s under the chip's power management (MI355X_MICROARCH.md "DVFS give-back" (7):
// for bf16 the 16x16 shape holds a higher clock than 32x32 at equal cycles per FLOP)?  Random operands, 2 waves per SIMD on
// every CU, operands in registers (MODE 0/1) or re-read from LDS for every instruction (MODE 2/3).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sp tools/probe/mfma_shape_probe.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, long long* __restrict__ clk, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = lds[lane + 64 * i]; b[i] = lds[4096 + lane + 64 * i]; }
  float s = 0.f;
  const long long t0 = clock64(), w0 = wall_clock64();
  if (MODE == 0 || MODE == 2) {          // 32x32x2: 6 accumulators (96 registers), 48 instructions of 64 clocks per iteration
    f32x16 acc[6];
    for (int c = 0; c < 6; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const float* p = lds + ((it & 7) << 9);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float av = a[kk], bv = b[kk];
        if (MODE == 2) { av = p[lane + 64 * (kk & 3)]; bv = p[256 + lane + 64 * (kk & 3)]; }
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, MODE == 2 ? bv : b[(kk + c) & 7], acc[c], 0, 0, 0);
      }
    }
    for (int c = 0; c < 6; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  } else {                               // 16x16x4: 24 accumulators (96 registers), 96 instructions of 32 clocks per iteration
    f32x4 acc[24];
    for (int c = 0; c < 24; ++c) for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const float* p = lds + ((it & 7) << 9);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float av = a[kk], bv = b[kk];
        if (MODE == 3) { av = p[lane + 64 * kk]; bv = p[256 + lane + 64 * kk]; }
#pragma unroll
        for (int c = 0; c < 24; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(MODE == 3 ? av : a[(kk + c) & 7], MODE == 3 ? bv : b[(kk + 2 * c) & 7], acc[c], 0, 0, 0);
      }
    }
    for (int c = 0; c < 24; ++c) for (int i = 0; i < 4; ++i) s += acc[c][i];
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
void run(const char* what, const float* src, float* out, long long* clk) {
  const int iters = 600, grid = 512;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    if (rep == 3) {
      const double flop = (double)iters * 48 * 4096.0 * 4 * grid;     // both variants: 48 x 4096 FLOP per wave and iteration
      printf("%-52s %.3f ms  %.1f TFLOP/s  clock %.3f GHz  clocks per iteration %.0f (floor 3072)\n", what, ms, flop / (ms * 1e-3) / 1e12,
             h[0] / (h[1] * 1e-8) / 1e9, (double)h[0] / iters);
    }
  }
}
int main() {
  float* src; float* out; long long* clk;
  (void)hipMalloc(&src, 8192 * 4); (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&clk, 16);
  float* h = (float*)malloc(8192 * 4);
  srand(1);
  for (int i = 0; i < 8192; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  (void)hipMemcpy(src, h, 8192 * 4, hipMemcpyHostToDevice);
  // warm the chip up first (power management reacts over milliseconds)
  for (int w = 0; w < 3; ++w) { run<0>("(warm-up) 32x32x2, operands in registers", src, out, clk); }
  run<0>("32x32x2, operands in registers", src, out, clk);
  run<1>("16x16x4, operands in registers", src, out, clk);
  run<2>("32x32x2, both operands re-read from LDS per K step", src, out, clk);
  run<3>("16x16x4, both operands re-read from LDS per K step", src, out, clk);
  run<0>("32x32x2, operands in registers (again)", src, out, clk);
  run<1>("16x16x4, operands in registers (again)", src, out, clk);
  return 0;
}
