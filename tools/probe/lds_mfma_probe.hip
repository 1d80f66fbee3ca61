// How fast can fp32 MFMAs run when every operand is read from LDS (ds_read_b32), as in k_lbs_fwd's blend loop?
//   variants: operand reads right before use (hipcc's natural schedule) / requested one K-pair ahead;
//   random operands; 2 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lp tools/probe/lds_mfma_probe.hip && /tmp/lp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* __restrict__ out, long long* __restrict__ clk,
                                            int iters) {
  __shared__ float lds[7168 * 2];
  for (int i = threadIdx.x; i < 7168 * 2; i += 256) lds[i] = src[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  f32x16 acc[3];
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    const float* buf = lds + (it & 1) * 7168;
    const float* dp = buf + half * 96 + l31;
    const float* fp = buf + 3072 + half * 128 + wave * 32 + l31;
    if (MODE == 0) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float f = fp[(2 * kk) * 128];
        const float d0 = dp[(2 * kk) * 96], d1 = dp[(2 * kk) * 96 + 32], d2 = dp[(2 * kk) * 96 + 64];
        acc[0] = MFMA(d0, f, acc[0]); acc[1] = MFMA(d1, f, acc[1]); acc[2] = MFMA(d2, f, acc[2]);
      }
    } else if (MODE == 1) {
      float f = fp[0], d0 = dp[0], d1 = dp[32], d2 = dp[64];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float fc = f, c0 = d0, c1 = d1, c2 = d2;
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = MFMA(c0, fc, acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < 16) { f = fp[(2 * kk + 2) * 128]; d0 = dp[(2 * kk + 2) * 96]; d1 = dp[(2 * kk + 2) * 96 + 32]; d2 = dp[(2 * kk + 2) * 96 + 64]; }
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = MFMA(c1, fc, acc[1]); acc[2] = MFMA(c2, fc, acc[2]);
      }
    } else {   // registers only (same MFMA count)
      const float f = fp[0], d0 = dp[0], d1 = dp[32], d2 = dp[64];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { acc[0] = MFMA(d0, f, acc[0]); acc[1] = MFMA(d1, f, acc[1]); acc[2] = MFMA(d2, f, acc[2]); }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int c = 0; c < 3; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
void run(const char* what, const float* src, float* out, long long* clk, int grid) {
  const int iters = 400;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, src, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    if (rep == 2) {
      const double mf = (double)iters * 48 * 4 * grid;   // wave-MFMAs
      printf("%-44s grid %4d  %.3f ms  %.1f TFLOP/s  clock %.3f GHz  ticks per MFMA per wave %.1f\n", what, grid, ms,
             mf * 4096.0 / (ms * 1e-3) / 1e12, h[0] / (h[1] * 1e-8) / 1e9, (double)h[0] / (iters * 48.0));
    }
  }
}
int main() {
  float* src; float* out; long long* clk;
  (void)hipMalloc(&src, 7168 * 2 * 4); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 16);
  float* h = (float*)malloc(7168 * 2 * 4);
  srand(1);
  for (int i = 0; i < 7168 * 2; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  (void)hipMemcpy(src, h, 7168 * 2 * 4, hipMemcpyHostToDevice);
  for (int grid : {512, 256}) {
    run<2>("operands in registers", src, out, clk, grid);
    run<0>("LDS operands, read right before use", src, out, clk, grid);
    run<1>("LDS operands, requested one K-pair ahead", src, out, clk, grid);
  }
  return 0;
}
