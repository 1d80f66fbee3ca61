// Probe: v_mfma_f32_16x16x1_4b_f32 (four independent 16x16 outer-product blocks per instruction, 8 passes) used as
//   dA[n][b] = sum_v W[v][n] * P[v][b],  n < 16 joint rows, b < 32 pose columns, v < 32 vertex rows,
// where P lives in the ACCUMULATOR layout of v_mfma_f32_32x32x2_f32 (lane = (half, b), register q = row acc_row(q, half)).
// Block g = lane / 16 of the 4-block instruction is (half = g / 2, column range = g % 2): one instruction per register q
// adds the two rows acc_row(q, 0), acc_row(q, 1) for all 32 columns; blocks 0 + 2 -> columns 0..15, 1 + 3 -> 16..31.
// Also times the instruction (shader clocks per MFMA, one wave, independent accumulators).
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern "C" __global__ void probe16(const float* A, const float* B, int K, const float* W /*[32 v][16 n]*/, float* Out /*[16][32]*/,
                                   long long* clk) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, n = lane & 15, g = lane >> 4;
  f32x16 x = {0};
  for (int k = 0; k < K; k += 2) x = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + h], B[(k + h) * 32 + l31], x, 0, 0, 0);
  f32x16 z = {0};
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;        // the row of P this lane's register q holds
    z = __builtin_amdgcn_mfma_f32_16x16x1f32(W[row * 16 + n], x[q], z, 0, 0, 0);
  }
  // z[4 blk + i] = block blk, row 4 * (lane / 16) + i, column lane % 16
  for (int i = 0; i < 4; ++i) {
    Out[(4 * g + i) * 32 + n] = z[i] + z[8 + i];
    Out[(4 * g + i) * 32 + 16 + n] = z[4 + i] + z[12 + i];
  }
  // timing: 4 independent accumulators, 256 instructions
  f32x16 t0 = {0}, t1 = {0}, t2 = {0}, t3 = {0};
  float a = W[lane & 15], b = x[0];
  long long c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) {
    t0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, t0, 0, 0, 0);
    t1 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, t1, 0, 0, 0);
    t2 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, t2, 0, 0, 0);
    t3 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, t3, 0, 0, 0);
  }
  long long c1 = __builtin_amdgcn_s_memtime();
  f32x16 u0 = {0}, u1 = {0};
  for (int i = 0; i < 128; ++i) {
    u0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, u0, 0, 0, 0);
    u1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, u1, 0, 0, 0);
  }
  long long c2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { clk[0] = c1 - c0; clk[1] = c2 - c1; }
  if (t0[0] + t1[1] + t2[2] + t3[3] + u0[0] + u1[1] == 12345.f) Out[0] = 0.f;     // keep the loops alive
}

extern "C" int probe16_run(const float* A, const float* B, int K, const float* W, float* Out, long long* clk, void* stream) {
  hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, K, W, Out, clk);
  return (int)hipGetLastError();
}
