// Probe: verifies the v_mfma_f32_32x32x2_f32 operand / accumulator lane maps that
// every MFMA kernel in csrc/ relies on, and the "accumulator tile as next B operand" idiom.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// D[32x32] = A[32xK] * B[Kx32], A row-major [32][K], B row-major [K][32]; dumps raw acc regs [64 lanes][16]
extern "C" __global__ void probe_mfma(const float* A, const float* B, int K, float* raw, float* Dout) {
  int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) {
    float a = A[l31 * K + k + h];
    float b = B[(k + h) * 32 + l31];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    raw[lane * 16 + r] = acc[r];
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    Dout[row * 32 + l31] = acc[r];
  }
}

// Z[32x32] = M[32x32] * X where X = A*B held in accumulators (sum over X's ROW index)
extern "C" __global__ void probe_chain(const float* A, const float* B, int K, const float* M, float* Zout) {
  int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  f32x16 x = {0};
  for (int k = 0; k < K; k += 2) {
    float a = A[l31 * K + k + h];
    float b = B[(k + h) * 32 + l31];
    x = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, x, 0, 0, 0);
  }
  f32x16 z = {0};
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // X row supplied by this lane half for reg r
    float m = M[l31 * 32 + row];                // A operand: M[i=l31][k=row]
    z = __builtin_amdgcn_mfma_f32_32x32x2f32(m, x[r], z, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    Zout[row * 32 + l31] = z[r];
  }
}

extern "C" int probe_run(const float* A, const float* B, int K, const float* M, float* raw, float* D, float* Z, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, s, A, B, K, raw, D);
  hipLaunchKernelGGL(probe_chain, dim3(1), dim3(64), 0, s, A, B, K, M, Z);
  return (int)hipGetLastError();
}
