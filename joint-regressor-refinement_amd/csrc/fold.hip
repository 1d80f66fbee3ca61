// "Folded" joint regression: an exact algebraic re-association of SMPL + J_regressor for the case
// where only the regressed joints are consumed (BASELINE configs 2-4: no vertices, no renderer).
//
//   joints[b,i,r] = sum_v Jn[i,v] sum_j W[v,j] ( sum_c A[b,j,r,c] vp_c[v,b] + A[b,j,r,3] ),   vp_c = D_c . F[b]
//                 = sum_j ( sum_c A[b,j,r,c] M[b,(i,j,c)] + A[b,j,r,3] G0[i,j] )
//   with  M[b,(i,j,c)] = sum_k H[(i,j,c)][k] F[b,k],
//         H[(i,j,c)][k] = sum_v Jn[i,v] W[v,j] D_c[k,v]   (1224 x 218, pose-independent, rebuilt when J changes)
//         G0[i,j]       = sum_v Jn[i,v] W[v,j].
// The per-iteration work drops from ~13.8 MFLOP/pose (dense skinning of 6890 vertices) to ~0.55 MFLOP/pose.
// It is a DIFFERENT algorithm with its own denominator: bench.py reports it
// separately and never as the headline value; results agree with the dense path to fp32 rounding.
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

constexpr int FP = 32;   // poses per block

// JW[v][m = i*24 + j] = Jn[i][v] * W[v][j]   (m < 408; columns 408..511 zero), v over the padded range
// (row v of the internal vertex order holds vertex p2v[v] of the regressor's columns; NULL = identity)
__global__ void k_fold_jw(const float* __restrict__ Jn, const float* __restrict__ Wjv, float* __restrict__ JW,
                          const int* __restrict__ p2v) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over VP * 512
  if (idx >= VP * FOLD_MJ) return;
  const int v = idx / FOLD_MJ, m = idx % FOLD_MJ;
  float val = 0.f;
  if (m < NH * NJ && v < V) {
    const int i = m / NJ, j = m % NJ;
    val = Jn[(size_t)i * V + (p2v ? p2v[v] : v)] * Wjv[((size_t)(v >> 5) * NJ + j) * 32 + (v & 31)];
  }
  JW[idx] = val;
}

// G0[m] = sum_v JW[v][m]
__global__ void k_fold_g0(const float* __restrict__ JW, float* __restrict__ G0) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= FOLD_MJ) return;
  float acc = 0.f;
  for (int v = 0; v < VP; ++v) acc += JW[(size_t)v * FOLD_MJ + m];
  G0[m] = acc;
}

// joints^T[r][i][b] = sum_j ( sum_c A[b,j,r,c] M[(i,j,c)][b] + A[b,j,r,3] G0[i,j] );  block = 32 poses x 17 joints
__global__ __launch_bounds__(FP * NH) void k_fold_fwd(const float* __restrict__ MT, const float* __restrict__ AT,
                                                      const float* __restrict__ G0, float* __restrict__ Jsum, int BP) {
  __shared__ float As[12 * NJ][FP];
  const int bl = threadIdx.x & (FP - 1), i = threadIdx.x / FP;
  const int b = blockIdx.x * FP + bl;
  for (int k = i; k < 12 * NJ; k += NH) As[k][bl] = AT[(size_t)k * BP + b];
  __syncthreads();
  float acc[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < NJ; ++j) {
    const float g0 = G0[i * NJ + j];
    float m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = MT[(size_t)((i * NJ + j) * 3 + c) * BP + b];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      acc[r] += As[(r * 4 + 0) * NJ + j][bl] * m[0] + As[(r * 4 + 1) * NJ + j][bl] * m[1] + As[(r * 4 + 2) * NJ + j][bl] * m[2] +
                As[(r * 4 + 3) * NJ + j][bl] * g0;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) Jsum[(size_t)(r * NH + i) * BP + b] = acc[r];
}

// dM^T[(i,j,c)][b] = sum_r dj[b,i,r] A[b,j,r,c]
__global__ __launch_bounds__(FP * NH) void k_fold_bwd_dM(const float* __restrict__ dJT, const float* __restrict__ AT,
                                                         float* __restrict__ dMT, int BP) {
  __shared__ float As[12 * NJ][FP];
  const int bl = threadIdx.x & (FP - 1), i = threadIdx.x / FP;
  const int b = blockIdx.x * FP + bl;
  for (int k = i; k < 12 * NJ; k += NH) As[k][bl] = AT[(size_t)k * BP + b];
  __syncthreads();
  float dj[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) dj[r] = dJT[(size_t)(r * NHP + i) * BP + b];
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      dMT[(size_t)((i * NJ + j) * 3 + c) * BP + b] =
          dj[0] * As[(0 * 4 + c) * NJ + j][bl] + dj[1] * As[(1 * 4 + c) * NJ + j][bl] + dj[2] * As[(2 * 4 + c) * NJ + j][bl];
}

// dA^T[(r,c)][j][b] = sum_i dj[b,i,r] M[(i,j,c)][b]  (c < 3) ;  dA^T[(r,3)][j][b] = sum_i dj[b,i,r] G0[i,j]
__global__ __launch_bounds__(FP * NJ) void k_fold_bwd_dA(const float* __restrict__ dJT, const float* __restrict__ MT,
                                                         const float* __restrict__ G0, float* __restrict__ dA, int BP) {
  __shared__ float djs[3 * NH][FP];
  const int bl = threadIdx.x & (FP - 1), j = threadIdx.x / FP;
  const int b = blockIdx.x * FP + bl;
  for (int k = j; k < 3 * NH; k += NJ) djs[k][bl] = dJT[(size_t)((k / NH) * NHP + (k % NH)) * BP + b];
  __syncthreads();
  float acc[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) acc[e] = 0.f;
  for (int i = 0; i < NH; ++i) {
    const float g0 = G0[i * NJ + j];
    float m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) m[c] = MT[(size_t)((i * NJ + j) * 3 + c) * BP + b];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float d = djs[r * NH + i][bl];
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r * 4 + c] = fmaf(d, m[c], acc[r * 4 + c]);
      acc[r * 4 + 3] = fmaf(d, g0, acc[r * 4 + 3]);
    }
  }
#pragma unroll
  for (int e = 0; e < 12; ++e) dA[(size_t)(e * NJ + j) * BP + b] = acc[e];
}

int launch_fold_jw(const float* Jn, const float* Wjv, float* JW, float* G0, const int* p2v, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_jw, dim3(VP * FOLD_MJ / 256), dim3(256), 0, s, Jn, Wjv, JW, p2v);
  hipLaunchKernelGGL(k_fold_g0, dim3(FOLD_MJ / 64), dim3(64), 0, s, JW, G0);
  return 0;
}
int launch_fold_fwd(const float* MT, const float* AT, const float* G0, float* Jsum, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_fwd, dim3(BP / FP), dim3(FP * NH), 0, s, MT, AT, G0, Jsum, BP);
  return 0;
}
int launch_fold_bwd(const float* dJT, const float* AT, const float* MT, const float* G0, float* dMT, float* dA, int BP,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_fold_bwd_dM, dim3(BP / FP), dim3(FP * NH), 0, s, dJT, AT, dMT, BP);
  hipLaunchKernelGGL(k_fold_bwd_dA, dim3(BP / FP), dim3(FP * NJ), 0, s, dJT, MT, G0, dA, BP);
  return 0;
}

}  // namespace jrr
