// MPJPE / Procrustes-aligned MPJPE on device:
//   evaluate                                   /root/reference/scripts/utils.py:117-145
//   batch_compute_similarity_transform_torch   /root/reference/scripts/eval_utils.py:7-58
// One pose per thread; the 3x3 SVD of K = X1 X2^T is done as a cyclic Jacobi eigen-decomposition of
// K^T K (V, sigma^2) followed by U = K V / sigma, with the reference's det-sign fix on the last axis.
#include "jrr_common.h"
#include "kernels.h"
#include <type_traits>

namespace jrr {

__device__ __forceinline__ void jacobi_rotate(float A[3][3], float V[3][3], int p, int q) {
  if (fabsf(A[p][q]) < 1e-30f) return;
  const float theta = (A[q][q] - A[p][p]) / (2.f * A[p][q]);
  const float t = copysignf(1.f, theta) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
  const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {   // A <- A J
    const float akp = A[k][p], akq = A[k][q];
    A[k][p] = c * akp - s * akq;
    A[k][q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {   // A <- J^T A
    const float apk = A[p][k], aqk = A[q][k];
    A[p][k] = c * apk - s * aqk;
    A[q][k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {   // V <- V J
    const float vkp = V[k][p], vkq = V[k][q];
    V[k][p] = c * vkp - s * vkq;
    V[k][q] = s * vkp + c * vkq;
  }
}

__device__ __forceinline__ float det3(const float M[3][3]) {
  return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
         M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}

__global__ __launch_bounds__(64) void k_evaluate(const float* __restrict__ pred, const float* __restrict__ target_mm, float* __restrict__ err,
                           float* __restrict__ err_pa, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float P[NH][3], Q[NH][3];
#pragma unroll
  for (int i = 0; i < NH; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      P[i][c] = pred[((size_t)b * NH + i) * 3 + c];
      Q[i][c] = target_mm[((size_t)b * NH + i) * 3 + c] / 1000.f;
    }
  // pelvis-centre both (utils.py:127-131), MPJPE
  float e = 0.f;
#pragma unroll
  for (int i = NH - 1; i >= 0; --i) {
    float d2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      P[i][c] -= P[0][c];
      Q[i][c] -= Q[0][c];
      const float d = P[i][c] - Q[i][c];
      d2 += d * d;
    }
    e += sqrtf(d2);
  }
  err[b] = e / NH;
  // Procrustes: remove means, K = sum_n x1 x2^T
  float mu1[3] = {0.f, 0.f, 0.f}, mu2[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NH; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) { mu1[c] += P[i][c]; mu2[c] += Q[i][c]; }
#pragma unroll
  for (int c = 0; c < 3; ++c) { mu1[c] /= NH; mu2[c] /= NH; }
  float K[3][3] = {{0.f}}, var1 = 0.f;
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    float x1[3], x2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x1[c] = P[i][c] - mu1[c]; x2[c] = Q[i][c] - mu2[c]; var1 += x1[c] * x1[c]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) K[r][c] += x1[r] * x2[c];
  }
  // eigen-decomposition of S = K^T K = V diag(s^2) V^T
  float S[3][3], V[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) S[r][c] = K[0][r] * K[0][c] + K[1][r] * K[1][c] + K[2][r] * K[2][c];
  for (int sweep = 0; sweep < 8; ++sweep) {
    jacobi_rotate(S, V, 0, 1);
    jacobi_rotate(S, V, 0, 2);
    jacobi_rotate(S, V, 1, 2);
  }
  // sort singular values descending (torch.svd order): the det fix applies to the SMALLEST axis.  A compare-exchange network on
  // (sigma^2, column of V) with static indices (an index array would put sigma^2 and V into scratch memory)
  float sig2[3] = {S[0][0], S[1][1], S[2][2]};
  auto cswap = [&](auto A_, auto B_) __attribute__((always_inline)) {
    constexpr int a = decltype(A_)::value, c = decltype(B_)::value;
    if (sig2[a] < sig2[c]) {
      const float t = sig2[a]; sig2[a] = sig2[c]; sig2[c] = t;
#pragma unroll
      for (int r = 0; r < 3; ++r) { const float u = V[r][a]; V[r][a] = V[r][c]; V[r][c] = u; }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  cswap(I0{}, I1{}); cswap(I1{}, I2{}); cswap(I0{}, I1{});
  float Vs[3][3], U[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float sg = sqrtf(fmaxf(sig2[k], 0.f));
#pragma unroll
    for (int r = 0; r < 3; ++r) Vs[r][k] = V[r][k];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      U[r][k] = (K[r][0] * V[0][k] + K[r][1] * V[1][k] + K[r][2] * V[2][k]) / fmaxf(sg, 1e-20f);
  }
  // R = V Z U^T with Z = diag(1, 1, sign(det(U V^T)))   (eval_utils.py:38-44)
  float UVt[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) UVt[r][c] = U[r][0] * Vs[c][0] + U[r][1] * Vs[c][1] + U[r][2] * Vs[c][2];
  const float z = (det3(UVt) < 0.f) ? -1.f : 1.f;
  float R[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r][c] = Vs[r][0] * U[c][0] + Vs[r][1] * U[c][1] + z * Vs[r][2] * U[c][2];
  // scale = trace(R K) / var1 ; t = mu2 - scale R mu1
  float tr = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r) tr += R[r][0] * K[0][r] + R[r][1] * K[1][r] + R[r][2] * K[2][r];
  const float scale = tr / var1;
  float t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = mu2[r] - scale * (R[r][0] * mu1[0] + R[r][1] * mu1[1] + R[r][2] * mu1[2]);
  float epa = 0.f;
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    float d2 = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float h = scale * (R[r][0] * P[i][0] + R[r][1] * P[i][1] + R[r][2] * P[i][2]) + t[r];
      const float d = h - Q[i][r];
      d2 += d * d;
    }
    epa += sqrtf(d2);
  }
  err_pa[b] = epa / NH;
}

int launch_evaluate(const float* pred, const float* target_mm, float* err, float* err_pa, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_evaluate, dim3((B + 63) / 64), dim3(64), 0, s, pred, target_mm, err, err_pa, B);
  return 0;
}

}  // namespace jrr
