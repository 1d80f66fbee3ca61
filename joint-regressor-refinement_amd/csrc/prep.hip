// Per-pose kernels (one pose per lane): 6-D rotation -> R, blend features, kinematic chain (forward
// and adjoint), joint loss, and the fused per-pose Adam update.
//
// Reference arithmetic restated here:
//   rot6d_to_rotmat            /root/reference/scripts/utils.py:190-204
//   rigid chain / A matrices   smplx 0.1.26 lbs.batch_rigid_transform
//   move_pelvis + MSELoss      scripts/utils.py:106-114, scripts/optimize.py:238-239
//   Adam                       torch.optim.Adam defaults, scripts/optimize.py:201-202,263-265
#include <cstdlib>

#include "jrr_common.h"
#include "kernels.h"
#include "dconv.h"
#include "proj.h"
#include "supk.h"

namespace jrr {

// ------------------------------------------------------------------------------------------
// 6-D rotation representation
// ------------------------------------------------------------------------------------------
struct Rot6 {
  float b1[3], b2[3], a2[3];
  float n1, nu, s;   // |a1|, |u|, b1.a2
};

__device__ __forceinline__ void rot6d_fwd(const float x[6], float R[9], Rot6& c) {
  const float eps = 1e-12f;
  float a1[3] = {x[0], x[2], x[4]};
  c.a2[0] = x[1]; c.a2[1] = x[3]; c.a2[2] = x[5];
  c.n1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
  float d1 = fmaxf(c.n1, eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) c.b1[i] = a1[i] / d1;
  c.s = c.b1[0] * c.a2[0] + c.b1[1] * c.a2[1] + c.b1[2] * c.a2[2];
  float u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = c.a2[i] - c.s * c.b1[i];
  c.nu = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  float d2 = fmaxf(c.nu, eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) c.b2[i] = u[i] / d2;
  float b3[3] = {c.b1[1] * c.b2[2] - c.b1[2] * c.b2[1], c.b1[2] * c.b2[0] - c.b1[0] * c.b2[2],
                 c.b1[0] * c.b2[1] - c.b1[1] * c.b2[0]};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    R[r * 3 + 0] = c.b1[r];
    R[r * 3 + 1] = c.b2[r];
    R[r * 3 + 2] = b3[r];
  }
}

__device__ __forceinline__ void rot6d_bwd(const Rot6& c, const float dR[9], float dx[6]) {
  const float eps = 1e-12f;
  float db1[3], db2[3], db3[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) { db1[r] = dR[r * 3 + 0]; db2[r] = dR[r * 3 + 1]; db3[r] = dR[r * 3 + 2]; }
  // b3 = b1 x b2
  db1[0] += c.b2[1] * db3[2] - c.b2[2] * db3[1];
  db1[1] += c.b2[2] * db3[0] - c.b2[0] * db3[2];
  db1[2] += c.b2[0] * db3[1] - c.b2[1] * db3[0];
  db2[0] += db3[1] * c.b1[2] - db3[2] * c.b1[1];
  db2[1] += db3[2] * c.b1[0] - db3[0] * c.b1[2];
  db2[2] += db3[0] * c.b1[1] - db3[1] * c.b1[0];
  // b2 = u / max(|u|, eps)
  float du[3];
  if (c.nu >= eps) {
    float t = c.b2[0] * db2[0] + c.b2[1] * db2[1] + c.b2[2] * db2[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) du[i] = (db2[i] - c.b2[i] * t) / c.nu;
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) du[i] = db2[i] / eps;
  }
  // u = a2 - (b1.a2) b1
  float t2 = c.b1[0] * du[0] + c.b1[1] * du[1] + c.b1[2] * du[2];
  float da2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    da2[i] = du[i] - c.b1[i] * t2;
    db1[i] += -c.a2[i] * t2 - c.s * du[i];
  }
  // b1 = a1 / max(|a1|, eps)
  float da1[3];
  if (c.n1 >= eps) {
    float t = c.b1[0] * db1[0] + c.b1[1] * db1[1] + c.b1[2] * db1[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) da1[i] = (db1[i] - c.b1[i] * t) / c.n1;
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) da1[i] = db1[i] / eps;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { dx[2 * i] = da1[i]; dx[2 * i + 1] = da2[i]; }
}

__global__ void k_rot6d_fwd(const float* __restrict__ x, float* __restrict__ R, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xv[6], Rv[9];
  Rot6 c;
#pragma unroll
  for (int k = 0; k < 6; ++k) xv[k] = x[(size_t)i * 6 + k];
  rot6d_fwd(xv, Rv, c);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = Rv[k];
}

__global__ void k_rot6d_bwd(const float* __restrict__ x, const float* __restrict__ dR, float* __restrict__ dx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xv[6], Rv[9], g[9], dxv[6];
  Rot6 c;
#pragma unroll
  for (int k = 0; k < 6; ++k) xv[k] = x[(size_t)i * 6 + k];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = dR[(size_t)i * 9 + k];
  rot6d_fwd(xv, Rv, c);
  rot6d_bwd(c, g, dxv);
#pragma unroll
  for (int k = 0; k < 6; ++k) dx[(size_t)i * 6 + k] = dxv[k];
}

// ------------------------------------------------------------------------------------------
// Rodrigues: axis-angle -> R (smplx 0.1.26 lbs.batch_rodrigues, the pose2rot=True branch of the SMPL
// operator: /root/reference/scripts/smpl.py:61-85 inherits smplx.SMPL.forward's default):
//   theta = |aa + 1e-8| ; r = aa / theta ; K = skew(r) ; R = I + sin(theta) K + (1 - cos(theta)) K^2
// (1 - cos(theta) is evaluated as 2 sin^2(theta/2): the same value without fp32 cancellation at small angles)
// and its adjoint (finite at aa = 0: sin(theta)/theta -> 1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void skew(const float r[3], float K[9]) {
  K[0] = 0.f;   K[1] = -r[2]; K[2] = r[1];
  K[3] = r[2];  K[4] = 0.f;   K[5] = -r[0];
  K[6] = -r[1]; K[7] = r[0];  K[8] = 0.f;
}
__device__ __forceinline__ void mat3_mul(const float A[9], const float Bm[9], float C[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * Bm[j] + A[i * 3 + 1] * Bm[3 + j] + A[i * 3 + 2] * Bm[6 + j];
}

__global__ void k_rodrigues_fwd(const float* __restrict__ aa, float* __restrict__ R, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a[3] = {aa[(size_t)i * 3], aa[(size_t)i * 3 + 1], aa[(size_t)i * 3 + 2]};
  const float e0 = a[0] + 1e-8f, e1 = a[1] + 1e-8f, e2 = a[2] + 1e-8f;
  const float th = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float r[3] = {a[0] / th, a[1] / th, a[2] / th};
  const float s = sinf(th), sh = sinf(0.5f * th), omc = 2.f * sh * sh;   // 1 - cos(theta) without cancellation at small theta
  float K[9], K2[9];
  skew(r, K);
  mat3_mul(K, K, K2);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f) + s * K[k] + omc * K2[k];
}

__global__ void k_rodrigues_bwd(const float* __restrict__ aa, const float* __restrict__ dR, float* __restrict__ daa, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a[3] = {aa[(size_t)i * 3], aa[(size_t)i * 3 + 1], aa[(size_t)i * 3 + 2]};
  const float e[3] = {a[0] + 1e-8f, a[1] + 1e-8f, a[2] + 1e-8f};
  const float th = sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const float r[3] = {a[0] / th, a[1] / th, a[2] / th};
  const float s = sinf(th), c = cosf(th), sh = sinf(0.5f * th), omc = 2.f * sh * sh;
  float K[9], K2[9], g[9];
  skew(r, K);
  mat3_mul(K, K, K2);
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = dR[(size_t)i * 9 + k];
  // through sin / cos
  float dth = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) dth += g[k] * (c * K[k] + s * K2[k]);
  // through K (linear in r): dL/dK = s g + (1-c) (g K^T + K^T g)
  float Kt[9], gKt[9], Ktg[9], dK[9];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) Kt[p * 3 + q] = K[q * 3 + p];
  mat3_mul(g, Kt, gKt);
  mat3_mul(Kt, g, Ktg);
#pragma unroll
  for (int k = 0; k < 9; ++k) dK[k] = s * g[k] + omc * (gKt[k] + Ktg[k]);
  const float dr[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
  // r = a / theta ; theta = |a + 1e-8|
  dth -= (dr[0] * a[0] + dr[1] * a[1] + dr[2] * a[2]) / (th * th);
#pragma unroll
  for (int k = 0; k < 3; ++k) daa[(size_t)i * 3 + k] = dr[k] / th + dth * e[k] / th;
}

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
// The per-(pose, joint) records of the pose-major API tensors ((B,24,6) poses / gradients, (B,154) Adam state) are 6
// contiguous floats at an even float offset: three 8-byte accesses instead of six 4-byte ones (lanes are 576 / 616 bytes
// apart, so every access is its own memory transaction: their count is what these kernels pay for).
__device__ __forceinline__ void load6(const float* __restrict__ p, float v[6]) {
  const f32x2* q = reinterpret_cast<const f32x2*>(p);
  const f32x2 a = q[0], b = q[1], c = q[2];
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1]; v[4] = c[0]; v[5] = c[1];
}
__device__ __forceinline__ void store6(float* __restrict__ p, const float v[6]) {
  f32x2* q = reinterpret_cast<f32x2*>(p);
  q[0] = f32x2{v[0], v[1]}; q[1] = f32x2{v[2], v[3]}; q[2] = f32x2{v[4], v[5]};
}
__device__ __forceinline__ void load_rot(const float* __restrict__ x6d, const float* __restrict__ Rin, int b, int j,
                                         float R[9], Rot6& c) {
  if (Rin) {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rin[((size_t)b * NJ + j) * 9 + k];
  } else {
    float xv[6];
    load6(x6d + ((size_t)b * NJ + j) * 6, xv);
    rot6d_fwd(xv, R, c);
  }
}

__device__ __forceinline__ void rest_joint(const float* __restrict__ Jt, const float* __restrict__ JS, int j,
                                           const float beta[NB], float J[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float acc = Jt[j * 3 + c];
#pragma unroll
    for (int l = 0; l < NB; ++l) acc = fmaf(JS[(j * 3 + c) * NB + l], beta[l], acc);
    J[c] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// k_prep_fwd: 6-D rotation -> R, blend features FT [KFP][BP], kinematic chain -> skinning transforms
// AT [12][24][BP] (A_j = G_j - [0 | G_j.R J_j], entry e = r*4+c), R0T [9][BP].
// One thread per (pose, joint): block = 32 poses x 24 joints, lanes run over poses (coalesced
// feature-major stores).  The chain is walked level by level of the kinematic tree (depth <= 9 for
// SMPL) with the world transforms exchanged through LDS.  Poses b >= B are written as zeros.
// ------------------------------------------------------------------------------------------
constexpr int PP = 32;   // poses per block of the (pose, joint)-parallel kernels
constexpr int PREP_FWD_LDS = NJ * 15 * PP;      // floats: world transforms [24][12][PP] + rest joints [24][3][PP]

__device__ __forceinline__ void prep_fwd_body(int blk, const float* __restrict__ x6d, const float* __restrict__ Rin,
                                              const float* __restrict__ betas, const float* __restrict__ Jt,
                                              const float* __restrict__ JS, const Parents& par, float* __restrict__ FT,
                                              float* __restrict__ AT, float* __restrict__ R0T, int B, int BP,
                                              int32_t* step_inc, float* __restrict__ FTq, float* __restrict__ sh) {
  // FTq: the same features in K-quads [KFP / 4][BP][4] -- the B operand of k_lbs_fwd's blend product (one 16-byte LDS read = four
  // K steps); the row-major FT stays for the chain adjoint and the folded-regressor product
  // sh: PREP_FWD_LDS floats of the caller's LDS (the composed kernel k_sup_step hands every phase the same pool)
  float (*Gs)[12][PP] = reinterpret_cast<float (*)[12][PP]>(sh);
  float (*Js)[3][PP] = reinterpret_cast<float (*)[3][PP]>(sh + NJ * 12 * PP);
  const int bl = threadIdx.x & (PP - 1), j = threadIdx.x / PP;
  const int b = blk * PP + bl;
  auto fq = [&](int k) -> float& { return FTq[((size_t)(k >> 2) * BP + b) * 4 + (k & 3)]; };
  if (step_inc && blk == 0 && threadIdx.x == 0) step_inc[0] += 1;   // Adam step count of this iteration
  const bool ok = b < B;
  float R[9], J[3] = {0.f, 0.f, 0.f}, beta[NB];
  Rot6 c;
  if (ok) {
    load_rot(x6d, Rin, b, j, R, c);
#pragma unroll
    for (int l = 0; l < NB; ++l) beta[l] = betas[(size_t)b * NB + l];
    rest_joint(Jt, JS, j, beta, J);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = 0.f;
#pragma unroll
    for (int l = 0; l < NB; ++l) beta[l] = 0.f;
  }
  // blend features: F[(j-1)*9 + k] = R_j[k] - I  (j >= 1), shape rows, template row, zero padding
  if (j > 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float f = ok ? R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f) : 0.f;
      FT[(size_t)((j - 1) * 9 + k) * BP + b] = f;
      fq((j - 1) * 9 + k) = f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) R0T[(size_t)k * BP + b] = R[k];
#pragma unroll
    for (int l = 0; l < NB; ++l) { FT[(size_t)(207 + l) * BP + b] = beta[l]; fq(207 + l) = beta[l]; }
    FT[(size_t)217 * BP + b] = ok ? 1.f : 0.f;
    fq(217) = ok ? 1.f : 0.f;
    for (int k = KF; k < KFP; ++k) { FT[(size_t)k * BP + b] = 0.f; fq(k) = 0.f; }
  }
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) Js[j][cc][bl] = J[cc];
  __syncthreads();
  // world transforms, one tree level per step
  float G[12];
  const int dj = par.depth[j];
  for (int d = 0; d <= par.maxd; ++d) {
    if (dj == d) {
      if (d == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) G[r * 4 + cc] = R[r * 3 + cc];
          G[r * 4 + 3] = J[r];
        }
      } else {
        const int p = par.p[j];
        float Gp[12], rel[3];
#pragma unroll
        for (int e = 0; e < 12; ++e) Gp[e] = Gs[p][e][bl];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) rel[cc] = J[cc] - Js[p][cc][bl];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            G[r * 4 + cc] = Gp[r * 4 + 0] * R[0 * 3 + cc] + Gp[r * 4 + 1] * R[1 * 3 + cc] + Gp[r * 4 + 2] * R[2 * 3 + cc];
          G[r * 4 + 3] = Gp[r * 4 + 0] * rel[0] + Gp[r * 4 + 1] * rel[1] + Gp[r * 4 + 2] * rel[2] + Gp[r * 4 + 3];
        }
      }
#pragma unroll
      for (int e = 0; e < 12; ++e) Gs[j][e][bl] = G[e];
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) AT[(size_t)((r * 4 + cc) * NJ + j) * BP + b] = ok ? G[r * 4 + cc] : 0.f;
    AT[(size_t)((r * 4 + 3) * NJ + j) * BP + b] =
        ok ? G[r * 4 + 3] - (G[r * 4 + 0] * J[0] + G[r * 4 + 1] * J[1] + G[r * 4 + 2] * J[2]) : 0.f;
  }
}

__global__ __launch_bounds__(PP * NJ) void k_prep_fwd(const float* __restrict__ x6d, const float* __restrict__ Rin,
                                                      const float* __restrict__ betas, const float* __restrict__ Jt,
                                                      const float* __restrict__ JS, Parents par, float* __restrict__ FT,
                                                      float* __restrict__ AT, float* __restrict__ R0T, int B, int BP,
                                                      int32_t* step_inc, float* __restrict__ FTq) {
  __shared__ float sh[PREP_FWD_LDS];
  prep_fwd_body(blockIdx.x, x6d, Rin, betas, Jt, JS, par, FT, AT, R0T, B, BP, step_inc, FTq, sh);
}

// The 24 posed SMPL joints of the most recent chain forward (smplx lbs(): J_transformed = G_j[:3, 3], the `joints` field of the
// operator's output, /root/reference/scripts/smpl.py:69-84): G_j t = A_j t + A_j R . J_j(beta) from the stored skinning transforms
// AT [12][24][BP] and the folded rest-joint tables.  One thread per (pose, joint); out (B,24,3).
__global__ __launch_bounds__(256) void k_posed_joints(const float* __restrict__ AT, const float* __restrict__ betas,
                                                      const float* __restrict__ Jt, const float* __restrict__ JS,
                                                      float* __restrict__ out, int B, int BP) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // b fastest: coalesced reads of AT
  if (idx >= B * NJ) return;
  const int b = idx % B, j = idx / B;
  float beta[NB], J[3];
#pragma unroll
  for (int l = 0; l < NB; ++l) beta[l] = betas[(size_t)b * NB + l];
  rest_joint(Jt, JS, j, beta, J);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float acc = AT[(size_t)((r * 4 + 3) * NJ + j) * BP + b];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc = fmaf(AT[(size_t)((r * 4 + c) * NJ + j) * BP + b], J[c], acc);
    out[((size_t)b * NJ + j) * 3 + r] = acc;
  }
}
// Adjoint of k_posed_joints: joints24[b][j][r] = A_{r,3}[j] + sum_c A_{r,c}[j] J_j(beta)[c]  ->  the skinning-transform adjoint dA^T
// [288][BP] (ONE slab for k_chain_bwd) and the direct shape term through the rest joints, gb (B,10) = sum_{j,c} (sum_r A_{r,c}[j] dj_r) JS
// (k_chain_bwd's gb_extra).  One thread per pose (a surface path: smpl(...).joints of scripts/smpl.py:69-84 is read by no caller).
__global__ __launch_bounds__(256) void k_posed_joints_bwd(const float* __restrict__ AT, const float* __restrict__ betas,
                                                          const float* __restrict__ Jt, const float* __restrict__ JS,
                                                          const float* __restrict__ dj24, float* __restrict__ dA,
                                                          float* __restrict__ gb, int B, int BP) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float beta[NB], g[NB];
#pragma unroll
  for (int l = 0; l < NB; ++l) { beta[l] = betas[(size_t)b * NB + l]; g[l] = 0.f; }
  for (int j = 0; j < NJ; ++j) {
    float J[3], d[3];
    rest_joint(Jt, JS, j, beta, J);
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = dj24[((size_t)b * NJ + j) * 3 + r];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dA[(size_t)((r * 4 + c) * NJ + j) * BP + b] = d[r] * J[c];
      dA[(size_t)((r * 4 + 3) * NJ + j) * BP + b] = d[r];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float dJ = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) dJ = fmaf(AT[(size_t)((r * 4 + c) * NJ + j) * BP + b], d[r], dJ);
#pragma unroll
      for (int l = 0; l < NB; ++l) g[l] = fmaf(dJ, JS[(j * 3 + c) * NB + l], g[l]);
    }
  }
#pragma unroll
  for (int l = 0; l < NB; ++l) gb[(size_t)b * NB + l] = g[l];
}
int launch_posed_joints_bwd(const Model& m, const float* AT, const float* betas, const float* dj24, float* dA, float* gb, int B, int BP,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_posed_joints_bwd, dim3((B + 255) / 256), dim3(256), 0, s, AT, betas, m.Jt, m.JS, dj24, dA, gb, B, BP);
  return 0;
}
int launch_posed_joints(const Model& m, const float* AT, const float* betas, float* out, int B, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_posed_joints, dim3((B * NJ + 255) / 256), dim3(256), 0, s, AT, betas, m.Jt, m.JS, out, B, BP);
  return 0;
}

// Horizontal fusion (fused inner loop with the pose discriminator): the chain forward and the discriminator's per-joint MLP
// both depend on x6d only and are both latency-bound with few workgroups (128 + 256 at 4096 poses) -- one launch, different
// CUs.  Every launch costs ~3.8 us of fixed time on this stack (a one-thread kernel measures that), so two independent small
// kernels side by side in one launch save one of those and the shorter kernel's whole duration.
struct DconvFwdArgs { const float* img; float* H2T; float* out; };
__global__ __launch_bounds__(PP * NJ) void k_prep_fwd_dconv(const float* __restrict__ x6d, const float* __restrict__ betas,
                                                            const float* __restrict__ Jt, const float* __restrict__ JS,
                                                            Parents par, float* __restrict__ FT, float* __restrict__ AT,
                                                            float* __restrict__ R0T, int B, int BP, int32_t* step_inc,
                                                            DconvFwdArgs d, int nprep, float* __restrict__ FTq) {
  // (one pool: a workgroup is either a chain-forward or a per-joint-MLP workgroup)
  __shared__ __attribute__((aligned(16))) float L[PREP_FWD_LDS > CL_FLOATS ? PREP_FWD_LDS : CL_FLOATS];
  if ((int)blockIdx.x < nprep) prep_fwd_body(blockIdx.x, x6d, nullptr, betas, Jt, JS, par, FT, AT, R0T, B, BP, step_inc, FTq, L);
  else dconv_fwd_body<true>(L, blockIdx.x - nprep, d.img, x6d, d.H2T, d.out, B, BP);
}

// ------------------------------------------------------------------------------------------
// Adam (torch single-tensor formula): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps), bias corrections evaluated in double like
// torch's Python scalars.
// ------------------------------------------------------------------------------------------
// (AdamScalars / adam_scalars / adam_update: jrr_common.h -- shared with the fused J-step update of lbs.hip)

// ------------------------------------------------------------------------------------------
// Perspective projection of the regressed joints (scripts/renderer.py:35-49 with pytorch3d 0.3.0
// PerspectiveCameras, R = I, T = cam, focal 5000/224 in NDC, principal point 0, 224x224 screen;
// SURVEY.md Appendix B):  X = -2x + tx, Y = -2y + ty, Z = 2z + tz ;  x_ndc = f X / Z ;
// x_screen = (W-1)/2 (1 - x_ndc)  (same for y).
// ------------------------------------------------------------------------------------------
// (project_point / project_point_bwd / Reproj: proj.h -- shared with the support-vertex iteration, supk.h)

// ------------------------------------------------------------------------------------------
// k_joints_loss: joints (from the reduced partials [3][17][BP]) -> joints (B,17,3), per-pose squared
// error, and the adjoint dJT [3][18][BP] of
//     weight * mean((move_pelvis(j) - gt/1000)^2)   [+ weight2d * mean((gt_j2d - project(j, cam))^2)]
//   scale = 2*weight/(batch_norm*51).  If djoints_in != NULL it is used as the adjoint instead
//   (operator-level backward), transposed into dJT.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PP * NH) void k_joints_loss(const float* __restrict__ JP, int nvc, int jp_rows,
                                                         const float* __restrict__ gt_mm,
                                                         const float* __restrict__ djoints_in, float scale,
                                                         float* __restrict__ joints_out, float* __restrict__ sqerr,
                                                         float* __restrict__ dJT, Reproj rp, int B, int BP,
                                                         const int* __restrict__ one_slab_flag) {
  // one thread per (pose, H36M joint): block = 32 poses x 17 joints, lanes over poses
  if (one_slab_flag && *one_slab_flag) nvc = 1;       // the support-restricted re-regression wrote ONE complete slab
  __shared__ float red[NH][8][PP];   // per-joint partials: 0..2 g, 3 err, 4 e2d, 5..7 gcam
  __shared__ float pel[3][PP];
  const int bl = threadIdx.x & (PP - 1), i = threadIdx.x / PP;
  const int b = blockIdx.x * PP + bl;
  const bool ok = b < B;
  float j[3] = {0.f, 0.f, 0.f};
  if (JP) {
    // per-vertex-chunk partials, summed in chunk order; four chunks' loads (12) in flight together
    int ch = 0;
    for (; ch + 4 <= nvc; ch += 4) {
      float t[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[u][c] = JP[(size_t)(((ch + u) * 3 + c) * jp_rows + i) * BP + b];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) j[c] += t[u][c];
    }
    for (; ch < nvc; ++ch)
#pragma unroll
      for (int c = 0; c < 3; ++c) j[c] += JP[(size_t)((ch * 3 + c) * jp_rows + i) * BP + b];
    if (joints_out && ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) joints_out[((size_t)b * NH + i) * 3 + c] = j[c];
    }
  }
  if (djoints_in) {      // operator-level backward: transpose the caller's adjoint
    if (dJT) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dJT[(size_t)(c * NHP + i) * BP + b] = ok ? djoints_in[((size_t)b * NH + i) * 3 + c] : 0.f;
        if (i == 0) dJT[(size_t)(c * NHP + NH) * BP + b] = 0.f;
      }
    }
    return;
  }
  if (!gt_mm) return;
  if (i == 0) { pel[0][bl] = j[0]; pel[1][bl] = j[1]; pel[2][bl] = j[2]; }
  __syncthreads();
  float g[3], err = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float gtv = ok ? gt_mm[((size_t)b * NH + i) * 3 + c] / 1000.f : 0.f;
    // joint 0: the centred value is identically 0; gt is pelvis-centred by the caller (optimize.py:162)
    const float d = (i == 0) ? -gtv : (j[c] - pel[c][bl]) - gtv;
    err += d * d;
    g[c] = (i == 0) ? 0.f : scale * d;
    red[i][c][bl] = g[c];
  }
  float e2 = 0.f, gc[3] = {0.f, 0.f, 0.f}, g2[3] = {0.f, 0.f, 0.f};
  if (rp.gt_j2d && ok) {     // 2-D reprojection term on the UN-centred joints (optimize.py:231-233)
    float t[3] = {rp.cam[(size_t)b * 3], rp.cam[(size_t)b * 3 + 1], rp.cam[(size_t)b * 3 + 2]};
    float xs, ys, invZ, X, Y;
    project_point(j, t, xs, ys, invZ, X, Y);
    const float dx = xs - rp.gt_j2d[((size_t)b * NH + i) * 2], dy = ys - rp.gt_j2d[((size_t)b * NH + i) * 2 + 1];
    e2 = dx * dx + dy * dy;
    project_point_bwd(rp.scale2d * dx, rp.scale2d * dy, invZ, X, Y, g2, gc);
  }
  red[i][3][bl] = err;
  red[i][4][bl] = e2;
#pragma unroll
  for (int c = 0; c < 3; ++c) red[i][5 + c][bl] = gc[c];
  __syncthreads();
  if (i == 0) {       // fixed-order sums over the 17 joints (deterministic)
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NH; ++q)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += red[q][k][bl];
    if (ok) {
      if (sqerr) sqerr[b] = s[3];
      if (rp.gt_j2d) {
        if (rp.sq2d) rp.sq2d[b] = s[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) rp.gcam[(size_t)b * 3 + c] = s[5 + c];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = -s[c];     // move_pelvis adjoint: -sum_i g_i
  }
  if (dJT) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dJT[(size_t)(c * NHP + i) * BP + b] = ok ? g[c] + g2[c] : 0.f;
      if (i == 0) dJT[(size_t)(c * NHP + NH) * BP + b] = 0.f;
    }
  }
}

// return_2d_joints core: joints (B,17,3), cam (B,3) -> screen coordinates (B,17,2)
__global__ void k_project_joints(const float* __restrict__ joints, const float* __restrict__ cam, float* __restrict__ out,
                                 int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over B*17
  if (idx >= n) return;
  const int b = idx / NH;
  float p[3] = {joints[(size_t)idx * 3], joints[(size_t)idx * 3 + 1], joints[(size_t)idx * 3 + 2]};
  float t[3] = {cam[(size_t)b * 3], cam[(size_t)b * 3 + 1], cam[(size_t)b * 3 + 2]};
  float xs, ys, invZ, X, Y;
  project_point(p, t, xs, ys, invZ, X, Y);
  out[(size_t)idx * 2] = xs;
  out[(size_t)idx * 2 + 1] = ys;
}

// Camera pre-fit (scripts/optimize.py:187-199): n Adam steps on the camera translation only, against the
// 2-D joints.  The reference re-runs the whole SMPL forward every step although the joints do not depend
// on the camera; here the joints are computed once and each thread runs its pose's n steps in registers.
__global__ __launch_bounds__(64) void k_camera_fit(const float* __restrict__ joints, const float* __restrict__ gt_j2d, float* __restrict__ cam,
                             float scale2d, int nsteps, float lr, float* __restrict__ sq2d, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float j[NH][3], q[NH][2];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) j[i][c] = joints[((size_t)b * NH + i) * 3 + c];
    q[i][0] = gt_j2d[((size_t)b * NH + i) * 2];
    q[i][1] = gt_j2d[((size_t)b * NH + i) * 2 + 1];
  }
  float t[3] = {cam[(size_t)b * 3], cam[(size_t)b * 3 + 1], cam[(size_t)b * 3 + 2]};
  float m[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f};
  double b1p = 1.0, b2p = 1.0;
  float e2 = 0.f;
  for (int s = 1; s <= nsteps; ++s) {
    float gt3[3] = {0.f, 0.f, 0.f}, dummy[3] = {0.f, 0.f, 0.f};
    e2 = 0.f;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      float xs, ys, invZ, X, Y;
      project_point(j[i], t, xs, ys, invZ, X, Y);
      const float dx = xs - q[i][0], dy = ys - q[i][1];
      e2 += dx * dx + dy * dy;
      project_point_bwd(scale2d * dx, scale2d * dy, invZ, X, Y, dummy, gt3);
    }
    b1p *= 0.9; b2p *= 0.999;
    AdamScalars sc;
    sc.step_size = (float)((double)lr / (1.0 - b1p));
    sc.bc2_sqrt = (float)sqrt(1.0 - b2p);
    sc.beta1 = 0.9f; sc.beta2 = 0.999f; sc.eps = 1e-8f;
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = adam_update(t[c], gt3[c], m[c], v[c], sc);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) cam[(size_t)b * 3 + c] = t[c];
  if (sq2d) sq2d[b] = e2;     // error at the LAST evaluated camera (before the final update), as the reference's loss
}

// standalone joint loss on (B,17,3) joints (operator-level API: jrr_joint_loss)
__global__ void k_joint_loss_plain(const float* __restrict__ joints, const float* __restrict__ gt_mm, float scale,
                                   float* __restrict__ sqerr, float* __restrict__ djoints, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float err = 0.f;
  for (int c = 0; c < 3; ++c) {
    float j0 = joints[((size_t)b * NH) * 3 + c];
    float gsum = 0.f;
    for (int i = 1; i < NH; ++i) {
      float d = (joints[((size_t)b * NH + i) * 3 + c] - j0) - gt_mm[((size_t)b * NH + i) * 3 + c] / 1000.f;
      err += d * d;
      float g = scale * d;
      gsum += g;
      if (djoints) djoints[((size_t)b * NH + i) * 3 + c] = g;
    }
    float d0 = -gt_mm[((size_t)b * NH) * 3 + c] / 1000.f;
    err += d0 * d0;
    if (djoints) djoints[((size_t)b * NH) * 3 + c] = -gsum;
  }
  if (sqerr) sqerr[b] = err;
}

__global__ void k_adam_flat(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, const int32_t* __restrict__ step, float lr, float beta1,
                            float beta2, float eps, int step_plus) {
  __shared__ AdamScalars sc;
  // step_plus = 1: the counter still holds the number of COMPLETED steps; a later launch of the same stream increments it (the J
  // step: k_jreg_rowsum -- one launch less than a separate increment kernel before this one)
  if (threadIdx.x == 0) sc = adam_scalars(step[0] + step_plus, lr, beta1, beta2, eps);
  __syncthreads();
  AdamScalars s = sc;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float mm = m[i], vv = v[i];
    p[i] = adam_update(p[i], g[i], mm, vv, s);
    m[i] = mm;
    v[i] = vv;
  }
}

// ------------------------------------------------------------------------------------------
// Adjoint of k_prep_fwd, in two kernels:
//
// k_chain_bwd (one pose per lane, children before parents): dL/dA^T [288][BP] and dL/dF^T [224][BP]
//   -> dL/dR^T [24*9][BP] and dL/dbeta^T [10][BP].  Every global access is pose-contiguous
//   (coalesced): R comes back from the feature rows F^T (+ identity) / R0T, the world rotations
//   G_j.R from the rotation part of A^T, so nothing of the forward chain is recomputed.  LDS holds
//   only the dG accumulators (12 floats per joint per lane).
//
// The same threads then finish the per-pose update (pose_update_*): 6-D rotation adjoint, extra (discriminator)
//   gradients, then either the gradient outputs or the fused Adam update -- dL/dR never leaves registers.
// ------------------------------------------------------------------------------------------
struct PoseUpdateArgs {
  const float* x6d_in;      // (B,24,6) or NULL (R mode)
  const float* gx_extra; const float* gb_extra;
  float* dx6d; float* dR; float* dbetas;                 // gradient outputs (nullable)
  float* x6d_io; float* betas_io; float* adam_m; float* adam_v; const int32_t* step;   // Adam (x6d_io nullable)
  float lr, beta1, beta2, eps;
  const float* gcam; float* cam_io; float* cam_m; float* cam_v;     // camera translation (2-D term), nullable
};

// joint j of pose b: dL/dR_j (9) -> 6-D rotation adjoint (+ the discriminator's gradient) -> gradient output or Adam
__device__ __forceinline__ void pose_update_joint(const PoseUpdateArgs& a, int b, int j, const float dRi[9], const AdamScalars& sc) {
  if (!a.x6d_in) {
    if (a.dR) {
#pragma unroll
      for (int k = 0; k < 9; ++k) a.dR[((size_t)b * NJ + j) * 9 + k] = dRi[k];
    }
    return;
  }
  float xv[6], R[9], dx[6];
  Rot6 c;
  load6(a.x6d_in + ((size_t)b * NJ + j) * 6, xv);
  rot6d_fwd(xv, R, c);
  rot6d_bwd(c, dRi, dx);
  if (a.gx_extra) {
    float gx[6];
    load6(a.gx_extra + ((size_t)b * NJ + j) * 6, gx);
#pragma unroll
    for (int k = 0; k < 6; ++k) dx[k] += gx[k];
  }
  if (a.dx6d) store6(a.dx6d + ((size_t)b * NJ + j) * 6, dx);
  if (a.x6d_io) {
    const size_t si = (size_t)b * NPARAM + j * 6;
    float mm[6], vv[6], xn[6];
    load6(a.adam_m + si, mm);
    load6(a.adam_v + si, vv);
#pragma unroll
    for (int k = 0; k < 6; ++k) xn[k] = adam_update(xv[k], dx[k], mm[k], vv[k], sc);
    store6(a.x6d_io + ((size_t)b * NJ + j) * 6, xn);
    store6(a.adam_m + si, mm);
    store6(a.adam_v + si, vv);
  }
}
// shape coefficient l of pose b
__device__ __forceinline__ void pose_update_beta(const PoseUpdateArgs& a, int b, int l, float g, const AdamScalars& sc) {
  if (a.gb_extra) g += a.gb_extra[(size_t)b * NB + l];
  if (a.dbetas) a.dbetas[(size_t)b * NB + l] = g;
  if (a.x6d_io) {
    const size_t si = (size_t)b * NPARAM + JRR_POSE6D + l;
    float mm = a.adam_m[si], vv = a.adam_v[si];
    a.betas_io[(size_t)b * NB + l] = adam_update(a.betas_io[(size_t)b * NB + l], g, mm, vv, sc);
    a.adam_m[si] = mm;
    a.adam_v[si] = vv;
  }
}
// camera component cc of pose b: Adam over [pose, orient, betas, cam] (optimize.py:201-202)
__device__ __forceinline__ void pose_update_cam(const PoseUpdateArgs& a, int b, int cc, const AdamScalars& sc) {
  const size_t si = (size_t)b * 3 + cc;
  float mm = a.cam_m[si], vv = a.cam_v[si];
  a.cam_io[si] = adam_update(a.cam_io[si], a.gcam[si], mm, vv, sc);
  a.cam_m[si] = mm;
  a.cam_v[si] = vv;
}

constexpr int PPB = 32;   // poses per block of k_chain_bwd (16 halves the coalescing width: measured 1.5x slower)
// LDS pool of the chain adjoint (floats): dG | Js | dBs | Jts | JSs | adam scalars (8) | child list (24 bytes in 8 floats)
constexpr int CB_DG = 0, CB_JS = CB_DG + NJ * 12 * PPB, CB_DBS = CB_JS + NJ * 3 * PPB, CB_JTS = CB_DBS + NJ * NB * PPB,
              CB_JSS = CB_JTS + NJ * 3, CB_ADAM = CB_JSS + NJ * 3 * NB, CB_CHILD = CB_ADAM + 8, CHAIN_BWD_LDS = CB_CHILD + 8;
static_assert(sizeof(AdamScalars) <= 8 * sizeof(float), "adam scalars slot");
__device__ __forceinline__ void chain_bwd_body(int blk, const float* __restrict__ FT, const float* __restrict__ R0T,
                                               const float* __restrict__ AT, const float* __restrict__ Jt,
                                               const float* __restrict__ JS, const Parents& par,
                                               const float* __restrict__ dA_, int nslabA, size_t strideA,
                                               const float* __restrict__ dF_, const PoseUpdateArgs& ua, int B, int BP,
                                               const unsigned* __restrict__ dmask, float* __restrict__ sh, int step_now = -1) {
  // step_now >= 0 (thread 0 only; the composed kernel k_sup_step): Adam's step count of this iteration, instead of ua.step[0]
  constexpr int PP = PPB;
  AdamScalars& adam_sc = *reinterpret_cast<AdamScalars*>(sh + CB_ADAM);
  // one thread per (pose, joint); levels of the kinematic tree are processed deepest first.  Each child
  // leaves its contribution to the parent's dG in its own LDS slot; the parent sums its children in
  // index order (no atomics: bitwise reproducible).
  float (*dG)[12][PP] = reinterpret_cast<float (*)[12][PP]>(sh + CB_DG);
  float (*Js)[3][PP] = reinterpret_cast<float (*)[3][PP]>(sh + CB_JS);
  float (*dBs)[NB][PP] = reinterpret_cast<float (*)[NB][PP]>(sh + CB_DBS);
  const int bl = threadIdx.x & (PP - 1), j = threadIdx.x / PP;
  const int b = blk * PP + bl;
  const bool ok = b < B;
  const size_t bb = ok ? (size_t)b : 0;      // padded lanes read pose 0 and write nothing
  const int p = par.p[j];
  // this joint's children (ascending) from the model's child lists.  (Searching them -- "for q > j: if parent[q] == j"
  // -- indexes the kernel arguments with a per-lane q: 23 dependent memory loads per working wave and tree level,
  // ~2 us per level, measured with wall-clock stamps: 17.6 of the kernel's 34 us.)
  const int c_lo = par.child_off[j], c_hi = par.child_off[j + 1];
  unsigned char* const childs = reinterpret_cast<unsigned char*>(sh + CB_CHILD);
  if (threadIdx.x < NJ) childs[threadIdx.x] = par.child[threadIdx.x];
  // the rest-joint tables (72 + 720 floats) are read with per-lane (joint) indices: from LDS, not from memory
  float* const Jts = sh + CB_JTS;
  float* const JSs = sh + CB_JSS;
  for (int i = threadIdx.x; i < NJ * 3 * NB; i += PP * NJ) JSs[i] = JS[i];
  if (threadIdx.x < NJ * 3) Jts[threadIdx.x] = Jt[threadIdx.x];
  float beta[NB], dbeta[NB], Ji[3];
  // Row addressing of the [row][BP] arrays: a scalar base + ONE 32-bit byte offset per lane, rows stepped by scalar
  // multiples of the row pitch (a 64-bit per-lane multiply-add per element -- what `array[(size_t)row * BP + b]` with a
  // per-lane row compiles to -- is quarter-rate: ~400 VALU instructions for this kernel's 94 loads).
  const unsigned pitch = (unsigned)BP * 4u, lane_b = (unsigned)bb * 4u;
  auto ld = [](const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
  };
#pragma unroll
  for (int l = 0; l < NB; ++l) { beta[l] = ld(FT, (unsigned)(207 + l) * pitch + lane_b); dbeta[l] = 0.f; }
  // everything this joint needs from global memory, issued up front (independent, pose-contiguous; joint 0 reads
  // valid rows it does not use instead of branching around the loads)
  float dA[12], GiR[9], R[9], Gp[9], dFj[9];
  const unsigned oj = (unsigned)j * pitch + lane_b;                         // row j
  const unsigned opj = (unsigned)(p > 0 ? p : 0) * pitch + lane_b;          // row of the parent
  const unsigned ofj = (unsigned)(j > 0 ? (j - 1) * 9 : 0) * pitch + lane_b;   // first pose-feature row of joint j
  // dA^T arrives as per-vertex-chunk partial slabs [nslabA][12][24][BP]: summed here, in slab order (deterministic)
  // (two slabs' loads -- 24 -- in flight together).  dmask (k_lbs_bwd16): bit j of word [slab][b / 64] says whether the slab holds
  // rows for joint j at all -- rows of joints its chunk never touched are neither written nor read (x + 0 = x: same sums)
  {
    const int n_bt = BP / 64, bt = (int)(blk * PP) / 64;
#pragma unroll
    for (int e = 0; e < 12; ++e) dA[e] = 0.f;
    for (int sl = 0; sl < nslabA; sl += 2) {
      float t[12], u[12];
      const bool two = sl + 1 < nslabA;
      const bool v0 = !dmask || ((dmask[sl * n_bt + bt] >> j) & 1u);
      const bool v1 = two && (!dmask || ((dmask[(sl + 1) * n_bt + bt] >> j) & 1u));
      const float* s0 = dA_ + (size_t)sl * strideA;
      const float* s1 = s0 + strideA;
#pragma unroll
      for (int e = 0; e < 12; ++e) t[e] = v0 ? ld(s0, oj + (unsigned)(e * NJ) * pitch) : 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) u[e] = v1 ? ld(s1, oj + (unsigned)(e * NJ) * pitch) : 0.f;
#pragma unroll
      for (int e = 0; e < 12; ++e) dA[e] += t[e];
      if (two) {
#pragma unroll
        for (int e = 0; e < 12; ++e) dA[e] += u[e];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      GiR[r * 3 + cc] = ld(AT, oj + (unsigned)((r * 4 + cc) * NJ) * pitch);
      const float gp = ld(AT, opj + (unsigned)((r * 4 + cc) * NJ) * pitch);
      Gp[r * 3 + cc] = (j > 0) ? gp : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float f = ld(FT, ofj + (unsigned)k * pitch), df = ld(dF_, ofj + (unsigned)k * pitch);
    R[k] = (j > 0) ? f + ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f) : 0.f;
    dFj[k] = (j > 0) ? df : 0.f;
  }
  // Adam's bias corrections (two fp64 pow: ~400 instructions, one thread) under the loads issued above, not before them
  if (ua.x6d_io && threadIdx.x == 0) adam_sc = adam_scalars(step_now >= 0 ? step_now : ua.step[0], ua.lr, ua.beta1, ua.beta2, ua.eps);
  __syncthreads();                              // tables staged
  rest_joint(Jts, JSs, j, beta, Ji);
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) Js[j][cc][bl] = Ji[cc];
  __syncthreads();

  float dRi[9];
  const int dj = par.depth[j];
  for (int d = par.maxd; d >= 0; --d) {
    if (dj == d) {
      // dG_i = own (A.R = G.R ; A.t = G.t - G.R J) + the contributions of the children (one level deeper)
      float dGi[12];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) dGi[r * 4 + cc] = dA[r * 4 + cc] - dA[r * 4 + 3] * Ji[cc];
        dGi[r * 4 + 3] = dA[r * 4 + 3];
      }
      for (int k = c_lo; k < c_hi; ++k) {
        const int q = childs[k];
#pragma unroll
        for (int e = 0; e < 12; ++e) dGi[e] += dG[q][e][bl];
      }
      float dJ[3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) dJ[cc] = -(GiR[0 * 3 + cc] * dA[3] + GiR[1 * 3 + cc] * dA[7] + GiR[2 * 3 + cc] * dA[11]);
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) dRi[r * 3 + cc] = dGi[r * 4 + cc];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) dJ[cc] += dGi[cc * 4 + 3];   // G_0.t = J_0
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
          for (int l = 0; l < NB; ++l) dbeta[l] = fmaf(dJ[cc], JSs[(0 * 3 + cc) * NB + l], dbeta[l]);
      } else {
        float rel[3], drel[3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) rel[cc] = Ji[cc] - Js[p][cc][bl];
        // G_i.R = Gp.R R_i ; G_i.t = Gp.R rel + Gp.t ;  F[(i-1)*9+k] = R_i[k] - I
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc)
            dRi[r * 3 + cc] = Gp[0 * 3 + r] * dGi[0 * 4 + cc] + Gp[1 * 3 + r] * dGi[1 * 4 + cc] + Gp[2 * 3 + r] * dGi[2 * 4 + cc] +
                              dFj[r * 3 + cc];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          drel[cc] = Gp[0 * 3 + cc] * dGi[0 * 4 + 3] + Gp[1 * 3 + cc] * dGi[1 * 4 + 3] + Gp[2 * 3 + cc] * dGi[2 * 4 + 3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            const float add = dGi[r * 4 + 0] * R[cc * 3 + 0] + dGi[r * 4 + 1] * R[cc * 3 + 1] + dGi[r * 4 + 2] * R[cc * 3 + 2] +
                              dGi[r * 4 + 3] * rel[cc];
            dG[j][r * 4 + cc][bl] = add;                 // this joint's contribution to dG of its parent
          }
          dG[j][r * 4 + 3][bl] = dGi[r * 4 + 3];
        }
        // J_i enters through -G_i.R J_i (dJ) and rel = J_i - J_p (drel)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
          for (int l = 0; l < NB; ++l)
            dbeta[l] = fmaf(dJ[cc] + drel[cc], JSs[(j * 3 + cc) * NB + l], fmaf(-drel[cc], JSs[(p * 3 + cc) * NB + l], dbeta[l]));
      }
    }
    __syncthreads();
  }
  // dL/dbeta: sum the per-joint contributions of each pose (fixed order: deterministic)
#pragma unroll
  for (int l = 0; l < NB; ++l) dBs[j][l][bl] = dbeta[l];
  __syncthreads();           // (also orders thread 0's adam_sc before its readers)
  // ---- fused per-pose update (formerly a second kernel over the same (pose, joint) threads): 6-D rotation adjoint,
  //      the discriminators' extra gradients, then either the gradient outputs or torch's Adam in place ----
  const AdamScalars sc = adam_sc;
  if (ok) {
    pose_update_joint(ua, b, j, dRi, sc);
    if (j < NB) {
      float acc = dF_[(size_t)(207 + j) * BP + b];
#pragma unroll
      for (int q = 0; q < NJ; ++q) acc += dBs[q][j][bl];
      pose_update_beta(ua, b, j, acc, sc);
    } else if (j < NB + 3 && ua.x6d_io && ua.cam_io) {
      pose_update_cam(ua, b, j - NB, sc);
    }
  }
}

__global__ __launch_bounds__(PPB * NJ) void k_chain_bwd(const float* __restrict__ FT, const float* __restrict__ R0T,
                                                        const float* __restrict__ AT, const float* __restrict__ Jt,
                                                        const float* __restrict__ JS, Parents par,
                                                        const float* __restrict__ dA_, int nslabA, size_t strideA,
                                                        const float* __restrict__ dF_, PoseUpdateArgs ua, int B, int BP,
                                                        const unsigned* __restrict__ dmask) {
  __shared__ __attribute__((aligned(16))) float sh[CHAIN_BWD_LDS];
  chain_bwd_body(blockIdx.x, FT, R0T, AT, Jt, JS, par, dA_, nslabA, strideA, dF_, ua, B, BP, dmask, sh);
}

// out[i] (+)= sum_s P[s*stride + i]  (split-K / vertex-chunk partial slabs), float4-wide
__global__ void k_reduce_slabs(const f32x4* __restrict__ P, int nslab, size_t stride4, f32x4* __restrict__ out,
                               size_t n4, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 acc = P[i];
    for (int s = 1; s < nslab; ++s) acc += P[(size_t)s * stride4 + i];
    if (accumulate) acc += out[i];
    out[i] = acc;
  }
}

// two independent slab sums in ONE launch (small batches: the dA^T slabs of k_lbs_bwd are too many for their consumer and get a
// sum of their own next to the dF^T one -- a launch costs ~4 us of stream time; batch 1024: 0.302 -> 0.292 ms per iteration):
// blocks [0, nb1) the first, the rest the second
__global__ void k_reduce_slabs2(const f32x4* __restrict__ P1, int nslab1, size_t stride1, f32x4* __restrict__ out1, size_t n1, int nb1,
                                const f32x4* __restrict__ P2, int nslab2, size_t stride2, f32x4* __restrict__ out2, size_t n2) {
  const bool first = (int)blockIdx.x < nb1;
  const f32x4* P = first ? P1 : P2;
  f32x4* out = first ? out1 : out2;
  const int nslab = first ? nslab1 : nslab2;
  const size_t stride4 = first ? stride1 : stride2, n4 = first ? n1 : n2;
  const size_t blk = first ? blockIdx.x : blockIdx.x - nb1, nb = first ? (size_t)nb1 : (size_t)gridDim.x - nb1;
  for (size_t i = blk * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x) {
    f32x4 acc = P[i];
    for (int s = 1; s < nslab; ++s) acc += P[(size_t)s * stride4 + i];
    out[i] = acc;
  }
}
int launch_reduce_slabs2(const float* P1, int nslab1, size_t stride1, float* out1, size_t n1, const float* P2, int nslab2,
                         size_t stride2, float* out2, size_t n2, hipStream_t s) {
  int b1 = (int)((n1 / 4 + 255) / 256), b2 = (int)((n2 / 4 + 255) / 256);
  if (b1 > 2048) b1 = 2048;
  if (b2 > 2048) b2 = 2048;
  hipLaunchKernelGGL(k_reduce_slabs2, dim3(b1 + b2), dim3(256), 0, s, (const f32x4*)P1, nslab1, stride1 / 4, (f32x4*)out1, n1 / 4, b1,
                     (const f32x4*)P2, nslab2, stride2 / 4, (f32x4*)out2, n2 / 4);
  return 0;
}

// Horizontal fusion of the two independent kernels that precede k_chain_bwd in the fused loop: the split-K slab sum of
// dF^T (bandwidth-bound, 58 MB) and the per-joint MLP adjoint of the discriminator (latency-bound).  Blocks [0, ndc) are the
// MLP workgroups (the longer dependency chain: dispatched first -- measured 2 us better than last), the rest stride over the
// slab sum.
struct DconvBwdArgs { const float* img; const float* x6d; const float* dH2T; const float* gout; float scale, target; float* gx; float* sqj; };
__global__ __launch_bounds__(256) void k_dconv_bwd_reduce(DconvBwdArgs d, int ndc, int B, int BP, const f32x4* __restrict__ P,
                                                          int nslab, size_t stride4, f32x4* __restrict__ out, size_t n4) {
  __shared__ __attribute__((aligned(16))) float L[CL_FLOATS];
  if ((int)blockIdx.x < ndc) {
    dconv_bwd_body<true>(L, blockIdx.x, d.img, d.x6d, d.dH2T, d.gout, d.scale, d.target, d.gx, B, BP, d.sqj);
  } else {
    const size_t nb = gridDim.x - ndc;
    for (size_t i = (size_t)(blockIdx.x - ndc) * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x) {
      f32x4 acc = P[i];
      for (int sl = 1; sl < nslab; ++sl) acc += P[(size_t)sl * stride4 + i];
      out[i] = acc;
    }
  }
}
int launch_dconv_bwd_reduce(const float* img, const float* x6d, const float* dH2T, const float* gout, float scale, float target,
                            float* gx, float* sqj, int B, int BP, const float* P, int nslab, size_t stride, float* out, size_t n,
                            hipStream_t s) {
  const size_t n4 = n / 4;
  int rblocks = (int)((n4 + 255) / 256);
  if (rblocks > 4096) rblocks = 4096;
  const int ndc = (BP / 32) * 6;
  DconvBwdArgs d{img, x6d, dH2T, gout, scale, target, gx, sqj};
  hipLaunchKernelGGL(k_dconv_bwd_reduce, dim3(ndc + rblocks), dim3(256), 0, s, d, ndc, B, BP, (const f32x4*)P, nslab, stride / 4,
                     (f32x4*)out, n4);
  return 0;
}

int launch_reduce_slabs(const float* P, int nslab, size_t stride, float* out, size_t n, hipStream_t s, int accumulate) {
  size_t n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_reduce_slabs, dim3(blocks), dim3(256), 0, s, (const f32x4*)P, nslab, stride / 4, (f32x4*)out, n4,
                     accumulate);
  return 0;
}

// ------------------------------------------------------------------------------------------
// k_sup_step: ONE inner iteration of a 32-pose group in one workgroup (round 6; supk.h) -- the support-vertex form of
// scripts/optimize.py:220-265 without the silhouette and 2-D terms:
//   chain forward (k_prep_fwd's body) -> support-vertex SMPL forward, joint loss, backward (sup_body) -> [per-joint MLP adjoint of the
//   pose discriminator, whose four GEMM launches PRECEDE this kernel] -> chain adjoint + 6-D rotation adjoint + Adam (k_chain_bwd's
//   body) -> [per-joint MLP forward of the NEXT iteration on the updated poses].
// The phases hand F^T / A^T / dA^T / dF^T / gx over through global memory (the arrays their stand-alone kernels use; a workgroup reads
// only what it wrote itself, behind a workgroup barrier) and share one LDS pool.  With the discriminator an iteration is 4 GEMM
// launches + this one (13 launches on the tile lists); without it, this launch alone.
// Adam's step count: every workgroup reads the pre-increment value; the LAST workgroup to arrive (an agent-scope counter in the
// engine workspace, reset by that workgroup) stores value + 1 -- after every other workgroup has read.
// ------------------------------------------------------------------------------------------
struct SupStepArgs {
  const float* x6d; const float* betas; float* FT; float* FTq; float* AT; float* R0T;
  SupArgs sup;
  const float* conv_img;     // LDS image of the per-joint MLP, NULL without the pose discriminator
  const float* dH2T; float dscale; float* gx; float* dsq; float* H2T_next;
  PoseUpdateArgs ua;
  int32_t* step; int* arrive;
  long long* stamps;         // experiments (tools/exp/sup_stamps.py): phase boundaries of workgroup 0 on the 100 MHz counter, NULL otherwise
};
constexpr int SUP_STEP_LDS = SUPL_FLOATS > CHAIN_BWD_LDS ? (SUPL_FLOATS > PREP_FWD_LDS ? SUPL_FLOATS : PREP_FWD_LDS)
                                                         : (CHAIN_BWD_LDS > PREP_FWD_LDS ? CHAIN_BWD_LDS : PREP_FWD_LDS);
static_assert(SUP_STEP_LDS >= CL_FLOATS && SUP_PP == PP && SUP_PP == PPB && SUP_THREADS == PP * NJ, "one geometry for every phase");
// PHASE 0: the whole iteration.  PHASE 1 / 2 (JRR_SUP_OVERLAP, api.hip): its two halves as launches of their own -- 1 = everything that
// does not read the discriminator GEMMs' results (chain forward, support-vertex forward / loss / backward), on a second stream beside
// those GEMMs; 2 = the rest (per-joint MLP adjoint, chain adjoint + Adam, the next iteration's per-joint MLP forward), behind both.
constexpr int SUP_TAIL_CONV = CHAIN_BWD_LDS > PREP_FWD_LDS ? CHAIN_BWD_LDS : PREP_FWD_LDS;      // phase 2: the MLP image behind the chain pool
constexpr int SUP_TAIL_LDS = SUP_TAIL_CONV + CL_FLOATS;
template <int PHASE>
__global__ __launch_bounds__(SUP_THREADS) void k_sup_step(SupStepArgs a, const float* __restrict__ Jt, const float* __restrict__ JS,
                                                          Parents par) {
  extern __shared__ __attribute__((aligned(16))) float pool[];
  const int blk = blockIdx.x, B = a.sup.B, BP = a.sup.BP;
  auto stamp = [&](int i) { if (PHASE == 0 && a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[i] = wall_clock64(); };
  stamp(0);
  int step_now = -1;
  if (PHASE != 1 && threadIdx.x == 0 && a.step) {
    const int s0 = __hip_atomic_load(a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the value is in: only now may this workgroup count as arrived
    step_now = s0 + 1;
    const int prev = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)gridDim.x - 1) {
      __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.step, step_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  constexpr int CONV_AT = PHASE == 2 ? SUP_TAIL_CONV : SUPL_CONV;
  if (PHASE != 2) {
    prep_fwd_body(blk, a.x6d, nullptr, a.betas, Jt, JS, par, a.FT, a.AT, a.R0T, B, BP, nullptr, a.FTq, pool);
    __syncthreads();
    stamp(1);
    sup_body(pool, blk, a.sup, a.stamps && blockIdx.x == 0 ? a.stamps + 8 : nullptr);
    if (PHASE == 1) return;
    __syncthreads();
    stamp(2);
    // (the per-joint MLP adjoint closed the support body: one interleaved pair of joints per wave, supk.h)
    stamp(3);
  } else if (a.conv_img) {      // phase 2 opens with that adjoint (the first phase's launch ran the support body without it)
    conv_stage_params(a.conv_img, pool + CONV_AT);
    __syncthreads();
    // (two single tiles per wave: outside sup_body the compiler's back end refuses dconv_bwd_pair, see k_tail_step)
    dconv_bwd_body<true, true>(pool + CONV_AT, 2 * blk, a.conv_img, a.x6d, a.dH2T, nullptr, a.dscale, 1.f, a.gx, B, BP, a.dsq);
    dconv_bwd_body<true, true>(pool + CONV_AT, 2 * blk + 1, a.conv_img, a.x6d, a.dH2T, nullptr, a.dscale, 1.f, a.gx, B, BP, a.dsq);
    __syncthreads();
  }
  chain_bwd_body(blk, a.FT, a.R0T, a.AT, Jt, JS, par, a.sup.dA, 1, 0, a.sup.dF, a.ua, B, BP, nullptr, pool, step_now);
  if (a.conv_img && a.H2T_next) {      // the updated poses of this group are complete: their per-joint MLP forward for the next iteration
    __syncthreads();
    stamp(4);
    // (the image the support body staged for the adjoint is still in place: the chain adjoint's pool ends far below it)
    static_assert(CHAIN_BWD_LDS <= SUPL_CONV && PREP_FWD_LDS <= SUPL_CONV, "the per-joint MLP image survives the chain phases");
    dconv_fwd_body<true, true>(pool + CONV_AT, 2 * blk, a.conv_img, a.ua.x6d_io, a.H2T_next, nullptr, B, BP);
    dconv_fwd_body<true, true>(pool + CONV_AT, 2 * blk + 1, a.conv_img, a.ua.x6d_io, a.H2T_next, nullptr, B, BP);
  }
  __syncthreads();
  stamp(5);
}

// ------------------------------------------------------------------------------------------
// k_tail_step (round 6): the END of an all-tiles iteration and the BEGINNING of the next one, per 32-pose group in one workgroup --
//   per-joint MLP adjoint (interleaved pairs) -> chain adjoint + 6-D rotation adjoint + Adam (k_chain_bwd's body)
//   -> [chain forward of the NEXT iteration on the updated poses (k_prep_fwd's body) -> its per-joint MLP forward].
// Replaces three launches of the loop (slab sum || MLP adjoint, k_chain_bwd, the next iteration's k_prep_fwd || MLP forward) by the slab
// sum + this one: the same device bodies, one LDS pool, the hand-overs through the arrays the stand-alone kernels use.  The step counter:
// every workgroup reads THIS iteration's count (already incremented by its chain forward, wherever that ran); when the next iteration's
// chain forward runs here, the last workgroup to arrive stores count + 1 for it -- after every other workgroup has read.
// ------------------------------------------------------------------------------------------
constexpr int TAIL_CONV = CHAIN_BWD_LDS > PREP_FWD_LDS ? CHAIN_BWD_LDS : PREP_FWD_LDS;      // the MLP image sits behind the chain pools
constexpr int TAIL_LDS = TAIL_CONV + CL_FLOATS;
struct TailStepArgs {
  const float* conv_img; const float* dH2T; float dscale; float* gx; float* dsq;      // per-joint MLP adjoint (conv_img NULL: none)
  const float* FT; const float* R0T; const float* AT;                                 // this iteration's chain forward
  const float* dA; int nslabA; size_t strideA; const float* dF; const unsigned* dmask;
  PoseUpdateArgs ua;
  int do_next; float* FTw; float* FTq; float* ATw; float* R0Tw; float* H2T_next;      // next iteration's chain forward (+ MLP forward)
  int32_t* step; int* arrive; int B, BP;
};
__global__ __launch_bounds__(PPB * NJ) void k_tail_step(TailStepArgs a, const float* __restrict__ Jt, const float* __restrict__ JS,
                                                        Parents par) {
  extern __shared__ __attribute__((aligned(16))) float pool[];
  const int blk = blockIdx.x, B = a.B, BP = a.BP;
  int step_now = -1;
  if (threadIdx.x == 0 && a.ua.x6d_io) {
    step_now = __hip_atomic_load(a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the value is in: only now may this workgroup count as arrived
    if (a.do_next) {
      const int prev = __hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == (int)gridDim.x - 1) {
        __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.step, step_now + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  float* const convL = pool + TAIL_CONV;
  if (a.conv_img) {
    conv_stage_params(a.conv_img, convL);
    __syncthreads();
    // (two single tiles per wave, not dconv_bwd_pair: in THIS kernel the compiler's back end refuses the pair's row pointers --
    // "illegal VGPR to SGPR copy" -- with any run-time joint index; the pair is 2.5 us faster where it compiles, k_sup_step)
    dconv_bwd_body<true, true>(convL, 2 * blk, a.conv_img, a.ua.x6d_in, a.dH2T, nullptr, a.dscale, 1.f, a.gx, B, BP, a.dsq);
    dconv_bwd_body<true, true>(convL, 2 * blk + 1, a.conv_img, a.ua.x6d_in, a.dH2T, nullptr, a.dscale, 1.f, a.gx, B, BP, a.dsq);
    __syncthreads();      // gx of this pose group is complete
  }
  chain_bwd_body(blk, a.FT, a.R0T, a.AT, Jt, JS, par, a.dA, a.nslabA, a.strideA, a.dF, a.ua, B, BP, a.dmask, pool, step_now);
  if (a.do_next) {
    __syncthreads();      // the updated poses of this group are complete, and nobody reads this iteration's F^T / A^T any more
    prep_fwd_body(blk, a.ua.x6d_io, nullptr, a.ua.betas_io, Jt, JS, par, a.FTw, a.ATw, a.R0Tw, B, BP, nullptr, a.FTq, pool);
    if (a.conv_img && a.H2T_next) {
      dconv_fwd_body<true, true>(convL, 2 * blk, a.conv_img, a.ua.x6d_io, a.H2T_next, nullptr, B, BP);
      dconv_fwd_body<true, true>(convL, 2 * blk + 1, a.conv_img, a.ua.x6d_io, a.H2T_next, nullptr, B, BP);
    }
  }
}

int launch_tail_step(const Model& m, const TailStepLaunch& q, const PrepBwdLaunch& L, hipStream_t s) {
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)k_tail_step, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS * 4) == hipSuccess;
  }();
  if (!attr) { jrr_set_error("k_tail_step: %d bytes of LDS refused", TAIL_LDS * 4); return JRR_ERR_HIP; }
  TailStepArgs a;
  a.conv_img = q.conv_img; a.dH2T = q.dH2T; a.dscale = q.dscale; a.gx = q.gx; a.dsq = q.dsq;
  a.FT = L.FT; a.R0T = L.R0T; a.AT = L.AT;
  a.dA = L.dATp; a.nslabA = L.nslabA; a.strideA = L.strideA; a.dF = L.dFTp; a.dmask = L.dmaskA;
  PoseUpdateArgs& u = a.ua;
  u.x6d_in = L.x6d_in; u.gx_extra = L.gx_extra; u.gb_extra = L.gb_extra;
  u.dx6d = L.dx6d; u.dR = L.dR; u.dbetas = L.dbetas;
  u.x6d_io = L.x6d_io; u.betas_io = L.betas_io; u.adam_m = L.adam_m; u.adam_v = L.adam_v; u.step = L.step;
  u.lr = L.lr; u.beta1 = L.beta1; u.beta2 = L.beta2; u.eps = L.eps;
  u.gcam = L.gcam; u.cam_io = L.cam_io; u.cam_m = L.cam_m; u.cam_v = L.cam_v;
  a.do_next = q.do_next ? 1 : 0; a.FTw = q.FTw; a.FTq = q.FTq; a.ATw = q.ATw; a.R0Tw = q.R0Tw; a.H2T_next = q.H2T_next;
  a.step = q.step; a.arrive = q.arrive; a.B = L.B; a.BP = L.BP;
  hipLaunchKernelGGL(k_tail_step, dim3((L.B + PPB - 1) / PPB), dim3(PPB * NJ), TAIL_LDS * 4, s, a, m.Jt, m.JS, m.parents);
  return 0;
}

int launch_sup_step(const Model& m, const SupStepLaunch& q, const PrepBwdLaunch& L, hipStream_t s, int phase) {
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)k_sup_step<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SUP_STEP_LDS * 4) == hipSuccess &&
           hipFuncSetAttribute((const void*)k_sup_step<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SUP_STEP_LDS * 4) == hipSuccess &&
           hipFuncSetAttribute((const void*)k_sup_step<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SUP_TAIL_LDS * 4) == hipSuccess;
  }();
  if (!attr) { jrr_set_error("k_sup_step: %d bytes of LDS refused", SUP_STEP_LDS * 4); return JRR_ERR_HIP; }
  SupStepArgs a;
  a.x6d = L.x6d_in; a.betas = L.betas_in; a.FT = q.FT; a.FTq = q.FTq; a.AT = q.AT; a.R0T = q.R0T;
  a.sup = SupArgs{q.t, q.nsv, q.Jn_vi, q.FTq, q.AT, q.gt_mm, q.scale, q.joints_out, q.sqerr, q.dA, q.dF, L.B, L.BP,
                  Reproj{q.gt_j2d, q.cam, q.gcam, q.sq2d, q.scale2d},
                  phase == 0 ? q.conv_img : nullptr, L.x6d_in, q.dH2T, q.dscale, q.gx, q.dsq};      // (phases 1 / 2: the adjoint opens phase 2)
  a.conv_img = q.conv_img; a.dH2T = q.dH2T; a.dscale = q.dscale; a.gx = q.gx; a.dsq = q.dsq; a.H2T_next = q.H2T_next;
  PoseUpdateArgs& u = a.ua;
  u.x6d_in = L.x6d_in; u.gx_extra = L.gx_extra; u.gb_extra = L.gb_extra;
  u.dx6d = L.dx6d; u.dR = L.dR; u.dbetas = L.dbetas;
  u.x6d_io = L.x6d_io; u.betas_io = L.betas_io; u.adam_m = L.adam_m; u.adam_v = L.adam_v; u.step = L.step;
  u.lr = L.lr; u.beta1 = L.beta1; u.beta2 = L.beta2; u.eps = L.eps;
  u.gcam = L.gcam; u.cam_io = L.cam_io; u.cam_m = L.cam_m; u.cam_v = L.cam_v;      // the camera translation steps with the poses (2-D term)
  a.step = q.step; a.arrive = q.arrive;
  static long long* const stamps = [] { const char* v = getenv("JRR_SUP_STAMPS"); return v ? (long long*)strtoull(v, nullptr, 0) : (long long*)nullptr; }();
  a.stamps = stamps;
  const dim3 grid((L.B + SUP_PP - 1) / SUP_PP), block(SUP_THREADS);
  if (phase == 1) hipLaunchKernelGGL(k_sup_step<1>, grid, block, SUP_STEP_LDS * 4, s, a, m.Jt, m.JS, m.parents);
  else if (phase == 2) hipLaunchKernelGGL(k_sup_step<2>, grid, block, SUP_TAIL_LDS * 4, s, a, m.Jt, m.JS, m.parents);
  else hipLaunchKernelGGL(k_sup_step<0>, grid, block, SUP_STEP_LDS * 4, s, a, m.Jt, m.JS, m.parents);
  return 0;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_rot6d_fwd(const float* x, float* R, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_rot6d_fwd, dim3((n + 255) / 256), dim3(256), 0, s, x, R, n);
  return 0;
}
int launch_rot6d_bwd(const float* x, const float* dR, float* dx, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_rot6d_bwd, dim3((n + 255) / 256), dim3(256), 0, s, x, dR, dx, n);
  return 0;
}

__global__ void k_step_inc(int32_t* step) { step[0] += 1; }
int launch_step_inc(int32_t* step, hipStream_t s) {
  hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, s, step);
  return 0;
}

int launch_rodrigues_fwd(const float* aa, float* R, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_rodrigues_fwd, dim3((n + 255) / 256), dim3(256), 0, s, aa, R, n);
  return 0;
}
int launch_rodrigues_bwd(const float* aa, const float* dR, float* daa, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_rodrigues_bwd, dim3((n + 255) / 256), dim3(256), 0, s, aa, dR, daa, n);
  return 0;
}

int launch_prep_fwd(const Model& m, const float* x6d, const float* Rin, const float* betas, float* FT, float* FTq, float* AT,
                    float* R0T, int B, int BP, int32_t* step_inc, hipStream_t s) {
  hipLaunchKernelGGL(k_prep_fwd, dim3(BP / PP), dim3(PP * NJ), 0, s, x6d, Rin, betas, m.Jt, m.JS, m.parents, FT, AT, R0T,
                     B, BP, step_inc, FTq);
  return 0;
}

int launch_prep_fwd_dconv(const Model& m, const float* x6d, const float* betas, float* FT, float* FTq, float* AT, float* R0T, int B,
                          int BP, int32_t* step_inc, const float* img, float* H2T, float* out, hipStream_t s) {
  const int nprep = BP / PP;
  DconvFwdArgs d{img, H2T, out};
  hipLaunchKernelGGL(k_prep_fwd_dconv, dim3(nprep + (BP / 32) * 2), dim3(PP * NJ), 0, s, x6d, betas, m.Jt, m.JS, m.parents, FT, AT,
                     R0T, B, BP, step_inc, d, nprep, FTq);
  return 0;
}

int launch_joints_loss(const float* JP, int nvc, const float* gt_mm, const float* djoints_in, float scale,
                       float* joints_out, float* sqerr, float* dJT, int B, int BP, hipStream_t s, const ReprojLaunch* r,
                       int jp_rows, const int* one_slab_flag) {
  Reproj rp;
  rp.gt_j2d = r ? r->gt_j2d : nullptr; rp.cam = r ? r->cam : nullptr; rp.gcam = r ? r->gcam : nullptr;
  rp.sq2d = r ? r->sq2d : nullptr; rp.scale2d = r ? r->scale2d : 0.f;
  hipLaunchKernelGGL(k_joints_loss, dim3(BP / PP), dim3(PP * NH), 0, s, JP, nvc, jp_rows, gt_mm, djoints_in, scale, joints_out,
                     sqerr, dJT, rp, B, BP, one_slab_flag);
  return 0;
}

int launch_project_joints(const float* joints, const float* cam, float* out, int B, hipStream_t s) {
  const int n = B * NH;
  hipLaunchKernelGGL(k_project_joints, dim3((n + 255) / 256), dim3(256), 0, s, joints, cam, out, n);
  return 0;
}

int launch_camera_fit(const float* joints, const float* gt_j2d, float* cam, float scale2d, int nsteps, float lr, float* sq2d,
                      int B, hipStream_t s) {
  hipLaunchKernelGGL(k_camera_fit, dim3((B + 63) / 64), dim3(64), 0, s, joints, gt_j2d, cam, scale2d, nsteps, lr, sq2d, B);
  return 0;
}

int launch_joint_loss_plain(const float* joints, const float* gt_mm, float scale, float* sqerr, float* djoints, int B,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_joint_loss_plain, dim3((B + 63) / 64), dim3(64), 0, s, joints, gt_mm, scale, sqerr, djoints, B);
  return 0;
}

int launch_prep_bwd(const PrepBwdLaunch& L, const Model& m, hipStream_t s) {
  PoseUpdateArgs a;
  a.x6d_in = L.x6d_in; a.gx_extra = L.gx_extra; a.gb_extra = L.gb_extra;
  a.dx6d = L.dx6d; a.dR = L.dR; a.dbetas = L.dbetas;
  a.x6d_io = L.x6d_io; a.betas_io = L.betas_io; a.adam_m = L.adam_m; a.adam_v = L.adam_v; a.step = L.step;
  a.lr = L.lr; a.beta1 = L.beta1; a.beta2 = L.beta2; a.eps = L.eps;
  a.gcam = L.gcam; a.cam_io = L.cam_io; a.cam_m = L.cam_m; a.cam_v = L.cam_v;
  hipLaunchKernelGGL(k_chain_bwd, dim3((L.B + PPB - 1) / PPB), dim3(PPB * NJ), 0, s, L.FT, L.R0T, L.AT, m.Jt, m.JS, m.parents,
                     L.dATp, L.nslabA, L.strideA, L.dFTp, a, L.B, L.BP, L.dmaskA);
  return 0;
}

int launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, const int32_t* step, float lr, float b1,
                     float b2, float eps, hipStream_t s, int step_plus) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_adam_flat, dim3(blocks), dim3(256), 0, s, p, g, m, v, n, step, lr, b1, b2, eps, step_plus);
  return 0;
}

}  // namespace jrr
