// Per-joint shared MLP of the pose discriminator on the matrix cores: device bodies shared by the stand-alone kernels
// (disc.hip) and the horizontally fused launches of the inner loop (prep.hip).
#pragma once
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// ------------------------------------------------------------------------------------------
// The per-joint shared MLP (1x1 convs 6 -> 32 -> 32, scripts/discriminator.py:15-18,32-36) and the 24 per-joint heads
// (:21-23,44-49) on the fp32 matrix cores.  One wave per (32 poses, joint): every layer is a 32 x 32 (channel, pose)
// tile in the accumulator layout of v_mfma_f32_32x32x2_f32 (rows = channels in registers, columns = poses on lanes), so a
// layer's output is the next layer's B operand without leaving registers (it sums over the tile's ROW index), forward
// and adjoint alike.  The weights sit in LDS in the orientation each product reads conflict-free.
//   forward : h1 = relu(W0 x + b0) [3 MFMA]   h2 = relu(W2 h1 + b2) [16]   z_j = wh_j . h2 + bh_j
//   adjoint : dh2 = relu'(h2) (dz_j wh_j + dH2)   dh1 = relu'(h1) W2^T dh2 [16]   gx = W0^T dh1 [16]
// ------------------------------------------------------------------------------------------
constexpr int CL_W0P = 0;                 // [8][32]   W0p[c][o] = conv0.w[o][c], rows 6,7 zero
constexpr int CL_B0 = CL_W0P + 256;       // [32]
constexpr int CL_B2 = CL_B0 + 32;         // [32]
constexpr int CL_W2T = CL_B2 + 32;        // [32][32]  W2T[c][o] = conv2.w[o][c]
constexpr int CL_W2 = CL_W2T + 1024;      // [32][32]  conv2.w[o][c]
constexpr int CL_W0A = CL_W2 + 1024;      // [32][32]  W0a[o][c6] = conv0.w[o][c6], columns 6.. zero
constexpr int CL_WH = CL_W0A + 1024;      // [24][33]  heads
constexpr int CL_FLOATS = CL_WH + 24 * 33;

// (the image itself is built once per parameter upload: k_conv_image, disc.hip)
__device__ __forceinline__ void conv_stage_params(const float* __restrict__ img, float* __restrict__ L) {
  const f32x4* src = reinterpret_cast<const f32x4*>(img);
  f32x4* dst = reinterpret_cast<f32x4*>(L);
  for (int i = threadIdx.x; i < CL_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
}

// h1, h2 (post-ReLU) of 32 poses x one joint, in accumulator layout
__device__ __forceinline__ void conv_mlp_tile(const float* __restrict__ L, const float* __restrict__ x6d, int b, bool ok, int j,
                                              int half, int l31, f32x16& h1, f32x16& h2) {
  f32x16 acc = zero16();
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const float xv = ok ? x6d[((size_t)b * NJ + j) * 6 + 2 * kk + half] : 0.f;
    acc = mfma(L[CL_W0P + (2 * kk + half) * 32 + l31], xv, acc);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) h1[q] = fmaxf(acc[q] + L[CL_B0 + acc_row(q, half)], 0.f);
  acc = zero16();
#pragma unroll
  for (int q = 0; q < 16; ++q) acc = mfma(L[CL_W2T + acc_row(q, half) * 32 + l31], h1[q], acc);
#pragma unroll
  for (int q = 0; q < 16; ++q) h2[q] = fmaxf(acc[q] + L[CL_B2 + acc_row(q, half)], 0.f);
}

// z_j = wh_j . h2 + bh_j for the lane's pose (both lane halves return the full sum)
__device__ __forceinline__ float conv_head(const float* __restrict__ L, int j, int half, const f32x16& h2) {
  float part = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) part = fmaf(L[CL_WH + j * 33 + acc_row(q, half)], h2[q], part);
  return part + __shfl_xor(part, 32) + L[CL_WH + j * 33 + 32];
}

// Workgroup `blk` of (BP / 32) * (24 / W) workgroups of W waves (W = blockDim.x / 64: 4 stand-alone, 12 inside the fused
// launches): wave w owns poses [32 bt, +32) and joint W * group + w.  L = the workgroup's LDS image (CL_FLOATS floats).
// STAGED: the image already sits in L (a second call of the composed kernel k_sup_step on the same pool)
template <bool QUAD, bool STAGED = false>
__device__ __forceinline__ void dconv_fwd_body(float* __restrict__ L, int blk, const float* __restrict__ img,
                                               const float* __restrict__ x6d, float* __restrict__ H2T,
                                               float* __restrict__ out, int B, int BP) {
  if (!STAGED) {
    conv_stage_params(img, L);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int W = blockDim.x >> 6, ngrp = NJ / W;
  const int bt = blk / ngrp, j = (blk % ngrp) * W + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = bt * 32 + l31;
  const bool ok = b < B;
  f32x16 h1, h2;
  conv_mlp_tile(L, x6d, b, ok, j, half, l31, h1, h2);
  if (QUAD) {      // [row/4][pose][4]: registers 4g .. 4g+3 are one quad of this lane's pose (jrr_common.h)
    const unsigned qoff = (unsigned)half * (unsigned)BP + (unsigned)b;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 t = {h2[4 * g], h2[4 * g + 1], h2[4 * g + 2], h2[4 * g + 3]};
      if (!ok) t = f32x4{0.f, 0.f, 0.f, 0.f};
      *quad_ptr(H2T, (size_t)j * 8, g, BP, qoff) = t;
    }
  } else {
    const unsigned voff = (unsigned)(4 * half) * (unsigned)BP + (unsigned)b;
#pragma unroll
    for (int q = 0; q < 16; ++q) urow(H2T, (size_t)(j * 32 + acc_row_u(q)), BP)[voff] = ok ? h2[q] : 0.f;   // padded poses: zeros
  }
  if (out) {
    const float z = conv_head(L, j, half, h2);
    if (ok && half == 0) out[(size_t)b * 25 + 1 + j] = sigmoidf(z);
  }
}

// input gradient of the per-joint MLP + heads; dH2T = gradient arriving from fc0 (may be NULL), gout (B,25) nullable
// the adjoint of ONE (32-pose tile bt, joint j) pair on the calling wave: no staging, no barrier (L holds the image)
template <bool QUAD>
__device__ __forceinline__ void dconv_bwd_tile(const float* __restrict__ L, int bt, int j, const float* __restrict__ x6d,
                                               const float* __restrict__ dH2T, const float* __restrict__ gout, float scale, float target,
                                               float* __restrict__ gx, int B, int BP, float* __restrict__ sqj) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int b = bt * 32 + l31;
  const bool ok = b < B;
  f32x16 h1, h2;
  conv_mlp_tile(L, x6d, b, ok, j, half, l31, h1, h2);
  const float z = conv_head(L, j, half, h2);
  const float sg = sigmoidf(z);
  if (sqj && ok && half == 0) sqj[(size_t)(1 + j) * BP + b] = (sg - target) * (sg - target);
  const float up = gout ? (ok ? gout[(size_t)b * 25 + 1 + j] : 0.f) : scale * (sg - target);
  const float dz = ok ? up * sg * (1.f - sg) : 0.f;
  const unsigned voff = (unsigned)(4 * half) * (unsigned)BP + (unsigned)b;
  f32x16 din = zero16();                 // gradient arriving from fc0
  if (dH2T) {
    if (QUAD) {
      const unsigned qoff = (unsigned)half * (unsigned)BP + (unsigned)b;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *quad_ptr(dH2T, (size_t)j * 8, g, BP, qoff);
        din[4 * g] = t[0]; din[4 * g + 1] = t[1]; din[4 * g + 2] = t[2]; din[4 * g + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) din[q] = urow(dH2T, (size_t)(j * 32 + acc_row_u(q)), BP)[voff];
    }
  }
  f32x16 acc = zero16();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float g = dz * L[CL_WH + j * 33 + acc_row(q, half)] + din[q];
    const float dh2 = (h2[q] > 0.f) ? g : 0.f;
    acc = mfma(L[CL_W2 + acc_row(q, half) * 32 + l31], dh2, acc);        // dh1[c] += W2[o][c] dh2[o]
  }
  f32x16 accx = zero16();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float dh1 = (h1[q] > 0.f) ? acc[q] : 0.f;
    accx = mfma(L[CL_W0A + acc_row(q, half) * 32 + l31], dh1, accx);     // gx[c6] += W0[o][c6] dh1[o]
  }
  if (ok) {      // rows 0..3 live in registers 0..3 of lane half 0, rows 4,5 in registers 0,1 of half 1
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2* dst = reinterpret_cast<f32x2*>(gx + ((size_t)b * NJ + j) * 6 + 4 * half);     // 8-byte aligned: even float offset
    dst[0] = f32x2{accx[0], accx[1]};
    if (half == 0) dst[1] = f32x2{accx[2], accx[3]};
  }
}

// TWO (tile, joint) adjoints on the calling wave, their dependent chains interleaved stage by stage (round 6): one tile is a chain of 51
// matrix instructions with an LDS operand read and a vector step between any two of them -- 7 us for a wave on its own, of which the
// matrix pipe is busy 1.4.  Two independent chains in one instruction stream fill each other's gaps.  Per tile the arithmetic and its
// order are those of dconv_bwd_tile (bit-identical results).  Quad layouts only (the loop's).
__device__ __forceinline__ void dconv_bwd_pair(const float* __restrict__ L, int bt, int j0, int j1, const float* __restrict__ x6d,
                                               const float* __restrict__ dH2T, float scale, float target, float* __restrict__ gx, int B,
                                               int BP, float* __restrict__ sqj) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  const int b = bt * 32 + l31;
  const bool ok = b < B;
  const int jj[2] = {j0, j1};
  const unsigned qoff = (unsigned)half * (unsigned)BP + (unsigned)b;
  // everything that comes from global memory, for both tiles, first
  float xv[2][3];
  f32x4 dq[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) xv[n][kk] = ok ? x6d[((size_t)b * NJ + jj[n]) * 6 + 2 * kk + half] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) dq[n][g] = *quad_ptr(dH2T, (size_t)jj[n] * 8, g, BP, qoff);
  }
  f32x16 acc[2] = {zero16(), zero16()}, h1[2], h2[2];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
    const float w = L[CL_W0P + (2 * kk + half) * 32 + l31];
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[n] = mfma(w, xv[n][kk], acc[n]);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float bq = L[CL_B0 + acc_row(q, half)];
#pragma unroll
    for (int n = 0; n < 2; ++n) h1[n][q] = fmaxf(acc[n][q] + bq, 0.f);
  }
  acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float w = L[CL_W2T + acc_row(q, half) * 32 + l31];
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[n] = mfma(w, h1[n][q], acc[n]);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float bq = L[CL_B2 + acc_row(q, half)];
#pragma unroll
    for (int n = 0; n < 2; ++n) h2[n][q] = fmaxf(acc[n][q] + bq, 0.f);
  }
  float dz[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const float z = conv_head(L, jj[n], half, h2[n]);
    const float sg = sigmoidf(z);
    if (sqj && ok && half == 0) sqj[(size_t)(1 + jj[n]) * BP + b] = (sg - target) * (sg - target);
    dz[n] = ok ? scale * (sg - target) * sg * (1.f - sg) : 0.f;
  }
  acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float w = L[CL_W2 + acc_row(q, half) * 32 + l31];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float g = dz[n] * L[CL_WH + jj[n] * 33 + acc_row(q, half)] + dq[n][q >> 2][q & 3];
      const float dh2 = (h2[n][q] > 0.f) ? g : 0.f;
      acc[n] = mfma(w, dh2, acc[n]);
    }
  }
  f32x16 accx[2] = {zero16(), zero16()};
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float w = L[CL_W0A + acc_row(q, half) * 32 + l31];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float dh1 = (h1[n][q] > 0.f) ? acc[n][q] : 0.f;
      accx[n] = mfma(w, dh1, accx[n]);
    }
  }
  if (ok) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      f32x2* dst = reinterpret_cast<f32x2*>(gx + ((size_t)b * NJ + jj[n]) * 6 + 4 * half);
      dst[0] = f32x2{accx[n][0], accx[n][1]};
      if (half == 0) dst[1] = f32x2{accx[n][2], accx[n][3]};
    }
  }
}

template <bool QUAD, bool STAGED = false>
__device__ __forceinline__ void dconv_bwd_body(float* __restrict__ L, int blk, const float* __restrict__ img,
                                               const float* __restrict__ x6d, const float* __restrict__ dH2T,
                                               const float* __restrict__ gout, float scale, float target,
                                               float* __restrict__ gx, int B, int BP, float* __restrict__ sqj) {
  if (!STAGED) {
    conv_stage_params(img, L);
    __syncthreads();
  }
  const int W = blockDim.x >> 6, ngrp = NJ / W;
  const int bt = blk / ngrp, j = (blk % ngrp) * W + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  dconv_bwd_tile<QUAD>(L, bt, j, x6d, dH2T, gout, scale, target, gx, B, BP, sqj);
}

}  // namespace jrr
