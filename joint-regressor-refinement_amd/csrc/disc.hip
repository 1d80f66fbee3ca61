// Pose / shape discriminators (/root/reference/scripts/discriminator.py:7-74) -- the VALU parts.
// The two wide layers (768->1024, 1024->1024) and their adjoints run on the MFMA GEMM (gemm.hip);
// here: the per-joint shared MLP (1x1 convs 6->32->32) + 24 per-joint heads, the 1024->1 output
// layer with sigmoid and the MSE adjoint, their input-gradients, and the tiny shape MLP.
//
// Flat parameter vector = state_dict order (offsets in floats):
//   conv0.w 0 (32x6)  conv0.b 192  conv2.w 224 (32x32)  conv2.b 1248
//   linears.i.{w,b} 1280+33i (+32)          i = 0..23
//   fc0.w 2072 (1024x768)  fc0.b 788504  fc2.w 789528 (1024x1024)  fc2.b 1838104
//   fc4.w 1839128 (1x1024) fc4.b 1840152
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }

// per-(pose, joint) shared MLP; lane = pose, blockIdx.y = joint
__device__ __forceinline__ void joint_mlp(const float* __restrict__ P, const float x[6], float h1[32], float h2[32]) {
  const float* w0 = P + DP_CONV0_W;
  const float* b0 = P + DP_CONV0_B;
  const float* w2 = P + DP_CONV2_W;
  const float* b2 = P + DP_CONV2_B;
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    float acc = b0[o];
#pragma unroll
    for (int c = 0; c < 6; ++c) acc = fmaf(w0[o * 6 + c], x[c], acc);
    h1[o] = fmaxf(acc, 0.f);
  }
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    float acc = b2[o];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc = fmaf(w2[o * 32 + c], h1[c], acc);
    h2[o] = fmaxf(acc, 0.f);
  }
}

__global__ __launch_bounds__(64) void k_disc_conv_fwd(const float* __restrict__ P, const float* __restrict__ x6d,
                                                      float* __restrict__ H2T, float* __restrict__ out, int B, int BP) {
  const int b = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
  if (b >= BP) return;
  if (b >= B) {
    for (int o = 0; o < 32; ++o) H2T[(size_t)(j * 32 + o) * BP + b] = 0.f;
    return;
  }
  float x[6], h1[32], h2[32];
#pragma unroll
  for (int c = 0; c < 6; ++c) x[c] = x6d[((size_t)b * NJ + j) * 6 + c];
  joint_mlp(P, x, h1, h2);
#pragma unroll
  for (int o = 0; o < 32; ++o) H2T[(size_t)(j * 32 + o) * BP + b] = h2[o];
  if (out) {
    const float* wh = P + DP_HEADS + 33 * j;
    float z = wh[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) z = fmaf(wh[o], h2[o], z);
    out[(size_t)b * 25 + 1 + j] = sigmoidf(z);
  }
}

// output layer: z = fc4.w . a2 + fc4.b ; s = sigmoid(z) ; dz = scale (s - target) s (1-s) ;
// dA2T[n][b] = relu'(a2[n][b]) * w[n] * dz.  Block = 1024 threads = 16 waves x 64 poses,
// wave q handles n in [64q, 64q+64).
__global__ __launch_bounds__(1024) void k_disc_out(const float* __restrict__ P, const float* __restrict__ A2T,
                                                   float* __restrict__ out, float* __restrict__ dA2T, float scale,
                                                   float target, int B, int BP) {
  __shared__ float red[16][64];
  __shared__ float dzs[64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const float* w = P + DP_FC4_W;
  float acc = 0.f;
  for (int n = q * 64; n < q * 64 + 64; ++n) acc = fmaf(w[n], A2T[(size_t)n * BP + b], acc);
  red[q][lane] = acc;
  __syncthreads();
  if (q == 0) {
    float z = P[DP_FC4_B];
#pragma unroll
    for (int i = 0; i < 16; ++i) z += red[i][lane];
    float s = sigmoidf(z);
    if (out && b < B) out[(size_t)b * 25] = s;
    dzs[lane] = (b < B) ? scale * (s - target) * s * (1.f - s) : 0.f;
  }
  __syncthreads();
  if (dA2T) {
    const float dz = dzs[lane];
    for (int n = q * 64; n < q * 64 + 64; ++n) {
      float a = A2T[(size_t)n * BP + b];
      dA2T[(size_t)n * BP + b] = (a > 0.f) ? w[n] * dz : 0.f;
    }
  }
}

// input gradient of the per-joint MLP + heads; dH2T is the gradient arriving from fc0 (may be NULL)
__global__ __launch_bounds__(64) void k_disc_conv_bwd(const float* __restrict__ P, const float* __restrict__ x6d,
                                                      const float* __restrict__ dH2T, float scale, float target,
                                                      float* __restrict__ gx, int B, int BP) {
  const int b = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
  if (b >= B) return;
  float x[6], h1[32], h2[32];
#pragma unroll
  for (int c = 0; c < 6; ++c) x[c] = x6d[((size_t)b * NJ + j) * 6 + c];
  joint_mlp(P, x, h1, h2);
  const float* wh = P + DP_HEADS + 33 * j;
  float z = wh[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) z = fmaf(wh[o], h2[o], z);
  const float s = sigmoidf(z);
  const float dz = scale * (s - target) * s * (1.f - s);
  float dh2[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    float g = dz * wh[o];
    if (dH2T) g += dH2T[(size_t)(j * 32 + o) * BP + b];
    dh2[o] = (h2[o] > 0.f) ? g : 0.f;
  }
  const float* w2 = P + DP_CONV2_W;
  const float* w0 = P + DP_CONV0_W;
  float dh1[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o) acc = fmaf(w2[o * 32 + c], dh2[o], acc);
    dh1[c] = (h1[c] > 0.f) ? acc : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o) acc = fmaf(w0[o * 6 + c], dh1[o], acc);
    gx[((size_t)b * NJ + j) * 6 + c] = acc;
  }
}

// shape discriminator 10 -> 10 -> 5 -> 1 (discriminator.py:57-74): forward + input gradient of
// weight*mean((s-target)^2).  Params: w0 0 (10x10) b0 100 w2 110 (5x10) b2 160 w4 165 (1x5) b4 170
__global__ void k_shape_disc(const float* __restrict__ P, const float* __restrict__ betas, float* __restrict__ out,
                             float* __restrict__ gb, float scale, float target, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float x[10], h1[10], h2[5];
#pragma unroll
  for (int i = 0; i < 10; ++i) x[i] = betas[(size_t)b * 10 + i];
#pragma unroll
  for (int o = 0; o < 10; ++o) {
    float acc = P[100 + o];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc = fmaf(P[o * 10 + i], x[i], acc);
    h1[o] = fmaxf(acc, 0.f);
  }
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    float acc = P[160 + o];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc = fmaf(P[110 + o * 10 + i], h1[i], acc);
    h2[o] = fmaxf(acc, 0.f);
  }
  float z = P[170];
#pragma unroll
  for (int i = 0; i < 5; ++i) z = fmaf(P[165 + i], h2[i], z);
  const float s = sigmoidf(z);
  if (out) out[b] = s;
  if (!gb) return;
  const float dz = scale * (s - target) * s * (1.f - s);
  float dh2[5], dh1[10];
#pragma unroll
  for (int i = 0; i < 5; ++i) dh2[i] = (h2[i] > 0.f) ? dz * P[165 + i] : 0.f;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < 5; ++o) acc = fmaf(P[110 + o * 10 + i], dh2[o], acc);
    dh1[i] = (h1[i] > 0.f) ? acc : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < 10; ++o) acc = fmaf(P[o * 10 + i], dh1[o], acc);
    gb[(size_t)b * 10 + i] = acc;
  }
}

// [rows][cols] -> [cols][rows]
__global__ void k_transpose(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float t[32][33];
  int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (r0 + i < rows && c < cols) t[i][threadIdx.x] = in[(size_t)(r0 + i) * cols + c];
  __syncthreads();
  int r = r0 + threadIdx.x, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (c0 + i < cols && r < rows) out[(size_t)(c0 + i) * rows + r] = t[threadIdx.x][i];
}

int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, in, out, rows, cols);
  return 0;
}

int launch_disc_conv_fwd(const float* P, const float* x6d, float* H2T, float* out, int B, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_disc_conv_fwd, dim3(BP / 64, NJ), dim3(64), 0, s, P, x6d, H2T, out, B, BP);
  return 0;
}
int launch_disc_out(const float* P, const float* A2T, float* out, float* dA2T, float scale, float target, int B, int BP,
                    hipStream_t s) {
  hipLaunchKernelGGL(k_disc_out, dim3(BP / 64), dim3(1024), 0, s, P, A2T, out, dA2T, scale, target, B, BP);
  return 0;
}
int launch_disc_conv_bwd(const float* P, const float* x6d, const float* dH2T, float scale, float target, float* gx,
                         int B, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_disc_conv_bwd, dim3((B + 63) / 64, NJ), dim3(64), 0, s, P, x6d, dH2T, scale, target, gx, B, BP);
  return 0;
}
int launch_shape_disc(const float* P, const float* betas, float* out, float* gb, float scale, float target, int B,
                      hipStream_t s) {
  hipLaunchKernelGGL(k_shape_disc, dim3((B + 63) / 64), dim3(64), 0, s, P, betas, out, gb, scale, target, B);
  return 0;
}

}  // namespace jrr
