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
#include "dconv.h"

namespace jrr {


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

// The LDS parameter image of the per-joint MLP kernels (layout CL_*), built ONCE per parameter upload (k_conv_image,
// jrr_engine_set_pose_disc).  Building it inside every workgroup -- gathers of conv0.w, two orientations of conv2.w, three
// dependent load rounds -- was 4.7 of k_dconv_fwd's 11.6 us (wall-clock stamps); copying the image is one round.
static_assert(CL_FLOATS % 4 == 0 && CL_FLOATS <= CONV_IMAGE_FLOATS, "image copied in 16-byte pieces");
__global__ void k_conv_image(const float* __restrict__ P, float* __restrict__ L) {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    const int c = i >> 5, o = i & 31;
    L[CL_W0P + i] = (c < 6) ? P[DP_CONV0_W + o * 6 + c] : 0.f;
  }
  for (int i = threadIdx.x; i < 32; i += blockDim.x) { L[CL_B0 + i] = P[DP_CONV0_B + i]; L[CL_B2 + i] = P[DP_CONV2_B + i]; }
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
    const int o = i >> 5, c = i & 31;
    const float w = P[DP_CONV2_W + i];          // conv2.w[o][c]
    L[CL_W2 + i] = w;
    L[CL_W2T + c * 32 + o] = w;
    L[CL_W0A + i] = (c < 6) ? P[DP_CONV0_W + o * 6 + c] : 0.f;
  }
  for (int i = threadIdx.x; i < 24 * 33; i += blockDim.x) L[CL_WH + i] = P[DP_HEADS + i];
}
int launch_conv_image(const float* P, float* img, hipStream_t s) {
  hipLaunchKernelGGL(k_conv_image, dim3(1), dim3(256), 0, s, P, img);
  return 0;
}
// stand-alone launches (operator-level entry points, outer-step paths): (BP / 32) * 6 workgroups of 4 waves
template <bool QUAD>
__global__ __launch_bounds__(256) void k_dconv_fwd(const float* __restrict__ img, const float* __restrict__ x6d,
                                                   float* __restrict__ H2T, float* __restrict__ out, int B, int BP) {
  __shared__ __attribute__((aligned(16))) float L[CL_FLOATS];
  dconv_fwd_body<QUAD>(L, blockIdx.x, img, x6d, H2T, out, B, BP);
}
template <bool QUAD>
__global__ __launch_bounds__(256) void k_dconv_bwd(const float* __restrict__ img, const float* __restrict__ x6d,
                                                   const float* __restrict__ dH2T, const float* __restrict__ gout,
                                                   float scale, float target, float* __restrict__ gx, int B, int BP,
                                                   float* __restrict__ sqj) {
  __shared__ __attribute__((aligned(16))) float L[CL_FLOATS];
  dconv_bwd_body<QUAD>(L, blockIdx.x, img, x6d, dH2T, gout, scale, target, gx, B, BP, sqj);
}
// output layer: z = fc4.w . a2 + fc4.b ; s = sigmoid(z) ; dz = scale (s - target) s (1-s) ;
// dA2T[n][b] = relu'(a2[n][b]) * w[n] * dz.  Block = 1024 threads = 16 waves x 64 poses,
// wave q handles n in [64q, 64q+64).
__global__ __launch_bounds__(1024) void k_disc_out(const float* __restrict__ P, const float* __restrict__ A2T,
                                                   float* __restrict__ out, float* __restrict__ dA2T,
                                                   const float* __restrict__ gout, float scale, float target, int B,
                                                   int BP, float* __restrict__ dz0, float* __restrict__ sq0) {
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
    if (sq0) sq0[b] = (b < B) ? (s - target) * (s - target) : 0.f;
    const float up = gout ? ((b < B) ? gout[(size_t)b * 25] : 0.f) : scale * (s - target);
    dzs[lane] = (b < B) ? up * s * (1.f - s) : 0.f;
    if (dz0) dz0[b] = dzs[lane];
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

// shape discriminator 10 -> 10 -> 5 -> 1 (discriminator.py:57-74): forward + input gradient of
// weight*mean((s-target)^2).  Params: w0 0 (10x10) b0 100 w2 110 (5x10) b2 160 w4 165 (1x5) b4 170
__global__ void k_shape_disc(const float* __restrict__ P, const float* __restrict__ betas, float* __restrict__ out,
                             float* __restrict__ gb, float scale, float target, int B, const float* __restrict__ gout,
                             float* __restrict__ sq) {
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
  if (sq) sq[b] = (s - target) * (s - target);
  if (!gb) return;
  const float dz = (gout ? gout[b] : scale * (s - target)) * s * (1.f - s);
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

// ---- weight gradients (outer-step discriminator update, scripts/optimize.py:276-293) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ void wave_atomic_add(float v, float* dst, int lane) {
  v = wave_sum(v);
  if (lane == 0) atomicAdd(dst, v);
}

// out[r] += sum_c M[r][c] * (vec ? vec[c] : 1)
__global__ void k_rowdot_accum(const float* __restrict__ M, int ld, const float* __restrict__ vec, float* __restrict__ out,
                               int cols) {
  __shared__ float red[256];
  const int r = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) acc += M[(size_t)r * ld + c] * (vec ? vec[c] : 1.f);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[r] += red[0];
}

// per-joint MLP + head weight gradients.  One wave per (64 poses, joint): lane = pose runs the joint MLP forward
// and backward in registers; the outer products over the 64 poses (dW2 = dh2^T h1, dW0 = dh1^T x, the bias and head
// sums) are taken from LDS-staged copies with lane = (output row o, half of the columns) and written as ONE partial
// slab per wave -- no atomics; the caller reduces the slabs.
// dH2T = gradient arriving from fc0.
constexpr int CW_LD = 36;                      // LDS row stride in floats (16-byte aligned rows)
__global__ __launch_bounds__(64) void k_disc_conv_bwd_params(const float* __restrict__ P, const float* __restrict__ x6d,
                                                             const float* __restrict__ dH2T, const float* __restrict__ gout,
                                                             float scale, float target, float* __restrict__ slab_shared,
                                                             float* __restrict__ slab_heads, int B, int BP) {
  __shared__ __attribute__((aligned(16))) float Sa[64 * CW_LD];
  __shared__ __attribute__((aligned(16))) float Sb[64 * CW_LD];
  const int lane = threadIdx.x, b = blockIdx.x * 64 + lane, j = blockIdx.y;
  const int o_ = lane & 31, hf = lane >> 5;
  const bool ok = b < B;
  float aW2[16], aW0[3] = {0.f, 0.f, 0.f}, ab2 = 0.f, ab0 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) aW2[k] = 0.f;
  {
    float x[6], h1[32], h2[32];
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = ok ? x6d[((size_t)b * NJ + j) * 6 + c] : 0.f;
    joint_mlp(P, x, h1, h2);
    const float* wh = P + DP_HEADS + 33 * j;
    float z = wh[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) z = fmaf(wh[o], h2[o], z);
    const float s = sigmoidf(z);
    const float dz = ok ? (gout ? gout[(size_t)b * 25 + 1 + j] : scale * (s - target)) * s * (1.f - s) : 0.f;
    float dh2[32], dh1[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) {
      float g = dz * wh[o];
      if (ok) g += dH2T[(size_t)(j * 32 + o) * BP + b];
      dh2[o] = (ok && h2[o] > 0.f) ? g : 0.f;
    }
    const float* w2 = P + DP_CONV2_W;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int o = 0; o < 32; ++o) acc = fmaf(w2[o * 32 + c], dh2[o], acc);
      dh1[c] = (h1[c] > 0.f) ? acc : 0.f;
    }
    // ---- head of joint j: dW[o] = sum_b dz h2[o], db = sum_b dz
#pragma unroll
    for (int o = 0; o < 32; ++o) Sa[lane * CW_LD + o] = dz * h2[o];
    Sa[lane * CW_LD + 32] = dz;
    __syncthreads();
    if (lane < 33) {
      float acc = 0.f;
      for (int bb = 0; bb < 64; ++bb) acc += Sa[bb * CW_LD + lane];
      slab_heads[(size_t)blockIdx.x * 792 + 33 * j + lane] = acc;
    }
    __syncthreads();
    // ---- conv2: dW2[o][c] += sum_b dh2[b][o] h1[b][c]; db2[o] += sum_b dh2[b][o]
#pragma unroll
    for (int o = 0; o < 32; ++o) { Sa[lane * CW_LD + o] = dh2[o]; Sb[lane * CW_LD + o] = h1[o]; }
    __syncthreads();
    for (int bb = 0; bb < 64; ++bb) {
      const float a = Sa[bb * CW_LD + o_];
      const f32x4* hr = reinterpret_cast<const f32x4*>(&Sb[bb * CW_LD + hf * 16]);
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const f32x4 r = hr[k4];
#pragma unroll
        for (int k = 0; k < 4; ++k) aW2[k4 * 4 + k] = fmaf(a, r[k], aW2[k4 * 4 + k]);
      }
      ab2 += a;
    }
    __syncthreads();
    // ---- conv0: dW0[o][c] += sum_b dh1[b][o] x[b][c]; db0[o] += sum_b dh1[b][o]
#pragma unroll
    for (int o = 0; o < 32; ++o) Sa[lane * CW_LD + o] = dh1[o];
#pragma unroll
    for (int c = 0; c < 6; ++c) Sb[lane * CW_LD + c] = x[c];
    __syncthreads();
    for (int bb = 0; bb < 64; ++bb) {
      const float a = Sa[bb * CW_LD + o_];
#pragma unroll
      for (int k = 0; k < 3; ++k) aW0[k] = fmaf(a, Sb[bb * CW_LD + hf * 3 + k], aW0[k]);
      ab0 += a;
    }
    __syncthreads();
  }
  float* slab = slab_shared + ((size_t)blockIdx.x * NJ + j) * 1280;
#pragma unroll
  for (int k = 0; k < 16; ++k) slab[DP_CONV2_W + o_ * 32 + hf * 16 + k] = aW2[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) slab[DP_CONV0_W + o_ * 6 + hf * 3 + k] = aW0[k];
  if (hf == 0) { slab[DP_CONV2_B + o_] = ab2; slab[DP_CONV0_B + o_] = ab0; }
}

// shape discriminator: forward, weight gradients of mean((s-target)^2), per-pose squared error
__global__ __launch_bounds__(64) void k_shape_disc_bwd_params(const float* __restrict__ P, const float* __restrict__ betas,
                                                              const float* __restrict__ gout, float scale, float target,
                                                              float* __restrict__ dP,
                                                              float* __restrict__ sqerr, int B) {
  const int lane = threadIdx.x, b = blockIdx.x * 64 + lane;
  const bool ok = b < B;
  float x[10], h1[10], h2[5];
#pragma unroll
  for (int i = 0; i < 10; ++i) x[i] = ok ? betas[(size_t)b * 10 + i] : 0.f;
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
  if (ok && sqerr) sqerr[b] = (s - target) * (s - target);
  const float dz = ok ? (gout ? gout[b] : scale * (s - target)) * s * (1.f - s) : 0.f;
  float dh2[5], dh1[10];
  wave_atomic_add(dz, dP + 170, lane);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    wave_atomic_add(dz * h2[i], dP + 165 + i, lane);
    dh2[i] = (h2[i] > 0.f) ? dz * P[165 + i] : 0.f;
    wave_atomic_add(dh2[i], dP + 160 + i, lane);
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < 5; ++o) {
      acc = fmaf(P[110 + o * 10 + i], dh2[o], acc);
      wave_atomic_add(dh2[o] * h1[i], dP + 110 + o * 10 + i, lane);
    }
    dh1[i] = (h1[i] > 0.f) ? acc : 0.f;
    wave_atomic_add(dh1[i], dP + 100 + i, lane);
  }
#pragma unroll
  for (int o = 0; o < 10; ++o)
#pragma unroll
    for (int i = 0; i < 10; ++i) wave_atomic_add(dh1[o] * x[i], dP + o * 10 + i, lane);
}

// sqerr[b] = sum_k (out[b][k] - target)^2
__global__ void k_sqerr_rows(const float* __restrict__ out, int ncol, float target, float* __restrict__ sqerr, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int k = 0; k < ncol; ++k) { const float d = out[(size_t)b * ncol + k] - target; acc += d * d; }
  sqerr[b] = acc;
}

// out[b*25] = sigmoid(fc4.b + sum_t zpart[t][b])   (operator-level forward: the fc4 dot comes from the fc2 epilogue)
__global__ void k_disc_z_finish(const float* __restrict__ zpart, int nz, int ld, const float* __restrict__ zbias,
                                float* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float z = zbias[0];
  for (int t = 0; t < nz; ++t) z += zpart[(size_t)t * ld + b];
  out[(size_t)b * 25] = sigmoidf(z);
}
int launch_disc_z_finish(const float* zpart, int nz, int ld, const float* zbias, float* out, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_disc_z_finish, dim3((B + 255) / 256), dim3(256), 0, s, zpart, nz, ld, zbias, out, B);
  return 0;
}
// out[r][c] = w[r] * in[r][c]
__global__ void k_scale_rows(const float* __restrict__ in, const float* __restrict__ w, float* __restrict__ out, int rows, int cols) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)rows * cols) out[i] = w[i / cols] * in[i];
}
int launch_scale_rows(const float* in, const float* w, float* out, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)(((size_t)rows * cols + 255) / 256)), dim3(256), 0, s, in, w, out, rows, cols);
  return 0;
}

// out[b] = sum_r M[r][b]   (rows of a [rows][ld] pose-contiguous array)
__global__ void k_colsum(const float* __restrict__ M, int rows, int ld, float* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int r = 0; r < rows; ++r) acc += M[(size_t)r * ld + b];
  out[b] = acc;
}
int launch_colsum(const float* M, int rows, int ld, float* out, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_colsum, dim3((B + 255) / 256), dim3(256), 0, s, M, rows, ld, out, B);
  return 0;
}

int launch_rowdot_accum(const float* M, int ld, const float* vec, float* out, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(k_rowdot_accum, dim3(rows), dim3(256), 0, s, M, ld, vec, out, cols);
  return 0;
}
int launch_disc_conv_bwd_params(const float* P, const float* x6d, const float* dH2T, const float* gout, float scale,
                                float target, float* slab_shared, float* slab_heads, int B, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_disc_conv_bwd_params, dim3(BP / 64, NJ), dim3(64), 0, s, P, x6d, dH2T, gout, scale, target, slab_shared,
                     slab_heads, B, BP);
  return 0;
}
int launch_shape_disc_bwd_params(const float* P, const float* betas, const float* gout, float scale, float target,
                                 float* dparams, float* sqerr, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_shape_disc_bwd_params, dim3((B + 63) / 64), dim3(64), 0, s, P, betas, gout, scale, target, dparams, sqerr, B);
  return 0;
}
int launch_sqerr_rows(const float* out, int ncol, float target, float* sqerr, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_sqerr_rows, dim3((B + 255) / 256), dim3(256), 0, s, out, ncol, target, sqerr, B);
  return 0;
}

// [rows][cols] -> [cols][rows]
__global__ void k_transpose(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int ldin, int ldout) {
  __shared__ float t[32][33];
  int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (r0 + i < rows && c < cols) t[i][threadIdx.x] = in[(size_t)(r0 + i) * ldin + c];
  __syncthreads();
  int r = r0 + threadIdx.x, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y)
    if (c0 + i < cols && r < rows) out[(size_t)(c0 + i) * ldout + r] = t[threadIdx.x][i];
}

int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s, int ldin, int ldout) {
  hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, in, out, rows, cols,
                     ldin ? ldin : cols, ldout ? ldout : rows);
  return 0;
}

int launch_disc_conv_fwd(const float* P, const float* x6d, float* H2T, float* out, int B, int BP, hipStream_t s, int quad) {
  if (quad) hipLaunchKernelGGL((k_dconv_fwd<true>), dim3((BP / 32) * 6), dim3(256), 0, s, P, x6d, H2T, out, B, BP);
  else hipLaunchKernelGGL((k_dconv_fwd<false>), dim3((BP / 32) * 6), dim3(256), 0, s, P, x6d, H2T, out, B, BP);
  return 0;
}
int launch_disc_out(const float* P, const float* A2T, float* out, float* dA2T, const float* gout, float scale,
                    float target, int B, int BP, hipStream_t s, float* dz0, float* sq0) {
  hipLaunchKernelGGL(k_disc_out, dim3(BP / 64), dim3(1024), 0, s, P, A2T, out, dA2T, gout, scale, target, B, BP, dz0, sq0);
  return 0;
}
int launch_disc_conv_bwd(const float* P, const float* x6d, const float* dH2T, const float* gout, float scale,
                         float target, float* gx, int B, int BP, hipStream_t s, float* sqj, int quad) {
  if (quad) hipLaunchKernelGGL((k_dconv_bwd<true>), dim3((BP / 32) * 6), dim3(256), 0, s, P, x6d, dH2T, gout, scale, target, gx, B, BP, sqj);
  else hipLaunchKernelGGL((k_dconv_bwd<false>), dim3((BP / 32) * 6), dim3(256), 0, s, P, x6d, dH2T, gout, scale, target, gx, B, BP, sqj);
  return 0;
}
int launch_shape_disc(const float* P, const float* betas, float* out, float* gb, float scale, float target, int B,
                      hipStream_t s, const float* gout, float* sq) {
  hipLaunchKernelGGL(k_shape_disc, dim3((B + 63) / 64), dim3(64), 0, s, P, betas, out, gb, scale, target, B, gout, sq);
  return 0;
}

}  // namespace jrr
