// Generic exact-fp32 MFMA GEMM in the "both operands K-major" form used everywhere in this path:
//     Out[m][n] = epilogue( sum_k A[k][m] * Bm[k][n] )
// Activations are kept feature-major ([feature][pose]), so every product of the pose
// discriminator (scripts/discriminator.py:20-29,32-54), its input-gradient, and the blend-basis
// adjoint dF = D . dVP is of this form with the pose index on the MFMA column (lane) axis.
// LDS-staged [16][BM] / [16][BN] chunks, register prefetch one chunk ahead, optional split-K
// (blockIdx.z) writing partial slabs that the consumer sums.
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

constexpr int GK = 16;   // K rows per staged chunk

template <int WM, int WN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void k_gemm_tn(GemmArgs g) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  constexpr int A4 = GK * BM / 4, B4 = GK * BN / 4;        // float4s per chunk
  constexpr int PA = (A4 + NT - 1) / NT, PB = (B4 + NT - 1) / NT;
  __shared__ float lds[GK * (BM + BN)];
  float* ldsA = lds;
  float* ldsB = lds + GK * BM;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int n_mt = (g.M + BM - 1) / BM, n_nt = g.N / BN;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L % n_mt, nt = L / n_mt;     // consecutive blocks share the Bm (activation) panel
  (void)n_nt;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nchunks = g.K / GK;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int c_begin = (int)((long)nchunks * split / nsplit), c_end = (int)((long)nchunks * (split + 1) / nsplit);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = zero16();

  f32x4 preA[PA], preB[PB];
  auto prefetch = [&](int ch) {
    const int k0 = ch * GK;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int f = tid + NT * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < A4) {
        int row = f / (BM / 4), col = (f % (BM / 4)) * 4;
        if (m0 + col < g.M) v = *reinterpret_cast<const f32x4*>(g.A + (size_t)(k0 + row) * g.lda + m0 + col);
      }
      preA[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int f = tid + NT * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < B4) {
        int row = f / (BN / 4), col = (f % (BN / 4)) * 4;
        v = *reinterpret_cast<const f32x4*>(g.Bm + (size_t)(k0 + row) * g.ldb + n0 + col);
      }
      preB[i] = v;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int f = tid + NT * i;
      if (f < A4) reinterpret_cast<f32x4*>(ldsA)[f] = preA[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int f = tid + NT * i;
      if (f < B4) reinterpret_cast<f32x4*>(ldsB)[f] = preB[i];
    }
  };

  if (c_begin < c_end) prefetch(c_begin);
  for (int ch = c_begin; ch < c_end; ++ch) {
    __syncthreads();
    commit();
    __syncthreads();
    if (ch + 1 < c_end) prefetch(ch + 1);
    const float* ap = ldsA + half * BM + wm * WM * 32 + l31;
    const float* bp = ldsB + half * BN + wn * WN * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < GK / 2; ++kk) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = ap[(2 * kk) * BM + i * 32];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = bp[(2 * kk) * BN + j * 32];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = mfma(a[i], b[j], acc[i][j]);
    }
  }

  float* out = g.Out + (size_t)split * g.split_stride;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + (wm * WM + i) * 32 + acc_row(q, half);
      if (m >= g.M) continue;
      float bias = 0.f;
      if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS) bias = g.bias[m];
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int n = n0 + (wn * WN + j) * 32 + l31;
        float v = acc[i][j][q];
        if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
        if (EPI == EPI_BIAS) v = v + bias;
        if (EPI == EPI_MASK) v = (g.mask[(size_t)m * g.ldo + n] > 0.f) ? v : 0.f;
        out[(size_t)m * g.ldo + n] = v;
      }
    }
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
static int launch_cfg(const GemmArgs& g, int epi, int nsplit, hipStream_t s) {
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  if (g.N % BN != 0 || g.K % GK != 0 || g.lda % 4 != 0 || g.ldb % 4 != 0) {
    jrr_set_error("gemm_tn: unsupported shape M=%d N=%d K=%d lda=%d ldb=%d", g.M, g.N, g.K, g.lda, g.ldb);
    return JRR_ERR_ARG;
  }
  dim3 grid(((g.M + BM - 1) / BM) * (g.N / BN), 1, nsplit), block(64 * WAVES_M * WAVES_N);
  switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, EPI_STORE>), grid, block, 0, s, g); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, EPI_BIAS_RELU>), grid, block, 0, s, g); break;
    case EPI_BIAS: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, EPI_BIAS>), grid, block, 0, s, g); break;
    case EPI_MASK: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, EPI_MASK>), grid, block, 0, s, g); break;
    default: return JRR_ERR_ARG;
  }
  return 0;
}

// 128x128 block tile (2x2 waves of 64x64): discriminator layers
int launch_gemm_128(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 2, 2, 2>(g, epi, nsplit, s); }
// 224x128 block tile (4 waves of 224x32): blend-basis adjoint, M = KFP = 224
int launch_gemm_224(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<7, 1, 1, 4>(g, epi, nsplit, s); }
// 32x128 block tile (4 waves of 32x32): skinny-M products (dJn: M = 17..64)
int launch_gemm_32(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<1, 1, 1, 4>(g, epi, nsplit, s); }

// 64x128 block tile (4 waves of 64x32): the J-regressor gradient product, M = 51 padded to 64
int launch_gemm_64(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 1, 1, 4>(g, epi, nsplit, s); }

}  // namespace jrr
