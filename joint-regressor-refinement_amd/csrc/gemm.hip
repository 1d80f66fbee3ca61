// Generic exact-fp32 MFMA GEMM in the "both operands K-major" form used everywhere in this path:
//     Out[m][n] = epilogue( sum_k A[k][m] * Bm[k][n] )
// Activations are kept feature-major ([feature][pose]), so every product of the pose
// discriminator (scripts/discriminator.py:20-29,32-54), its input-gradient, and the blend-basis
// adjoint dF = D . dVP is of this form with the pose index on the MFMA column (lane) axis.
// Optional split-K (blockIdx.z) writes partial slabs that the consumer sums.
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {


// Operand chunks ([GK][BM] of A, [GK][BN] of Bm) are staged by LDS-DMA (global_load_lds_dwordx4)
// into a 2-deep ring, one chunk ahead of the MFMAs, one workgroup barrier per chunk.
template <int WM, int WN, int WAVES_M, int WAVES_N, int GK, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void k_gemm_tn(GemmArgs g) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  constexpr int A16 = GK * BM / 4, B16 = GK * BN / 4;       // 16-byte pieces per chunk
  constexpr int OPA = (A16 + 63) / 64, OPB = (B16 + 63) / 64;   // wave-instructions per chunk
  constexpr int PA = (OPA + NW - 1) / NW, PB = (OPB + NW - 1) / NW;
  constexpr int SLOT = GK * (BM + BN);
  __shared__ float lds[2 * SLOT];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int n_mt = (g.M + BM - 1) / BM;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L % n_mt, nt = L / n_mt;     // consecutive blocks share the Bm (activation) panel
  const int m0 = mt * BM, n0 = nt * BN;
  const int nchunks = g.K / GK;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int c_begin = (int)((long)nchunks * split / nsplit), c_end = (int)((long)nchunks * (split + 1) / nsplit);

  // per-lane source offsets (floats) of this wave's DMA pieces; loop-invariant
  unsigned offA[PA], offB[PB];
  bool okA[PA], okB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int p = (wave + NW * i) * 64 + lane;
    const int row = p / (BM / 4), col = (p % (BM / 4)) * 4;
    okA[i] = (wave + NW * i) < OPA && p < A16 && (m0 + col) < g.M;
    offA[i] = (unsigned)row * (unsigned)g.lda + (unsigned)col;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int p = (wave + NW * i) * 64 + lane;
    const int row = p / (BN / 4), col = (p % (BN / 4)) * 4;
    okB[i] = (wave + NW * i) < OPB && p < B16;
    offB[i] = (unsigned)row * (unsigned)g.ldb + (unsigned)col;
  }
  const float* const Abase = g.A + m0;
  const float* const Bbase = g.Bm + n0;

  auto issue = [&](int ch, int slot) {
    const float* a = Abase + (size_t)ch * GK * g.lda;
    const float* b = Bbase + (size_t)ch * GK * g.ldb;
    asm volatile("" : "+s"(a));
    asm volatile("" : "+s"(b));
    float* dA = lds + slot * SLOT;
    float* dB = dA + GK * BM;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      if (okA[i]) __builtin_amdgcn_global_load_lds(JRR_GLB(a + offA[i]), JRR_LDS(dA + (wave + NW * i) * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (okB[i]) __builtin_amdgcn_global_load_lds(JRR_GLB(b + offB[i]), JRR_LDS(dB + (wave + NW * i) * 256), 16, 0, 0);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = zero16();

  if (c_begin < c_end) issue(c_begin, 0);
  for (int ch = c_begin; ch < c_end; ++ch) {
    __syncthreads();   // chunk ch landed (vmcnt(0) + barrier); the other slot is free
    if (ch + 1 < c_end) issue(ch + 1, (ch - c_begin + 1) & 1);
    const float* ap = lds + ((ch - c_begin) & 1) * SLOT + half * BM + wm * WM * 32 + l31;
    const float* bp = lds + ((ch - c_begin) & 1) * SLOT + GK * BM + half * BN + wn * WN * 32 + l31;
    // The operands of K-pair kk+1 are requested right after the first MFMA of pair kk has issued, so the reads
    // complete under this pair's MFMAs (the compiler's own schedule reads right before use and exposes the LDS
    // latency once per pair).
    float a[WM], b[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = ap[i * 32];
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = bp[j * 32];
#pragma unroll
    for (int kk = 0; kk < GK / 2; ++kk) {
      float ca[WM], cb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) ca[i] = a[i];
#pragma unroll
      for (int j = 0; j < WN; ++j) cb[j] = b[j];
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = mfma(ca[0], cb[0], acc[0][0]);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 1 < GK / 2) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = ap[(2 * kk + 2) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = bp[(2 * kk + 2) * BN + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
          if (i + j > 0) acc[i][j] = mfma(ca[i], cb[j], acc[i][j]);
    }
  }

  // epilogue: the row part of every address is wave-uniform (kept in SGPRs), the lane part (4 * half rows + column)
  // is one 32-bit offset -- one store per element instead of five VALU instructions of address arithmetic
  float* out = g.Out + (size_t)split * g.split_stride;
  const unsigned lane_off = (unsigned)(4 * half) * (unsigned)g.ldo + (unsigned)(n0 + l31);
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int mu = m0 + (wm * WM + i) * 32 + (q & 3) + 8 * (q >> 2);   // acc_row(q, half) = mu-part + 4 half
      const int m = mu + 4 * half;
      if (m >= g.M) continue;
      float bias = 0.f;
      if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS) bias = g.bias[m];
      float* orow = out + (size_t)mu * g.ldo;
      asm volatile("" : "+s"(orow));
      const float* mrow = (EPI == EPI_MASK) ? g.mask + (size_t)mu * g.ldo : nullptr;
      if (EPI == EPI_MASK) asm volatile("" : "+s"(mrow));
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const unsigned off = lane_off + (unsigned)((wn * WN + j) * 32);
        float v = acc[i][j][q];
        if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
        if (EPI == EPI_BIAS) v = v + bias;
        if (EPI == EPI_MASK) v = (mrow[off] > 0.f) ? v : 0.f;
        if (EPI == EPI_ACCUM) v += orow[off];
        orow[off] = v;
      }
    }
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int GK>
static int launch_cfg(const GemmArgs& g, int epi, int nsplit, hipStream_t s) {
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  if (g.N % BN != 0 || g.K % GK != 0 || g.lda % 4 != 0 || g.ldb % 4 != 0 || g.M % 4 != 0) {
    jrr_set_error("gemm_tn: unsupported shape M=%d N=%d K=%d lda=%d ldb=%d", g.M, g.N, g.K, g.lda, g.ldb);
    return JRR_ERR_ARG;
  }
  dim3 grid(((g.M + BM - 1) / BM) * (g.N / BN), 1, nsplit), block(64 * WAVES_M * WAVES_N);
  switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, GK, EPI_STORE>), grid, block, 0, s, g); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, GK, EPI_BIAS_RELU>), grid, block, 0, s, g); break;
    case EPI_BIAS: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, GK, EPI_BIAS>), grid, block, 0, s, g); break;
    case EPI_MASK: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, GK, EPI_MASK>), grid, block, 0, s, g); break;
    case EPI_ACCUM: hipLaunchKernelGGL((k_gemm_tn<WM, WN, WAVES_M, WAVES_N, GK, EPI_ACCUM>), grid, block, 0, s, g); break;
    default: return JRR_ERR_ARG;
  }
  return 0;
}

// 128x128 block tile (2x2 waves of 64x64), 32-deep chunks: discriminator layers
int launch_gemm_128(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 2, 2, 2, 32>(g, epi, nsplit, s); }
// 128x64 block tile (2x2 waves of 64x32): twice the workgroups of the 128x128 tile (2 per CU at N = 4096)
int launch_gemm_128x64(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 1, 2, 2, 32>(g, epi, nsplit, s); }
// 128x128 block tile, 8 waves (2x4 of 64x32): 2 waves per SIMD with one workgroup per CU
int launch_gemm_128w8(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 1, 2, 4, 32>(g, epi, nsplit, s); }
// 128x32 block tile (4 waves stacked in M): narrow-N products (folded-regressor table, N = 224 per plane)
int launch_gemm_128x32(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<1, 1, 4, 1, 16>(g, epi, nsplit, s); }
// 224x128 block tile (4 waves of 224x32), 16-deep chunks: blend-basis adjoint, M = KFP = 224
int launch_gemm_224(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<7, 1, 1, 4, 16>(g, epi, nsplit, s); }
// 64x128 block tile (4 waves of 64x32): the J-regressor gradient product, M = 51 padded to 64
int launch_gemm_64(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 1, 1, 4, 16>(g, epi, nsplit, s); }

}  // namespace jrr
