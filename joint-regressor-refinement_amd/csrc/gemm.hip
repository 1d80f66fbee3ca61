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
// into a 2-deep ring, one chunk ahead of the MFMAs, one workgroup barrier per chunk; partial tiles are masked per lane.
// (The hot products of the loop have their own kernels on quad-layout operands: k_blend_adjoint, k_disc_gemm.  This
// generic kernel serves the folded-regressor tables, the outer step's weight-gradient GEMMs and the joint re-regression
// from stored vertices.)  blockIdx.y batches independent products, blockIdx.z splits K.
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
  const float* const Abase = g.A + (size_t)blockIdx.y * g.batchA + m0;
  const float* const Bbase = g.Bm + (size_t)blockIdx.y * g.batchB + n0;

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
  float* out = g.Out + (size_t)split * g.split_stride + (size_t)blockIdx.y * g.batchO;
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
static int launch_cfg(const GemmArgs& g, int epi, int nsplit, hipStream_t s, int nbatch = 1) {
  constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
  if (g.N % BN != 0 || g.K % GK != 0 || g.lda % 4 != 0 || g.ldb % 4 != 0 || g.M % 4 != 0) {
    jrr_set_error("gemm_tn: unsupported shape M=%d N=%d K=%d lda=%d ldb=%d", g.M, g.N, g.K, g.lda, g.ldb);
    return JRR_ERR_ARG;
  }
  dim3 grid(((g.M + BM - 1) / BM) * (g.N / BN), nbatch, nsplit), block(64 * WAVES_M * WAVES_N);
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

// 128x128 block tile (2x2 waves of 64x64), 32-deep chunks: outer-step weight gradients
int launch_gemm_128(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 2, 2, 2, 32>(g, epi, nsplit, s); }
// 128x64 block tile (2x2 waves of 64x32): row-major discriminator layers of the outer step, folded-regressor product
int launch_gemm_128x64(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<2, 1, 2, 2, 32>(g, epi, nsplit, s); }
// 128x32 block tile (4 waves stacked in M): narrow-N products (folded-regressor table, N = 224 per plane)
int launch_gemm_128x32(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<1, 1, 4, 1, 16>(g, epi, nsplit, s); }
// 224x128 block tile (4 waves of 224x32), 16-deep chunks: folded-regressor adjoint, M = KFP = 224
int launch_gemm_224(const GemmArgs& g, int epi, int nsplit, hipStream_t s) { return launch_cfg<7, 1, 1, 4, 16>(g, epi, nsplit, s); }
// 32x128 block tile (4 waves of 32x32 side by side), batched over gridDim.y: joints = Jn . verts from stored vertices
int launch_gemm_32x128(const GemmArgs& g, int epi, int nsplit, int nbatch, hipStream_t s) {
  return launch_cfg<1, 1, 1, 4, 16>(g, epi, nsplit, s, nbatch);
}

// ------------------------------------------------------------------------------------------------------------------
// "NT" product for operands whose REDUCTION index is the contiguous one (the pose index of the [feature][pose] arrays):
//     Out[split][m][n] = sum_{k in split} A_c[m][k] * B_c[n][k]        m < 32, c = coordinate plane of the split
// J-regressor gradient dJn[i][v] = sum_{c,b} dj[c][i][b] verts_c[v][b]  (scripts/optimize.py:300-312) straight from the
// coordinate-major vertex tiles k_lbs_fwd writes -- no pose-major copy of the 340 MB vertex buffer.
// The MFMA wants the reduction index on the K axis with one value per lane, the data has it contiguous.  Two facts make
// that cheap: (1) a K-pair may take ANY two k as long as both operands agree, so step t of an 8-group takes
// k = 8g + 4*half + t and a lane's four steps are ONE 16-byte piece; (2) an LDS-DMA instruction copies 64 arbitrary
// 16-byte pieces into 1 KB of LDS, so the gather [row][4 k] -> [k-quad][row] is done by the copy itself, and the
// ds_read_b128 operand reads are contiguous per half-wave (conflict-free).
// Tile: 32 (m) x 128 (n), four waves side by side, 32-deep chunks, 2-deep ring.
// ------------------------------------------------------------------------------------------------------------------
struct NtArgs {
  const float* A; size_t planeA; int ldA; int rowsA;   // A_c = A + c*planeA; rows m >= rowsA read row rowsA - 1 (a zero row)
  const float* Bm; size_t planeB; int ldB;             // B_c = Bm + c*planeB; row n at n*ldB
  float* Out; int ldo; size_t split_stride;            // Out[split][32][ldo]
  int K, ksplit;                                        // reduction length per plane, splits per plane
};
__global__ __launch_bounds__(256) void k_gemm_nt32(NtArgs g) {
  constexpr int SA = 8 * 32 * 4, SB = 8 * 128 * 4, SLOT = SA + SB;     // floats: [k-quad 8][row][4]
  __shared__ __attribute__((aligned(16))) float lds[2 * SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * 128;
  const int split = blockIdx.y, c = split / g.ksplit, ks = split % g.ksplit;
  const int nch = g.K / 32;
  const int c_begin = (int)((long)nch * ks / g.ksplit), c_end = (int)((long)nch * (ks + 1) / g.ksplit);
  const float* Ac = g.A + (size_t)c * g.planeA;
  const float* Bc = g.Bm + (size_t)c * g.planeB;
  // this wave's copies per chunk: 4 of B (pieces p = (wave*4 + i)*64 + lane: k-quad p / 128, row p % 128), 1 of A
  unsigned offB[4], offA;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (wave * 4 + i) * 64 + lane;
    offB[i] = (unsigned)(n0 + (p & 127)) * (unsigned)g.ldB + 4u * (unsigned)(p >> 7);
  }
  {
    const int p = wave * 64 + lane, row = p & 31;
    offA = (unsigned)(row < g.rowsA ? row : g.rowsA - 1) * (unsigned)g.ldA + 4u * (unsigned)(p >> 5);
  }
  auto issue = [&](int ch, int slot) {
    const float* a = Ac + (size_t)ch * 32;
    const float* b = Bc + (size_t)ch * 32;
    asm volatile("" : "+s"(a));
    asm volatile("" : "+s"(b));
    float* dA = lds + slot * SLOT;
    float* dB = dA + SA;
    __builtin_amdgcn_global_load_lds(JRR_GLB(a + offA), JRR_LDS(dA + wave * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(JRR_GLB(b + offB[i]), JRR_LDS(dB + (wave * 4 + i) * 256), 16, 0, 0);
  };
  f32x16 acc = zero16();
  if (c_begin < c_end) issue(c_begin, 0);
  for (int ch = c_begin; ch < c_end; ++ch) {
    __syncthreads();
    const int slot = (ch - c_begin) & 1;
    if (ch + 1 < c_end) issue(ch + 1, slot ^ 1);
    const f32x4* la = reinterpret_cast<const f32x4*>(lds + slot * SLOT);
    const f32x4* lb = reinterpret_cast<const f32x4*>(lds + slot * SLOT + SA);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 a4 = la[(gq * 2 + half) * 32 + l31];
      const f32x4 b4 = lb[(gq * 2 + half) * 128 + wave * 32 + l31];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = mfma(a4[t], b4[t], acc);
    }
  }
  float* out = g.Out + (size_t)split * g.split_stride;
#pragma unroll
  for (int q = 0; q < 16; ++q) out[(size_t)acc_row(q, half) * g.ldo + n0 + wave * 32 + l31] = acc[q];
}

int launch_gemm_nt32(const float* A, size_t planeA, int ldA, int rowsA, const float* Bm, size_t planeB, int ldB, float* Out,
                     int ldo, size_t split_stride, int N, int K, int nplanes, int ksplit, hipStream_t s) {
  if (N % 128 != 0 || K % 32 != 0 || ldA % 4 != 0 || ldB % 4 != 0 || rowsA < 1 || rowsA > 32) {
    jrr_set_error("gemm_nt32: unsupported shape N=%d K=%d ldA=%d ldB=%d", N, K, ldA, ldB);
    return JRR_ERR_ARG;
  }
  NtArgs g{A, planeA, ldA, rowsA, Bm, planeB, ldB, Out, ldo, split_stride, K, ksplit};
  hipLaunchKernelGGL(k_gemm_nt32, dim3(N / 128, nplanes * ksplit), dim3(256), 0, s, g);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Blend-basis adjoint  dF^T[m][b] = sum_{c,v} D_c[v][m] dvp_c[v][b]   (M = 224 features, N = poses, K = 3 x 6912 vertices).
// Both operands arrive in VERTEX QUADS ([v/4][column][4]): D from the model upload (Dq), dvp as k_lbs_bwd writes it.
// A K-pair may take any two k as long as A and B agree, so step t of a quad pair takes vertex 4*(2g + half) + t: a lane's
// four steps are ONE 16-byte LDS read per operand tile (7 + 1 ds_read_b128 per 28 MFMAs instead of 32 ds_read_b32).
// Tile 224 x 128, four waves of 224 x 32 (7 accumulator tiles), chunks of 4 quads (16 vertices: A 14 KB -- one
// contiguous block of Dq -- + B 8 KB), 3-deep LDS-DMA ring with counted waits (waves 0,1 issue 6 copies per chunk, waves
// 2,3 issue 5), split-K over blockIdx.y.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BA_QUADS = 4;                                   // quads per chunk
constexpr int BA_SA = BA_QUADS * KFP * 4, BA_SB = BA_QUADS * 128 * 4, BA_SLOT = BA_SA + BA_SB;   // floats
__global__ __launch_bounds__(256, 2) void k_blend_adjoint(const float* __restrict__ Dq, const float* __restrict__ DVPq,
                                                          float* __restrict__ dFTp, size_t split_stride, int BP) {
  __shared__ __attribute__((aligned(16))) float lds[3 * BA_SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nt = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = nt * 128;
  const int split = blockIdx.y, nsplit = gridDim.y;
  constexpr int NCH = 3 * (VP / 4) / BA_QUADS;                // 1296 chunks over the three coordinate planes
  const int c_begin = (int)((long)NCH * split / nsplit), c_end = (int)((long)NCH * (split + 1) / nsplit);
  // copies per chunk: A = 14 linear 1 KB pieces (pieces wave, wave + 4, ...), B = 8 pieces (quad p / 2, poses (p % 2) * 64 + lane)
  auto issue = [&](int ch, int slot) {
    const float* a = Dq + (size_t)ch * BA_SA;                 // chunk ch = quads [4 ch, 4 ch + 4) of the flattened (plane, quad) axis
    const float* b = DVPq + ((size_t)ch * BA_QUADS * BP + n0) * 4;
    asm volatile("" : "+s"(a));
    asm volatile("" : "+s"(b));
    float* dA = lds + slot * BA_SLOT;
    float* dB = dA + BA_SA;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave + 4 * i;
      if (p < 14) __builtin_amdgcn_global_load_lds(JRR_GLB(a + p * 256 + lane * 4), JRR_LDS(dA + p * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave + 4 * i;
      __builtin_amdgcn_global_load_lds(JRR_GLB(b + ((size_t)(p >> 1) * BP + (p & 1) * 64 + lane) * 4), JRR_LDS(dB + p * 256), 16, 0, 0);
    }
  };
  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = zero16();
  if (c_begin < c_end) issue(c_begin, 0);
  if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
  int slot = 0;
  for (int ch = c_begin; ch < c_end; ++ch) {
    // this wave's copies of chunk ch have landed once at most the next chunk's are outstanding (6 for waves 0,1; 5 for 2,3)
    if (ch + 1 < c_end) {
      if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ch + 2 < c_end) issue(ch + 2, slot >= 1 ? slot - 1 : 2);
    const f32x4* la = reinterpret_cast<const f32x4*>(lds + slot * BA_SLOT);
    const f32x4* lb = reinterpret_cast<const f32x4*>(lds + slot * BA_SLOT + BA_SA);
    slot = (slot == 2) ? 0 : slot + 1;
    // operands of quad pair 1 are requested right after the first MFMA of pair 0
    f32x4 a4[7], b4;
#pragma unroll
    for (int i = 0; i < 7; ++i) a4[i] = la[half * KFP + 32 * i + l31];
    b4 = lb[half * 128 + wave * 32 + l31];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      f32x4 ca[7], cb = b4;
#pragma unroll
      for (int i = 0; i < 7; ++i) ca[i] = a4[i];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma(ca[0][0], cb[0], acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (gq == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) a4[i] = la[(2 + half) * KFP + 32 * i + l31];
        b4 = lb[(2 + half) * 128 + wave * 32 + l31];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (t + i > 0) acc[i] = mfma(ca[i][t], cb[t], acc[i]);
    }
  }
  float* out = dFTp + (size_t)split * split_stride;
  const unsigned lane_off = (unsigned)(4 * half) * (unsigned)BP + (unsigned)(n0 + wave * 32 + l31);
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) urow(out, (size_t)(32 * i + acc_row_u(q)), BP)[lane_off] = acc[i][q];
}

int launch_blend_adjoint(const float* Dq, const float* DVPq, float* dFTp, size_t split_stride, int BP, int nsplit, hipStream_t s) {
  hipLaunchKernelGGL(k_blend_adjoint, dim3(BP / 128, nsplit), dim3(256), 0, s, Dq, DVPq, dFTp, split_stride, BP);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Pose-discriminator layers on QUAD-layout operands: Out[m][n] = epilogue(sum_k A[k][m] B[k][n]) with every matrix stored
// as [row/4][column][4] (weights re-laid once per upload; activations written that way by the producing epilogue, whose
// lanes own four consecutive rows 8g + 4 half + {0..3} of their column = one quad = 16 bytes).  Same tricks as
// k_blend_adjoint: a lane's four K steps are one ds_read_b128 per operand tile, stores / mask loads are dwordx4,
// exact tiles, 3-deep LDS-DMA ring with counted waits, operand prefetch one quad pair ahead.
//   tile (32 WM WAVES_M) x (32 WAVES_N), 32-deep chunks (8 quads); EPI / BTR as in k_gemm_tn.
// ------------------------------------------------------------------------------------------------------------------
template <int WM, int WAVES_M, int WAVES_N, int EPI, int BTR>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void k_disc_gemm(GemmArgs g) {
  constexpr int NW = WAVES_M * WAVES_N, BM = 32 * WM * WAVES_M, BN = 32 * WAVES_N;
  constexpr int SA = 8 * BM * 4, SB = 8 * BN * 4, SLOT = SA + SB;            // floats per ring slot
  constexpr int OPA = SA / 256, OPB = SB / 256, PA = OPA / NW, PB = OPB / NW;  // 1 KB copies per chunk / per wave
  static_assert(OPA % NW == 0 && OPB % NW == 0, "exact tiling required");
  __shared__ __attribute__((aligned(16))) float lds[3 * SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int n_mt = g.M / BM;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L % n_mt, nt = L / n_mt;     // consecutive blocks share the activation panel
  const int m0 = mt * BM, n0 = nt * BN;
  const int nch = g.K / 32;
  unsigned offA[PA], offB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int e = (wave + NW * i) * 64 + lane;             // linear 16-byte piece of the [8 quads][BM][4] image
    offA[i] = ((unsigned)(e / BM) * (unsigned)g.M + (unsigned)(m0 + e % BM)) * 4u;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int e = (wave + NW * i) * 64 + lane;
    offB[i] = ((unsigned)(e / BN) * (unsigned)g.ldb + (unsigned)(n0 + e % BN)) * 4u;
  }
  auto issue = [&](int ch, int slot) {
    const float* a = g.A + (size_t)ch * 8 * g.M * 4;
    const float* b = g.Bm + (size_t)ch * 8 * g.ldb * 4;
    asm volatile("" : "+s"(a));
    asm volatile("" : "+s"(b));
    float* dA = lds + slot * SLOT;
    float* dB = dA + SA;
#pragma unroll
    for (int i = 0; i < PA; ++i) __builtin_amdgcn_global_load_lds(JRR_GLB(a + offA[i]), JRR_LDS(dA + (wave + NW * i) * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i) __builtin_amdgcn_global_load_lds(JRR_GLB(b + offB[i]), JRR_LDS(dB + (wave + NW * i) * 256), 16, 0, 0);
  };
  issue(0, 0);
  if (nch > 1) issue(1, 1);
  float cscale = 0.f;                                        // BTR = 2: dz of this lane's column (computed while the copies fly)
  if (BTR == 2) {
    const int n = n0 + wn * 32 + l31;
    float z = g.zbias[0];
    for (int t = 0; t < g.nzpart; ++t) z += g.zpart[(size_t)t * g.ldb + n];
    const float sg = 1.f / (1.f + expf(-z));
    const bool ok = n < g.nvalid;
    const float up = g.gout ? (ok ? g.gout[(size_t)n * g.gout_ld] : 0.f) : g.scale * (sg - g.target);
    cscale = ok ? up * sg * (1.f - sg) : 0.f;
    if (mt == 0 && wm == 0 && half == 0 && ok) {
      if (g.sq0) g.sq0[n] = (sg - g.target) * (sg - g.target);
      if (g.out0) g.out0[(size_t)n * g.out0_ld] = sg;
    }
  }
  f32x16 acc[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) acc[i] = zero16();
  int slot = 0;
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + 2 < nch) issue(ch + 2, slot >= 1 ? slot - 1 : 2);
    const f32x4* la = reinterpret_cast<const f32x4*>(lds + slot * SLOT) + wm * WM * 32 + l31;
    const f32x4* lb = reinterpret_cast<const f32x4*>(lds + slot * SLOT + SA) + wn * 32 + l31;
    slot = (slot == 2) ? 0 : slot + 1;
    f32x4 a4[WM], b4;
#pragma unroll
    for (int i = 0; i < WM; ++i) a4[i] = la[half * BM + 32 * i];
    b4 = lb[half * BN];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      f32x4 ca[WM], cb;
#pragma unroll
      for (int i = 0; i < WM; ++i) ca[i] = a4[i];
#pragma unroll
      for (int t = 0; t < 4; ++t) cb[t] = BTR ? ((b4[t] > 0.f) ? cscale : 0.f) : b4[t];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma(ca[0][0], cb[0], acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (gq + 1 < 4) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a4[i] = la[(2 * gq + 2 + half) * BM + 32 * i];
        b4 = lb[(2 * gq + 2 + half) * BN];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          if (t + i > 0) acc[i] = mfma(ca[i][t], cb[t], acc[i]);
    }
  }
  // epilogue: registers 4q4 .. 4q4+3 of tile i = rows r0 + {0..3}, r0 = m0 + (wm WM + i) 32 + 8 q4 + 4 half: one output quad
  const int n = n0 + wn * 32 + l31;
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int r0u = m0 + (wm * WM + i) * 32 + 8 * q4;            // wave-uniform part of the row
      const int r0 = r0u + 4 * half;
      f32x4 v = {acc[i][4 * q4], acc[i][4 * q4 + 1], acc[i][4 * q4 + 2], acc[i][4 * q4 + 3]};
      if (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS_RELU_DOT) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(g.bias + r0);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u] + bb[u], 0.f);
      }
      if (EPI == EPI_BIAS_RELU_DOT) {
        const f32x4 dw = *reinterpret_cast<const f32x4*>(g.dotw + r0);
#pragma unroll
        for (int u = 0; u < 4; ++u) dot = fmaf(dw[u], v[u], dot);
      }
      const size_t qi = ((size_t)(r0u / 4 + half) * g.ldo + n);     // quad index of (row r0, column n)
      if (EPI == EPI_MASK) {
        const f32x4 mk = reinterpret_cast<const f32x4*>(g.mask)[qi];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (mk[u] > 0.f) ? v[u] : 0.f;
      }
      reinterpret_cast<f32x4*>(g.Out)[qi] = v;
    }
  if (EPI == EPI_BIAS_RELU_DOT) {      // rows of the two lane halves are disjoint: add them, one store per column
    const float t = dot + __shfl_xor(dot, 32);
    if (half == 0) g.dot_out[(size_t)(mt * WAVES_M + wm) * g.ldo + n] = t;
  }
}

// quad-layout discriminator GEMM; returns the number of partial-dot slabs per column through *ndot (EPI_BIAS_RELU_DOT)
int launch_disc_gemm_q(const GemmArgs& g, int epi, int btr, hipStream_t s, int* ndot) {
  if (g.K % 32 != 0 || g.N % 64 != 0 || g.ldb % 4 != 0) { jrr_set_error("disc_gemm_q: unsupported shape M=%d N=%d K=%d", g.M, g.N, g.K); return JRR_ERR_ARG; }
  // 128x64 tiles (4 waves of 64x32); when those give fewer than 512 workgroups but 96x64 tiles (2 waves of 96x32) reach it: those
  const bool t96 = g.M % 96 == 0 && (g.M % 128 != 0 || ((g.M / 128) * (g.N / 64) < 512 && (g.M / 96) * (g.N / 64) >= 512));
  if (!t96 && g.M % 128 != 0) { jrr_set_error("disc_gemm_q: M=%d is neither a multiple of 128 nor of 96", g.M); return JRR_ERR_ARG; }
  if (ndot) *ndot = t96 ? g.M / 96 : (g.M / 128) * 2;
  if (t96) {
    dim3 grid((g.M / 96) * (g.N / 64)), block(128);
    if (epi == EPI_STORE && !btr) hipLaunchKernelGGL((k_disc_gemm<3, 1, 2, EPI_STORE, 0>), grid, block, 0, s, g);
    else { jrr_set_error("disc_gemm_q: 96-row tiles serve the plain store epilogue only"); return JRR_ERR_ARG; }
    return 0;
  }
  dim3 grid((g.M / 128) * (g.N / 64)), block(256);
  if (epi == EPI_BIAS_RELU && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_BIAS_RELU, 0>), grid, block, 0, s, g);
  else if (epi == EPI_BIAS_RELU_DOT && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_BIAS_RELU_DOT, 0>), grid, block, 0, s, g);
  else if (epi == EPI_STORE && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_STORE, 0>), grid, block, 0, s, g);
  else if (epi == EPI_MASK && btr == 2) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_MASK, 2>), grid, block, 0, s, g);
  else { jrr_set_error("disc_gemm_q: unsupported epilogue %d / transform %d", epi, btr); return JRR_ERR_ARG; }
  return 0;
}

// [K][M] (row stride ld) -> quads [K/4][M][4]
__global__ void k_to_quads(const float* __restrict__ in, int ld, float* __restrict__ out, int K, int M) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)K * M) return;
  const int k = (int)(i / M), m = (int)(i % M);
  out[((size_t)(k >> 2) * M + m) * 4 + (k & 3)] = in[(size_t)k * ld + m];
}
int launch_to_quads(const float* in, int ld, float* out, int K, int M, hipStream_t s) {
  hipLaunchKernelGGL(k_to_quads, dim3((unsigned)(((size_t)K * M + 255) / 256)), dim3(256), 0, s, in, ld, out, K, M);
  return 0;
}

}  // namespace jrr
