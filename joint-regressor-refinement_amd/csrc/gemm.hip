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

// ------------------------------------------------------------------------------------------------------------------
// Products against the STORED vertices (row quads [3][VP/4][BP][4], written by k_lbs_fwd) -- the 340 MB buffer is read
// once per product, in 1 KB runs, never transposed:
//
// k_gemm_q32   joints^T[i][b] = sum_v Jn[i][v] verts_c[v][b]          (reduction over the vertex = the QUAD index)
//     Both operands are K-quads [k/4][row][4]: a lane's ds_read_b128 is four K-steps (a K-pair may take any two k as
//     long as both operands agree: step t of an 8-group takes k = 8g + 4*half + t).  Tile 32 (m) x 128 (n), four waves
//     side by side, 16-deep chunks, 4-deep LDS-DMA ring with counted vmcnt, split over k.
//
// k_jgrad_q    dJn[i][v] = sum_{c,b} dj[c][i][b] verts_c[v][b]        (scripts/optimize.py:300-312; reduction over the
//     POSE index, the vertex quads are the OUTPUT columns).  A 16-byte piece of the vertex buffer is four vertices of one
//     pose = four output columns of one K row: the lane with column index l reads the piece of vertex quad l and feeds
//     element u to accumulator u, whose column l therefore is vertex 4 l + u -- four accumulators are a 32 x 128 tile
//     with permuted columns, and the epilogue's 16-byte store puts them back in order.  The joint adjoint dj^T [18][BP]
//     (pose-contiguous rows) is the A operand: the LDS-DMA gathers [row][4 poses] pieces into K-quads.  A workgroup
//     owns 128 vertices of one plane and a range of poses; its four waves split every 32-pose chunk (8 poses each) and
//     add their tiles through LDS at the end.  The LDS-DMA stores every vertex-quad row rotated by its own index, so the
//     32 lanes of a half-wave (32 different rows, same pose) read conflict-free without padding; 4-deep ring of 20 KB
//     slots (three chunks in flight: throughput of this stream = bytes in flight / ~5.7 us), counted vmcnt, two workgroups per CU = one round of 162 x 3 workgroups.
// ------------------------------------------------------------------------------------------------------------------
struct Q32Args {
  const float* A; int ldA; size_t planeA;     // A quads [K/4][ldA][4], rows 0..31 used; plane c at A + c*planeA
  const float* Bm; int ldB; size_t planeB;    // B quads [K/4][ldB][4]
  float* Out; int ldo; size_t ks_stride, plane_stride;   // tile (k-split ks, plane c) at Out + ks*ks_stride + c*plane_stride
  int K, ksplit;
  const int* skip;                            // nullable: *skip != 0 -> nothing to do
};
__global__ __launch_bounds__(256) void k_gemm_q32(Q32Args g) {
  if (g.skip && *g.skip) return;
  // 16-deep chunks (4 k-quads): A 2 KB + B 8 KB per slot, 4-deep ring (three chunks in flight: the product is a
  // stream over the 340 MB vertex buffer, 32 MFMA-rows of work per 128-byte column -- bytes in flight are what counts)
  constexpr int SA = 4 * 32 * 4, SB = 4 * 128 * 4, SLOT = SA + SB, RING = 4;
  __shared__ __attribute__((aligned(16))) float lds[RING * SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * 128;
  const int split = blockIdx.y, c = split / g.ksplit, ks = split % g.ksplit;
  const int nch = g.K / 16;
  const int c_begin = (int)((long)nch * ks / g.ksplit), c_end = (int)((long)nch * (ks + 1) / g.ksplit);
  const float* Ac = g.A + (size_t)c * g.planeA;
  const float* Bc = g.Bm + (size_t)c * g.planeB;
  // this wave's copies per chunk: 2 of B (pieces p = (wave*2 + i)*64 + lane: k-quad p / 128, row p % 128) and -- waves
  // 0 and 1 -- 1 of A (pieces p = wave*64 + lane: k-quad p / 32, row p % 32)
  unsigned offB[2], offA;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (wave * 2 + i) * 64 + lane;
    offB[i] = ((unsigned)(p >> 7) * (unsigned)g.ldB + (unsigned)(n0 + (p & 127))) * 4u;
  }
  {
    const int p = (wave & 1) * 64 + lane;
    offA = ((unsigned)(p >> 5) * (unsigned)g.ldA + (unsigned)(p & 31)) * 4u;
  }
  const bool copiesA = wave < 2;
  auto issue = [&](int ch, int slot) {
    const float* a = Ac + (size_t)ch * 4 * g.ldA * 4;
    const float* b = Bc + (size_t)ch * 4 * g.ldB * 4;
    asm volatile("" : "+s"(a));
    asm volatile("" : "+s"(b));
    float* dA = lds + slot * SLOT;
    float* dB = dA + SA;
    if (copiesA) __builtin_amdgcn_global_load_lds(JRR_GLB(a + offA), JRR_LDS(dA + wave * 256), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds(JRR_GLB(b + offB[i]), JRR_LDS(dB + (wave * 2 + i) * 256), 16, 0, 0);
  };
  f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
  for (int i = 0; i < RING - 1; ++i)
    if (c_begin + i < c_end) issue(c_begin + i, i);
  int slot = 0;
  for (int ch = c_begin; ch < c_end; ++ch) {
    // chunk ch landed: all but the copies of the (up to) two younger chunks; everybody is done with chunk ch - 1
    const int younger = (c_end - 1 - ch < 2) ? c_end - 1 - ch : 2;
    if (copiesA) { if (younger == 2) barrier_keep_vm<6>(); else if (younger == 1) barrier_keep_vm<3>(); else barrier_keep_vm<0>(); }
    else         { if (younger == 2) barrier_keep_vm<4>(); else if (younger == 1) barrier_keep_vm<2>(); else barrier_keep_vm<0>(); }
    if (ch + RING - 1 < c_end) issue(ch + RING - 1, (slot + RING - 1) & (RING - 1));
    const f32x4* la = reinterpret_cast<const f32x4*>(lds + slot * SLOT);
    const f32x4* lb = reinterpret_cast<const f32x4*>(lds + slot * SLOT + SA);
    const f32x4 a0 = la[half * 32 + l31], b0 = lb[half * 128 + wave * 32 + l31];
    const f32x4 a1 = la[(2 + half) * 32 + l31], b1 = lb[(2 + half) * 128 + wave * 32 + l31];
#pragma unroll
    for (int t = 0; t < 4; ++t) {       // two accumulator chains (a dependent MFMA issues every 112 clocks, not 64)
      acc0 = mfma(a0[t], b0[t], acc0);
      acc1 = mfma(a1[t], b1[t], acc1);
    }
    slot = (slot + 1) & (RING - 1);
  }
  float* out = g.Out + (size_t)ks * g.ks_stride + (size_t)c * g.plane_stride;
#pragma unroll
  for (int q = 0; q < 16; ++q) out[(size_t)acc_row(q, half) * g.ldo + n0 + wave * 32 + l31] = acc0[q] + acc1[q];
}

int launch_gemm_q32(const float* A, int ldA, size_t planeA, const float* Bm, int ldB, size_t planeB, float* Out, int ldo,
                    size_t ks_stride, size_t plane_stride, int N, int K, int nplanes, int ksplit, hipStream_t s,
                    const int* skip_flag) {
  if (N % 128 != 0 || K % 16 != 0 || ldA < 32 || ldB < N) {
    jrr_set_error("gemm_q32: unsupported shape N=%d K=%d ldA=%d ldB=%d", N, K, ldA, ldB);
    return JRR_ERR_ARG;
  }
  Q32Args g{A, ldA, planeA, Bm, ldB, planeB, Out, ldo, ks_stride, plane_stride, K, ksplit, skip_flag};
  hipLaunchKernelGGL(k_gemm_q32, dim3(N / 128, nplanes * ksplit), dim3(256), 0, s, g);
  return 0;
}

struct JgArgs {
  const float* dJT;    // [3][NHP][BP] joint adjoint, row NHP - 1 of every plane is zero
  const float* VTq;    // [3][VP/4][BP][4]
  float* Out;          // [3 * ksplit][32][VP] partial slabs
  int BP, ksplit;
  const int* skip;     // nullable: *skip != 0 -> nothing to do
};
constexpr int JG_SB = 32 * 32 * 4, JG_SA = 8 * 32 * 4, JG_SLOT = JG_SB + JG_SA, JG_RING = 4;   // 16 KB + 4 KB per slot
__global__ __launch_bounds__(256, 2) void k_jgrad_q(JgArgs g) {
  if (g.skip && *g.skip) return;
  __shared__ __attribute__((aligned(16))) float lds[JG_RING * JG_SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int vq0 = blockIdx.x * 32;                 // first vertex quad of this workgroup's 128 vertices
  const int split = blockIdx.y, c = split / g.ksplit, ks = split % g.ksplit;
  const int nch = g.BP / 32;
  const int c_begin = (int)((long)nch * ks / g.ksplit), c_end = (int)((long)nch * (ks + 1) / g.ksplit);
  const float* Ac = g.dJT + (size_t)c * NHP * g.BP;
  const float* Bc = g.VTq + ((size_t)c * (VP / 4) + vq0) * g.BP * 4;
  // Per chunk (32 poses) this wave copies 8 vertex-quad rows -- instruction i: rows r = 2 (4 wave + i) + half, the lane's
  // piece is pose (l31 + r) % 32 of that row: each row lands ROTATED by its own index, which is what makes the reads
  // below (32 rows, one pose) hit 16 different bank groups per 16 lanes without padding -- and one instruction of
  // gathered dj^T pieces (piece p = wave*64 + lane: pose quad p / 32, joint row p % 32; rows >= 18 read the zero row 17).
  unsigned offB[4], offA;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 2 * (wave * 4 + i) + half;
    offB[i] = ((unsigned)r * (unsigned)g.BP + (unsigned)((l31 + r) & 31)) * 4u;
  }
  {
    const int p = wave * 64 + lane, row = p & 31;
    offA = (unsigned)(row < NHP ? row : NHP - 1) * (unsigned)g.BP + 4u * (unsigned)(p >> 5);
  }
  auto issue = [&](int ch, int slot) {
    const float* a = Ac + (size_t)ch * 32;
    const float* b = Bc + (size_t)ch * 128;
    asm volatile("" : "+s"(a));
    asm volatile("" : "+s"(b));
    float* dB = lds + slot * JG_SLOT;
    float* dA = dB + JG_SB;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(JRR_GLB(b + offB[i]), JRR_LDS(dB + (wave * 4 + i) * 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(JRR_GLB(a + offA), JRR_LDS(dA + wave * 256), 16, 0, 0);
  };
  f32x16 acc[4] = {zero16(), zero16(), zero16(), zero16()};
  // All workgroups of a pose range would walk it in lockstep, i.e. request, at any moment, addresses that differ by
  // multiples of the 64 KB+ row stride -- the same few HBM channels.  Each vertex tile therefore starts its (cyclic) walk
  // at a different chunk; the order of the sum is still fixed per workgroup.
  const int ncl = c_end - c_begin;
  const int start = ncl > 0 ? (int)(blockIdx.x * 5u % (unsigned)ncl) : 0;
  auto chunk_at = [&](int i) { const int o = start + i; return c_begin + (o >= ncl ? o - ncl : o); };
#pragma unroll
  for (int i = 0; i < JG_RING - 1; ++i)
    if (i < ncl) issue(chunk_at(i), i);
  const int kq = wave * 2 + half;                        // pose quad of this lane's four K-steps (this wave's 8 poses)
  const int rot = (kq * 4 - l31) & 31;                   // where pose 4 kq sits in row l31
  int slot = 0;
  for (int i = 0; i < ncl; ++i) {
    // chunk i landed (its 5 copies are older than the 5 of each younger chunk in flight); everybody is done with i - 1
    if (i + 2 < ncl) barrier_keep_vm<10>();
    else if (i + 1 < ncl) barrier_keep_vm<5>();
    else barrier_keep_vm<0>();
    if (i + JG_RING - 1 < ncl) issue(chunk_at(i + JG_RING - 1), (slot + JG_RING - 1) & (JG_RING - 1));
    const float* lb = lds + slot * JG_SLOT + l31 * 128;
    const f32x4 a4 = reinterpret_cast<const f32x4*>(lds + slot * JG_SLOT + JG_SB)[kq * 32 + l31];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(lb + ((rot + t) & 31) * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = mfma(a4[t], b4[u], acc[u]);
    }
    slot = (slot + 1) & (JG_RING - 1);
  }
  // add the four waves' tiles, two accumulators at a time: red[wave][q * 2 + uu][lane]  (32 KB)
  float* red = lds;
  f32x4 o[4];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    __syncthreads();
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int q = 0; q < 16; ++q) red[((wave * 32) + q * 2 + uu) * 64 + lane] = acc[2 * h2 + uu][q];
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        const int q = wave * 4 + qq;                  // this wave finishes accumulator registers 4 wave .. 4 wave + 3
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sacc += red[((w * 32) + q * 2 + uu) * 64 + lane];
        o[qq][2 * h2 + uu] = sacc;
      }
  }
  float* out = g.Out + (size_t)split * 32 * VP;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq)
    *reinterpret_cast<f32x4*>(out + (size_t)acc_row(wave * 4 + qq, half) * VP + (size_t)(vq0 + l31) * 4) = o[qq];
}

int launch_jgrad_q(const float* dJT, const float* VTq, float* Out, int BP, int ksplit, hipStream_t s, const int* skip_flag) {
  if (BP % 32 != 0 || ksplit < 1 || ksplit > BP / 32) { jrr_set_error("jgrad_q: unsupported shape BP=%d ksplit=%d", BP, ksplit); return JRR_ERR_ARG; }
  JgArgs g{dJT, VTq, Out, BP, ksplit, skip_flag};
  hipLaunchKernelGGL(k_jgrad_q, dim3(VP / 128, 3 * ksplit), dim3(256), 0, s, g);
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
template <bool LIST>      // LIST: the K range is the rows of the ntl tiles tl[] (the others hold a zero dvp nobody wrote: k_lbs_bwd16<LIST>)
__global__ __launch_bounds__(256, 2) void k_blend_adjoint(const float* __restrict__ Dq, const float* __restrict__ DVPq,
                                                          float* __restrict__ dFTp, size_t split_stride, int BP,
                                                          const int* __restrict__ tl, int ntl) {
  __shared__ __attribute__((aligned(16))) float lds[3 * BA_SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nt = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = nt * 128;
  const int split = blockIdx.y, nsplit = gridDim.y;
  constexpr int NCH_ALL = 3 * (VP / 4) / BA_QUADS;            // 1296 chunks over the three coordinate planes (two per tile and plane)
  const int NCH = LIST ? 6 * ntl : NCH_ALL;
  const int c_begin = (int)((long)NCH * split / nsplit), c_end = (int)((long)NCH * (split + 1) / nsplit);
  // listed chunk k = (plane k / (2 ntl), tile tl[(k % (2 ntl)) / 2], half k % 2)
  auto chunk_of = [&](int k) { return LIST ? (k / (2 * ntl)) * (NCH_ALL / 3) + 2 * tl[(k % (2 * ntl)) >> 1] + (k & 1) : k; };
  // copies per chunk: A = 14 linear 1 KB pieces (pieces wave, wave + 4, ...), B = 8 pieces (quad p / 2, poses (p % 2) * 64 + lane)
  auto issue = [&](int kch, int slot) {
    const int ch = chunk_of(kch);
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
      // dvp is read exactly once (340 MB): non-temporal (aux bit 1), so that it does not push the basis D and this kernel's own
      // output slabs out of L2 / the Infinity Cache (measured: the slab sum that follows 52 -> 46 us, this kernel -1 us)
      __builtin_amdgcn_global_load_lds(JRR_GLB(b + ((size_t)(p >> 1) * BP + (p & 1) * 64 + lane) * 4), JRR_LDS(dB + p * 256), 16, 0, 2);
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

int launch_blend_adjoint(const float* Dq, const float* DVPq, float* dFTp, size_t split_stride, int BP, int nsplit, hipStream_t s,
                         const int* tl, int ntl) {
  if (tl) hipLaunchKernelGGL(k_blend_adjoint<true>, dim3(BP / 128, nsplit), dim3(256), 0, s, Dq, DVPq, dFTp, split_stride, BP, tl, ntl);
  else hipLaunchKernelGGL(k_blend_adjoint<false>, dim3(BP / 128, nsplit), dim3(256), 0, s, Dq, DVPq, dFTp, split_stride, BP, nullptr, 0);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// SIDE MODE (JRR_FLAG_BLEND_BF16X3) -- NOT the reference's arithmetic, never the default, never what bench.py's `value` runs.
// The blend-basis adjoint as a SPLIT-bf16 product with fp32 accumulation: every fp32 operand x is taken as hi + lo,
// hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest-even; what is dropped is <= 2^-17 |x|), and
//     dF^T += D_hi dvp_hi + D_hi dvp_lo + D_lo dvp_hi        (the lo x lo term, <= 2^-16 of the product, is dropped)
// in three v_mfma_f32_32x32x16_bf16 per accumulator tile and 16 vertices -- where the exact-fp32 kernel above issues eight
// 32x32x2 instructions of twice the duration each: 5.3 x less matrix-pipe time, which leaves this kernel bound by its operand
// traffic (the same 22 KB per chunk through the same LDS-DMA ring).
//   D is split ONCE (k_split_blend_basis, at engine creation) into the layout a lane reads with one ds_read_b128 per tile and part:
//   chunk ch (16 vertices = one matrix-instruction K step) = [part hi | lo][K half h][224 rows][8 bf16], element j of (h, row m) = vertex
//   16 ch + 8 h + j -- 14 336 bytes per chunk, the size of the fp32 chunk it replaces.  dvp is split in registers, per chunk:
//   a lane's eight K values of its pose column are two 16-byte LDS reads (quads 2 h, 2 h + 1).
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BX_CHUNKS = 3 * (VP / 4) / BA_QUADS;            // 1296 chunks of 16 vertices over the three coordinate planes
__global__ __launch_bounds__(256) void k_split_blend_basis(const float* __restrict__ Dq, bf16x8* __restrict__ Ds) {
  const int i = blockIdx.x * 256 + threadIdx.x;               // (chunk, K half, row)
  if (i >= BX_CHUNKS * 2 * KFP) return;
  const int m = i % KFP, h = (i / KFP) & 1, ch = i / (2 * KFP);
  const f32x4* q = reinterpret_cast<const f32x4*>(Dq) + ((size_t)ch * BA_QUADS + 2 * h) * KFP + m;
  const f32x4 x0 = q[0], x1 = q[KFP];
  bf16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? x0[j] : x1[j - 4];
    const __bf16 t = (__bf16)x;
    hi[j] = t;
    lo[j] = (__bf16)(x - (float)t);
  }
  bf16x8* o = Ds + (size_t)ch * (4 * KFP);
  o[(0 * 2 + h) * KFP + m] = hi;
  o[(1 * 2 + h) * KFP + m] = lo;
}
int launch_split_blend_basis(const float* Dq, void* Ds, hipStream_t s) {
  hipLaunchKernelGGL(k_split_blend_basis, dim3((BX_CHUNKS * 2 * KFP + 255) / 256), dim3(256), 0, s, Dq, reinterpret_cast<bf16x8*>(Ds));
  return 0;
}
size_t blend_basis_split_bytes() { return (size_t)BX_CHUNKS * 4 * KFP * sizeof(bf16x8); }

__global__ __launch_bounds__(256, 2) void k_blend_adjoint_bf16x3(const bf16x8* __restrict__ Ds, const float* __restrict__ DVPq,
                                                                 float* __restrict__ dFTp, size_t split_stride, int BP) {
  static_assert(4 * KFP * sizeof(bf16x8) == BA_SA * sizeof(float), "a split chunk of D is as large as the fp32 chunk it replaces");
  __shared__ __attribute__((aligned(16))) float lds[3 * BA_SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int nt = xcd_remap(blockIdx.x, gridDim.x);
  const int n0 = nt * 128;
  const int split = blockIdx.y, nsplit = gridDim.y;
  const int c_begin = (int)((long)BX_CHUNKS * split / nsplit), c_end = (int)((long)BX_CHUNKS * (split + 1) / nsplit);
  auto issue = [&](int ch, int slot) {        // the fp32 kernel's ring: A = 14 linear 1 KB pieces, B = 8 pieces (quad p / 2, poses (p % 2) * 64 + lane)
    const float* a = reinterpret_cast<const float*>(Ds) + (size_t)ch * BA_SA;
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
      __builtin_amdgcn_global_load_lds(JRR_GLB(b + ((size_t)(p >> 1) * BP + (p & 1) * 64 + lane) * 4), JRR_LDS(dB + p * 256), 16, 0, 2);
    }
  };
  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) acc[i] = zero16();
  if (c_begin < c_end) issue(c_begin, 0);
  if (c_begin + 1 < c_end) issue(c_begin + 1, 1);
  int slot = 0;
  for (int ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) {
      if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (ch + 2 < c_end) issue(ch + 2, slot >= 1 ? slot - 1 : 2);
    const bf16x8* la = reinterpret_cast<const bf16x8*>(lds + slot * BA_SLOT) + half * KFP + l31;      // + part * 2 KFP + 32 i
    const f32x4* lb = reinterpret_cast<const f32x4*>(lds + slot * BA_SLOT + BA_SA) + (2 * half) * 128 + wave * 32 + l31;
    slot = (slot == 2) ? 0 : slot + 1;
    const f32x4 b0 = lb[0], b1 = lb[128];
    bf16x8 ah[7], al[7], bh, bl;
#pragma unroll
    for (int i = 0; i < 7; ++i) ah[i] = la[32 * i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = j < 4 ? b0[j] : b1[j - 4];
      const __bf16 t = (__bf16)x;
      bh[j] = t;
      bl[j] = (__bf16)(x - (float)t);
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) al[i] = la[2 * KFP + 32 * i];
    // three rounds over the seven tiles: consecutive matrix instructions never share an accumulator
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i], 0, 0, 0);
  }
  float* out = dFTp + (size_t)split * split_stride;
  const unsigned lane_off = (unsigned)(4 * half) * (unsigned)BP + (unsigned)(n0 + wave * 32 + l31);
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) urow(out, (size_t)(32 * i + acc_row_u(q)), BP)[lane_off] = acc[i][q];
}
int launch_blend_adjoint_bf16x3(const void* Ds, const float* DVPq, float* dFTp, size_t split_stride, int BP, int nsplit, hipStream_t s) {
  hipLaunchKernelGGL(k_blend_adjoint_bf16x3, dim3(BP / 128, nsplit), dim3(256), 0, s, reinterpret_cast<const bf16x8*>(Ds), DVPq, dFTp, split_stride, BP);
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
// KS = 2: two waves per (wm, wn) position split every 32-deep chunk's K steps between them (quad pairs 0-1 / 2-3) and add their
// accumulators through LDS at the end -- twice the waves on the same tile and the same LDS footprint: the 96 x 64 tile (M = 768:
// 512 workgroups of two waves = ONE wave per SIMD) gets the second wave per SIMD that covers an LDS wait.
template <int WM, int WAVES_M, int WAVES_N, int EPI, int BTR, int KS = 1>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * KS) void k_disc_gemm(GemmArgs g) {
  constexpr int NW = WAVES_M * WAVES_N * KS, BM = 32 * WM * WAVES_M, BN = 32 * WAVES_N;
  constexpr int SA = 8 * BM * 4, SB = 8 * BN * 4, SLOT = SA + SB;            // floats per ring slot
  constexpr int OPA = SA / 256, OPB = SB / 256, PA = OPA / NW, PB = OPB / NW;  // 1 KB copies per chunk / per wave
  static_assert(OPA % NW == 0 && OPB % NW == 0, "exact tiling required");
  __shared__ __attribute__((aligned(16))) float lds[3 * SLOT];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int ks = wave / (WAVES_M * WAVES_N), wpos = wave % (WAVES_M * WAVES_N);
  const int wm = wpos / WAVES_N, wn = wpos % WAVES_N;
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
  // BTR = 2 (the adjoint of fc2, scripts/discriminator.py:47-52 backwards): B holds relu'(a2) as 0 / 1 -- written by the
  // EPI_BIAS_RELU_DOT epilogue of the forward product -- and the column factor dz, which does not depend on the summation index,
  // multiplies the finished sums in the epilogue: no per-element transform between the LDS read and the matrix instruction.
  float cscale = 0.f;                                        // dz of this lane's column (computed while the copies fly)
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
  // Every variant sums K in the SAME order -- per chunk the quad pairs 0-1 into one accumulator set and 2-3 into another, the two
  // added at the end -- so that the K-split tile (KS = 2: the two sets live in two waves) and the one-wave-per-position tiles give
  // bit-identical results: which tile shape a product gets depends on the batch size, and engines of different batch sizes are
  // compared bit for bit (tests/test_gpu_parity.py)
  constexpr int NACC = (KS == 1) ? 2 : 1;
  f32x16 accs[NACC][WM];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < WM; ++i) accs[a][i] = zero16();
  int slot = 0;
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ch + 2 < nch) issue(ch + 2, slot >= 1 ? slot - 1 : 2);
    const f32x4* la = reinterpret_cast<const f32x4*>(lds + slot * SLOT) + wm * WM * 32 + l31;
    const f32x4* lb = reinterpret_cast<const f32x4*>(lds + slot * SLOT + SA) + wn * 32 + l31;
    slot = (slot == 2) ? 0 : slot + 1;
    constexpr int GQ = 4 / KS;                            // quad pairs of a chunk this wave multiplies: ks GQ .. ks GQ + GQ - 1
    const int gq0 = ks * GQ;
    f32x4 a4[WM], b4;
#pragma unroll
    for (int i = 0; i < WM; ++i) a4[i] = la[(2 * gq0 + half) * BM + 32 * i];
    b4 = lb[(2 * gq0 + half) * BN];
#pragma unroll
    for (int gq = 0; gq < GQ; ++gq) {
      f32x4 ca[WM], cb;
#pragma unroll
      for (int i = 0; i < WM; ++i) ca[i] = a4[i];
#pragma unroll
      for (int t = 0; t < 4; ++t) cb[t] = b4[t];
      f32x16 (&acc)[WM] = accs[(KS == 1) ? gq / 2 : 0];
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma(ca[0][0], cb[0], acc[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (gq + 1 < GQ) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a4[i] = la[(2 * (gq0 + gq) + 2 + half) * BM + 32 * i];
        b4 = lb[(2 * (gq0 + gq) + 2 + half) * BN];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          if (t + i > 0) acc[i] = mfma(ca[i][t], cb[t], acc[i]);
    }
  }
  f32x16 (&acc)[WM] = accs[0];
  if (KS == 1) {
#pragma unroll
    for (int i = 0; i < WM; ++i) acc[i] += accs[NACC - 1][i];
  }
  if (KS == 2) {      // the K halves meet: the ks = 1 waves leave their tiles in the (now idle) ring, the ks = 0 waves add and finish
    __syncthreads();                                      // every wave is done reading the last chunk
    float* red = lds + wpos * (WM * 16 * 64);
    if (ks == 1) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(i * 16 + q) * 64 + lane] = acc[i][q];
    }
    __syncthreads();
    if (ks == 1) return;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] += red[(i * 16 + q) * 64 + lane];
  }
  // epilogue: registers 4q4 .. 4q4+3 of tile i = rows r0 + {0..3}, r0 = m0 + (wm WM + i) 32 + 8 q4 + 4 half: one output quad
  const int n = n0 + wn * 32 + l31;
  float dot = 0.f;
  // 64-row tiles (WM = 1, two row waves): the fc4 partial dot of a 64-row group is ONE chain over its two 32-row tiles in the wide tile's
  // wave -- here the second row wave CONTINUES the first one's chain, lane by lane (through LDS, one barrier: the K-split partner waves
  // have left), so that the group's partial is bit-identical whatever the tile shape
  constexpr bool DOT_CHAIN = EPI == EPI_BIAS_RELU_DOT && WM == 1 && WAVES_M == 2 && WAVES_N == 1;
  float* const dotx = lds + 2 * (WM * 16 * 64) + 1024;      // behind the K-split reduction's slots
  if (DOT_CHAIN && wm == 1) { __syncthreads(); dot = dotx[lane]; }
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
        for (int u = 0; u < 4; ++u) v[u] = (mk[u] > 0.f) ? (BTR == 2 ? v[u] * cscale : v[u]) : 0.f;
      }
      if (EPI == EPI_BIAS_RELU_DOT) {      // the only reader of this output is the BTR = 2 product: it needs relu'(a2), not a2
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (v[u] > 0.f) ? 1.f : 0.f;
      }
      // streamed: the next launch reads it from the Infinity Cache either way, and what is written through during the launch is not
      // left for the write-back at its end (measured: the four GEMMs 0.2596 -> 0.2553 ms)
      __builtin_nontemporal_store(v, &reinterpret_cast<f32x4*>(g.Out)[qi]);
    }
  if (DOT_CHAIN) {
    if (wm == 0) { dotx[lane] = dot; __syncthreads(); }      // hand the chain to the second row wave
    else {
      const float t = dot + __shfl_xor(dot, 32);
      if (half == 0) g.dot_out[(size_t)mt * g.ldo + n] = t;      // one partial per 64-row group, as the wide tile's (mt WAVES_M + wm)
    }
  } else if (EPI == EPI_BIAS_RELU_DOT) {      // rows of the two lane halves are disjoint: add them, one store per column
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
    // the 96-row tile on four waves, two per (row block, column half), splitting K inside the workgroup: two waves per SIMD instead of
    // one (round 5: the four launches 0.2685 -> 0.2646 ms); JRR_DISC_KS=1 (experiments): two waves, as in rounds 2-4
    static const bool ks2 = [] { const char* e = getenv("JRR_DISC_KS"); return !(e && atoi(e) == 1); }();
    dim3 grid((g.M / 96) * (g.N / 64)), block(ks2 ? 256 : 128);
    if (epi == EPI_STORE && !btr && ks2) hipLaunchKernelGGL((k_disc_gemm<3, 1, 2, EPI_STORE, 0, 2>), grid, block, 0, s, g);
    else if (epi == EPI_STORE && !btr) hipLaunchKernelGGL((k_disc_gemm<3, 1, 2, EPI_STORE, 0>), grid, block, 0, s, g);
    else { jrr_set_error("disc_gemm_q: 96-row tiles serve the plain store epilogue only"); return JRR_ERR_ARG; }
    return 0;
  }
  // (256 x 64 tiles on eight waves -- one workgroup per CU, two waves per SIMD behind ONE barrier -- measured equal: 0.2676-0.2689 vs
  // 0.2684-0.2685 ms for the four launches, round 5; not kept)
  // SMALL BATCHES (round 6): below 512 workgroups of 128 x 64 -- fewer than 4096 pose columns: the reference's default batch of 256, a
  // 512-pose shard of a strong-scaling run, BASELINE configs[1]'s 1024 -- most CUs would hold no workgroup at all (1024 poses: 128 of them
  // on 256 CUs).  Those launches take 128 x 32 tiles: twice the workgroups, the same four waves each (two K halves per row block: KS = 2),
  // and per output element the SAME sequence of operations as the wide tile -- a wave still owns 64 rows x 32 columns, K is still summed in
  // the two-set order, the fc4 partial dots keep their 64-row groups -- so engines of different batch sizes stay bit-identical.
  // JRR_DISC_NARROW=0 (experiments) keeps the wide tiles.
  static const bool narrow_ok = [] { const char* e = getenv("JRR_DISC_NARROW"); return !(e && e[0] == '0'); }();
  if (narrow_ok && (g.M / 128) * (g.N / 64) < 512) {
    // ... and below 2048 pose columns, where even those are fewer than 512 workgroups, 64 x 32 tiles: a wave owns ONE 32 x 32 tile and half
    // the K steps, so its own chain of matrix instructions -- what a launch of a few dozen workgroups lasts -- is half as long.  Per
    // output element still the same operations in the same order; the fc4 partial dots keep their 64-row groups as one chain (DOT_CHAIN
    // in the kernel).  JRR_DISC_NARROW=1 keeps 128 x 32 everywhere.
    static const bool tiny_ok = [] { const char* e = getenv("JRR_DISC_NARROW"); return !(e && e[0] == '1'); }();
    if (tiny_ok && (g.M / 128) * (g.N / 32) < 512 && g.M % 64 == 0) {
      dim3 gridt((g.M / 64) * (g.N / 32)), blockt(256);
      if (epi == EPI_BIAS_RELU_DOT && !btr) hipLaunchKernelGGL((k_disc_gemm<1, 2, 1, EPI_BIAS_RELU_DOT, 0, 2>), gridt, blockt, 0, s, g);
      else if (epi == EPI_BIAS_RELU && !btr) hipLaunchKernelGGL((k_disc_gemm<1, 2, 1, EPI_BIAS_RELU, 0, 2>), gridt, blockt, 0, s, g);
      else if (epi == EPI_STORE && !btr) hipLaunchKernelGGL((k_disc_gemm<1, 2, 1, EPI_STORE, 0, 2>), gridt, blockt, 0, s, g);
      else if (epi == EPI_MASK && btr == 2) hipLaunchKernelGGL((k_disc_gemm<1, 2, 1, EPI_MASK, 2, 2>), gridt, blockt, 0, s, g);
      else { jrr_set_error("disc_gemm_q: unsupported epilogue %d / transform %d", epi, btr); return JRR_ERR_ARG; }
      return 0;
    }
    dim3 gridn((g.M / 128) * (g.N / 32)), blockn(256);
    if (epi == EPI_BIAS_RELU && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 1, EPI_BIAS_RELU, 0, 2>), gridn, blockn, 0, s, g);
    else if (epi == EPI_BIAS_RELU_DOT && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 1, EPI_BIAS_RELU_DOT, 0, 2>), gridn, blockn, 0, s, g);
    else if (epi == EPI_STORE && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 1, EPI_STORE, 0, 2>), gridn, blockn, 0, s, g);
    else if (epi == EPI_MASK && btr == 2) hipLaunchKernelGGL((k_disc_gemm<2, 2, 1, EPI_MASK, 2, 2>), gridn, blockn, 0, s, g);
    else { jrr_set_error("disc_gemm_q: unsupported epilogue %d / transform %d", epi, btr); return JRR_ERR_ARG; }
    return 0;
  }
  dim3 grid((g.M / 128) * (g.N / 64)), block(256);
  // JRR_DISC_KS128=2 (experiment, round 6): the full-size 128 x 64 tile on EIGHT waves, two K halves per position like the 96-row tile
  // (same tile, same order of the sums: bit-identical) -- four waves per SIMD instead of two
  static const bool ks128 = [] { const char* e = getenv("JRR_DISC_KS128"); return e && e[0] == '2'; }();
  if (ks128) {
    dim3 block8(512);
    if (epi == EPI_BIAS_RELU && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_BIAS_RELU, 0, 2>), grid, block8, 0, s, g);
    else if (epi == EPI_BIAS_RELU_DOT && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_BIAS_RELU_DOT, 0, 2>), grid, block8, 0, s, g);
    else if (epi == EPI_STORE && !btr) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_STORE, 0, 2>), grid, block8, 0, s, g);
    else if (epi == EPI_MASK && btr == 2) hipLaunchKernelGGL((k_disc_gemm<2, 2, 2, EPI_MASK, 2, 2>), grid, block8, 0, s, g);
    else { jrr_set_error("disc_gemm_q: unsupported epilogue %d / transform %d", epi, btr); return JRR_ERR_ARG; }
    return 0;
  }
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
