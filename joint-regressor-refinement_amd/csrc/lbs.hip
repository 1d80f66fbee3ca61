// SMPL linear blend skinning + H36M joint regression on the fp32 matrix cores.
//
// Restates (SURVEY.md Appendix A steps 1,3,5 = smplx 0.1.26 lbs(); /root/reference call sites
// scripts/utils.py:94-98):
//   v_posed = v_template + shapedirs.beta + posedirs.(R-I)      -> one K=218 product  D . F
//   T_v     = sum_j W[v,j] A_j                                   -> K=24 products      W . A
//   verts   = T_v[:3,:3] v_posed + T_v[:3,3]
//   joints  = Jn @ verts                                         -> K=6890 product
//
// Every tile lives in the accumulator layout of v_mfma_f32_32x32x2_f32 with
//   rows = 32 vertices (registers), columns = 32 poses (lanes),
// so the three products chain without leaving registers: the blend product and the skinning
// product both produce (vertex, pose) tiles, the per-element affine combine is plain VALU on
// matching registers, and the regressor product consumes the vertex tile as its B operand
// (it sums over the tile's ROW index, guide section 3 "accumulator tile as the next operand").
#include <cstdlib>

#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

// ------------------------------------------------------------------------------------------
// forward
//   grid.x = (BP/128) * nvc workgroups of 256 threads; wave w owns poses [bg*128 + 32w, +32)
//   and loops over the vertex tiles of chunk vc.
//
//   EVERY MFMA operand is read from LDS.  Operands are staged by LDS-DMA (global_load_lds_dwordx4:
//   no VGPR round trip, asynchronous) into a 2-deep ring, one stage ahead of the MFMAs, one
//   workgroup barrier per stage.  Per vertex tile there are 13 stages:
//     s = 0..6   blend product, K chunk kc = s: basis rows D[32 k][3][32 v] (12 KB, shared by the
//                four waves) + feature rows F^T[32 k][128 poses] (16 KB)        48 MFMA / wave
//     s = 7..12  skinning, half-stage (r, h): two A^T blocks [24 j][128 poses] (24 KB):
//                h = 0: translation column c = 3 and c = 0;  h = 1: c = 1, 2, then the
//                regressor product  joints^T += Jn . verts_r                 24 (+16) MFMA / wave
//   The skinning-weight / regressor tiles (W^T, Jn) ride with stage 0 into a per-tile-parity
//   region.  outputs: JP [nvc][3][17][BP] joint partials; optional VPb [3][VP/4][BP][4] (v_posed in row quads,
//   kept for the backward pass) and VTb [3][VP/4][BP][4] (the vertices, same layout; k_verts_untranspose turns
//   them into the reference's (B,6890,3) and/or projects them for the silhouette renderer).
// ------------------------------------------------------------------------------------------
constexpr int W_FLOATS = NJ * 32;             // 768
constexpr int JN_FLOATS = 32 * 32;            // 1024
constexpr int STG_FLOATS = KCH * 96 + KCH * BG;   // 3072 (D chunk) + 4096 (F chunk) = 7168
constexpr int WJ_FLOATS = W_FLOATS + JN_FLOATS;   // 1792
constexpr int NSTAGE = NKCH + 6;                  // 13
// v_mfma_f32_32x32x2_f32 instructions per wave and vertex tile: 109 K-pairs x 3 planes + 12 x 24 / 2 + 3 x 16
constexpr int LBS_FWD_MFMA_PER_TILE = (KF / 2) * 3 + 6 * 24 + 3 * 16;
// per-tile operand record of the backward kernel: [Jn 18x32 | W^T 24x32 | W 32x32(j) | pad] = 10 KB
constexpr int TB_JN = 0, TB_WJV = NHP * 32, TB_WVJ = TB_WJV + W_FLOATS, TB_FLOATS = 2560;
// joint-sparse variants: the W block holds the segment-window weights W16 [16 n][36] (576 floats, jrr_common.h) instead of
// [32 v][32 j]; the record then ends at 1920 floats: 8 of the 10 DMA pieces
constexpr int TB_W16_FLOATS = 16 * 36, TB_PIECES_SPARSE = 8;
// record of k_lbs_bwd16 (below): JN | WT | WD | JL, 7 DMA pieces
constexpr int R16_JN = 0, R16_WT = 640, R16_WD = 1024, R16_JL = 1536, R16_WX = 1552, R16_FLOATS = 1792, R16_PIECES = 7;   // WX: K step 3 of WT
// v_mfma_f32_16x16x1_4b_f32: four independent 16x16 outer products per instruction (block = lane / 16), 8 passes = half the
// issue time of v_mfma_f32_32x32x2_f32 (33 vs 64 clocks measured, tools/probe/mfma16_probe.hip):
//   A: lane l holds A_blk[row l % 16],  B: lane l holds B_blk[col l % 16],  blk = l / 16
//   D: register 4 blk + i of lane l holds D_blk[row 4 (l / 16) + i][col l % 16]
__device__ __forceinline__ f32x16 mfma16(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0); }

// one wave-instruction: 64 lanes x 16 B from per-lane global addresses into LDS [dst, dst + 1 KB)
__device__ __forceinline__ void dma16(const float* gsrc_lane, float* lds_dst_wave) {
  __builtin_amdgcn_global_load_lds(JRR_GLB(gsrc_lane), JRR_LDS(lds_dst_wave), 16, 0, 0);
}
// same with the address split as (wave-uniform base, 32-bit per-lane offset in floats).  The base is
// laundered through an SGPR so the compiler keeps it scalar (saddr form) and cannot fold it into a
// per-copy VGPR offset or hoist it out of the tile loop (either costs ~2 VGPRs per copy).
__device__ __forceinline__ void dma16u(const float* base_uniform, unsigned lane_off, float* lds_dst_wave) {
  asm volatile("" : "+s"(base_uniform));
  __builtin_amdgcn_global_load_lds(JRR_GLB(base_uniform + lane_off), JRR_LDS(lds_dst_wave), 16, 0, 0);
}

// KJ > 0 (joint-sparse skinning, Model::kjs = 8 or 12): a tile's skinning product only runs over the tile's own <= KJ
// joints -- Wjv then is the compacted W^T [VT][KJ][32], jl the joint lists -- in THREE stages per tile (r = 0, 1, 2:
// the four A^T blocks (r, c) of the listed joints, 16 KB; T_{r,3}, T_{r,0}, T_{r,1}, T_{r,2} = 4 x KJ/2 MFMA; combine;
// regressor 16 MFMA) instead of six half-stages of 24 (+16) MFMA: with KJ = 8, 423 instead of 519 MFMA per tile and 48
// instead of 144 KB of A^T streamed per tile (KJ = 12: 447 MFMA, 72 KB).  Skipped terms are exact zeros: the result is the dense product's up to the
// grouping of the K pairs.
// WIDE: the model has tiles with more than KJ joints (per-tile classes: such a tile runs a second pass over slots KJ .. 2 KJ - 1).
// A separate instantiation: the branches of the second pass cost the stage loop ~2 % even when never taken (measured: 0.361 vs
// 0.369 ms at B = 4096), so a model without wide tiles runs the kernel without them.
template <bool STORE_VP, bool STORE_VERTS, int KJ, bool WIDE, bool LIST = false>
__global__ __launch_bounds__(256, 2) void k_lbs_fwd(const float* __restrict__ Dk, const float* __restrict__ Wjv,
                                                    const float* __restrict__ Jn_vi, const float* __restrict__ FT,
                                                    const float* __restrict__ AT, float* __restrict__ VPb,
                                                    float* __restrict__ JP, float* __restrict__ VTb, int B, int BP,
                                                    int nvc, long long* __restrict__ probe, int paired,
                                                    const int* __restrict__ jl, const int* __restrict__ tnj,
                                                    const int* __restrict__ vmask, const int* __restrict__ tl, int ntl, int vpm) {
  // vpm (STORE_VERTS only): VTb is a POSE-MAJOR buffer [BP][3][VP] -- what the fused rasteriser reads (round 5).  A pose's 32 rows of
  // a tile sit in two lanes (l and l + 32: rows 8 g + 4 half + 0..3), whose four 16-byte stores per plane complete one 128-byte
  // line of that pose; the rasteriser then loads a pose's 83 KB contiguously instead of 5184 pieces out of 5184 cache lines.
  // LIST: the launch covers the ntl tiles tl[0 .. ntl) only (ascending) -- the tiles that hold an entry of the regressor's support.
  // Every other tile multiplies its vertices by a zero block of the regressor: it adds exact zeros to the joints, and nothing else
  // of it is read by the joint-loss iteration (the backward kernels skip the same tiles).
  const int NT = LIST ? ntl : VT;
  // vmask (STORE_VERTS only, nullable): vertices leave the chip only for the tiles with vmask[tile] != 0 -- the J step over the
  // regressor's support reads a few dozen vertex rows, not 340 MB
  constexpr bool SPARSE = KJ > 0;
  constexpr int KJS = SPARSE ? KJ : 8;                       // (8: keeps the dead sparse branch of the dense variant well-formed)
  constexpr int NST = SPARSE ? NKCH + 3 : NSTAGE;            // stages per vertex tile
  constexpr int WT_FLOATS = SPARSE ? KJS * 32 : W_FLOATS;    // W^T rows staged per tile (a WIDE tile: 2 KJS rows)
  constexpr int WT_COPIES = (WT_FLOATS + 255) / 256, WT_COPIES_WIDE = (2 * WT_FLOATS + 255) / 256;
  __shared__ float lds[2 * STG_FLOATS + 2 * WJ_FLOATS];
  // shader-clock probe (profiling only), see jrr_engine_probe_read
  const long long probe_t0 = probe ? clock64() : 0, probe_w0 = probe ? wall_clock64() : 0;
  float* const ring = lds;
  float* const wj = lds + 2 * STG_FLOATS;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  int bg = L / nvc, vc = L % nvc;
  int t_begin = (int)((long)NT * vc / nvc), t_end = (int)((long)NT * (vc + 1) / nvc);
  if (paired) {
    // One full round of 512 workgroups = two per CU, and the two do NOT progress evenly: the first-dispatched one
    // wins the issue arbitration and would finish ~20 % earlier, leaving its partner alone (one wave per SIMD) for
    // the tail (measured with per-workgroup s_memrealtime stamps).  So the two workgroups of a CU
    // -- dispatch slots j and j + per_xcd/2 of an XCD -- share one PAIR of vertex chunks of the same pose group and
    // the first one takes the larger share of the pair's tiles (16 : 11 at B = 4096, launcher).  Static, hence still bitwise reproducible;
    // if the hardware paired differently the split would merely be uneven.
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int p = j % (per_xcd / 2), slot = j / (per_xcd / 2);
    const int npair = nvc / 2, bg_per_xcd = (per_xcd / 2) / npair;
    bg = xcd * bg_per_xcd + p / npair;
    const int vcp = p % npair;
    vc = 2 * vcp + slot;
    const int P0 = (int)((long)NT * vcp / npair), P1 = (int)((long)NT * (vcp + 1) / npair);
    const int mid = P0 + ((P1 - P0) * paired + 500) / 1000;      // `paired` = the first workgroup's share in thousandths
    t_begin = slot ? mid : P0;
    t_end = slot ? P1 : mid;
  }
  const int b0 = bg * BG + wave * BT;
  const size_t bcol = (size_t)b0 + l31;
  const unsigned qoff = (unsigned)half * (unsigned)BP + (unsigned)bcol;          // lane part of a (row quad, pose) address, in quads
  // DMA addressing = wave-uniform base pointer (SGPR pair) + one 32-bit per-lane offset (VGPR), so
  // the 78 copies per tile cost scalar address arithmetic only.  Row-pair pattern: lanes 0-31
  // fetch row 2o, lanes 32-63 row 2o+1, 16 B per lane.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane_rp = (unsigned)half * (unsigned)BP + (unsigned)l31 * 4u;   // floats
  const unsigned lane_ln = (unsigned)lane * 4u;                                  // floats
  const float* const Fbg = FT + (size_t)bg * BG * 4;            // FT in K-quads [KFP / 4][BP][4] (k_prep_fwd writes it beside the row-major copy)
  const float* const Abg = AT + (size_t)bg * BG;
  unsigned lane_jl[2] = {0, 0};   // SPARSE: (row of this lane's joint of the current tile) * BP + pose offset, floats
  unsigned lane_jx[2] = {0, 0};   // ... of the SECOND pass of a wide tile (slots KJS .. 2 KJS - 1)
  int wide = 0, wide_next = 0;    // this tile / the next tile has more than KJS joints (wave-uniform)
  long long n_extra = 0;          // second passes run (probe)

  // ---- DMA issue for stage s of tile vt into ring slot `slot` ----
  auto issue = [&](int vt, int s, int slot, int pass = 0, int wide_w = 0, int par = 0) {      // par: W / Jn buffer of the tile (s == 0)
    float* dst = ring + slot * STG_FLOATS;
    if (s < NKCH) {
      // K chunk s of the basis, K-QUADS [8 quads][3][32 v][4]: one contiguous 12 KB block of Dk
      const float* dsrc = Dk + ((size_t)vt * KFP + s * KCH) * 96;
#pragma unroll
      for (int i = 0; i < 3; ++i) { const int o = wv + 4 * i; dma16u(dsrc + o * 256, lane_ln, dst + o * 256); }
      // ... and of the features, K-quads [8 quads][128 poses][4]: piece o = (quad o / 2, poses (o % 2) * 64 + lane)
      float* fdst = dst + KCH * 96;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = wv + 4 * i;
        dma16u(Fbg + ((size_t)(s * (KCH / 4) + (o >> 1)) * BP + (o & 1) * 64) * 4, lane_ln, fdst + o * 256);
      }
      if (s == 0) {
        float* wdst = wj + par * WJ_FLOATS;
        // (the table holds NJ slot rows per tile; a wide tile -- `wide_w`, wave-uniform -- stages 2 KJS of them)
        if (wv < ((SPARSE && wide_w) ? WT_COPIES_WIDE : WT_COPIES)) dma16u(Wjv + (size_t)vt * W_FLOATS + wv * 256, lane_ln, wdst + wv * 256);
        dma16u(Jn_vi + (size_t)vt * JN_FLOATS + wv * 256, lane_ln, wdst + W_FLOATS + wv * 256);
      }
    } else if (SPARSE) {
      // four blocks [KJS joints][128 poses] of A^T, components (r, 3), (r, 0), (r, 1), (r, 2); this wave copies row pair
      // wv of each -- the rows of joints jl[2 wv], jl[2 wv + 1] (lane offset `lane_jl[0]`, set per tile) -- and, with 12
      // joints, waves 0 and 1 also row pair 4 + wv (`lane_jl[1]`)
      const int r = s - NKCH;
      const unsigned o0 = pass ? lane_jx[0] : lane_jl[0], o1 = pass ? lane_jx[1] : lane_jl[1];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const int c = (ci == 0) ? 3 : ci - 1;
        dma16u(Abg + (size_t)((r * 4 + c) * NJ) * BP, o0, dst + ci * (KJS * BG) + wv * 256);
        if (KJS > 8 && wv < KJS / 2 - 4) dma16u(Abg + (size_t)((r * 4 + c) * NJ) * BP, o1, dst + ci * (KJS * BG) + (4 + wv) * 256);
      }
    } else {
      const int h = s - NKCH, r = h >> 1;
      const int c0 = (h & 1) ? 1 : 3, c1 = (h & 1) ? 2 : 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int o = wv + 4 * i;   // 12 row pairs per block
        dma16u(Abg + (size_t)((r * 4 + c0) * NJ + 2 * o) * BP, lane_rp, dst + o * 256);
        dma16u(Abg + (size_t)((r * 4 + c1) * NJ + 2 * o) * BP, lane_rp, dst + NJ * BG + o * 256);
      }
    }
  };

  // H36M joints 0..15 as block accumulators of v_mfma_f32_16x16x1_4b_f32 (17 joint rows in a 32-row product waste 47 % of
  // it; 16 rows on the four-block instruction take half the issue time), joint 16 as a per-lane partial on the vector ALU
  f32x16 jacc[3] = {zero16(), zero16(), zero16()};
  float j16[3] = {0.f, 0.f, 0.f};
  f32x16 vp[3];
  f32x16 vr;

  // joints^T[i, b] += sum_v Jn[i, v] verts_r[v, b] for the tile in `vr`: A operand Jn[i = lane % 16][v = acc_row(q, half)]
  // (the [v][32 i] tile of the record, read with i = lane % 16), B operand register q of the vertex tile; one four-block
  // instruction adds the two rows acc_row(q, 0), acc_row(q, 1) for all 32 pose columns (blocks 0 + 2: columns 0..15,
  // 1 + 3: 16..31).  Operands four steps ahead (an instruction is 32 clocks).
  // (the first four A operands are requested by the caller -- `ra`, before anything that would fence the schedule: in the WIDE
  // instantiation the second-pass branch sits between the skinning products and this product, and operands requested behind it
  // exposed ~200 clocks of LDS latency per stage)
  auto regress_operands = [&](const float* ldsJ, float (&ra)[4]) __attribute__((always_inline)) {
    const float* jp = ldsJ + half * 128 + (lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = jp[acc_row_u(q) * 32];
  };
  auto regress = [&](auto R_, const float* ldsJ, const float (&ra)[4]) __attribute__((always_inline)) {
    constexpr int r = decltype(R_)::value;
    const float* jp = ldsJ + half * 128 + (lane & 15);              // row acc_row(q, half) = acc_row_u(q) + 4 half
    float a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3];
    // (a second accumulator chain -- even / odd rows, added at the end -- fits since the quad-operand blend stage and was measured
    // SLOWER: 0.3641 vs 0.3614 ms; the single chain's dependent issue is covered by the SIMD's other wave)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float ac = a0;
      a0 = a1; a1 = a2; a2 = a3;
      __builtin_amdgcn_sched_barrier(0);
      jacc[r] = mfma16(ac, vr[q], jacc[r]);
      __builtin_amdgcn_sched_barrier(0);
      if (q + 4 < 16) a3 = jp[acc_row_u(q + 4) * 32];
    }
    // joint 16 on the vector ALU: Jn[16][v] verts[v][b] over this lane's rows (the two lane halves are summed at the end)
    const float* sp = ldsJ + half * 128 + 16;
    float p16 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) p16 = fmaf(sp[acc_row_u(q) * 32], vr[q], p16);
    j16[r] += p16;
  };

  // pose-major vertex store (vpm): wave-uniform row base + a 32-bit lane offset formed AT the store from a laundered lane index
  // (hoisted out of the tile loop it costs the WIDE instantiations registers they do not have)
  auto pm_store = [&](int r, int vt, int g4, const f32x4& t) __attribute__((always_inline)) {
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned lo = ((unsigned)b0 + (unsigned)(ln & 31)) * (3u * VP) + 4u * (unsigned)(ln >> 5);
    float* base = VTb + ((size_t)r * VP + vt * 32 + 8 * g4);
    asm volatile("" : "+s"(base));
    *reinterpret_cast<f32x4*>(base + lo) = t;
  };
  const int t_first = (t_begin < t_end) ? (LIST ? tl[t_begin] : t_begin) : 0;
  if (SPARSE && WIDE && t_begin < t_end) wide_next = tnj[t_first] > KJS;
  if (t_begin < t_end) issue(t_first, 0, 0, 0, wide_next, t_begin & 1);
  int g = 0;   // global stage counter: ring slot = g & 1
  for (int ix = t_begin; ix < t_end; ++ix) {
    const int vt = LIST ? tl[ix] : ix;                                       // the tile (wave-uniform)
    const int vtn = LIST ? ((ix + 1 < t_end) ? tl[ix + 1] : 0) : ix + 1;      // the next one (used under ix + 1 < t_end only)
    const float* ldsW = wj + (ix & 1) * WJ_FLOATS;
    const float* ldsJ = ldsW + W_FLOATS;
    const bool stv = STORE_VERTS && (!vmask || vmask[vt] != 0);      // wave-uniform
    if (SPARSE) {      // the skinning copies of this tile are issued from its stage NKCH - 1 on
      if (WIDE) {
        wide = wide_next;
        wide_next = (ix + 1 < t_end) ? (tnj[vtn] > KJS) : 0;                    // wave-uniform: scalar loads
      }
      const int j0 = jl[vt * NJ + 2 * wv], j1 = jl[vt * NJ + 2 * wv + 1];
      lane_jl[0] = (unsigned)(half ? j1 : j0) * (unsigned)BP + (unsigned)l31 * 4u;
      if (KJS > 8 && wv < KJS / 2 - 4) {
        const int j2 = jl[vt * NJ + 8 + 2 * wv], j3 = jl[vt * NJ + 8 + 2 * wv + 1];
        lane_jl[1] = (unsigned)(half ? j3 : j2) * (unsigned)BP + (unsigned)l31 * 4u;
      }
      if (WIDE && wide) {      // second pass: slots KJS .. 2 KJS - 1
        const int k0 = jl[vt * NJ + KJS + 2 * wv], k1 = jl[vt * NJ + KJS + 2 * wv + 1];
        lane_jx[0] = (unsigned)(half ? k1 : k0) * (unsigned)BP + (unsigned)l31 * 4u;
        if (KJS > 8 && wv < KJS / 2 - 4) {
          const int k2 = jl[vt * NJ + KJS + 8 + 2 * wv], k3 = jl[vt * NJ + KJS + 8 + 2 * wv + 1];
          lane_jx[1] = (unsigned)(half ? k3 : k2) * (unsigned)BP + (unsigned)l31 * 4u;
        }
        ++n_extra;
      }
    }
    static_for<0, NST>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      // Stage g landed; slot (g+1)&1 is free again.  The copies of stage g were issued at the start of the previous
      // stage, BEFORE that stage's v_posed / vertex stores: exactly those stores are left in flight (they get one more
      // stage to retire instead of stalling this barrier; barrier_keep_vm).  First stage of the kernel: nothing younger.
      {
        constexpr int sp = (s == 0) ? NST - 1 : s - 1;                    // previous stage
        constexpr int hp = sp - NKCH;                                      // its skinning (half-)stage, if any
        constexpr int nst = (hp < 0) ? 0
                            : SPARSE ? (STORE_VP ? 4 : 0) + (STORE_VERTS ? 4 : 0)
                                     : (STORE_VP ? 2 : 0) + ((STORE_VERTS && (hp & 1)) ? 4 : 0);
        // (with a tile mask the vertex stores of the previous stage may not exist: count the v_posed stores only -- waiting for
        // more than necessary is always safe)
        constexpr int nst_vp = (hp < 0) ? 0 : SPARSE ? (STORE_VP ? 4 : 0) : (STORE_VP ? 2 : 0);
        if (s == 0 && ix == t_begin) barrier_keep_vm<0>();
        else if (STORE_VERTS && nst != nst_vp && vmask) barrier_keep_vm<nst_vp>();
        else barrier_keep_vm<nst>();
      }
      // the stage that follows this one: a wide tile's skinning stage r is followed by its second pass over the same r
      if (SPARSE && WIDE && s >= NKCH && wide) issue(vt, s, (g + 1) & 1, 1);
      else if (s + 1 < NST) issue(vt, s + 1, (g + 1) & 1);
      else if (ix + 1 < t_end) issue(vtn, 0, (g + 1) & 1, 0, wide_next, (ix + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);   // the counted wait above relies on: copies first, this stage's stores after
      const float* buf = ring + (g & 1) * STG_FLOATS;
      if constexpr (s < NKCH) {
        if (s == 0) { vp[0] = zero16(); vp[1] = zero16(); vp[2] = zero16(); }
        // Both operands arrive in K-QUADS: a lane's ds_read_b128 is FOUR K steps (a K pair may take any two k as long as both
        // operands agree: step t of group g takes k = 8 g + t and 8 g + 4 + t, the lane half picks the quad 2 g + half) --
        // 4 LDS reads per 12 matrix instructions instead of 16.  The last chunk holds k = 192 .. 217: its group 3 has only the
        // steps t = 0, 1 (k = 216, 217 against zero rows).  Operands of group g + 1 are requested right after the first
        // instruction of group g.
        constexpr int ngroups = 4;
        const f32x4* dq = reinterpret_cast<const f32x4*>(buf) + half * 96 + l31;                          // quad q, plane c: + (q * 3 + c) * 32
        const f32x4* fq = reinterpret_cast<const f32x4*>(buf + KCH * 96) + half * BG + wave * BT + l31;   // quad q: + q * 128
        f32x4 a0 = dq[0], a1 = dq[32], a2 = dq[64], bq = fq[0];
        f32x4 n0 = a0, n1 = a1, n2 = a2, nb = bq;
#pragma unroll
        for (int gq = 0; gq < ngroups; ++gq) {
          constexpr int last_steps = KF - (NKCH - 1) * KCH - 24;      // valid steps of group 3 of the last chunk: 218 - 192 - 24 = 2
          const int nsteps = (s == NKCH - 1 && gq == ngroups - 1) ? last_steps : 4;
          const f32x4 c0 = a0, c1 = a1, c2 = a2, cb = bq;
          __builtin_amdgcn_sched_barrier(0);
          vp[0] = mfma(c0[0], cb[0], vp[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (gq + 1 < ngroups) {
            n0 = dq[(2 * gq + 2) * 96]; n1 = dq[(2 * gq + 2) * 96 + 32]; n2 = dq[(2 * gq + 2) * 96 + 64];
            nb = fq[(2 * gq + 2) * BG];
          }
          __builtin_amdgcn_sched_barrier(0);
          vp[1] = mfma(c1[0], cb[0], vp[1]);
          vp[2] = mfma(c2[0], cb[0], vp[2]);
#pragma unroll
          for (int t = 1; t < 4; ++t) {
            if (t >= nsteps) break;
            vp[0] = mfma(c0[t], cb[t], vp[0]);
            vp[1] = mfma(c1[t], cb[t], vp[1]);
            vp[2] = mfma(c2[t], cb[t], vp[2]);
          }
          a0 = n0; a1 = n1; a2 = n2; bq = nb;
        }
      } else if constexpr (SPARSE) {
        constexpr int r = s - NKCH;
        if (STORE_VP) {   // v_posed plane r leaves with skinning stage r: four 16-byte row quads
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 t = {vp[r][4 * g4], vp[r][4 * g4 + 1], vp[r][4 * g4 + 2], vp[r][4 * g4 + 3]};
            *quad_ptr(VPb, (size_t)r * (VP / 4) + vt * 8, g4, BP, qoff) = t;
          }
        }
        const float* wp = ldsW + half * 32 + l31;                       // W^T rows of the listed joints: row 2 p + half
        const float* ab = buf + half * BG + wave * BT + l31;             // block ci at + ci * KJS * BG, row pair p at + 2 p BG
        float w[KJS / 2], x0[KJS / 2], x1[KJS / 2];
#pragma unroll
        for (int pq = 0; pq < KJS / 2; ++pq) {
          w[pq] = wp[2 * pq * 32];
          x0[pq] = (WIDE && KJS > 8) ? 0.f : ab[2 * pq * BG];
          x1[pq] = (WIDE && KJS > 8) ? 0.f : ab[KJS * BG + 2 * pq * BG];
        }
        f32x16 T = zero16(), U = zero16();
        // TIGHT (12 slots AND wide tiles: a body whose best order still has a 13-joint tile): 30 operand registers per block pair
        // and the regressor operands held across the second-pass branch do not fit 256 registers (7 .. 12 spilled) -- that one
        // instantiation reads the second block pair's operands where they are used and requests the regressor operands behind the
        // branch (exposed LDS latency instead of scratch traffic; no other instantiation changes)
        constexpr bool TIGHT = WIDE && KJS > 8;
        float y0[KJS / 2], y1[KJS / 2];
#pragma unroll
        for (int pq = 0; pq < KJS / 2; ++pq) {                            // T_{r,3}, T_{r,0}; the next two blocks' operands meanwhile
          if (TIGHT) { x0[pq] = ab[2 * pq * BG]; x1[pq] = ab[KJS * BG + 2 * pq * BG]; }
          T = mfma(w[pq], x0[pq], T);
          if (!TIGHT) { y0[pq] = ab[2 * KJS * BG + 2 * pq * BG]; y1[pq] = ab[3 * KJS * BG + 2 * pq * BG]; }
          U = mfma(w[pq], x1[pq], U);
        }
        vr = T + U * vp[0];                      // T_{r,3} + T_{r,0} v_x
        T = zero16(); U = zero16();
#pragma unroll
        for (int pq = 0; pq < KJS / 2; ++pq) {                            // T_{r,1}, T_{r,2}
          if (TIGHT) { y0[pq] = ab[2 * KJS * BG + 2 * pq * BG]; y1[pq] = ab[3 * KJS * BG + 2 * pq * BG]; }
          T = mfma(w[pq], y0[pq], T);
          U = mfma(w[pq], y1[pq], U);
        }
        vr += T * vp[1];                         // T_{r,1} v_y
        vr += U * vp[2];                         // T_{r,2} v_z
        float ra[4];
        if (!TIGHT) regress_operands(ldsJ, ra);
        if (WIDE && wide) {
          // ---- second pass of a WIDE tile (more than KJS joints): the same four products over slots KJS .. 2 KJS - 1,
          //      added to vr.  Its own ring stage: every copy and store of this wave is drained at its barrier (rare path).
          ++g;
          barrier_keep_vm<0>();
          if (s + 1 < NST) issue(vt, s + 1, (g + 1) & 1);
          else if (ix + 1 < t_end) issue(vtn, 0, (g + 1) & 1, 0, wide_next, (ix + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          const float* bx = ring + (g & 1) * STG_FLOATS + half * BG + wave * BT + l31;
          const float* wx = ldsW + (KJS + half) * 32 + l31;
          // (two accumulators at a time, like the first pass: the register budget of the hot path is the kernel's)
          T = zero16(); U = zero16();
#pragma unroll
          for (int pq = 0; pq < KJS / 2; ++pq) {
            const float wq = wx[2 * pq * 32];
            T = mfma(wq, bx[2 * pq * BG], T);
            U = mfma(wq, bx[KJS * BG + 2 * pq * BG], U);
          }
          vr += T;
          vr += U * vp[0];
          T = zero16(); U = zero16();
#pragma unroll
          for (int pq = 0; pq < KJS / 2; ++pq) {
            const float wq = wx[2 * pq * 32];
            T = mfma(wq, bx[2 * KJS * BG + 2 * pq * BG], T);
            U = mfma(wq, bx[3 * KJS * BG + 2 * pq * BG], U);
          }
          vr += T * vp[1];
          vr += U * vp[2];
        }
        if (STORE_VERTS && stv) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 t = {vr[4 * g4], vr[4 * g4 + 1], vr[4 * g4 + 2], vr[4 * g4 + 3]};
            if (vpm) pm_store(r, vt, g4, t);
            else *quad_ptr(VTb, (size_t)r * (VP / 4) + vt * 8, g4, BP, qoff) = t;
          }
        }
        if (TIGHT) regress_operands(ldsJ, ra);
        regress(std::integral_constant<int, r>{}, ldsJ, ra);
        // after a second pass the next stage's counted wait (which assumes this stage's stores are all younger than its
        // copies) no longer holds: drain
        if (WIDE && wide) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        constexpr int h = s - NKCH, r = h >> 1;
        if (STORE_VP) {   // spread the v_posed stores over the six skinning stages: two 16-byte row quads each
          constexpr int c = h >> 1, g0 = (h & 1) * 2;
#pragma unroll
          for (int g = g0; g < g0 + 2; ++g) {
            const f32x4 t = {vp[c][4 * g], vp[c][4 * g + 1], vp[c][4 * g + 2], vp[c][4 * g + 3]};
            *quad_ptr(VPb, (size_t)c * (VP / 4) + vt * 8, g, BP, qoff) = t;
          }
        }
        const float* wp = ldsW + half * 32 + l31;
        const float* a0 = buf + half * BG + wave * BT + l31;
        const float* a1 = a0 + NJ * BG;
        // two accumulator chains per half-stage, interleaved (a back-to-back MFMA on the SAME accumulator issues
        // every 112 clocks instead of 64: tools/probe/clock_probe.hip), operands of joint pair jp+1 requested
        // right after the first MFMA of pair jp
        f32x16 T = zero16();
        f32x16 U = zero16();
        {   // operands two joint pairs ahead (a pair is only two MFMAs = 128 clocks)
          float w0 = wp[0], x00 = a0[0], x01 = a1[0];
          float w1 = wp[2 * 32], x10 = a0[2 * BG], x11 = a1[2 * BG];
#pragma unroll
          for (int jp = 0; jp < 12; ++jp) {
            const float wc = w0, c0 = x00, c1 = x01;
            w0 = w1; x00 = x10; x01 = x11;
            __builtin_amdgcn_sched_barrier(0);
            T = mfma(wc, c0, T);
            __builtin_amdgcn_sched_barrier(0);
            if (jp + 2 < 12) { w1 = wp[(2 * jp + 4) * 32]; x10 = a0[(2 * jp + 4) * BG]; x11 = a1[(2 * jp + 4) * BG]; }
            __builtin_amdgcn_sched_barrier(0);
            U = mfma(wc, c1, U);
          }
        }
        if ((h & 1) == 0) {
          vr = T + U * vp[0];                    // T_{r,3} + T_{r,0} v_x
        } else {
          vr += T * vp[1];                       // T_{r,1} v_y
          vr += U * vp[2];                       // T_{r,2} v_z
          if (STORE_VERTS && stv) {      // vertices, in the row-quad layout of v_posed: four 16-byte stores per lane
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 t = {vr[4 * g], vr[4 * g + 1], vr[4 * g + 2], vr[4 * g + 3]};
              if (vpm) pm_store(r, vt, g, t);
              else *quad_ptr(VTb, (size_t)r * (VP / 4) + vt * 8, g, BP, qoff) = t;
            }
          }
          float ra[4];
          regress_operands(ldsJ, ra);
          regress(std::integral_constant<int, r>{}, ldsJ, ra);
        }
      }
      ++g;
    });
  }
  // (the output addresses are formed HERE, from the hardware lane count and the scalar wave index: hoisted above the tile loop --
  // or from a thread index kept alive across it -- they cost the 12-slot WIDE instantiations spilled registers)
  const int le = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // the lane index, from no live register
  const size_t b0e = (size_t)(bg * BG + wv * BT);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int u = 0; u < 4; ++u) {        // block layout: register 4 blk + u = joint 4 (lane / 16) + u, pose column lane % 16 (+ 16)
      const int i = 4 * (le >> 4) + u;
      float* dst = JP + ((size_t)(vc * 3 + r) * NH + i) * BP + b0e + (le & 15);
      dst[0] = jacc[r][u] + jacc[r][8 + u];
      dst[16] = jacc[r][4 + u] + jacc[r][12 + u];
    }
#pragma unroll
  for (int r = 0; r < 3; ++r) {          // joint 16: the two lane halves hold the partial sums over their rows
    const float t = j16[r] + __shfl_xor(j16[r], 32);
    if (le < 32) JP[((size_t)(vc * 3 + r) * NH + 16) * BP + b0e + (le & 31)] = t;
  }
  if (probe && blockIdx.x == 0 && wv == 0 && le == 0) {
    probe[0] = clock64() - probe_t0;                 // shader clocks this wave was resident
    probe[1] = (long long)(t_end - t_begin) * (SPARSE ? (KF / 2) * 3 + 3 * 2 * KJS + 3 * 16 : LBS_FWD_MFMA_PER_TILE) + n_extra * 3 * 2 * KJS;   // MFMA instructions it issued
    probe[2] = wall_clock64() - probe_w0;            // the same interval on the constant 100 MHz counter
  }
}

// ------------------------------------------------------------------------------------------
// backward (to v_posed and to the skinning transforms)
//   one workgroup of four waves per (pose tile bt of 32 poses, vertex chunk vc); per vertex tile:
//     dverts_r[v,b] = sum_i Jn[i,v] dj[b,i,r]                (K = 18)        -- and / or loaded (DV, row quads)
//     T_{r,c}[v,b]  = sum_j W[v,j] A[b,j,r,c]                (K = 24, recomputed)
//     dvp_c[v,b]    = sum_r T_{r,c} dverts_r                  -> DVP [3][VP/4][BP][4] (row quads)
//     dA_{r,c}[j,b] += sum_v W[v,j] dverts_r[v,b] vp_c[v,b]   (sums over the tile's ROW index)
//     dA_{c,3}[j,b] += sum_v W[v,j] dverts_c[v,b]
//   Wave roles.  Waves 0..2 are the PLANE waves c = 0, 1, 2: T_{r,c} for the three r (36 MFMA), dvp_c, dA_{r,c} for the
//   three r (48 MFMA) -- they need the vertex adjoint dverts_r of all three r.  Wave 3 is the VERTEX-ADJOINT wave: it
//   computes dverts_r once per tile for the whole workgroup (27 MFMA; each plane wave used to recompute all of it),
//   hands the three tiles to the plane waves through LDS (48 floats per lane, 16-byte pieces, conflict-free), and
//   takes the translation column dA_{c,3} of all three c off them (48 MFMA; it only needs dverts).  84 : 84 : 84 : 75
//   MFMA per tile instead of 4 x 127 per 4/3 tiles.  The vertex-adjoint wave runs ONE TILE AHEAD of the plane waves:
//   while they work on tile t it computes dverts (and dA_{.,3}) of tile t + 1, so the hand-off costs two short barriers
//   per tile (tile published / tile consumed) and no waiting on MFMA work.  The per-tile operand records `Tb` = [Jn 18x32 | W^T 24x32 | W 32x32(j) | pad] (10 KB) ride a 3-deep
//   LDS-DMA ring (tile t for the plane waves, t + 1 for wave 3, t + 2 in flight); each plane wave keeps its slice of
//   A^T (36 operand vectors) in LDS for the whole kernel and register-prefetches its v_posed tile one tile ahead.
//   outputs: DVP, dATp [nvc][12][24][BP] partials.
// ------------------------------------------------------------------------------------------
// DV: 0 = vertex adjoint from the joint adjoint dJT (Jn^T dj), 1 = loaded from dVT, 2 = both (sum)
constexpr int BWD_RING = 3;
constexpr int DVBUF_FLOATS = 3 * 16 * 64;        // three 32x32 tiles as [r][register quad][lane][4]
// KJ > 0 (Model::kjs): T_{r,c} is recomputed over the tile's own <= KJ joints only (the record's W^T block then
// holds the KJ compacted rows, written by k_bwd_tab_static): with KJ = 8, 12 instead of 36 MFMA per plane wave and tile.
template <int DV, int KJ>
__global__ __launch_bounds__(256, 2) void k_lbs_bwd(const float* __restrict__ Tb, const float* __restrict__ AT,
                                                    const float* __restrict__ VPb, const float* __restrict__ dJT,
                                                    const float* __restrict__ dVT, float* __restrict__ DVP,
                                                    float* __restrict__ dATp, int BP, int nvc, int n_bt,
                                                    const int* __restrict__ jl, const int* __restrict__ segid,
                                                    const int* __restrict__ segj) {
  constexpr bool SPARSE = KJ > 0;
  constexpr int KJS = SPARSE ? KJ : 8;
  __shared__ __attribute__((aligned(16))) float lds[BWD_RING * TB_FLOATS + DVBUF_FLOATS + 3 * 36 * 64 + 27 * 64];
  float* const ring = lds;
  f32x4* const dvbuf = reinterpret_cast<f32x4*>(lds + BWD_RING * TB_FLOATS);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool plane = wv < 3;
  const int c = wv;                                   // coordinate plane of a plane wave
  float* const ldsA = lds + BWD_RING * TB_FLOATS + DVBUF_FLOATS + (plane ? wv : 0) * 36 * 64;
  float* const ldsDj = lds + BWD_RING * TB_FLOATS + DVBUF_FLOATS + 3 * 36 * 64;   // wave 3: dj operands [r * 9 + ip][lane]
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int vc = L / n_bt, bt = L % n_bt;
  // (a 5 : 4 split of chunk pairs between the two workgroups of a CU, which gains 6 % in k_lbs_fwd, measured equal here)
  const int t_begin = (int)((long)VT * vc / nvc), t_end = (int)((long)VT * (vc + 1) / nvc);
  const int b0 = bt * BT;
  const size_t bcol = (size_t)b0 + l31;
  const unsigned lane_ln = (unsigned)lane * 4u;

  auto issue = [&](int vt, int slot) {
    const float* src = Tb + (size_t)vt * TB_FLOATS;
    float* dst = ring + slot * TB_FLOATS;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int o = wv + 4 * i;
      if (o < (SPARSE ? TB_PIECES_SPARSE : TB_FLOATS / 256)) dma16u(src + o * 256, lane_ln, dst + o * 256);
    }
  };
  // The two roles are two separate code paths (each with its own accumulators: the register allocation is the larger
  // of the two, not their union); both execute the same barrier sequence: one __syncthreads, then two per tile.
  issue(t_begin, 0);
  if (t_begin + 1 < t_end) issue(t_begin + 1, 1);

  if (plane) {
    // ================= plane wave c: T_{r,c}, dvp_c, dA_{r,c} (joint-sparse: + dA_{c,3}) =================
    f32x16 acc3[3] = {zero16(), zero16(), zero16()};
    // joint-sparse: the four products (0,c), (1,c), (2,c), (c,3) over the segment's 16-row joint window, as block
    // accumulators of v_mfma_f32_16x16x1_4b_f32 (blocks 0 + 2 = pose columns 0..15, blocks 1 + 3 = 16..31)
    f32x16 acc4 = zero16();
    int cur_seg = SPARSE ? segid[t_begin] : 0;
    int seg_tile = t_begin;                  // a tile of the open segment (its window is segj[seg_tile])
    unsigned seen = 0u;                      // joints whose slab rows this wave has written
    // close a segment: add its accumulators into this workgroup's dA slab (first touch of a joint stores, later ones add;
    // the slab rows belong to this wave alone) and start the next one
    auto flush_window = [&]() {
      if constexpr (SPARSE) {
        if (seen != 0u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // earlier slab stores of this wave have landed (first flush: nothing to re-read)
        const int* sj = segj + seg_tile * 16;
        const int col = b0 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int j = sj[4 * (lane >> 4) + i];
          if (j >= 0) {
            const bool add = (seen >> j) & 1u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const f32x16& a = p == 0 ? acc3[0] : p == 1 ? acc3[1] : p == 2 ? acc3[2] : acc4;
              const int e = p < 3 ? p * 4 + c : c * 4 + 3;
              float* dst = dATp + ((size_t)(vc * 12 + e) * NJ + j) * BP + col;
              float lo = a[i] + a[8 + i], hi = a[4 + i] + a[12 + i];
              if (add) {      // agent-scope loads (L1 bypassed): what an earlier flush of this wave stored, whichever lane stored it
                lo += __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi += __hip_atomic_load(dst + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
              // agent-scope stores: a later flush of this wave re-reads them (possibly from another lane) with agent-scope loads
              __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(dst + 16, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          const int j = sj[n];                                       // wave-uniform
          if (j >= 0) seen |= 1u << j;
        }
        acc3[0] = zero16(); acc3[1] = zero16(); acc3[2] = zero16(); acc4 = zero16();
      }
    };
    // this wave's A^T operand vectors (r, j-pair) -> LDS, once
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int jp = 0; jp < 12; ++jp)
        ldsA[(r * 12 + jp) * 64 + lane] = AT[(size_t)((r * 4 + c) * NJ + 2 * jp + half) * BP + bcol];
    const unsigned qoff = (unsigned)half * (unsigned)BP + (unsigned)bcol;     // lane part of a (row quad, pose) address
    auto load_vp = [&](int vt, f32x16& dstv) {      // four 16-byte row quads (jrr_common.h)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 t = *quad_ptr(VPb, (size_t)c * (VP / 4) + vt * 8, g, BP, qoff);
        dstv[4 * g] = t[0]; dstv[4 * g + 1] = t[1]; dstv[4 * g + 2] = t[2]; dstv[4 * g + 3] = t[3];
      }
    };
    f32x16 vpA = zero16(), vpB = zero16();            // v_posed tiles, ping-pong: the next tile's loads fly for a whole tile
    load_vp(t_begin, vpA);
    __syncthreads();                                  // records of the first two tiles landed
    int slot = 0;                                     // ring slot of tile vt
    auto tile = [&](int vt, const f32x16& vpc, f32x16& vpn) {
      // dverts of tile vt published; record vt + 1 landed; everybody is done with tile vt - 1.  This wave's record
      // copies are older than its 4 v_posed prefetch loads and its 4 dvp stores of the previous tile: both stay in
      // flight across the barrier (the prefetch is consumed in the second half of THIS tile, the stores never).
      barrier_keep_vm<8>();
      const int slot1 = (slot + 1 == BWD_RING) ? 0 : slot + 1, slot2 = (slot1 + 1 == BWD_RING) ? 0 : slot1 + 1;
      if (vt + 2 < t_end) issue(vt + 2, slot2);
      __builtin_amdgcn_sched_barrier(0);     // the counted wait relies on: record copies first, loads / stores after
      const float* tab = ring + slot * TB_FLOATS;
      f32x16 dv[3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 t = dvbuf[(r * 4 + g) * 64 + lane];
          dv[r][4 * g] = t[0]; dv[r][4 * g + 1] = t[1]; dv[r][4 * g + 2] = t[2]; dv[r][4 * g + 3] = t[3];
        }
      if (vt + 1 < t_end) load_vp(vt + 1, vpn);
      // every plane wave holds dverts of tile vt in registers: the buffer may take tile vt + 1 (short barrier: the
      // plane waves reach it right after their 12 LDS reads, wave 3 at once)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // T_{r,c} for the three r as three interleaved accumulator chains sharing the W operand; operands of joint pair
      // jp + 1 are requested right after the first MFMA of pair jp (a wave never parks on lgkmcnt between products)
      f32x16 T0 = zero16(), T1 = zero16(), T2 = zero16();
      if constexpr (SPARSE) {
        // operand of joint j for this lane half: ldsA[(r * 12 + j / 2) * 64 + (j % 2) * 32 + l31] = ldsA[r * 768 + 32 j + l31]
        const float* wp = tab + TB_WJV + half * 32 + l31;
        float w[KJS / 2], a0[KJS / 2], a1[KJS / 2], a2[KJS / 2];
#pragma unroll
        for (int pq = 0; pq < KJS / 2; ++pq) {
          const int j0 = jl[vt * NJ + 2 * pq], j1 = jl[vt * NJ + 2 * pq + 1];            // wave-uniform: scalar loads
          const float* ap = ldsA + (half ? j1 : j0) * 32 + l31;
          w[pq] = wp[2 * pq * 32]; a0[pq] = ap[0]; a1[pq] = ap[12 * 64]; a2[pq] = ap[24 * 64];
        }
#pragma unroll
        for (int pq = 0; pq < KJS / 2; ++pq) {
          T0 = mfma(w[pq], a0[pq], T0);
          T1 = mfma(w[pq], a1[pq], T1);
          T2 = mfma(w[pq], a2[pq], T2);
        }
      } else {
        const float* wp = tab + TB_WJV + half * 32 + l31;
        const float* ap = ldsA + lane;
        float w = wp[0], a0 = ap[0], a1 = ap[12 * 64], a2 = ap[24 * 64];
#pragma unroll
        for (int jp = 0; jp < 12; ++jp) {
          const float wc = w, c0 = a0, c1 = a1, c2 = a2;
          __builtin_amdgcn_sched_barrier(0);
          T0 = mfma(wc, c0, T0);
          __builtin_amdgcn_sched_barrier(0);
          if (jp + 1 < 12) {
            w = wp[(2 * jp + 2) * 32];
            a0 = ap[(jp + 1) * 64]; a1 = ap[(12 + jp + 1) * 64]; a2 = ap[(24 + jp + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
          T1 = mfma(wc, c1, T1);
          T2 = mfma(wc, c2, T2);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {                      // dvp_c = sum_r T_{r,c} dverts_r, stored as 16-byte row quads
        f32x4 t;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = 4 * g + u;
          t[u] = fmaf(T2[q], dv[2][q], fmaf(T1[q], dv[1][q], T0[q] * dv[0][q]));
        }
        *quad_ptr(DVP, (size_t)c * (VP / 4) + vt * 8, g, BP, qoff) = t;
      }
      if constexpr (SPARSE) {
        // dA over the segment's 16-row joint window.  A operand: W16[n = lane % 16][v = acc_row(q, half)], four 16-byte
        // reads per tile (rows padded to 36 floats: conflict-free); B operand: this lane's register q of the accumulator
        // layout; one 4-block instruction adds rows acc_row(q, 0) and acc_row(q, 1) for all 32 pose columns.
        const int sg = segid[vt];                                    // wave-uniform
        if (sg != cur_seg) { flush_window(); cur_seg = sg; }
        seg_tile = vt;
        const f32x4* w16 = reinterpret_cast<const f32x4*>(tab + TB_WVJ + (lane & 15) * 36 + 4 * half);
        f32x4 wq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wq[g] = w16[2 * g];
        const f32x16& dvc = c == 0 ? dv[0] : c == 1 ? dv[1] : dv[2];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float wvj = wq[q >> 2][q & 3];
          const float p0 = dv[0][q] * vpc[q], p1 = dv[1][q] * vpc[q], p2 = dv[2][q] * vpc[q];
          acc3[0] = mfma16(wvj, p0, acc3[0]);
          acc3[1] = mfma16(wvj, p1, acc3[1]);
          acc3[2] = mfma16(wvj, p2, acc3[2]);
          acc4 = mfma16(wvj, dvc[q], acc4);
        }
      } else {
        const float* wvp = tab + TB_WVJ + l31;
        float wn = wvp[acc_row(0, half) * 32];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float wvj = wn;
          const float p0 = dv[0][q] * vpc[q], p1 = dv[1][q] * vpc[q], p2 = dv[2][q] * vpc[q];
          __builtin_amdgcn_sched_barrier(0);
          acc3[0] = mfma(wvj, p0, acc3[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (q + 1 < 16) wn = wvp[acc_row(q + 1, half) * 32];
          __builtin_amdgcn_sched_barrier(0);
          acc3[1] = mfma(wvj, p1, acc3[1]);
          acc3[2] = mfma(wvj, p2, acc3[2]);
        }
      }
      slot = slot1;
    };
    for (int vt = t_begin; vt < t_end; vt += 2) {
      tile(vt, vpA, vpB);
      if (vt + 1 < t_end) tile(vt + 1, vpB, vpA);
    }
    if constexpr (SPARSE) {
      flush_window();
      // joints no tile of this chunk touches: their slab rows are zero
#pragma unroll 1
      for (int j = 0; j < NJ; ++j) {
        if ((seen >> j) & 1u) continue;
        if (lane < 32) {
#pragma unroll
          for (int p = 0; p < 4; ++p) dATp[((size_t)(vc * 12 + (p < 3 ? p * 4 + c : c * 4 + 3)) * NJ + j) * BP + bcol] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = acc_row(q, half);
        if (j < NJ) {
#pragma unroll
          for (int r = 0; r < 3; ++r) dATp[((size_t)(vc * 12 + r * 4 + c) * NJ + j) * BP + bcol] = acc3[r][q];
        }
      }
    }
  } else {
    // ================= wave 3: dverts of the next tile for everybody, dA_{r,3} =================
    f32x16 acc3[3] = {zero16(), zero16(), zero16()};
    f32x16 dv[3];
    const unsigned qoffh = (unsigned)half * (unsigned)BP + (unsigned)bcol;    // lane part of a (row quad, pose) address
    // vertex adjoint of tile vt: Jn^T dj on the matrix cores and / or the caller's adjoint from memory
    auto compute_dv = [&](int vt, const float* tab) {
      if (DV != 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 t = *quad_ptr(dVT, (size_t)r * (VP / 4) + vt * 8, g, BP, qoffh);
            dv[r][4 * g] = t[0]; dv[r][4 * g + 1] = t[1]; dv[r][4 * g + 2] = t[2]; dv[r][4 * g + 3] = t[3];
          }
      } else {
        dv[0] = zero16(); dv[1] = zero16(); dv[2] = zero16();
      }
      if (DV != 1) {
        const float* jp_ = tab + TB_JN + half * 32 + l31;
        const float* dp_ = ldsDj + lane;
        float jn = jp_[0], d0 = dp_[0], d1 = dp_[9 * 64], d2 = dp_[18 * 64];
#pragma unroll
        for (int ip = 0; ip < 9; ++ip) {
          const float jc = jn, c0 = d0, c1 = d1, c2 = d2;
          __builtin_amdgcn_sched_barrier(0);
          dv[0] = mfma(jc, c0, dv[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (ip + 1 < 9) {
            jn = jp_[(2 * ip + 2) * 32];
            d0 = dp_[(ip + 1) * 64]; d1 = dp_[(9 + ip + 1) * 64]; d2 = dp_[(18 + ip + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
          dv[1] = mfma(jc, c1, dv[1]);
          dv[2] = mfma(jc, c2, dv[2]);
        }
      }
    };
    auto publish_dv = [&]() {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 t = {dv[r][4 * g], dv[r][4 * g + 1], dv[r][4 * g + 2], dv[r][4 * g + 3]};
          dvbuf[(r * 4 + g) * 64 + lane] = t;
        }
    };
    // dA_{r,3}[j,b] += sum_v W[v,j] dverts_r[v,b]
    auto accumulate_dA3 = [&](const float* tab) {
      const float* wvp = tab + TB_WVJ + l31;
      float wn = wvp[acc_row(0, half) * 32];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float wvj = wn;
        __builtin_amdgcn_sched_barrier(0);
        acc3[0] = mfma(wvj, dv[0][q], acc3[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < 16) wn = wvp[acc_row(q + 1, half) * 32];
        __builtin_amdgcn_sched_barrier(0);
        acc3[1] = mfma(wvj, dv[1][q], acc3[1]);
        acc3[2] = mfma(wvj, dv[2][q], acc3[2]);
      }
    };
    if (DV != 1) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int ip = 0; ip < 9; ++ip) ldsDj[(r * 9 + ip) * 64 + lane] = dJT[(size_t)(r * NHP + 2 * ip + half) * BP + bcol];
    }
    __syncthreads();                                  // records of the first two tiles landed
    compute_dv(t_begin, ring);
    publish_dv();
    if constexpr (!SPARSE) accumulate_dA3(ring);      // joint-sparse: the plane waves take dA_{c,3} (16-row products)
    int slot = 0;
    for (int vt = t_begin; vt < t_end; ++vt) {
      barrier_keep_vm<0>();                           // nothing younger than this wave's record copies is in flight
      const int slot1 = (slot + 1 == BWD_RING) ? 0 : slot + 1, slot2 = (slot1 + 1 == BWD_RING) ? 0 : slot1 + 1;
      if (vt + 2 < t_end) issue(vt + 2, slot2);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (vt + 1 < t_end) {      // this wave's whole tile (27 + 48 MFMA) runs beside the plane waves' 84
        compute_dv(vt + 1, ring + slot1 * TB_FLOATS);
        publish_dv();
        if constexpr (!SPARSE) accumulate_dA3(ring + slot1 * TB_FLOATS);
      }
      slot = slot1;
    }
    if constexpr (!SPARSE) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = acc_row(q, half);
        if (j < NJ) {
#pragma unroll
          for (int r = 0; r < 3; ++r) dATp[((size_t)(vc * 12 + r * 4 + 3) * NJ + j) * BP + bcol] = acc3[r][q];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, joint-sparse models: FOUR SYMMETRIC WAVES, each alone with 16 poses
//   The role kernel above splits a 32-pose tile over three plane waves and a vertex-adjoint wave: one wave is always late,
//   and two barriers per tile tie the four together.  With 16x16 matrix tiles (v_mfma_f32_16x16x4_f32, the same FLOP per clock
//   as 32x32x2) a (32 vertices x 16 poses) tile is 8 registers instead of 16 per lane, the twelve dA accumulators of a
//   16-row joint window are 48 registers instead of 192 -- and ONE wave can hold everything for its own 16 poses: vertex
//   adjoint, T recompute, dvp, all twelve dA products.  No roles, no hand-off through LDS, no imbalance; the four waves of a
//   workgroup (64 poses) only share the per-tile operand records (LDS-DMA ring, ONE barrier per tile).
//   16x16x4:  A: lane l holds A[m = l % 16][k = l / 16]   B: lane l holds B[k = l / 16][n = l % 16]
//             D: register i of lane l holds D[row 4 (l / 16) + i][col l % 16]
//   so register i of a lane group g = l / 16 is vertex row 16 blk + 4 g + i of its pose column: four consecutive rows = one
//   16-byte row quad of VPb / DVP (one dwordx4 per (plane, blk)), and -- as the B operand of a product that sums over the
//   tile's rows -- K index g of the step (blk, i), whose A operand therefore is W16[n][16 blk + 4 g + i].
//   Record R16 of a tile (k_jreg_tiles / k_bwd_tab_static write it in place of the role kernel's):
//     JN [blk][s < 5][64 lanes]   Jn[i = 4 s + g][16 blk + m]            A operands of dverts  (i >= 17: 0)
//     WT [blk][s < 3][64 lanes]   W[16 blk + m][joint slot 4 s + g]       A operands of T       (slots >= KJ: 0)
//     WD [blk][g][n][4]           W16[n][16 blk + 4 g + 0..3]             A operands of dA, one ds_read_b128 per blk
//     JL [16]                     16 * joint of slot k (ints)              row offsets into the wave's A^T slice
//     WX [blk][64 lanes]          K step 3 of WT (joint slots 12 .. 15)   a WIDE tile's T recompute runs ceil(joints / 4) K steps:
//   the kernel is built for S = KJ / 4 of them; a tile with more joints (tnj[vt] > KJ, at most 16) runs the extra steps in a
//   rare wave-uniform branch -- it costs itself, nobody else.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// NG (round 5): 16-pose column groups per wave.  1: the round-3 shape -- two workgroups per CU, two waves per SIMD at 256 registers.
// 2: ONE workgroup per CU, one wave per SIMD with the 512-register budget, each wave holding TWO column groups: every A operand of
// the three products (Jn, the tile's W^T rows, the W16 window) is pose-independent, so one LDS read feeds two matrix instructions
// (28 -> 14 LDS reads per 162 instructions), the two groups are two independent accumulator chains the wave interleaves itself --
// what the SIMD's second wave did, without its duplicate barrier, prologue and flush -- and the next tile's v_posed quads of both
// groups wait in registers.  A workgroup then covers 128 poses.
template <int DV, int KJ, bool WIDE, bool LIST = false, int NG = 1>      // WIDE, LIST: as in k_lbs_fwd (WIDE: 0.184 vs 0.187 ms)
__global__ __launch_bounds__(256, NG == 2 ? 1 : 2) void k_lbs_bwd16(const float* __restrict__ Tb, const float* __restrict__ AT,
                                                      const float* __restrict__ VPb, const float* __restrict__ dJT,
                                                      const float* __restrict__ dVT, float* __restrict__ DVP,
                                                      float* __restrict__ dATp, int BP, int nvc, int n_bt,
                                                      const int* __restrict__ segid, const int* __restrict__ segj, int paired,
                                                      const int* __restrict__ tnj, const int* __restrict__ tl, int ntl,
                                                      unsigned* __restrict__ dmask, int rev) {
  // rev (experiment, JRR_BWD16_REV=1; DESIGN.md section 8 "stage B"): walk the chunk's tiles from the LAST to the first -- the tiles
  // k_lbs_fwd wrote last are then read first, while they may still sit in the 256 MB Infinity Cache
  // dmask (nullable): [chunk][64-pose group] bit j = this workgroup wrote the rows of joint j of its dA slab.  The rows of the joints
  // no tile of the chunk touches are then NOT zero-filled (and not read: k_chain_bwd takes the mask) -- at 4096 poses the eight slabs
  // are 37.7 MB, about half of it rows of zeros (timing-only ablation of flush + fill: 18 us of the launch).
  // LIST: the ntl tiles tl[] only.  A tile without a support entry of the regressor has a zero vertex adjoint (DV == 0: no adjoint
  // from outside): its dvp and its dA contributions are exact zeros, and the blend adjoint skips its rows of DVP as well.
  const int NT = LIST ? ntl : VT;
  constexpr int S = KJ / 4;                          // K steps of the T product (4 joint slots each)
  constexpr int ASL = 9 * NJ * 16;                   // floats of one wave's A^T slice [(r,c)][24 joints][16 poses]
  __shared__ __attribute__((aligned(16))) float lds[BWD_RING * R16_FLOATS + 4 * NG * ASL];
  float* const ring = lds;
  const int tid = threadIdx.x;
  const int lane = tid & 63, n = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const ldsA = lds + BWD_RING * R16_FLOATS + wv * NG * ASL;      // group gi at + gi * ASL
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  // default mapping: an XCD's contiguous range of L = one vertex chunk x all pose groups (paired == -2: chunk-major inside a
  // pose group instead; measured equal or slower: 0.200 vs 0.203 ms at B = 4096, 0.069 vs 0.066 ms at B = 1024)
  int vc = (paired == -2) ? L % nvc : L / n_bt, bt = (paired == -2) ? L / nvc : L % n_bt;
  int t_begin = (int)((long)NT * vc / nvc), t_end = (int)((long)NT * (vc + 1) / nvc);
  if (paired > 0) {
    // Exactly one round of two workgroups per CU (k_lbs_fwd's geometry, see there): the two workgroups of a CU -- dispatch
    // slots j and j + per_xcd / 2 of an XCD -- take the two halves of a PAIR of vertex chunks of the SAME pose group.
    // Measured (B = 4096): 0.204 -> 0.184 ms; what matters is that the two share the pose group (the chunk-major mapping
    // without pairing gains nothing), the split itself is flat around even: `paired` = the first-dispatched workgroup's
    // share of the pair's tiles in thousandths (launcher default).  Static: bitwise reproducible.
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int p = j % (per_xcd / 2), slot_ = j / (per_xcd / 2);
    const int npair = nvc / 2, bt_per_xcd = (per_xcd / 2) / npair;
    bt = xcd * bt_per_xcd + p / npair;
    const int vcp = p % npair;
    vc = 2 * vcp + slot_;
    const int P0 = (int)((long)NT * vcp / npair), P1 = (int)((long)NT * (vcp + 1) / npair);
    const int mid = P0 + ((P1 - P0) * paired + 500) / 1000;       // `paired` = the first workgroup's share in thousandths
    t_begin = slot_ ? mid : P0;
    t_end = slot_ ? P1 : mid;
  }
  size_t bcol[NG];                                            // this lane's pose column in each of the wave's groups
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) bcol[gi] = (size_t)bt * (64 * NG) + wv * (16 * NG) + gi * 16 + n;
  const unsigned lane_ln = (unsigned)lane * 4u;

  auto issue = [&](int vt, int slot) {
    const float* src = Tb + (size_t)vt * TB_FLOATS;
    float* dst = ring + slot * R16_FLOATS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = wv + 4 * i;
      if (o < R16_PIECES) dma16u(src + o * 256, lane_ln, dst + o * 256);
    }
  };
  auto tile_of = [&](int ix) { const int k = rev ? t_begin + t_end - 1 - ix : ix; return LIST ? tl[k] : k; };      // wave-uniform (scalar load)
  if (t_begin < t_end) issue(tile_of(t_begin), 0);
  if (t_begin + 1 < t_end) issue(tile_of(t_begin + 1), 1);

  // this wave's A^T slice (the 3x3 rotation part; the translation column is not needed for T) -> LDS, once
#pragma unroll
  for (int rc = 0; rc < 9; ++rc)
#pragma unroll
    for (int jj = 0; jj < NJ / 4; ++jj) {
      const int j = 4 * jj + g;
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) ldsA[gi * ASL + (rc * NJ + j) * 16 + n] = AT[(size_t)(((rc / 3) * 4 + rc % 3) * NJ + j) * BP + bcol[gi]];
    }
  // joint adjoint as B operands of the vertex-adjoint product: dj[r][s] = dJ^T[r][i = 4 s + g][pose]
  float dj[NG][3][5];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const int i = 4 * s5 + g;
        dj[gi][r][s5] = (DV != 1 && i < NHP) ? dJT[(size_t)(r * NHP + i) * BP + bcol[gi]] : 0.f;
      }

  // row quad (plane, tile, blk) of this lane: rows 32 vt + 16 blk + 4 g + 0..3 of its pose
  auto qidx = [&](int plane, int vt, int blk, int gi) { return ((size_t)plane * (VP / 4) + vt * 8 + 4 * blk + g) * BP + bcol[gi]; };
  const f32x4* const VP4 = reinterpret_cast<const f32x4*>(VPb);
  const f32x4* const dV4 = reinterpret_cast<const f32x4*>(dVT);
  f32x4* const DVP4 = reinterpret_cast<f32x4*>(DVP);
  auto load_vp = [&](int vt, f32x4 (&dst)[NG][3][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) dst[gi][c][blk] = __builtin_nontemporal_load(&VP4[qidx(c, vt, blk, gi)]);   // read once: do not displace the operands in L2
  };

  f32x4 acc[NG][12];                               // dA_{r,c} at 3 r + c (c < 3), dA_{r,3} at 9 + r: 16 window rows x 16 poses
#pragma unroll
  for (int gi = 0; gi < NG; ++gi)
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[gi][e] = zero4();
  int cur_seg = (t_begin < t_end) ? segid[tile_of(t_begin)] : 0;
  int seg_tile = (t_begin < t_end) ? tile_of(t_begin) : 0;
  unsigned seen = 0u;
  // close a segment: add the accumulators into this workgroup's dA slab (first touch of a joint stores, later ones add;
  // the 16 pose columns belong to this wave alone)
  auto flush_window = [&]() __attribute__((always_inline)) {
    // a later flush re-reads slab rows an earlier one stored: those stores must have landed.  The FIRST flush of a workgroup (seen == 0,
    // wave-uniform: most workgroups lie inside one segment and flush once, at the end) reads nothing and need not wait for the tile's
    // last dvp stores
    if (seen != 0u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int* sj = segj + seg_tile * 16;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
    // rows an earlier flush of this workgroup stored are read back FIRST, all of them in flight together (one agent-scope load per
    // row and entry: issued one by one between the stores they feed, 48 dependent L2 round trips made the flush the largest fixed
    // cost of the launch), then everything is stored
    int jr[4];
    float* rowp[4];
    float oldv[4][12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      jr[i] = sj[4 * g + i];
      rowp[i] = dATp + ((size_t)(vc * 12) * NJ + (jr[i] >= 0 ? jr[i] : 0)) * BP + bcol[gi];
      const bool add = jr[i] >= 0 && ((seen >> jr[i]) & 1u);
#pragma unroll
      for (int e = 0; e < 12; ++e) {
        const int ent = e < 9 ? (e / 3) * 4 + e % 3 : (e - 9) * 4 + 3;
        // L1 bypassed: an earlier flush of this wave stored it, possibly from another lane
        oldv[i][e] = add ? __hip_atomic_load(rowp[i] + (size_t)ent * NJ * BP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (jr[i] >= 0) {
        const bool add = (seen >> jr[i]) & 1u;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
          const int ent = e < 9 ? (e / 3) * 4 + e % 3 : (e - 9) * 4 + 3;
          const float val = add ? acc[gi][e][i] + oldv[i][e] : acc[gi][e][i];
          __hip_atomic_store(rowp[i] + (size_t)ent * NJ * BP, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // paired with the agent-scope re-read of a later flush
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[gi][e] = zero4();
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int j = sj[k];
      if (j >= 0) seen |= 1u << j;
    }
  };

  f32x4 vpA[NG][3][2], vpB[NG][3][2];
  if (t_begin < t_end) load_vp(tile_of(t_begin), vpA);
  int slot = 0;
  auto tile = [&](int ix, const f32x4 (&vpc)[NG][3][2], f32x4 (&vpn)[NG][3][2]) __attribute__((always_inline)) {
    const int vt = tile_of(ix);
    // record vt has landed (it was issued two tiles ago: older than the 6 prefetch loads and the 6 dvp stores of the previous
    // tile, which stay in flight) and every wave is done with the slot the next copy overwrites
    barrier_keep_vm<12 * NG>();
    const int slot1 = (slot + 1 == BWD_RING) ? 0 : slot + 1, slot2 = (slot1 + 1 == BWD_RING) ? 0 : slot1 + 1;
    if (ix + 2 < t_end) issue(tile_of(ix + 2), slot2);
    __builtin_amdgcn_sched_barrier(0);
    const float* tab = ring + slot * R16_FLOATS;
    if (ix + 1 < t_end) load_vp(tile_of(ix + 1), vpn);
    // ---- vertex adjoint of the tile: dverts_r = Jn^T dj_r and / or the caller's ----
    f32x4 dv[NG][3][2];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) dv[gi][r][blk] = (DV != 0) ? dV4[qidx(r, vt, blk, gi)] : zero4();
    if (DV != 1) {
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const float a = tab[R16_JN + (blk * 5 + s5) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) dv[gi][r][blk] = mfma4(a, dj[gi][r][s5], dv[gi][r][blk]);
        }
    }
    // ---- operands shared by the planes: W^T (A of T), W16 (A of dA), the A^T row offsets of the tile's joint slots ----
    float wt[2][S];
    int jo[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      jo[s] = reinterpret_cast<const int*>(tab + R16_JL)[4 * s + g];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) wt[blk][s] = tab[R16_WT + (blk * 3 + s) * 64 + lane];
    }
    f32x4 wd[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) wd[blk] = *reinterpret_cast<const f32x4*>(tab + R16_WD + ((blk * 4 + g) * 16 + n) * 4);
    const int sg = segid[vt];                                      // wave-uniform
    const int s_tile = WIDE ? (tnj[vt] + 3) >> 2 : 0;              // K steps this tile's joints need (wave-uniform)
    if (sg != cur_seg) { flush_window(); cur_seg = sg; }
    seg_tile = vt;
    // ---- translation column: dA_{r,3} += W16^T dverts_r ----
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int gi = 0; gi < NG; ++gi) acc[gi][9 + r] = mfma4(wd[blk][i], dv[gi][r][blk][i], acc[gi][9 + r]);
    // ---- per plane c: T_{r,c} (recomputed), dvp_c = sum_r T_{r,c} dverts_r, dA_{r,c} += W16^T (dverts_r * vp_c) ----
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {      // (one group's T at a time: 24 live registers, not 24 NG -- the W^T operands stay in registers)
        f32x4 T[3][2];
#pragma unroll
        for (int r = 0; r < 3; ++r) { T[r][0] = zero4(); T[r][1] = zero4(); }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float at = ldsA[gi * ASL + (r * 3 + c) * (NJ * 16) + jo[s] + n];
            T[r][0] = mfma4(wt[0][s], at, T[r][0]);
            T[r][1] = mfma4(wt[1][s], at, T[r][1]);
          }
        if (WIDE && s_tile > S) {                                     // WIDE tile: the K steps beyond the kernel's S (slots 4 S .. 15)
#pragma unroll 1
          for (int s = S; s < s_tile && s < 4; ++s) {
            const int jox = reinterpret_cast<const int*>(tab + R16_JL)[4 * s + g];
            const float w0 = s < 3 ? tab[R16_WT + (0 * 3 + s) * 64 + lane] : tab[R16_WX + lane];
            const float w1 = s < 3 ? tab[R16_WT + (1 * 3 + s) * 64 + lane] : tab[R16_WX + 64 + lane];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
              const float at = ldsA[gi * ASL + (r * 3 + c) * (NJ * 16) + jox + n];
              T[r][0] = mfma4(w0, at, T[r][0]);
              T[r][1] = mfma4(w1, at, T[r][1]);
            }
          }
        }
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          f32x4 t;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            t[i] = fmaf(T[2][blk][i], dv[gi][2][blk][i], fmaf(T[1][blk][i], dv[gi][1][blk][i], T[0][blk][i] * dv[gi][0][blk][i]));
          __builtin_nontemporal_store(t, &DVP4[qidx(c, vt, blk, gi)]);      // streamed (340 MB per launch): nt, measured -2.5 us
        }
      }
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) acc[gi][3 * r + c] = mfma4(wd[blk][i], dv[gi][r][blk][i] * vpc[gi][c][blk][i], acc[gi][3 * r + c]);
    }
    slot = slot1;
  };
  for (int ix = t_begin; ix < t_end; ix += 2) {
    tile(ix, vpA, vpB);
    if (ix + 1 < t_end) tile(ix + 1, vpB, vpA);
  }
  flush_window();
  if (dmask) {      // one word per (chunk, 64-pose group): this workgroup's NG groups of 64 poses touched the same joints
    if (tid < NG) dmask[vc * (n_bt * NG) + bt * NG + tid] = seen;
    return;
  }
  // joints no tile of this chunk touches: zero rows
#pragma unroll 1
  for (int j = 0; j < NJ; ++j) {
    if ((seen >> j) & 1u) continue;
    if (lane < 16) {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int ent = 0; ent < 12; ++ent) dATp[((size_t)(vc * 12 + ent) * NJ + j) * BP + bcol[gi]] = 0.f;
    }
  }
}

// static (W) parts of the per-tile backward operand records
// (Wc != NULL: the W^T block takes the kjs compacted rows of the joint-sparse path, zeros behind them)
__global__ void k_bwd_tab_static(const float* __restrict__ Wjv, const float* __restrict__ Wvj, float* __restrict__ Tb,
                                 const float* __restrict__ Wc, int kjs, const float* __restrict__ W16,
                                 const int* __restrict__ jl, int r16) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over VT * 2048
  if (idx >= VT * 2048) return;
  const int vt = idx >> 11, k = idx & 2047;
  float* dst = Tb + (size_t)vt * TB_FLOATS;
  if (r16) {      // record R16 of k_lbs_bwd16 (everything but its JN block, which k_jreg_tiles writes)
    if (k >= R16_FLOATS - R16_WT) return;
    const int o = R16_WT + k;
    // joint slots: the NJ-slot tables of the model (Wc [VT][NJ][32], jl [VT][NJ]); 16 of them fit the record
    if (o < R16_WD) {               // WT [blk][s < 3][g][m] = W[16 blk + m][slot 4 s + g]
      const int q = o - R16_WT, blk = q / 192, s = (q / 64) % 3, g = (q >> 4) & 3, m = q & 15;
      const int slot = 4 * s + g;
      dst[o] = Wc[((size_t)vt * NJ + slot) * 32 + 16 * blk + m];
    } else if (o < R16_JL) {        // WD [blk][g][n][i] = W16[n][16 blk + 4 g + i]
      const int q = o - R16_WD, blk = q >> 8, g = (q >> 6) & 3, n = (q >> 2) & 15, i = q & 3;
      dst[o] = W16[(size_t)vt * TB_W16_FLOATS + n * 36 + 16 * blk + 4 * g + i];
    } else if (o < R16_WX) {        // JL [16]: 16 * joint of slot k (row offset into a wave's A^T slice), as an int
      const int q = o - R16_JL;
      reinterpret_cast<int*>(dst)[o] = 16 * jl[vt * NJ + q];
    } else if (o < R16_WX + 128) {  // WX [blk][g][m] = W[16 blk + m][slot 12 + g]: K step 3
      const int q = o - R16_WX, blk = q >> 6, g = (q >> 4) & 3, m = q & 15;
      dst[o] = Wc[((size_t)vt * NJ + 12 + g) * 32 + 16 * blk + m];
    } else dst[o] = 0.f;
    return;
  }
  if (k < W_FLOATS) dst[TB_WJV + k] = Wc ? (k < kjs * 32 ? Wc[(size_t)vt * W_FLOATS + k] : 0.f) : Wjv[(size_t)vt * W_FLOATS + k];
  else if (k < W_FLOATS + 1024) {
    const int kk = k - W_FLOATS;
    dst[TB_WVJ + kk] = Wc ? (kk < TB_W16_FLOATS ? W16[(size_t)vt * TB_W16_FLOATS + kk] : 0.f) : Wvj[(size_t)vt * 1024 + kk];
  }
  else if (k - W_FLOATS - 1024 < TB_FLOATS - TB_WVJ - 1024) dst[TB_WVJ + 1024 + (k - W_FLOATS - 1024)] = 0.f;
}

// [3][VP/4][BP][4] vertices (row quads) -> pose-major (B, ldv) rows of (v, xyz) and/or projected (x_ndc, y_ndc, Z, 0)
// records for the silhouette renderer (scripts/optimize.py:80-82 flip/scale + scripts/mesh_renderer.py:52-57 camera).
// One 32-vertex x 32-pose tile per block through LDS; every global access is contiguous (16 bytes per thread on the
// quad side).
// (p2v != NULL: row v of the buffer is vertex p2v[v] of the caller's mesh -- the store then is per vertex, 12 bytes)
__global__ void k_verts_untranspose(const float* __restrict__ VTb, float* __restrict__ verts, int ldv, int vlimit,
                                    const float* __restrict__ cam, f32x4* __restrict__ ndc, float focal, int B, int BP,
                                    const int* __restrict__ p2v) {
  __shared__ float tile[32][97];
  const int v0 = blockIdx.x * 32, bb0 = blockIdx.y * 32;
  const f32x4* VTq = reinterpret_cast<const f32x4*>(VTb);
  for (int idx = threadIdx.x; idx < 3 * 8 * 32; idx += blockDim.x) {
    const int r = idx >> 8, vq = (idx >> 5) & 7, bl = idx & 31;
    const f32x4 t = VTq[((size_t)r * (VP / 4) + (v0 >> 2) + vq) * BP + bb0 + bl];
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[bl][(vq * 4 + u) * 3 + r] = t[u];
  }
  __syncthreads();
  if (verts) {
    for (int idx = threadIdx.x; idx < 32 * 96; idx += blockDim.x) {
      const int bl = idx / 96, rem = idx % 96;
      const int b = bb0 + bl, v = v0 + rem / 3;
      if (p2v) {
        const int vo = v < V ? p2v[v] : -1;
        if (b < B && vo >= 0 && vo < vlimit) verts[(size_t)b * ldv + vo * 3 + rem % 3] = tile[bl][rem];
      } else if (b < B && v < vlimit) verts[(size_t)b * ldv + v0 * 3 + rem] = tile[bl][rem];
    }
  }
  if (ndc) {
    for (int idx = threadIdx.x; idx < 32 * 32; idx += blockDim.x) {
      const int bl = idx / 32, vv = idx % 32;
      const int b = bb0 + bl, v = v0 + vv;
      const int vo = (v < V) ? (p2v ? p2v[v] : v) : -1;
      if (b < B && vo >= 0) {
        const float X = -2.f * tile[bl][vv * 3] + cam[(size_t)b * 3], Y = -2.f * tile[bl][vv * 3 + 1] + cam[(size_t)b * 3 + 1];
        const float Z = 2.f * tile[bl][vv * 3 + 2] + cam[(size_t)b * 3 + 2];
        f32x4 o = {focal * X / Z, focal * Y / Z, Z, 0.f};
        ndc[(size_t)b * V + vo] = o;
      }
    }
  }
}

int launch_verts_untranspose(const float* VTb, float* verts, int ldv, int vlimit, const float* cam, float* ndc, int B, int BP,
                             hipStream_t s, const int* p2v) {
  hipLaunchKernelGGL(k_verts_untranspose, dim3(VT, BP / 32), dim3(256), 0, s, VTb, verts, ldv, vlimit, cam, (f32x4*)ndc,
                     5000.f / 224.f, B, BP, p2v);
  return 0;
}

// (B,6890,3) -> [3][VP/4][BP][4] (row quads) transpose of an external vertex adjoint (operator-level SMPL backward,
// silhouette adjoint)
__global__ void k_dverts_transpose(const float* __restrict__ dverts, int ldv, float* __restrict__ dVT, int B, int BP,
                                   const int* __restrict__ p2v) {
  __shared__ float tile[32][97];
  const int v0 = blockIdx.x * 32, bb0 = blockIdx.y * 32;
  for (int idx = threadIdx.x; idx < 32 * 96; idx += blockDim.x) {
    int bl = idx / 96, rem = idx % 96;   // rem = vv*3 + r
    int b = bb0 + bl, v = v0 + rem / 3;
    float val = 0.f;
    if (b < B && v < V) val = p2v ? dverts[(size_t)b * ldv + p2v[v] * 3 + rem % 3] : dverts[(size_t)b * ldv + v0 * 3 + rem];
    tile[bl][rem] = val;
  }
  __syncthreads();
  f32x4* dVq = reinterpret_cast<f32x4*>(dVT);
  for (int idx = threadIdx.x; idx < 3 * 8 * 32; idx += blockDim.x) {
    const int r = idx >> 8, vq = (idx >> 5) & 7, bl = idx & 31;
    f32x4 t;
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = tile[bl][(vq * 4 + u) * 3 + r];
    dVq[((size_t)r * (VP / 4) + (v0 >> 2) + vq) * BP + bb0 + bl] = t;
  }
}

// ------------------------------------------------------------------------------------------
// J_regressor normalisation (scripts/utils.py:87-92) into the tile layouts, and its adjoint
// ------------------------------------------------------------------------------------------
constexpr int JREG_THREADS = 1024;     // one block per regressor row: 6890 columns, 7 per thread
__global__ __launch_bounds__(JREG_THREADS) void k_jreg_rowsum(const float* __restrict__ J, const float* __restrict__ mask,
                                                               float* __restrict__ rowsum, int* __restrict__ sup_flag,
                                                               int32_t* __restrict__ step_inc) {
  __shared__ float red[JREG_THREADS];
  const int i = blockIdx.x;
  if (sup_flag && i == 0 && threadIdx.x == 0) *sup_flag = 1;      // k_jreg_support (two launches later) may clear it
  if (step_inc && i == 0 && threadIdx.x == 0) step_inc[0] += 1;   // the J step's Adam count (k_adam_flat, the launch before, used + 1)
  float acc = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float x = J[(size_t)i * V + v];
    if (mask) x *= mask[(size_t)i * V + v];
    acc += fmaxf(x, 0.f);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = JREG_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowsum[i] = red[0];
}

__global__ void k_jreg_tiles(const float* __restrict__ J, const float* __restrict__ mask,
                             const float* __restrict__ rowsum, float* __restrict__ Jn, float* __restrict__ Jn_vi,
                             float* __restrict__ Jn_iv, float* __restrict__ Jn_q, const int* __restrict__ p2v, int r16) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over VT*32*32 (tile, vv, i)
  if (idx >= VT * 1024) return;
  const int vt = idx >> 10, vv = (idx >> 5) & 31, i = idx & 31;
  const int v = vt * 32 + vv;                  // row of the internal vertex order
  const int vo = (v < V) ? (p2v ? p2v[v] : v) : -1;   // the regressor column stored there
  float val = 0.f;
  if (i < NH && vo >= 0) {
    float x = J[(size_t)i * V + vo];
    if (mask) x *= mask[(size_t)i * V + vo];
    val = fmaxf(x, 0.f) / rowsum[i];
    Jn[(size_t)i * V + vo] = val;            // Jn keeps the regressor's own column order
  }
  Jn_vi[(size_t)vt * 1024 + vv * 32 + i] = val;
  Jn_q[((size_t)(v >> 2) * 32 + i) * 4 + (v & 3)] = val;      // vertex quads [VP/4][32][4]: A operand of k_gemm_q32
  // backward operand record: the role kernel's [i][32 v], or R16's JN [blk][s = i / 4][g = i % 4][m] (k_lbs_bwd16)
  if (r16) { if (i < 20) Jn_iv[(size_t)vt * TB_FLOATS + R16_JN + ((vv >> 4) * 5 + (i >> 2)) * 64 + (i & 3) * 16 + (vv & 15)] = val; }
  else if (i < NHP) Jn_iv[(size_t)vt * TB_FLOATS + TB_JN + i * 32 + vv] = val;
}

// ------------------------------------------------------------------------------------------
// J step over the regressor's SUPPORT (scripts/optimize.py:300-312).  dJ_raw = mask relu'(J mask) (dJn - <dJn, Jn>) / rowsum is
// exactly zero wherever J mask <= 0, and <dJn, Jn> only meets dJn where Jn > 0: the J step needs dJn -- a (17 x 6890) x B product
// over the 340 MB of stored vertices -- on the positive entries of J only (62 of 117 130 for the shipped checkpoint, and Adam
// never re-activates an entry: its gradient is zero while it is <= 0).  The same holds for the re-regression of the joints with
// the stepped regressor.  Both become gathers of a few vertex rows; rows with more than JSUP_CAP positive entries switch the
// whole step back to the dense products (flag = 0, decided on the device: no host synchronisation).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JREG_THREADS) void k_jreg_support(const float* __restrict__ Jn, const int* __restrict__ v2p, JSupport sup) {
  __shared__ int wcount[JREG_THREADS / 64];
  __shared__ int base;
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int v0 = 0; v0 < V; v0 += JREG_THREADS) {            // ascending vertex order: deterministic lists
    const int v = v0 + threadIdx.x;
    const float w = (v < V) ? Jn[(size_t)i * V + v] : 0.f;
    const bool on = w > 0.f;
    const unsigned long long bal = __ballot(on);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int q = 0; q < wave; ++q) off += wcount[q];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (on && off < JSUP_CAP) { sup.col[i * JSUP_CAP + off] = v2p ? v2p[v] : v; sup.val[i * JSUP_CAP + off] = w; }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < JREG_THREADS / 64; ++q) t += wcount[q]; base += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sup.cnt[i] = base < JSUP_CAP ? base : JSUP_CAP;
    if (base > JSUP_CAP) atomicMin(sup.flag, 0);            // (flag is set to 1 by k_jreg_rowsum, the launch before)
  }
}

// tmask[tile] = 1 for the 32-vertex tiles that hold a support entry: the J step's forward stores the vertices of those tiles only
__global__ __launch_bounds__(256) void k_jsup_tilemask(JSupport sup) {
  __shared__ int m[VT];
  for (int t = threadIdx.x; t < VT; t += 256) m[t] = 0;
  __syncthreads();
  for (int idx = threadIdx.x; idx < NH * JSUP_CAP; idx += 256) {
    const int i = idx / JSUP_CAP, e = idx % JSUP_CAP;
    if (e < sup.cnt[i]) m[sup.col[idx] >> 5] = 1;
  }
  __syncthreads();
  const bool known = sup.flag[JSUP_KNOWN] != 0;
  for (int t = threadIdx.x; t < VT; t += 256) {
    sup.tmask[t] = m[t];
    // the host has been told which tiles hold the support (and enqueues support-restricted work only): a J step can only shrink
    // it (ReLU' = 0 outside it) -- unless the caller changed J or the mask in place.  Loud, not silently wrong: sticky error word,
    // reported by the next jrr_j_support_info
    if (known && m[t] && !sup.tknown[t]) atomicOr(&sup.flag[JSUP_ERR], 1);
  }
  if (known && threadIdx.x == 0 && sup.flag[0] == 0) atomicOr(&sup.flag[JSUP_ERR], 2);
}
int launch_jsup_tilemask(const JSupport& sup, hipStream_t s) {
  hipLaunchKernelGGL(k_jsup_tilemask, dim3(1), dim3(256), 0, s, sup);
  return 0;
}

// one workgroup per (row i, entry slot): dJn[i][row] = sum_b sum_r dj_r[i][b] verts_r[row][b]; fixed-order sums.
// (1024 threads: the sum over the poses is a chain of dependent load rounds -- 4 per thread at 4096 poses instead of 16: 17 -> ~6 us)
constexpr int JGS_THREADS = 1024;
__global__ __launch_bounds__(JGS_THREADS) void k_jgrad_sparse(JSupport sup, const float* __restrict__ dJT, const float* __restrict__ VTq,
                                                              float* __restrict__ dJn, int BP) {
  if (*sup.flag == 0) return;
  __shared__ float red[JGS_THREADS];
  const int i = blockIdx.x;
  for (int e = blockIdx.y; e < sup.cnt[i]; e += gridDim.y) {
    const int row = sup.col[i * JSUP_CAP + e];
    float acc = 0.f;
    for (int b = threadIdx.x; b < BP; b += JGS_THREADS) {          // padded poses carry dj = 0
#pragma unroll
      for (int r = 0; r < 3; ++r)
        acc = fmaf(dJT[(size_t)(r * NHP + i) * BP + b], VTq[(((size_t)r * (VP / 4) + (row >> 2)) * BP + b) * 4 + (row & 3)], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = JGS_THREADS / 2; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) dJn[(size_t)i * VP + row] = red[0];
    __syncthreads();
  }
}

// joints^T slab [3][32][BP] (rows i < 17) of the stored vertices with the current regressor: one thread per (pose, row i)
__global__ __launch_bounds__(256) void k_rejoints_sparse(JSupport sup, const float* __restrict__ VTq, float* __restrict__ out, int BP,
                                                         int32_t* __restrict__ step_inc) {
  if (step_inc && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) step_inc[0] += 1;      // before the early return
  if (*sup.flag == 0) return;
  const int b = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (b >= BP) return;
  float acc[3] = {0.f, 0.f, 0.f};
  const int n = sup.cnt[i];
  for (int e = 0; e < n; ++e) {
    const int row = sup.col[i * JSUP_CAP + e];
    const float w = sup.val[i * JSUP_CAP + e];
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[r] = fmaf(w, VTq[(((size_t)r * (VP / 4) + (row >> 2)) * BP + b) * 4 + (row & 3)], acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) out[(size_t)(r * 32 + i) * BP + b] = acc[r];
}

// The J step's gradient restricted to the regressor's support, for the data-parallel all-reduce: dJ (17 x 6890, file order of
// the columns) is exactly zero outside the support, so the ranks exchange [17][JSUP_CAP] floats instead of 468 520 bytes.
// The compact values come out of k_jreg_bwd (zeros behind the row's count); scatter: their way back into a zero-filled dJ.
__global__ __launch_bounds__(JSUP_CAP) void k_jsup_scatter(JSupport sup, const float* __restrict__ in, const int* __restrict__ p2v,
                                                           float* __restrict__ dJ) {
  const int i = blockIdx.x, e = threadIdx.x;
  if (e < sup.cnt[i]) { const int row = sup.col[i * JSUP_CAP + e]; dJ[(size_t)i * V + (p2v ? p2v[row] : row)] = in[i * JSUP_CAP + e]; }
}
int launch_jsup_scatter(const JSupport& sup, const float* in, const int* p2v, float* dJ, hipStream_t s) {
  hipLaunchKernelGGL(k_jsup_scatter, dim3(NH), dim3(JSUP_CAP), 0, s, sup, in, p2v, dJ);
  return 0;
}

int launch_jgrad_sparse(const JSupport& sup, const float* dJT, const float* VTq, float* dJn, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_jgrad_sparse, dim3(NH, 16), dim3(JGS_THREADS), 0, s, sup, dJT, VTq, dJn, BP);
  return 0;
}
int launch_rejoints_sparse(const JSupport& sup, const float* VTq, float* out, int BP, hipStream_t s, int32_t* step_inc) {
  hipLaunchKernelGGL(k_rejoints_sparse, dim3((BP + 255) / 256, NH), dim3(256), 0, s, sup, VTq, out, BP, step_inc);   // BP is a multiple of 128 only: round UP (the kernel guards b >= BP)
  return 0;
}

// dJ_raw = mask * relu'(J*mask) * (dJn - sum_v(dJn*Jn)) / rowsum      (dJn given as [17][ldn])
__global__ __launch_bounds__(JREG_THREADS) void k_jreg_bwd(const float* __restrict__ J, const float* __restrict__ mask,
                                                            const float* __restrict__ Jn, const float* __restrict__ rowsum,
                                                            const float* __restrict__ dJn, int ldn, float* __restrict__ dJ,
                                                            const int* __restrict__ v2p, JSupport sup, const int* __restrict__ p2v,
                                                            float* __restrict__ dJs) {
  __shared__ float red[JREG_THREADS];
  const int i = blockIdx.x;
  float acc = 0.f;
  // dJn comes in the internal vertex order (row v2p[v] holds column v; NULL = identity)
  for (int v = threadIdx.x; v < V; v += blockDim.x) acc += dJn[(size_t)i * ldn + (v2p ? v2p[v] : v)] * Jn[(size_t)i * V + v];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = JREG_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float dot = red[0], rs = rowsum[i];
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float mk = mask ? mask[(size_t)i * V + v] : 1.f;
    float x = J[(size_t)i * V + v] * mk;
    float g = (x > 0.f) ? (dJn[(size_t)i * ldn + (v2p ? v2p[v] : v)] - dot) / rs * mk : 0.f;
    dJ[(size_t)i * V + v] = g;
  }
  // dJs (nullable): the same gradient on the regressor's support, [17][JSUP_CAP] with zeros behind the row's count -- the payload of
  // the data-parallel all-reduce (jrr_j_regressor_grad_support).  Entry e of the list is internal row col[e]; its vertex is p2v[col[e]].
  if (dJs) {
    const int n = sup.cnt[i];
    for (int e = threadIdx.x; e < JSUP_CAP; e += blockDim.x) {
      float g = 0.f;
      if (e < n) {
        const int row = sup.col[i * JSUP_CAP + e];
        const int vo = p2v ? p2v[row] : row;
        const float mk = mask ? mask[(size_t)i * V + vo] : 1.f;
        const float x = J[(size_t)i * V + vo] * mk;
        g = (x > 0.f) ? (dJn[(size_t)i * ldn + row] - dot) / rs * mk : 0.f;
      }
      dJs[i * JSUP_CAP + e] = g;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The second half of the J step in ONE launch (scripts/optimize.py:312 `J_Regressor_optimizer.step()` + the re-normalisation of the
// next find_joints, scripts/utils.py:87-92): one workgroup per regressor row does what used to be five launches -- torch's Adam on
// the row (k_adam_flat), the engine's copy of the raw parameter, the row sum, the normalised row in all layouts (k_jreg_tiles) and the
// row's support list (k_jreg_support) -- with the arithmetic and the summation orders of those kernels (bit-identical results).  The
// gradient arrives dense (dJ) or on the OLD support lists (dJs, the all-reduced payload of jrr_j_step_apply_support: scattered through
// LDS, zero elsewhere -- what the dense gradient holds there).  The workgroup that finishes LAST increments the step counter (every
// workgroup read it at its start) and publishes the fits-the-lists flag.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(JREG_THREADS) void k_jstep_update(JStepUpdate a) {
  __shared__ float red[JREG_THREADS];
  __shared__ float gl[V];
  __shared__ AdamScalars sc;
  __shared__ int wcount[JREG_THREADS / 64];
  __shared__ int base, islast;
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { sc = adam_scalars(a.step[0] + 1, a.lr, 0.9f, 0.999f, 1e-8f); base = 0; }
  if (a.dJs) {
    for (int v = tid; v < V; v += JREG_THREADS) gl[v] = 0.f;
    __syncthreads();
    if (tid < a.sup.cnt[i]) { const int row = a.sup.col[i * JSUP_CAP + tid]; gl[a.p2v ? a.p2v[row] : row] = a.dJs[i * JSUP_CAP + tid]; }
  }
  __syncthreads();
  const AdamScalars s = sc;
  constexpr int NK = (V + JREG_THREADS - 1) / JREG_THREADS;      // 7 columns per thread
  float x[NK];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int v = tid + k * JREG_THREADS;
    x[k] = 0.f;
    if (v < V) {
      const size_t o = (size_t)i * V + v;
      const float g = a.dJs ? gl[v] : a.dJ[o];
      float mm = a.m[o], vv = a.v[o];
      const float pn = adam_update(a.J[o], g, mm, vv, s);
      a.J[o] = pn; a.m[o] = mm; a.v[o] = vv;
      a.Jraw[o] = pn;
      float mk = 1.f;
      if (a.mask) { mk = a.mask[o]; a.Jmask[o] = mk; }
      x[k] = fmaxf(pn * mk, 0.f);
      acc += x[k];
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int w = JREG_THREADS / 2; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const float rs = red[0];
  if (tid == 0) a.rowsum[i] = rs;
  // the normalised row in every layout (k_jreg_tiles), then its support list in ascending FILE order of the vertices (k_jreg_support)
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int v = tid + k * JREG_THREADS;
    const float val = (v < V) ? x[k] / rs : 0.f;
    int r = 0;
    if (v < V) {
      r = a.v2p ? a.v2p[v] : v;                      // internal row of the vertex
      const int vt = r >> 5, vv = r & 31;
      a.Jn[(size_t)i * V + v] = val;
      a.Jn_vi[(size_t)vt * 1024 + vv * 32 + i] = val;
      a.Jn_q[((size_t)(r >> 2) * 32 + i) * 4 + (r & 3)] = val;
      if (a.r16) a.Jn_iv[(size_t)vt * TB_FLOATS + R16_JN + ((vv >> 4) * 5 + (i >> 2)) * 64 + (i & 3) * 16 + (vv & 15)] = val;
      else a.Jn_iv[(size_t)vt * TB_FLOATS + TB_JN + i * 32 + vv] = val;
    }
    const bool on = val > 0.f;
    const unsigned long long bal = __ballot(on);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int q = 0; q < wave; ++q) off += wcount[q];
    off += __popcll(bal & ((1ull << lane) - 1ull));
    if (on && off < JSUP_CAP) { a.sup.col[i * JSUP_CAP + off] = r; a.sup.val[i * JSUP_CAP + off] = val; }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int q = 0; q < JREG_THREADS / 64; ++q) t += wcount[q]; base += t; }
    __syncthreads();
  }
  if (tid == 0) {
    a.sup.cnt[i] = base < JSUP_CAP ? base : JSUP_CAP;
    if (base > JSUP_CAP) atomicAdd(&a.sync[1], 1);
    __threadfence();
    islast = atomicAdd(&a.sync[0], 1) == (int)gridDim.x - 1;
    if (islast) {
      __threadfence();
      *a.sup.flag = atomicAdd(&a.sync[1], 0) == 0 ? 1 : 0;
      a.step[0] += 1;
      a.sync[0] = 0; a.sync[1] = 0;
    }
  }
}
int launch_jstep_update(const JStepUpdate& a, hipStream_t s) {
  hipLaunchKernelGGL(k_jstep_update, dim3(NH), dim3(JREG_THREADS), 0, s, a);
  launch_jsup_tilemask(a.sup, s);
  return 0;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_lbs_fwd(const Model& m, const float* Jn_vi, const float* FT, const float* AT, float* VPb, float* JP,
                   float* verts, int B, int BP, int nvc, hipStream_t s, long long* probe, const int* vmask, const int* tl, int ntl,
                   int verts_pose_major) {
  dim3 grid((BP / BG) * nvc), block(256);
  if (tl && !(m.kjs && VPb)) { jrr_set_error("lbs_fwd: the tile list serves the joint-sparse kernels with v_posed kept"); return JRR_ERR_ARG; }
  // exactly one round of two workgroups per CU, an even number of chunks and whole pose groups per XCD
  static const int fwd_split = [] { const char* e = getenv("JRR_FWD_SPLIT"); return e ? atoi(e) : 593; }();   // 16 : 11 tiles at B = 4096 (re-swept late in round 4: 0.3613 ms against 0.3633 at 15 : 12 and 0.3630 at 17 : 10)
  const int paired = (grid.x == 512 && (nvc & 1) == 0 && (32 % (nvc / 2)) == 0) ? fwd_split : 0;
#define JRR_LBS_FWD_K(SVP, SVT, KJV, WD, WT)                                                                                      \
  hipLaunchKernelGGL((k_lbs_fwd<SVP, SVT, KJV, WD>), grid, block, 0, s, m.Dk, WT, Jn_vi, FT, AT, VPb, JP, verts, B, BP, nvc, probe,  \
                     paired, m.jl, m.tnj, verts ? vmask : nullptr, nullptr, 0, verts_pose_major)
#define JRR_LBS_FWD_L(SVT, KJV, WD)                                                                                               \
  hipLaunchKernelGGL((k_lbs_fwd<true, SVT, KJV, WD, true>), grid, block, 0, s, m.Dk, m.Wc, Jn_vi, FT, AT, VPb, JP, verts, B, BP, nvc,  \
                     probe, paired, m.jl, m.tnj, nullptr, tl, ntl, verts_pose_major)
#define JRR_LBS_FWD(SVP, SVT)                                                                                                     \
  do {                                                                                                                            \
    if (m.kjs == 8 && !m.wide_tiles) JRR_LBS_FWD_K(SVP, SVT, 8, false, m.Wc);                                                     \
    else if (m.kjs == 8) JRR_LBS_FWD_K(SVP, SVT, 8, true, m.Wc);                                                                  \
    else if (m.kjs == 12 && !m.wide_tiles) JRR_LBS_FWD_K(SVP, SVT, 12, false, m.Wc);                                              \
    else if (m.kjs == 12) JRR_LBS_FWD_K(SVP, SVT, 12, true, m.Wc);                                                                \
    else JRR_LBS_FWD_K(SVP, SVT, 0, false, m.Wjv);                                                                                \
  } while (0)
  if (tl) {      // the listed tiles only (all of them keep v_posed; with `verts`, their vertices)
    if (verts) {
      if (m.kjs == 8 && !m.wide_tiles) JRR_LBS_FWD_L(true, 8, false);
      else if (m.kjs == 8) JRR_LBS_FWD_L(true, 8, true);
      else if (!m.wide_tiles) JRR_LBS_FWD_L(true, 12, false);
      else JRR_LBS_FWD_L(true, 12, true);
    } else {
      if (m.kjs == 8 && !m.wide_tiles) JRR_LBS_FWD_L(false, 8, false);
      else if (m.kjs == 8) JRR_LBS_FWD_L(false, 8, true);
      else if (!m.wide_tiles) JRR_LBS_FWD_L(false, 12, false);
      else JRR_LBS_FWD_L(false, 12, true);
    }
    return 0;
  }
  if (VPb && verts) JRR_LBS_FWD(true, true);
  else if (VPb) JRR_LBS_FWD(true, false);
  else if (verts) JRR_LBS_FWD(false, true);
  else JRR_LBS_FWD(false, false);
#undef JRR_LBS_FWD
#undef JRR_LBS_FWD_K
#undef JRR_LBS_FWD_L
  return 0;
}

int launch_lbs_bwd(const Model& m, const float* Tb, const float* AT, const float* VPb, const float* dJT,
                   const float* dVT, float* DVP, float* dATp, int BP, int nvc, hipStream_t s, const int* tl, int ntl, unsigned* dmask) {
  if (tl && !(m.kjs && m.bwd16 && !dVT)) { jrr_set_error("lbs_bwd: the tile list serves k_lbs_bwd16 without an outside vertex adjoint"); return JRR_ERR_ARG; }
  if (m.kjs && m.bwd16) {                       // four symmetric waves: one workgroup per (64 NG poses, vertex chunk)
    // JRR_BWD16_NG (experiments): 16-pose column groups per wave -- 1 (default): the round-3 shape, two workgroups of 64 poses per CU;
    // 2: one workgroup of 128 poses per CU, one wave per SIMD with 512 registers (round 5: built, parity-green, measured SLOWER --
    // 0.251 vs 0.178 ms at 4096 poses: a lone wave per SIMD does not cover its own LDS / VALU latencies with the compiler's schedule)
    // (NG = 2 is built for the loop's own case -- vertex adjoint from the joints only, no wide tile; with a vertex adjoint from outside or
    // the wide-tile branch the two groups' registers no longer fit the 512 and the compiler spills 8 .. 53 of them: those run NG = 1)
    static const int ng_env = [] { const char* e = getenv("JRR_BWD16_NG"); return (e && e[0] == '2') ? 2 : 1; }();
    const int ng = (ng_env == 2 && !dVT && !m.wide_tiles && m.kjs == 8) ? 2 : 1;
    const int n_bt16 = BP / (64 * ng);
    dim3 grid16(n_bt16 * nvc), block16(256);
    // JRR_BWD16_PAIRED (experiments, NG = 1): share of the first-dispatched workgroup of a CU in thousandths (default 540 since the slab flush
    // reads back in one batch: 0.1777 ms against 0.1797 at 520, 0.1782 at 560; before that 520: round 4, with
    // the non-temporal streams -- 0.1797 ms against 0.1815 at 480, 0.186 at 560; shader-clock stamps: the first-dispatched workgroup's
    // tiles take ~12 300 clocks, its partner's ~14 000 while both run and 7 400 once it is alone), 0 = no
    // pairing, -2 = chunk-major mapping without pairing
    static const int pair_env = [] { const char* e = getenv("JRR_BWD16_PAIRED"); return e ? atoi(e) : 540; }();
    static const int rev16 = [] { const char* e = getenv("JRR_BWD16_REV"); return (e && e[0] == '1') ? 1 : 0; }();
    const int paired16 = ng == 2 ? 0 : (pair_env > 0) ? ((grid16.x == 512 && (nvc & 1) == 0 && (32 % (nvc / 2)) == 0) ? pair_env : 0) : pair_env;
#define JRR_LBS_BWD16_K(DVM, KJV, WD)                                                                                           \
  do {                                                                                                                          \
    if constexpr (DVM == 0 && !WD) {                                                                                            \
      if (ng == 2) {                                                                                                            \
        hipLaunchKernelGGL((k_lbs_bwd16<0, 8, false, false, 2>), grid16, block16, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt16, \
                           m.segid, m.segj, paired16, m.tnj, nullptr, 0, dmask, rev16);                                                \
        break;                                                                                                                  \
      }                                                                                                                         \
    }                                                                                                                           \
    hipLaunchKernelGGL((k_lbs_bwd16<DVM, KJV, WD, false, 1>), grid16, block16, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt16, \
                       m.segid, m.segj, paired16, m.tnj, nullptr, 0, dmask, rev16);                                                    \
  } while (0)
#define JRR_LBS_BWD16_L(KJV, WD)                                                                                                \
  do {                                                                                                                          \
    if constexpr (!WD) {                                                                                                        \
      if (ng == 2) {                                                                                                            \
        hipLaunchKernelGGL((k_lbs_bwd16<0, 8, false, true, 2>), grid16, block16, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt16, \
                           m.segid, m.segj, paired16, m.tnj, tl, ntl, dmask, rev16);                                                   \
        break;                                                                                                                  \
      }                                                                                                                         \
    }                                                                                                                           \
    hipLaunchKernelGGL((k_lbs_bwd16<0, KJV, WD, true, 1>), grid16, block16, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt16, \
                       m.segid, m.segj, paired16, m.tnj, tl, ntl, dmask, rev16);                                                       \
  } while (0)
#define JRR_LBS_BWD16(DVM)                                                                                                      \
  do {                                                                                                                          \
    if (m.kjs == 8 && !m.wide_tiles) JRR_LBS_BWD16_K(DVM, 8, false);                                                            \
    else if (m.kjs == 8) JRR_LBS_BWD16_K(DVM, 8, true);                                                                         \
    else if (!m.wide_tiles) JRR_LBS_BWD16_K(DVM, 12, false);                                                                    \
    else JRR_LBS_BWD16_K(DVM, 12, true);                                                                                        \
  } while (0)
    if (tl) {
      if (m.kjs == 8 && !m.wide_tiles) JRR_LBS_BWD16_L(8, false);
      else if (m.kjs == 8) JRR_LBS_BWD16_L(8, true);
      else if (!m.wide_tiles) JRR_LBS_BWD16_L(12, false);
      else JRR_LBS_BWD16_L(12, true);
      return 0;
    }
    if (dVT && dJT) JRR_LBS_BWD16(2);
    else if (dVT) JRR_LBS_BWD16(1);
    else JRR_LBS_BWD16(0);
#undef JRR_LBS_BWD16
#undef JRR_LBS_BWD16_K
#undef JRR_LBS_BWD16_L
    return 0;
  }
  const int n_bt = BP / BT;                     // one workgroup per (pose tile, vertex chunk)
  dim3 grid(n_bt * nvc), block(256);
  const int role_kjs = m.role_kjs;              // (the role kernel has no per-tile classes: one wide tile sends it to its dense form)
#define JRR_LBS_BWD(DVM)                                                                                                        \
  do {                                                                                                                          \
    if (role_kjs == 8)                                                                                                          \
      hipLaunchKernelGGL((k_lbs_bwd<DVM, 8>), grid, block, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt, m.jl, m.segid, m.segj);        \
    else if (role_kjs == 12)                                                                                                    \
      hipLaunchKernelGGL((k_lbs_bwd<DVM, 12>), grid, block, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt, m.jl, m.segid, m.segj);       \
    else                                                                                                                        \
      hipLaunchKernelGGL((k_lbs_bwd<DVM, 0>), grid, block, 0, s, Tb, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, n_bt, m.jl, m.segid, m.segj);        \
  } while (0)
  if (dVT && dJT) JRR_LBS_BWD(2);
  else if (dVT) JRR_LBS_BWD(1);
  else JRR_LBS_BWD(0);
#undef JRR_LBS_BWD
  return 0;
}

int launch_dverts_transpose(const float* dverts, int ldv, float* dVT, int B, int BP, hipStream_t s, const int* p2v) {
  hipLaunchKernelGGL(k_dverts_transpose, dim3(VT, BP / 32), dim3(256), 0, s, dverts, ldv, dVT, B, BP, p2v);
  return 0;
}

int launch_jreg_normalize(const float* J, const float* mask, float* rowsum, float* Jn, float* Jn_vi, float* Jn_iv,
                          float* Jn_q, const int* p2v, hipStream_t s, int r16, const int* v2p, const JSupport* sup, int32_t* step_inc) {
  hipLaunchKernelGGL(k_jreg_rowsum, dim3(NH), dim3(JREG_THREADS), 0, s, J, mask, rowsum, sup ? sup->flag : nullptr, step_inc);
  hipLaunchKernelGGL(k_jreg_tiles, dim3(VT * 1024 / 256), dim3(256), 0, s, J, mask, rowsum, Jn, Jn_vi, Jn_iv, Jn_q, p2v, r16);
  if (sup) {
    hipLaunchKernelGGL(k_jreg_support, dim3(NH), dim3(JREG_THREADS), 0, s, Jn, v2p, *sup);
    launch_jsup_tilemask(*sup, s);
  }
  return 0;
}

int launch_bwd_tab_static(const Model& m, float* Tb, hipStream_t s) {
  const bool r16 = m.kjs && m.bwd16;
  hipLaunchKernelGGL(k_bwd_tab_static, dim3(VT * 2048 / 256), dim3(256), 0, s, m.Wjv, m.Wvj, Tb, (r16 || m.role_kjs) ? m.Wc : nullptr,
                     r16 ? m.kjs : m.role_kjs, m.W16, m.jl, r16 ? 1 : 0);
  return 0;
}

int launch_jreg_bwd(const float* J, const float* mask, const float* Jn, const float* rowsum, const float* dJn, int ldn,
                    float* dJ, const int* v2p, hipStream_t s, const JSupport* sup, const int* p2v, float* dJs) {
  const JSupport none{nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(k_jreg_bwd, dim3(NH), dim3(JREG_THREADS), 0, s, J, mask, Jn, rowsum, dJn, ldn, dJ, v2p, sup ? *sup : none, p2v,
                     sup ? dJs : nullptr);
  return 0;
}

}  // namespace jrr
