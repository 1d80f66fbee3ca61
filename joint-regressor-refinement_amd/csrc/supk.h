// The joint-loss iteration on the regressor's SUPPORT VERTICES, one workgroup per 32-pose group (round 6).
//
// scripts/optimize.py:220-265 without the silhouette term reads the SMPL vertices through J_regressor x V only
// (scripts/utils.py:87-98), and the H36M regressor is positive in a few dozen of its 117 130 entries: the joints depend on
// those vertices alone and every other vertex receives a zero adjoint.  The tile-restricted iteration of round 4
// (JRR_FLAG_SUPPORT_TILES) still ran the three large LBS kernels -- over 6 tiles of 32 vertices for 58 support vertices -- as
// separate launches with the pose group's v_posed / dvp / partial slabs travelling through HBM between them.  Here the restriction
// is per VERTEX (n_sv <= 64 of them = 192 coordinate rows) and a 32-pose group never leaves its workgroup:
//
//   v_posed  [192 rows][32 poses] = Ds . F          6 row tiles x 112 v_mfma_f32_32x32x2_f32; A operand = the basis rows of the
//                                                   support vertices, gathered once per support (k_sup_gather), read 16 B per lane
//                                                   straight from L2; B operand = the pose group's feature quads in LDS
//   T_v = sum_j W[v,j] A_j , verts = T v_posed      vector ALU, one thread per (pose, vertex): <= 4 joints per SMPL vertex
//   joints = Jn[:, support] verts, loss, dj         one thread per (pose, H36M joint)            (scripts/utils.py:106-114)
//   dverts = Jn^T dj , dvp = T^T dverts             one thread per (pose, vertex)
//   dA_j  += W[v,j] dverts (x) [v_posed; 1]         one thread per (pose, SMPL joint), over the joint's vertex list (no atomics)
//   dF [224][32] = Ds^T . dvp                       7 row tiles x 96 matrix instructions
//
// Everything is summed in a fixed order (bitwise reproducible).  Inputs F^T (K-quads) / A^T and outputs dA^T / dF^T are the
// arrays k_prep_fwd writes and k_chain_bwd reads: the body composes with them inside one launch (prep.hip, k_sup_step) or runs
// between them as a launch of its own (sup.hip, k_sup_iter).
#pragma once
#include "jrr_common.h"
#include "kernels.h"
#include "proj.h"
#include "dconv.h"

namespace jrr {

constexpr int SUP_ROWS = 3 * SUP_NSV;        // 192 coordinate rows (row 3 s + c = coordinate c of support vertex s)
constexpr int SUP_RT = SUP_ROWS / 32;        // 6 row tiles of the forward product
constexpr int SUP_FG = KFP / 8;              // 28 K groups (8 features: four matrix instructions) of the forward product
constexpr int SUP_BG = SUP_ROWS / 8;         // 24 K groups of the adjoint product
constexpr int SUP_MT = KFP / 32;             // 7 row tiles of the adjoint product
constexpr int SUP_PP = 32;                   // poses per workgroup
constexpr int SUP_THREADS = SUP_PP * NJ;     // 768: (pose, joint) threads, as k_prep_fwd / k_chain_bwd
constexpr int SUP_JS = SUP_NSV;              // row pitch of the regressor's support columns in LDS
constexpr int SUP_DSF_FLOATS = SUP_RT * SUP_FG * 256, SUP_DSB_FLOATS = SUP_MT * SUP_BG * 256;

// LDS image (floats).  Fq is dead after the forward product and then holds dverts; the vertices are dead after the joints and
// then hold dvp in row quads (the B operand of the adjoint product).
constexpr int SUPL_FQ = 0;                              // [56 quads][32 poses][4]      -> dverts [192][32]
constexpr int SUPL_AL = SUPL_FQ + (KFP / 4) * 32 * 4;   // [288][32]   A^T of the pose group
constexpr int SUPL_VP = SUPL_AL + 12 * NJ * 32;         // [192][32]   v_posed
constexpr int SUPL_V = SUPL_VP + SUP_ROWS * 32;         // [192][32]   vertices                   -> dvp [48 quads][32][4]
constexpr int SUPL_JN = SUPL_V + SUP_ROWS * 32;         // [17][64]    regressor columns of the support vertices
constexpr int SUPL_RED = SUPL_JN + NH * SUP_JS;         // [17][8][32] per-joint loss partials: 0..2 g, 3 err, 4 e2d, 5..7 gcam
constexpr int SUPL_PEL = SUPL_RED + NH * 8 * 32;        // [3][32]
constexpr int SUPL_DJ = SUPL_PEL + 3 * 32;              // [3][17][32] joint adjoint
constexpr int SUPL_MASK = SUPL_DJ + 3 * NH * 32;        // [17][2] row masks + [64] column masks of the regressor's non-zeros (32-bit words)
constexpr int SUPL_CONV = SUPL_MASK + 2 * 32 + SUP_NSV;  // LDS image of the pose discriminator's per-joint MLP (dconv.h), staged with the operands
constexpr int SUPL_FLOATS = SUPL_CONV + CL_FLOATS;
static_assert(SUPL_FLOATS * 4 <= 160 * 1024, "LDS image of the support iteration");

struct SupArgs {
  SupTables t; int nsv;
  const float* Jn_vi;        // normalised regressor [VP][32 i] (tile-major [VT][32 v][32 i]): live values, the J step rewrites them
  const float* FTq;          // [KFP/4][BP][4]
  const float* AT;           // [288][BP]
  const float* gt_mm;        // (B,17,3) millimetres, pelvis-centred by the caller
  float scale;               // 2 * weight / (batch_norm * 51)
  float* joints_out;         // (B,17,3), nullable
  float* sqerr;              // (B), nullable
  float* dA;                 // [288][BP] out
  float* dF;                 // [224][BP] out
  int B, BP;
  Reproj rp;                 // 2-D reprojection term on the un-centred joints (scripts/optimize.py:231-233); gt_j2d == NULL: off
  // The pose discriminator's per-joint MLP adjoint (dconv.h) closes the body: its LDS image is staged with the operands.
  // conv_img == NULL: no discriminator / the caller runs the adjoint itself.
  const float* conv_img; const float* x6d; const float* dH2T; float dscale; float* gx; float* dsq;
};

// stamps (nullable; experiments): phase boundaries on the 100 MHz counter, written by thread 0
__device__ __forceinline__ void sup_body(float* __restrict__ lds, int blk, const SupArgs& a, long long* stamps = nullptr) {
  auto stamp = [&](int i) { if (stamps && threadIdx.x == 0) stamps[i] = wall_clock64(); };
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = tid & 31, jt = tid >> 5;               // pose column; SMPL joint / vertex slot / H36M joint of this thread
  const int b0 = blk * SUP_PP, b = b0 + p;
  const bool ok = b < a.B;
  const int nsv = a.nsv, BP = a.BP;
  float* const Fq = lds + SUPL_FQ;
  float* const Al = lds + SUPL_AL;
  float* const vpL = lds + SUPL_VP;
  float* const vL = lds + SUPL_V;
  float* const JnS = lds + SUPL_JN;
  float* const red = lds + SUPL_RED;
  float* const pel = lds + SUPL_PEL;
  float* const djL = lds + SUPL_DJ;
  unsigned* const rowm = reinterpret_cast<unsigned*>(lds + SUPL_MASK);      // rowm[2 i], rowm[2 i + 1]: support vertices 0..31 / 32..63 of joint i
  unsigned* const colm = rowm + 2 * 32;                                     // colm[s]: the H36M joints that read support vertex s

  // ---- the pose group's operands -> LDS ----
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.FTq);
    f32x4* dst = reinterpret_cast<f32x4*>(Fq);
    for (int i = tid; i < (KFP / 4) * 32; i += SUP_THREADS) dst[i] = src[(size_t)(i >> 5) * BP + b0 + (i & 31)];
  }
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.AT);
    f32x4* dst = reinterpret_cast<f32x4*>(Al);
    for (int i = tid; i < 12 * NJ * 8; i += SUP_THREADS) dst[i] = src[((size_t)(i >> 3) * BP + b0) / 4 + (i & 7)];
  }
  for (int i = tid; i < NH * SUP_JS; i += SUP_THREADS) {
    const int ii = i / SUP_JS, s = i % SUP_JS;
    JnS[i] = s < nsv ? a.Jn_vi[(size_t)a.t.rows[s] * 32 + ii] : 0.f;
  }
  float* const convL = lds + SUPL_CONV;
  if (a.conv_img) conv_stage_params(a.conv_img, convL);
  __syncthreads();
  stamp(0);

  // ---- v_posed = Ds . F : wave w < 6 owns row tile w; two accumulator chains (even / odd K groups), added at the end ----
  if (wv < SUP_RT) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a.t.Dsf) + (size_t)wv * SUP_FG * 64 + lane;
    const f32x4* b4 = reinterpret_cast<const f32x4*>(Fq) + half * 32 + l31;      // quad 2 g + half of K group g
    f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
    for (int g = 0; g < SUP_FG; g += 2) {
      const f32x4 x0 = a4[g * 64], x1 = a4[(g + 1) * 64];
      const f32x4 y0 = b4[(2 * g) * 32], y1 = b4[(2 * g + 2) * 32];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = mfma(x0[t], y0[t], acc0);
        acc1 = mfma(x1[t], y1[t], acc1);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) vpL[(32 * wv + acc_row(r, half)) * 32 + l31] = acc0[r] + acc1[r];
  } else {
    // the waves without a row tile: where the regressor's support columns are non-zero (62 of 17 x 58 entries with the H36M
    // regressor), as bit masks -- the joints and the vertex adjoint then visit those entries only, in the same ascending order
    for (int i = wv - SUP_RT; i < NH; i += SUP_THREADS / 64 - SUP_RT) {
      const unsigned long long m = __ballot(JnS[i * SUP_JS + lane] != 0.f);
      if (lane == 0) { rowm[2 * i] = (unsigned)m; rowm[2 * i + 1] = (unsigned)(m >> 32); }
    }
    if (wv == SUP_THREADS / 64 - 1) {
      unsigned m = 0u;
      for (int i = 0; i < NH; ++i) m |= (JnS[i * SUP_JS + lane] != 0.f ? 1u : 0u) << i;
      colm[lane] = m;
    }
  }
  __syncthreads();
  stamp(1);

  // ---- skinning: T = sum_j W[v,j] A_j over the vertex's own joints, verts = T [v_posed; 1] ----
  for (int s = jt; s < nsv; s += NJ) {
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    const int cnt = a.t.sk_cnt[s];
    for (int k = 0; k < cnt; ++k) {
      const int j = a.t.sk_j[s * NJ + k];
      const float w = a.t.sk_w[s * NJ + k];
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] = fmaf(w, Al[(e * NJ + j) * 32 + p], T[e]);
    }
    const float vx = vpL[(3 * s) * 32 + p], vy = vpL[(3 * s + 1) * 32 + p], vz = vpL[(3 * s + 2) * 32 + p];
#pragma unroll
    for (int r = 0; r < 3; ++r) vL[(3 * s + r) * 32 + p] = fmaf(T[r * 4 + 2], vz, fmaf(T[r * 4 + 1], vy, fmaf(T[r * 4], vx, T[r * 4 + 3])));
  }
  __syncthreads();
  stamp(2);

  // ---- joints (ascending vertex row), pelvis-centred squared error and its adjoint (k_joints_loss, prep.hip) ----
  float j3[3] = {0.f, 0.f, 0.f}, g3[3] = {0.f, 0.f, 0.f};
  if (jt < NH) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned m = rowm[2 * jt + h];
      while (m) {
        const int s = 32 * h + __builtin_ctz(m);
        m &= m - 1u;
        const float w = JnS[jt * SUP_JS + s];
#pragma unroll
        for (int c = 0; c < 3; ++c) j3[c] = fmaf(w, vL[(3 * s + c) * 32 + p], j3[c]);
      }
    }
    if (a.joints_out && ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.joints_out[((size_t)b * NH + jt) * 3 + c] = j3[c];
    }
    if (jt == 0) { pel[p] = j3[0]; pel[32 + p] = j3[1]; pel[64 + p] = j3[2]; }
  }
  __syncthreads();
  float g2[3] = {0.f, 0.f, 0.f};      // adjoint of the 2-D term w.r.t. this joint
  if (jt < NH) {
    float err = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float gtv = ok ? a.gt_mm[((size_t)b * NH + jt) * 3 + c] / 1000.f : 0.f;
      const float d = (jt == 0) ? -gtv : (j3[c] - pel[c * 32 + p]) - gtv;      // joint 0 is identically 0 after centring
      err += d * d;
      g3[c] = (jt == 0) ? 0.f : a.scale * d;
      red[(jt * 8 + c) * 32 + p] = g3[c];
    }
    float e2 = 0.f, gc[3] = {0.f, 0.f, 0.f};
    if (a.rp.gt_j2d && ok) {      // k_joints_loss's statements (prep.hip)
      float t[3] = {a.rp.cam[(size_t)b * 3], a.rp.cam[(size_t)b * 3 + 1], a.rp.cam[(size_t)b * 3 + 2]};
      float xs, ys, invZ, X, Y;
      project_point(j3, t, xs, ys, invZ, X, Y);
      const float dx = xs - a.rp.gt_j2d[((size_t)b * NH + jt) * 2], dy = ys - a.rp.gt_j2d[((size_t)b * NH + jt) * 2 + 1];
      e2 = dx * dx + dy * dy;
      project_point_bwd(a.rp.scale2d * dx, a.rp.scale2d * dy, invZ, X, Y, g2, gc);
    }
    red[(jt * 8 + 3) * 32 + p] = err;
    red[(jt * 8 + 4) * 32 + p] = e2;
#pragma unroll
    for (int c = 0; c < 3; ++c) red[(jt * 8 + 5 + c) * 32 + p] = gc[c];
  }
  __syncthreads();
  if (jt < NH) {
    if (jt == 0) {      // fixed-order sums over the 17 joints
      float sm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1      // (fully unrolled, its 136 LDS reads in flight cost the composed kernel 22 spilled registers)
      for (int q = 0; q < NH; ++q)
#pragma unroll
        for (int k = 0; k < 8; ++k) sm[k] += red[(q * 8 + k) * 32 + p];
      if (ok && a.sqerr) a.sqerr[b] = sm[3];
      if (ok && a.rp.gt_j2d) {
        if (a.rp.sq2d) a.rp.sq2d[b] = sm[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) a.rp.gcam[(size_t)b * 3 + c] = sm[5 + c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) g3[c] = -sm[c];      // move_pelvis adjoint: -sum_i g_i
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) djL[(c * NH + jt) * 32 + p] = ok ? g3[c] + g2[c] : 0.f;
  }
  __syncthreads();
  stamp(3);

  // ---- vertex adjoint dverts = Jn^T dj, dvp = T^T dverts (T recomputed) ----
  float* const dvL = Fq;       // [192][32]
  float* const dvq = vL;       // row quads [48][32][4]
  for (int s = jt; s < nsv; s += NJ) {
    float dv[3] = {0.f, 0.f, 0.f};
    for (unsigned m = colm[s]; m; m &= m - 1u) {
      const int i = __builtin_ctz(m);
      const float w = JnS[i * SUP_JS + s];
#pragma unroll
      for (int r = 0; r < 3; ++r) dv[r] = fmaf(w, djL[(r * NH + i) * 32 + p], dv[r]);
    }
    float T[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) T[e] = 0.f;
    const int cnt = a.t.sk_cnt[s];
    for (int k = 0; k < cnt; ++k) {
      const int j = a.t.sk_j[s * NJ + k];
      const float w = a.t.sk_w[s * NJ + k];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r * 3 + c] = fmaf(w, Al[((r * 4 + c) * NJ + j) * 32 + p], T[r * 3 + c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int rho = 3 * s + c;
      dvq[((rho >> 2) * 32 + p) * 4 + (rho & 3)] = fmaf(T[6 + c], dv[2], fmaf(T[3 + c], dv[1], T[c] * dv[0]));
      dvL[rho * 32 + p] = dv[c];
    }
  }
  for (int i = tid; i < (SUP_ROWS - 3 * nsv) * 32; i += SUP_THREADS) {      // padding rows of the adjoint product's K range
    const int rho = 3 * nsv + (i >> 5);
    dvq[((rho >> 2) * 32 + (i & 31)) * 4 + (rho & 3)] = 0.f;
  }
  __syncthreads();
  stamp(4);

  // ---- dA_j = sum_v W[v,j] dverts_v (x) [v_posed_v; 1]: one thread per (pose, joint), the joint's vertices in ascending order ----
  {
    float acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 0.f;
    const int cnt = a.t.jl_cnt[jt];
    for (int k = 0; k < cnt; ++k) {
      const int s = a.t.jl_s[jt * SUP_NSV + k];
      const float w = a.t.jl_w[jt * SUP_NSV + k];
      const float vx = vpL[(3 * s) * 32 + p], vy = vpL[(3 * s + 1) * 32 + p], vz = vpL[(3 * s + 2) * 32 + p];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float wd = w * dvL[(3 * s + r) * 32 + p];
        acc[r * 4] = fmaf(wd, vx, acc[r * 4]);
        acc[r * 4 + 1] = fmaf(wd, vy, acc[r * 4 + 1]);
        acc[r * 4 + 2] = fmaf(wd, vz, acc[r * 4 + 2]);
        acc[r * 4 + 3] += wd;
      }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) a.dA[(size_t)(e * NJ + jt) * BP + b] = acc[e];
  }
  stamp(5);

  // ---- dF = Ds^T . dvp : wave w < 7 owns feature tile w ----
  if (wv < SUP_MT) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a.t.Dsb) + (size_t)wv * SUP_BG * 64 + lane;
    const f32x4* b4 = reinterpret_cast<const f32x4*>(dvq) + half * 32 + l31;
    f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
    for (int g = 0; g < SUP_BG; g += 2) {
      const f32x4 x0 = a4[g * 64], x1 = a4[(g + 1) * 64];
      const f32x4 y0 = b4[(2 * g) * 32], y1 = b4[(2 * g + 2) * 32];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = mfma(x0[t], y0[t], acc0);
        acc1 = mfma(x1[t], y1[t], acc1);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a.dF[(size_t)(32 * wv + acc_row(r, half)) * BP + b0 + l31] = acc0[r] + acc1[r];
  }
  // ---- the pose discriminator's per-joint MLP adjoint: wave w takes joints w and w + 12 as ONE interleaved pair (dconv.h).  (Measured
  //      first, and dropped: the same tiles on the waves that own no row tile during the two matrix phases -- a tile is a 7 us dependent
  //      chain for a lone wave, so two tiles per idle wave doubled those phases: 63.6 -> 68.9 us for the kernel.) ----
  if (a.conv_img) dconv_bwd_pair(convL, blk, wv, wv + 12, a.x6d, a.dH2T, a.dscale, 1.f, a.gx, a.B, BP, a.dsq);
}

}  // namespace jrr
