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
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

// ------------------------------------------------------------------------------------------
// forward
//   grid.x = (BP/128) * nvc workgroups of 256 threads; wave w owns poses [bg*128 + 32w, +32)
//   and loops over the vertex tiles of chunk vc.  The blend-basis tile is staged through LDS in
//   7 chunks of 32 feature rows (12 KB), prefetched into registers one chunk ahead.
//   outputs: JP [nvc][3][17][BP] joint partials; optional VPb [3][VP][BP] (v_posed, kept for
//   the backward pass) and verts (B,6890,3).
// ------------------------------------------------------------------------------------------
constexpr int DCH_FLOATS = KCH * 96;          // 3072 floats per staged basis chunk
constexpr int W_FLOATS = NJ * 32;             // 768
constexpr int JN_FLOATS = 32 * 32;            // 1024

template <bool STORE_VP, bool STORE_VERTS>
__global__ __launch_bounds__(256, 2) void k_lbs_fwd(const float* __restrict__ Dk, const float* __restrict__ Wjv,
                                                    const float* __restrict__ Jn_vi, const float* __restrict__ FT,
                                                    const float* __restrict__ AT, float* __restrict__ VPb,
                                                    float* __restrict__ JP, float* __restrict__ verts, int ldv, int B,
                                                    int BP, int nvc) {
  __shared__ float lds[DCH_FLOATS + W_FLOATS + JN_FLOATS];
  float* ldsD = lds;
  float* ldsW = lds + DCH_FLOATS;
  float* ldsJ = ldsW + W_FLOATS;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int bg = L / nvc, vc = L % nvc;
  const int b0 = bg * BG + wave * BT;
  const int t_begin = (int)((long)VT * vc / nvc), t_end = (int)((long)VT * (vc + 1) / nvc);
  const size_t bcol = (size_t)b0 + l31;

  f32x16 jacc[3] = {zero16(), zero16(), zero16()};

  // register prefetch buffers for the staged data
  f32x4 preD[3], preJ, preW;
  auto prefetch = [&](int vt, int kc) {
    const f32x4* src = reinterpret_cast<const f32x4*>(Dk + ((size_t)vt * KFP + kc * KCH) * 96);
#pragma unroll
    for (int i = 0; i < 3; ++i) preD[i] = src[tid + 256 * i];
    if (kc == 0) {
      preJ = reinterpret_cast<const f32x4*>(Jn_vi + (size_t)vt * JN_FLOATS)[tid];
      if (tid < W_FLOATS / 4) preW = reinterpret_cast<const f32x4*>(Wjv + (size_t)vt * W_FLOATS)[tid];
    }
  };
  auto commit = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 3; ++i) reinterpret_cast<f32x4*>(ldsD)[tid + 256 * i] = preD[i];
    if (kc == 0) {
      reinterpret_cast<f32x4*>(ldsJ)[tid] = preJ;
      if (tid < W_FLOATS / 4) reinterpret_cast<f32x4*>(ldsW)[tid] = preW;
    }
  };

  if (t_begin < t_end) prefetch(t_begin, 0);
  for (int vt = t_begin; vt < t_end; ++vt) {
    f32x16 vp[3] = {zero16(), zero16(), zero16()};
    for (int kc = 0; kc < NKCH; ++kc) {
      __syncthreads();          // previous stage fully consumed
      commit(kc);
      __syncthreads();
      if (kc + 1 < NKCH) prefetch(vt, kc + 1);
      else if (vt + 1 < t_end) prefetch(vt + 1, 0);
      const int npairs = (kc == NKCH - 1) ? (KF - (NKCH - 1) * KCH) / 2 : KCH / 2;   // 13 : 16
      const float* fp = FT + (size_t)(kc * KCH + half) * BP + bcol;
      const float* dp = ldsD + half * 96 + l31;
#pragma unroll 4
      for (int kk = 0; kk < npairs; ++kk) {
        float f = fp[(size_t)(2 * kk) * BP];
        float d0 = dp[(2 * kk) * 96], d1 = dp[(2 * kk) * 96 + 32], d2 = dp[(2 * kk) * 96 + 64];
        vp[0] = mfma(d0, f, vp[0]);
        vp[1] = mfma(d1, f, vp[1]);
        vp[2] = mfma(d2, f, vp[2]);
      }
    }
    if (STORE_VP) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          VPb[((size_t)c * VP + vt * 32 + acc_row(r, half)) * BP + bcol] = vp[c][r];
    }
    // skinning transforms and vertices, one output coordinate r at a time.
    // The A^T operands do not depend on the vertex tile; launder the pointer so the compiler
    // re-loads them from L1/L2 per tile instead of hoisting 144 registers out of the loop
    // (which it then spills to scratch).
    int at_off = 0;
    asm volatile("" : "+s"(at_off));
    const float* ATl = AT + at_off;
    float w[12];
#pragma unroll
    for (int jp = 0; jp < 12; ++jp) w[jp] = ldsW[(2 * jp + half) * 32 + l31];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // translation column first: vr = T_{r,3}
      f32x16 vr = zero16();
      {
        const float* ap = ATl + (size_t)((r * 4 + 3) * NJ + half) * BP + bcol;
#pragma unroll
        for (int jp = 0; jp < 12; ++jp) vr = mfma(w[jp], ap[(size_t)(2 * jp) * BP], vr);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        f32x16 T = zero16();
        const float* ap = ATl + (size_t)((r * 4 + c) * NJ + half) * BP + bcol;
#pragma unroll
        for (int jp = 0; jp < 12; ++jp) T = mfma(w[jp], ap[(size_t)(2 * jp) * BP], T);
        vr += T * vp[c];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STORE_VERTS) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          int v = vt * 32 + acc_row(q, half);
          if (v < V && b0 + l31 < B) verts[(size_t)(b0 + l31) * ldv + v * 3 + r] = vr[q];
        }
      }
      // joints^T[i, b] += sum_v Jn[i, v] verts_r[v, b]
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float jn = ldsJ[acc_row(q, half) * 32 + l31];
        jacc[r] = mfma(jn, vr[q], jacc[r]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      int i = acc_row(q, half);
      if (i < NH) JP[((size_t)(vc * 3 + r) * NH + i) * BP + bcol] = jacc[r][q];
    }
}

// ------------------------------------------------------------------------------------------
// backward (to v_posed and to the skinning transforms)
//   one independent wave per (pose tile bt, coordinate plane c in {0,1,2}, vertex chunk vc):
//     dverts_r[v,b] = sum_i Jn[i,v] dj[b,i,r]                (K = 18)        -- or loaded (DVERTS_MEM)
//     T_{r,c}[v,b]  = sum_j W[v,j] A[b,j,r,c]                (K = 24, recomputed)
//     dvp_c[v,b]    = sum_r T_{r,c} dverts_r                  -> DVP [3][VP][BP]
//     dA_{r,c}[j,b] += sum_v W[v,j] dverts_r[v,b] vp_c[v,b]   (sums over the tile's ROW index)
//     dA_{c,3}[j,b] += sum_v W[v,j] dverts_c[v,b]
//   No LDS and no barriers: all operands are tiny per-tile tables read straight from L2.
//   outputs: DVP, dATp [nvc][12][24][BP] partials.
// ------------------------------------------------------------------------------------------
template <bool DVERTS_MEM>
__global__ __launch_bounds__(256, 2) void k_lbs_bwd(const float* __restrict__ Wjv, const float* __restrict__ Wvj,
                                                    const float* __restrict__ Jn_iv, const float* __restrict__ AT,
                                                    const float* __restrict__ VPb, const float* __restrict__ dJT,
                                                    const float* __restrict__ dVT, float* __restrict__ DVP,
                                                    float* __restrict__ dATp, int BP, int nvc, int nitems) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int item = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (item >= nitems) return;
  const int c = item % 3;
  const int rest = item / 3;
  const int vc = rest % nvc, bt = rest / nvc;
  const int b0 = bt * BT;
  const size_t bcol = (size_t)b0 + l31;
  const int t_begin = (int)((long)VT * vc / nvc), t_end = (int)((long)VT * (vc + 1) / nvc);

  float dj[3][9];
  if (!DVERTS_MEM) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int ip = 0; ip < 9; ++ip) dj[r][ip] = dJT[(size_t)(r * NHP + 2 * ip + half) * BP + bcol];
  }
  float a_rc[3][12];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int jp = 0; jp < 12; ++jp) a_rc[r][jp] = AT[(size_t)((r * 4 + c) * NJ + 2 * jp + half) * BP + bcol];

  f32x16 dA[3] = {zero16(), zero16(), zero16()};
  f32x16 dA3 = zero16();

  for (int vt = t_begin; vt < t_end; ++vt) {
    f32x16 dv[3];
    if (DVERTS_MEM) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) dv[r][q] = dVT[((size_t)r * VP + vt * 32 + acc_row(q, half)) * BP + bcol];
    } else {
      dv[0] = zero16(); dv[1] = zero16(); dv[2] = zero16();
      const float* jp_ = Jn_iv + (size_t)vt * NHP * 32 + half * 32 + l31;
#pragma unroll
      for (int ip = 0; ip < 9; ++ip) {
        float jn = jp_[(2 * ip) * 32];
#pragma unroll
        for (int r = 0; r < 3; ++r) dv[r] = mfma(jn, dj[r][ip], dv[r]);
      }
    }
    f32x16 vp;
#pragma unroll
    for (int q = 0; q < 16; ++q) vp[q] = VPb[((size_t)c * VP + vt * 32 + acc_row(q, half)) * BP + bcol];

    f32x16 dvp = zero16();
    {
      float w[12];
      const float* wp = Wjv + (size_t)vt * W_FLOATS + half * 32 + l31;
#pragma unroll
      for (int jp = 0; jp < 12; ++jp) w[jp] = wp[(2 * jp) * 32];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        f32x16 T = zero16();
#pragma unroll
        for (int jp = 0; jp < 12; ++jp) T = mfma(w[jp], a_rc[r][jp], T);
        dvp += T * dv[r];
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) DVP[((size_t)c * VP + vt * 32 + acc_row(q, half)) * BP + bcol] = dvp[q];

    const float* wvp = Wvj + (size_t)vt * 1024 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float wv = wvp[acc_row(q, half) * 32];
#pragma unroll
      for (int r = 0; r < 3; ++r) dA[r] = mfma(wv, dv[r][q] * vp[q], dA[r]);
      float dvc = (c == 0) ? dv[0][q] : (c == 1) ? dv[1][q] : dv[2][q];
      dA3 = mfma(wv, dvc, dA3);
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    int j = acc_row(q, half);
    if (j < NJ) {
#pragma unroll
      for (int r = 0; r < 3; ++r) dATp[((size_t)(vc * 12 + r * 4 + c) * NJ + j) * BP + bcol] = dA[r][q];
      dATp[((size_t)(vc * 12 + c * 4 + 3) * NJ + j) * BP + bcol] = dA3[q];
    }
  }
}

// (B,6890,3) -> [3][VP][BP] transpose of an external vertex adjoint (operator-level SMPL backward)
__global__ void k_dverts_transpose(const float* __restrict__ dverts, float* __restrict__ dVT, int B, int BP) {
  __shared__ float tile[32][97];
  const int v0 = blockIdx.x * 32, bb0 = blockIdx.y * 32;
  for (int idx = threadIdx.x; idx < 32 * 96; idx += blockDim.x) {
    int bl = idx / 96, rem = idx % 96;   // rem = vv*3 + r
    int b = bb0 + bl, v = v0 + rem / 3;
    float val = 0.f;
    if (b < B && v < V) val = dverts[((size_t)b * V + v0) * 3 + rem];
    tile[bl][rem] = val;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 96 * 32; idx += blockDim.x) {
    int rem = idx / 32, bl = idx % 32;
    int vv = rem / 3, r = rem % 3;
    dVT[((size_t)r * VP + v0 + vv) * BP + bb0 + bl] = tile[bl][rem];
  }
}

// ------------------------------------------------------------------------------------------
// J_regressor normalisation (scripts/utils.py:87-92) into the tile layouts, and its adjoint
// ------------------------------------------------------------------------------------------
__global__ void k_jreg_rowsum(const float* __restrict__ J, const float* __restrict__ mask, float* __restrict__ rowsum) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  float acc = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float x = J[(size_t)i * V + v];
    if (mask) x *= mask[(size_t)i * V + v];
    acc += fmaxf(x, 0.f);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowsum[i] = red[0];
}

__global__ void k_jreg_tiles(const float* __restrict__ J, const float* __restrict__ mask,
                             const float* __restrict__ rowsum, float* __restrict__ Jn, float* __restrict__ Jn_vi,
                             float* __restrict__ Jn_iv) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over VT*32*32 (tile, vv, i)
  if (idx >= VT * 1024) return;
  const int vt = idx >> 10, vv = (idx >> 5) & 31, i = idx & 31;
  const int v = vt * 32 + vv;
  float val = 0.f;
  if (i < NH && v < V) {
    float x = J[(size_t)i * V + v];
    if (mask) x *= mask[(size_t)i * V + v];
    val = fmaxf(x, 0.f) / rowsum[i];
    Jn[(size_t)i * V + v] = val;
  }
  Jn_vi[(size_t)vt * 1024 + vv * 32 + i] = val;
  if (i < NHP) Jn_iv[(size_t)vt * NHP * 32 + i * 32 + vv] = val;
}

// dJ_raw = mask * relu'(J*mask) * (dJn - sum_v(dJn*Jn)) / rowsum      (dJn given as [17][ldn])
__global__ void k_jreg_bwd(const float* __restrict__ J, const float* __restrict__ mask, const float* __restrict__ Jn,
                           const float* __restrict__ rowsum, const float* __restrict__ dJn, int ldn,
                           float* __restrict__ dJ) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  float acc = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) acc += dJn[(size_t)i * ldn + v] * Jn[(size_t)i * V + v];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float dot = red[0], rs = rowsum[i];
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float mk = mask ? mask[(size_t)i * V + v] : 1.f;
    float x = J[(size_t)i * V + v] * mk;
    float g = (x > 0.f) ? (dJn[(size_t)i * ldn + v] - dot) / rs * mk : 0.f;
    dJ[(size_t)i * V + v] = g;
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
int launch_lbs_fwd(const Model& m, const float* Jn_vi, const float* FT, const float* AT, float* VPb, float* JP,
                   float* verts, int ldv, int B, int BP, int nvc, hipStream_t s) {
  dim3 grid((BP / BG) * nvc), block(256);
  if (VPb && verts)
    hipLaunchKernelGGL((k_lbs_fwd<true, true>), grid, block, 0, s, m.Dk, m.Wjv, Jn_vi, FT, AT, VPb, JP, verts, ldv, B, BP, nvc);
  else if (VPb)
    hipLaunchKernelGGL((k_lbs_fwd<true, false>), grid, block, 0, s, m.Dk, m.Wjv, Jn_vi, FT, AT, VPb, JP, verts, ldv, B, BP, nvc);
  else if (verts)
    hipLaunchKernelGGL((k_lbs_fwd<false, true>), grid, block, 0, s, m.Dk, m.Wjv, Jn_vi, FT, AT, VPb, JP, verts, ldv, B, BP, nvc);
  else
    hipLaunchKernelGGL((k_lbs_fwd<false, false>), grid, block, 0, s, m.Dk, m.Wjv, Jn_vi, FT, AT, VPb, JP, verts, ldv, B, BP, nvc);
  return 0;
}

int launch_lbs_bwd(const Model& m, const float* Jn_iv, const float* AT, const float* VPb, const float* dJT,
                   const float* dVT, float* DVP, float* dATp, int BP, int nvc, hipStream_t s) {
  const int nitems = (BP / BT) * 3 * nvc;
  dim3 grid((nitems + 3) / 4), block(256);
  if (dVT)
    hipLaunchKernelGGL((k_lbs_bwd<true>), grid, block, 0, s, m.Wjv, m.Wvj, Jn_iv, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, nitems);
  else
    hipLaunchKernelGGL((k_lbs_bwd<false>), grid, block, 0, s, m.Wjv, m.Wvj, Jn_iv, AT, VPb, dJT, dVT, DVP, dATp, BP, nvc, nitems);
  return 0;
}

int launch_dverts_transpose(const float* dverts, float* dVT, int B, int BP, hipStream_t s) {
  hipLaunchKernelGGL(k_dverts_transpose, dim3(VT, BP / 32), dim3(256), 0, s, dverts, dVT, B, BP);
  return 0;
}

int launch_jreg_normalize(const float* J, const float* mask, float* rowsum, float* Jn, float* Jn_vi, float* Jn_iv,
                          hipStream_t s) {
  hipLaunchKernelGGL(k_jreg_rowsum, dim3(NH), dim3(256), 0, s, J, mask, rowsum);
  hipLaunchKernelGGL(k_jreg_tiles, dim3(VT * 1024 / 256), dim3(256), 0, s, J, mask, rowsum, Jn, Jn_vi, Jn_iv);
  return 0;
}

int launch_jreg_bwd(const float* J, const float* mask, const float* Jn, const float* rowsum, const float* dJn, int ldn,
                    float* dJ, hipStream_t s) {
  hipLaunchKernelGGL(k_jreg_bwd, dim3(NH), dim3(256), 0, s, J, mask, Jn, rowsum, dJn, ldn, dJ);
  return 0;
}

}  // namespace jrr
