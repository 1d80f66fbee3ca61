// Support-vertex tables and the stand-alone launch of the support iteration (supk.h).
//
// Reference: scripts/utils.py:87-98 (J_regressor * mask -> ReLU -> row-normalise, J x V), scripts/optimize.py:220-265.
#include "supk.h"

namespace jrr {

size_t sup_tables_floats() {
  return (size_t)SUP_DSF_FLOATS + SUP_DSB_FLOATS + SUP_NSV /*rows*/ + SUP_NSV /*sk_cnt*/ + 2 * (size_t)SUP_NSV * NJ + 32 /*jl_cnt*/ +
         2 * (size_t)NJ * SUP_NSV;
}
void sup_tables_carve(SupTables& t, float* base) {
  float* p = base;
  t.Dsf = p; p += SUP_DSF_FLOATS;
  t.Dsb = p; p += SUP_DSB_FLOATS;
  t.rows = (int*)p; p += SUP_NSV;
  t.sk_cnt = (int*)p; p += SUP_NSV;
  t.sk_j = (int*)p; p += SUP_NSV * NJ;
  t.sk_w = p; p += SUP_NSV * NJ;
  t.jl_cnt = (int*)p; p += 32;
  t.jl_s = (int*)p; p += NJ * SUP_NSV;
  t.jl_w = p; p += NJ * SUP_NSV;
}

// Blend-basis rows of the support vertices in the two operand images, and the skinning lists.  Dn [3][VP][KFP] and Wjv [VT][24][32]
// are the model's feature-contiguous basis and W^T tiles, both in the library's internal vertex order (as t.rows).
__global__ __launch_bounds__(256) void k_sup_gather(const float* __restrict__ Dn, const float* __restrict__ Wjv, SupTables t, int nsv) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  auto Dval = [&](int rho, int k) -> float {
    if (rho >= 3 * nsv || k >= KF) return 0.f;
    return Dn[((size_t)(rho % 3) * VP + t.rows[rho / 3]) * KFP + k];
  };
  if (idx < SUP_DSF_FLOATS) {
    const int tt = idx & 3, l = (idx >> 2) & 63, g = (idx >> 8) % SUP_FG, rt = idx / (256 * SUP_FG);
    t.Dsf[idx] = Dval(32 * rt + (l & 31), 8 * g + 4 * (l >> 5) + tt);
  } else if (idx < SUP_DSF_FLOATS + SUP_DSB_FLOATS) {
    const int q = idx - SUP_DSF_FLOATS;
    const int tt = q & 3, l = (q >> 2) & 63, rg = (q >> 8) % SUP_BG, mt = q / (256 * SUP_BG);
    t.Dsb[q] = Dval(8 * rg + 4 * (l >> 5) + tt, 32 * mt + (l & 31));
  } else {
    const int q = idx - SUP_DSF_FLOATS - SUP_DSB_FLOATS;
    auto W = [&](int s, int j) -> float { const int row = t.rows[s]; return Wjv[((size_t)(row >> 5) * NJ + j) * 32 + (row & 31)]; };
    if (q < SUP_NSV) {                    // vertex q: its joints
      int cnt = 0;
      if (q < nsv)
        for (int j = 0; j < NJ; ++j) {
          const float w = W(q, j);
          if (w != 0.f) { t.sk_j[q * NJ + cnt] = j; t.sk_w[q * NJ + cnt] = w; ++cnt; }
        }
      t.sk_cnt[q] = cnt;
    } else if (q < SUP_NSV + NJ) {        // joint j: its vertices
      const int j = q - SUP_NSV;
      int cnt = 0;
      for (int s = 0; s < nsv; ++s) {
        const float w = W(s, j);
        if (w != 0.f) { t.jl_s[j * SUP_NSV + cnt] = s; t.jl_w[j * SUP_NSV + cnt] = w; ++cnt; }
      }
      t.jl_cnt[j] = cnt;
    }
  }
}

int launch_sup_gather(const Model& m, const SupTables& t, int nsv, hipStream_t s) {
  const int n = SUP_DSF_FLOATS + SUP_DSB_FLOATS + SUP_NSV + NJ;
  hipLaunchKernelGGL(k_sup_gather, dim3((n + 255) / 256), dim3(256), 0, s, m.Dn, m.Wjv, t, nsv);
  return 0;
}

__global__ __launch_bounds__(SUP_THREADS) void k_sup_iter(SupArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sup_lds[];
  sup_body(sup_lds, blockIdx.x, a);
}

int launch_sup_iter(const SupTables& t, int nsv, const float* Jn_vi, const float* FTq, const float* AT, const float* gt_mm, float scale,
                    float* joints_out, float* sqerr, float* dA, float* dF, int B, int BP, hipStream_t s, const ReprojLaunch* r) {
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)k_sup_iter, hipFuncAttributeMaxDynamicSharedMemorySize, SUPL_FLOATS * 4) == hipSuccess;
  }();
  if (!attr) { jrr_set_error("k_sup_iter: %d bytes of LDS refused", SUPL_FLOATS * 4); return JRR_ERR_HIP; }
  SupArgs a{t, nsv, Jn_vi, FTq, AT, gt_mm, scale, joints_out, sqerr, dA, dF, B, BP,
            r ? Reproj{r->gt_j2d, r->cam, r->gcam, r->sq2d, r->scale2d} : Reproj{nullptr, nullptr, nullptr, nullptr, 0.f},
            nullptr, nullptr, nullptr, 0.f, nullptr, nullptr};      // (the per-joint MLP adjoint is its own launch in this form)
  hipLaunchKernelGGL(k_sup_iter, dim3((B + SUP_PP - 1) / SUP_PP), dim3(SUP_THREADS), SUPL_FLOATS * 4, s, a);
  return 0;
}

}  // namespace jrr
