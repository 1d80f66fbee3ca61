// Internal launcher interface between api.hip and the kernel translation units.
#pragma once
#include "jrr_common.h"

namespace jrr {

// flat pose-discriminator parameter offsets (state_dict order, scripts/discriminator.py:7-30)
constexpr int DP_CONV0_W = 0, DP_CONV0_B = 192, DP_CONV2_W = 224, DP_CONV2_B = 1248, DP_HEADS = 1280;
constexpr int DP_FC0_W = 2072, DP_FC0_B = 788504, DP_FC2_W = 789528, DP_FC2_B = 1838104;
constexpr int DP_FC4_W = 1839128, DP_FC4_B = 1840152, DP_TOTAL = 1840153;

enum { EPI_STORE = 0, EPI_BIAS_RELU = 1, EPI_MASK = 2, EPI_BIAS = 3, EPI_ACCUM = 4, EPI_BIAS_RELU_DOT = 5 };   // 5: k_disc_gemm only

struct GemmArgs {
  const float* A; int lda;      // A[k][m]
  const float* Bm; int ldb;     // Bm[k][n]
  float* Out; int ldo;          // Out[m][n]
  const float* bias;            // [m]      (EPI_BIAS_RELU / EPI_BIAS)
  const float* mask;            // [m][ldo] (EPI_MASK: keep where mask > 0)
  int M, N, K;
  size_t split_stride;          // floats between split-K partial slabs
  size_t batchA = 0, batchB = 0, batchO = 0;   // floats between the operands / outputs of gridDim.y batched products
  // k_disc_gemm, BTR = 2: the B operand x becomes (x > 0) ? dz[n] : 0 on its way to the MFMA, where dz[n] is the pose
  // discriminator's output-layer adjoint of column n, computed in the prologue from the
  // partial dots of the fc2 launch: z = zbias[0] + sum_t zpart[t][n]; s = sigmoid(z);
  // dz = (gout ? gout[n * gout_ld] : scale * (s - target)) * s * (1 - s), 0 for n >= nvalid.  Workgroups of the first
  // m-tile also write sq0[n] = (s - target)^2 and out0[n * out0_ld] = s when those pointers are set.
  const float* zpart = nullptr; int nzpart = 0; const float* zbias = nullptr;
  const float* gout = nullptr; int gout_ld = 0; float scale = 0.f, target = 0.f; int nvalid = 0;
  float* sq0 = nullptr; float* out0 = nullptr; int out0_ld = 0;
  const float* dotw = nullptr;       // [m]   (EPI_BIAS_RELU_DOT)
  float* dot_out = nullptr;          // [(M/BM) * WAVES_M][ldo] partial column dots (EPI_BIAS_RELU_DOT)
};

struct PrepBwdLaunch {
  const float* x6d_in = nullptr; const float* R_in = nullptr; const float* betas_in = nullptr;
  const float* dATp = nullptr; const float* dFTp = nullptr;    // adjoints dA^T [nslabA][288][BP] (partial slabs, summed
  int nslabA = 1; size_t strideA = 0;                           // by k_chain_bwd), dF^T [224][BP] (reduced)
  const unsigned* dmaskA = nullptr;                             // [nslabA][BP / 64] valid-joint masks of the slabs (NULL: all rows valid)
  const float* FT = nullptr; const float* R0T = nullptr; const float* AT = nullptr;   // saved by k_prep_fwd
  float* dRT = nullptr; float* dbT = nullptr;                   // scratch [216][BP], [10][BP]
  const float* gx_extra = nullptr; const float* gb_extra = nullptr;
  float* dx6d = nullptr; float* dR = nullptr; float* dbetas = nullptr;
  float* x6d_io = nullptr; float* betas_io = nullptr; float* adam_m = nullptr; float* adam_v = nullptr;
  const int32_t* step = nullptr;
  float lr = 0, beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
  int B = 0, BP = 0;
  const float* gcam = nullptr; float* cam_io = nullptr; float* cam_m = nullptr; float* cam_v = nullptr;
};

struct ReprojLaunch { const float* gt_j2d; const float* cam; float* gcam; float* sq2d; float scale2d; };

// prep.hip
int launch_rot6d_fwd(const float* x, float* R, int n, hipStream_t s);
int launch_rot6d_bwd(const float* x, const float* dR, float* dx, int n, hipStream_t s);
int launch_rodrigues_fwd(const float* aa, float* R, int n, hipStream_t s);
int launch_rodrigues_bwd(const float* aa, const float* dR, float* daa, int n, hipStream_t s);
// FT: blend features row-major [KFP][BP] (chain adjoint, folded product); FTq: the same in K-quads [KFP/4][BP][4] (k_lbs_fwd)
int launch_prep_fwd(const Model& m, const float* x6d, const float* Rin, const float* betas, float* FT, float* FTq, float* AT,
                    float* R0T, int B, int BP, int32_t* step_inc, hipStream_t s);
int launch_posed_joints(const Model& m, const float* AT, const float* betas, float* out, int B, int BP, hipStream_t s);
// its adjoint: dj24 (B,24,3) -> dA^T [288][BP] (one slab) and the direct shape term gb (B,10)
int launch_posed_joints_bwd(const Model& m, const float* AT, const float* betas, const float* dj24, float* dA, float* gb, int B, int BP,
                            hipStream_t s);
// horizontally fused launches of the inner loop (prep.hip): chain forward || per-joint MLP forward, and
// per-joint MLP adjoint || dF^T slab sum
int launch_prep_fwd_dconv(const Model& m, const float* x6d, const float* betas, float* FT, float* FTq, float* AT, float* R0T, int B,
                          int BP, int32_t* step_inc, const float* img, float* H2T, float* out, hipStream_t s);
int launch_dconv_bwd_reduce(const float* img, const float* x6d, const float* dH2T, const float* gout, float scale, float target,
                            float* gx, float* sqj, int B, int BP, const float* P, int nslab, size_t stride, float* out, size_t n,
                            hipStream_t s);
int launch_joints_loss(const float* JP, int nvc, const float* gt_mm, const float* djoints_in, float scale,
                       float* joints_out, float* sqerr, float* dJT, int B, int BP, hipStream_t s,
                       const ReprojLaunch* r = nullptr, int jp_rows = NH, const int* one_slab_flag = nullptr);
int launch_step_inc(int32_t* step, hipStream_t s);
int launch_project_joints(const float* joints, const float* cam, float* out, int B, hipStream_t s);
int launch_camera_fit(const float* joints, const float* gt_j2d, float* cam, float scale2d, int nsteps, float lr, float* sq2d,
                      int B, hipStream_t s);
int launch_joint_loss_plain(const float* joints, const float* gt_mm, float scale, float* sqerr, float* djoints, int B,
                            hipStream_t s);
int launch_prep_bwd(const PrepBwdLaunch& L, const Model& m, hipStream_t s);
int launch_reduce_slabs(const float* P, int nslab, size_t stride, float* out, size_t n, hipStream_t s, int accumulate = 0);
int launch_reduce_slabs2(const float* P1, int nslab1, size_t stride1, float* out1, size_t n1, const float* P2, int nslab2,
                         size_t stride2, float* out2, size_t n2, hipStream_t s);
int launch_adam_flat(float* p, const float* g, float* m, float* v, size_t n, const int32_t* step, float lr, float b1,
                     float b2, float eps, hipStream_t s, int step_plus = 0);

// lbs.hip
int launch_lbs_fwd(const Model& m, const float* Jn_vi, const float* FT, const float* AT, float* VPb, float* JP,
                   float* VTb, int B, int BP, int nvc, hipStream_t s, long long* probe = nullptr, const int* vmask = nullptr,
                   const int* tl = nullptr, int ntl = 0, int verts_pose_major = 0);
// verts_pose_major: VTb is [BP][3][VP] (a pose's vertices contiguous: the fused rasteriser's input) instead of row quads
// tl / ntl (nullable; joint-sparse classes, VPb kept): run the ntl listed tiles only (the regressor's support tiles: every other
// tile adds exact zeros to the joints); also launch_lbs_bwd and launch_blend_adjoint
// vmask (nullable, with VTb): store the vertices of the tiles with vmask[tile] != 0 only (JSupport::tmask)
int launch_verts_untranspose(const float* VTb, float* verts, int ldv, int vlimit, const float* cam, float* ndc, int B, int BP,
                             hipStream_t s, const int* p2v = nullptr);
int launch_bwd_tab_static(const Model& m, float* Tb, hipStream_t s);
int launch_lbs_bwd(const Model& m, const float* Tb, const float* AT, const float* VPb, const float* dJT,
                   const float* dVT, float* DVP, float* dATp, int BP, int nvc, hipStream_t s, const int* tl = nullptr, int ntl = 0,
                   unsigned* dmask = nullptr);
// dmask (nullable; k_lbs_bwd16 only -- ignored by the role kernel, which fills its slabs): [nvc][BP / 64] joint masks of the rows written
// (rows of untouched joints are then not zero-filled; consumer: PrepBwdLaunch::dmaskA)
int launch_dverts_transpose(const float* dverts, int ldv, float* dVT, int B, int BP, hipStream_t s, const int* p2v = nullptr);
// Support of the normalised regressor (the J step's products only need it: ReLU' makes dJ exactly zero elsewhere).
// JSupport lives in the engine workspace: cnt[i] positive entries of row i, their internal vertex rows / weights in ascending
// vertex order, flag[0] = 1 when every row has at most JSUP_CAP of them (else the dense products run).
constexpr int JSUP_CAP = 128;
struct JSupport { int* flag; int* cnt; int* col; float* val; int* tmask; int* tknown; };     // flag[64], cnt[32], col[17][JSUP_CAP], val[17][JSUP_CAP], tmask[VT], tknown[VT]
// flag[0] fits, flag[1..2] counters of k_jstep_update, flag[JSUP_ERR] STICKY error word, flag[JSUP_KNOWN] = 1 while the host relies on
// tknown = the tile mask jrr_j_support_info reported: a support that then leaves it (bit 0) or stops fitting the lists (bit 1) --
// only possible when the caller edits J or the mask in place behind the engine's back -- sets the error word (k_jsup_tilemask)
constexpr int JSUP_ERR = 3, JSUP_KNOWN = 4;
// tmask[tile] = 1 when a support entry lives in that 32-vertex tile (rebuilt with the lists)
int launch_jreg_normalize(const float* J, const float* mask, float* rowsum, float* Jn, float* Jn_vi, float* Jn_iv, float* Jn_q,
                          const int* p2v, hipStream_t s, int r16 = 0, const int* v2p = nullptr, const JSupport* sup = nullptr,
                          int32_t* step_inc = nullptr);
// dJn[i][row] = sum_{b, r} dj_r[i][b] verts_r[row][b] for the support entries only (sup.flag != 0; the other entries of dJn keep
// whatever finite value they hold: they meet Jn = 0 and relu' = 0 in k_jreg_bwd)
int launch_jgrad_sparse(const JSupport& sup, const float* dJT, const float* VTq, float* dJn, int BP, hipStream_t s);
// joints of the stored vertices with the current regressor, one slab [3][32][BP] (rows i < 17), support entries only
// step_inc (nullable): incremented by one thread of the launch (the reuse iteration's Adam step count: one launch less)
int launch_rejoints_sparse(const JSupport& sup, const float* VTq, float* out, int BP, hipStream_t s, int32_t* step_inc = nullptr);
int launch_jsup_tilemask(const JSupport& sup, hipStream_t s);
int launch_jsup_scatter(const JSupport& sup, const float* in, const int* p2v, float* dJ, hipStream_t s);
// sup / p2v / dJs (nullable): also deliver the gradient on the support lists, dJs [17][JSUP_CAP]
int launch_jreg_bwd(const float* J, const float* mask, const float* Jn, const float* rowsum, const float* dJn, int ldn,
                    float* dJ, const int* v2p, hipStream_t s, const JSupport* sup = nullptr, const int* p2v = nullptr, float* dJs = nullptr);
// the J step's second half in one launch (lbs.hip k_jstep_update)
struct JStepUpdate {
  float* J; const float* dJ; const float* dJs; float* m; float* v; int32_t* step; float lr;
  const float* mask; float* Jraw; float* Jmask; float* rowsum; float* Jn; float* Jn_vi; float* Jn_iv; float* Jn_q;
  const int* p2v; const int* v2p; int r16;
  JSupport sup; int* sync;          // sync[0] workgroups done, sync[1] rows above the list capacity (both zero between launches)
};
int launch_jstep_update(const JStepUpdate& a, hipStream_t s);

// gemm.hip
int launch_disc_gemm_q(const GemmArgs& g, int epi, int btr, hipStream_t s, int* ndot = nullptr);
int launch_to_quads(const float* in, int ld, float* out, int K, int M, hipStream_t s);
int launch_gemm_128(const GemmArgs& g, int epi, int nsplit, hipStream_t s);
int launch_gemm_128x64(const GemmArgs& g, int epi, int nsplit, hipStream_t s);
int launch_gemm_128x32(const GemmArgs& g, int epi, int nsplit, hipStream_t s);
int launch_gemm_224(const GemmArgs& g, int epi, int nsplit, hipStream_t s);
// blend-basis adjoint dF^T[split][224][BP] = sum_{c, v in split} D_c[v][.] dvp_c[v][.], both operands in vertex quads
int launch_blend_adjoint(const float* Dq, const float* DVPq, float* dFTp, size_t split_stride, int BP, int nsplit, hipStream_t s,
                         const int* tl = nullptr, int ntl = 0);
// side mode JRR_FLAG_BLEND_BF16X3: the same product from split-bf16 operands (three bf16 matrix instructions, fp32 accumulation)
int launch_split_blend_basis(const float* Dq, void* Ds, hipStream_t s);
size_t blend_basis_split_bytes();
int launch_blend_adjoint_bf16x3(const void* Ds, const float* DVPq, float* dFTp, size_t split_stride, int BP, int nsplit, hipStream_t s);
// skip_flag (device, nullable): the launch returns at once when *skip_flag != 0 (the support-restricted kernels did the work)
int launch_gemm_q32(const float* A, int ldA, size_t planeA, const float* Bm, int ldB, size_t planeB, float* Out, int ldo,
                    size_t ks_stride, size_t plane_stride, int N, int K, int nplanes, int ksplit, hipStream_t s,
                    const int* skip_flag = nullptr);
int launch_jgrad_q(const float* dJT, const float* VTq, float* Out, int BP, int ksplit, hipStream_t s, const int* skip_flag = nullptr);

// sil.hip
// S = image size (224 or 256; focal length 5000 / S)
int launch_sil_project(const float* verts, int ldv, const float* cam, float* ndc, int B, hipStream_t s, int S = 224);
// cover [B][S*S] / ncover [B]: the covered pixels of each pose (pixel << 14 | winning face), written by the
// rasteriser and consumed by the adjoint
int launch_sil_raster(const float* ndc, const unsigned* faces_pk, int nfaces, unsigned* cover, int* ncover, float* alpha, int B,
                      hipStream_t s, int S = 224);
// smask [B] = per-pose sum(mask^2) over the image (launch_mask_sq): the rasteriser only visits the mesh's pixel box
int launch_mask_sq(const float* mask, float* smask, int B, hipStream_t s, int S = 224);
int launch_sil_raster_adj(float* VQ, int BP, const float* cam, const unsigned* faces_pk, int nfaces, const float* mask, const float* smask,
                          unsigned* cover, int* ncover, float* sqsil, float scale, float* gcam, int accumulate_cam, int B,
                          hipStream_t s, int S = 224, const float* VPM = nullptr);
// VPM (nullable): the vertices pose-major [BP][3][VP] (launch_lbs_fwd with verts_pose_major); NULL: read from the row quads of VQ
int launch_sil_pix_to_face(const unsigned* cover, const int* ncover, int* p2f, int B, hipStream_t s, int S = 224);
int launch_sil_bwd(const float* ndc, const int* faces, const unsigned* cover, const int* ncover, const float* mask,
                   const float* galpha, float scale, float* dverts, int ldv, float* gcam, int accumulate_cam, int B,
                   hipStream_t s, int S = 224);

// sup.hip / supk.h: the joint-loss iteration on the regressor's support VERTICES, one workgroup per 32-pose group
constexpr int SUP_NSV = 64;       // most support vertices the fused iteration is built for (192 coordinate rows)
// tables of a support (engine workspace; built by launch_sup_gather whenever the engine learns a support, jrr_j_support_info):
//   rows [SUP_NSV]                internal vertex rows of the support vertices, ascending
//   Dsf  [6 tiles][28][64][4]     their blend-basis rows as the A operand of v_posed = Ds F   (lane l: row 32 t + l % 32, k = 8 g + 4 (l / 32) + 0..3)
//   Dsb  [7 tiles][24][64][4]     the same as the A operand of dF = Ds^T dvp                  (lane l: k = 32 t + l % 32, row 8 g + 4 (l / 32) + 0..3)
//   sk_* [SUP_NSV][24]            per vertex: its joints with a non-zero skinning weight (ascending) and the weights
//   jl_* [24][SUP_NSV]            per joint: the support vertices it skins (ascending) and the weights
struct SupTables { float* Dsf; float* Dsb; int* rows; int* sk_cnt; int* sk_j; float* sk_w; int* jl_cnt; int* jl_s; float* jl_w; };
size_t sup_tables_floats();
void sup_tables_carve(SupTables& t, float* base);
int launch_sup_gather(const Model& m, const SupTables& t, int nsv, hipStream_t s);
// one joint-loss forward + backward of the pose groups: F^T (K-quads) / A^T of k_prep_fwd in, joints / squared error / dA^T [288][BP] /
// dF^T [224][BP] out (what k_chain_bwd reads as ONE slab)
int launch_sup_iter(const SupTables& t, int nsv, const float* Jn_vi, const float* FTq, const float* AT, const float* gt_mm, float scale,
                    float* joints_out, float* sqerr, float* dA, float* dF, int B, int BP, hipStream_t s, const ReprojLaunch* r = nullptr);

// ONE inner iteration per 32-pose group in one launch (prep.hip k_sup_step): chain forward, support-vertex forward / loss / backward,
// [per-joint MLP adjoint], chain adjoint + Adam, [per-joint MLP forward of the next iteration].  The pose-update half takes a PrepBwdLaunch
// (x6d_in / betas_in / gx_extra / gb_extra / x6d_io / betas_io / Adam state / step / lr / B / BP).
struct SupStepLaunch {
  SupTables t; int nsv; const float* Jn_vi; const float* gt_mm; float scale;
  float* FT; float* FTq; float* AT; float* R0T; float* joints_out; float* sqerr; float* dA; float* dF;
  const float* conv_img = nullptr; const float* dH2T = nullptr; float dscale = 0.f; float* gx = nullptr; float* dsq = nullptr;
  float* H2T_next = nullptr;
  int32_t* step = nullptr; int* arrive = nullptr;      // Adam's step counter (incremented once per launch) and the arrival counter
  // 2-D reprojection term (nullable gt_j2d: off); the camera's Adam state travels in the PrepBwdLaunch (gcam / cam_io / cam_m / cam_v)
  const float* gt_j2d = nullptr; const float* cam = nullptr; float* gcam = nullptr; float* sq2d = nullptr; float scale2d = 0.f;
};
// phase 0: the whole iteration; 1 / 2: its halves before / after the point where the discriminator GEMMs' results are read
int launch_sup_step(const Model& m, const SupStepLaunch& q, const PrepBwdLaunch& L, hipStream_t s, int phase = 0);

// the END of an all-tiles iteration + the BEGINNING of the next in one launch (prep.hip k_tail_step): per-joint MLP adjoint, chain adjoint +
// Adam (the PrepBwdLaunch, as launch_prep_bwd takes it), then -- do_next -- the chain forward [+ MLP forward] of the next iteration
struct TailStepLaunch {
  const float* conv_img = nullptr; const float* dH2T = nullptr; float dscale = 0.f; float* gx = nullptr; float* dsq = nullptr;
  bool do_next = false; float* FTw = nullptr; float* FTq = nullptr; float* ATw = nullptr; float* R0Tw = nullptr; float* H2T_next = nullptr;
  int32_t* step = nullptr; int* arrive = nullptr;
};
int launch_tail_step(const Model& m, const TailStepLaunch& q, const PrepBwdLaunch& L, hipStream_t s);

// fold.hip
int launch_fold_jw(const float* Jn, const float* Wjv, float* JW, float* G0, const int* p2v, hipStream_t s);
int launch_fold_fwd(const float* MT, const float* AT, const float* G0, float* Jsum, int BP, hipStream_t s);
int launch_fold_bwd(const float* dJT, const float* AT, const float* MT, const float* G0, float* dMT, float* dA, int BP,
                    hipStream_t s);

// eval.hip
int launch_evaluate(const float* pred, const float* target_mm, float* err, float* err_pa, int B, hipStream_t s);

// disc.hip
int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t s, int ldin = 0, int ldout = 0);
constexpr int CONV_IMAGE_FLOATS = 4224;     // LDS parameter image of the per-joint MLP kernels (disc.hip CL_*), rounded up
int launch_conv_image(const float* P, float* img, hipStream_t s);
// (the per-joint MLP launchers take the IMAGE, not the flat parameter vector)
int launch_disc_conv_fwd(const float* P, const float* x6d, float* H2T, float* out, int B, int BP, hipStream_t s, int quad = 0);
int launch_disc_out(const float* P, const float* A2T, float* out, float* dA2T, const float* gout, float scale,
                    float target, int B, int BP, hipStream_t s, float* dz0 = nullptr, float* sq0 = nullptr);
int launch_disc_z_finish(const float* zpart, int nz, int ld, const float* zbias, float* out, int B, hipStream_t s);
int launch_scale_rows(const float* in, const float* w, float* out, int rows, int cols, hipStream_t s);
int launch_colsum(const float* M, int rows, int ld, float* out, int B, hipStream_t s);
int launch_rowdot_accum(const float* M, int ld, const float* vec, float* out, int rows, int cols, hipStream_t s);
// conv / per-joint-head weight gradients as partial slabs: shared [24 * BP/64][1280] (conv0 W,b | conv2 W,b in the
// DP_* order) and heads [BP/64][792]; the caller reduces them into the flat gradient
int launch_disc_conv_bwd_params(const float* P, const float* x6d, const float* dH2T, const float* gout, float scale,
                                float target, float* slab_shared, float* slab_heads, int B, int BP, hipStream_t s);
int launch_shape_disc_bwd_params(const float* P, const float* betas, const float* gout, float scale, float target,
                                 float* dparams, float* sqerr, int B, hipStream_t s);
int launch_sqerr_rows(const float* out, int ncol, float target, float* sqerr, int B, hipStream_t s);
int launch_disc_conv_bwd(const float* P, const float* x6d, const float* dH2T, const float* gout, float scale,
                         float target, float* gx, int B, int BP, hipStream_t s, float* sqj = nullptr, int quad = 0);
int launch_shape_disc(const float* P, const float* betas, float* out, float* gb, float scale, float target, int B,
                      hipStream_t s, const float* gout = nullptr, float* sq = nullptr);

}  // namespace jrr
