/*
 * jrr.h -- C ABI of the MI355X (gfx950) pose-refinement hot path.
 *
 * Drop-in boundary for the inner loop of the reference's scripts/optimize.py
 * (ubc-vision/joint-regressor-refinement).  The reference has no FFI layer: its boundary is
 * Python-level.  Each entry point below names the reference
 * interface (file:line in /root/reference) whose arithmetic it replaces; the Python host
 * code in joint-regressor-refinement_amd/ binds these through ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / C++ types cross the boundary.
 *  - every `*_dev` pointer is device memory owned by the caller (PyTorch's allocator), 16-byte aligned (whole
 *    tensors and pose-granular contiguous slices of them are);
 *    every `*_host` pointer is host memory read synchronously during the call.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *    kernels are enqueued on it, no entry point synchronises the device unless stated.  (jrr_refine_run* may put part of an iteration on an engine-owned side stream, forked from and joined back into `stream` by events within the call: ordering as seen from `stream` is unchanged.)
 *  - return value: 0 on success, negative jrr_status otherwise; nothing throws.
 *  - one caller thread per engine; engines on different devices/processes are independent.
 *  - all floating point is IEEE fp32 (the reference runs `.float()`, optimize.py:160);
 *    matrix products use the exact-fp32 MFMA v_mfma_f32_32x32x2_f32.
 */
#ifndef JRR_H
#define JRR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  JRR_OK = 0,
  JRR_ERR_ARG = -1,       /* null pointer / bad size */
  JRR_ERR_HIP = -2,       /* a HIP runtime call failed; see jrr_last_error() */
  JRR_ERR_WORKSPACE = -3, /* workspace too small */
  JRR_ERR_STATE = -4      /* engine not configured for the requested op */
} jrr_status;

enum {
  JRR_NUM_VERTS = 6890,
  JRR_NUM_JOINTS = 24,
  JRR_NUM_H36M = 17,
  JRR_NUM_BETAS = 10,
  JRR_POSE6D = 144,        /* 24 joints x 6 */
  JRR_DISC_PARAMS = 1840153,
  JRR_SHAPE_DISC_PARAMS = 171,
  JRR_SIL_SIZE = 224       /* silhouette image size (square) of an engine without JRR_FLAG_SIL_256; see jrr_silhouette_forward */
};

/* engine flags */
enum {
  JRR_FLAG_POSE_DISC = 1,   /* allocate / run the pose-discriminator term (optimize.py:241-247) */
  JRR_FLAG_SHAPE_DISC = 2,  /* shape-discriminator term (optimize.py:244,249-250) */
  JRR_FLAG_KEEP_VERTS = 4,  /* reserve a (B,6890,3) vertex buffer for return_verts / J step */
  JRR_FLAG_FOLDED = 8,      /* reserve the folded-regressor tables (jrr_engine_set_folded) */
  JRR_FLAG_SILHOUETTE = 16, /* reserve the soft-silhouette buffers (needs JRR_FLAG_KEEP_VERTS and model faces): per pose the projected
                               vertices of the stand-alone API (110 KB), the covered-pixel list (200 KB at 224 x 224) and a pose-major
                               copy of the vertices for the fused rasteriser (83 KB) */
  JRR_FLAG_NO_MODEL = 32,   /* discriminator-only engine (model == NULL): the SMPL sections (~230 KB per pose) are not
                               part of the workspace; only JRR_FLAG_POSE_DISC / JRR_FLAG_SHAPE_DISC may accompany it */
  JRR_FLAG_SIL_256 = 64,    /* with JRR_FLAG_SILHOUETTE: 256 x 256 silhouettes (the reference constructor's default,
                               scripts/mesh_renderer.py:25; focal length 5000 / 256) instead of 224 x 224 (scripts/optimize.py:110):
                               every (B,224,224) below then reads (B,256,256) */
  JRR_FLAG_SIL_SIZE_MASK = 15 << 16, /* with JRR_FLAG_SILHOUETTE: JRR_FLAG_SIL_SIZE(size) below -- an explicit image size, any multiple of 32
                               up to 256 (Mesh_Renderer(image_size), scripts/mesh_renderer.py:25,34-38; focal length 5000 / size).  Sizes other
                               than 224 and 256 serve the stand-alone renderer (jrr_silhouette_forward / _backward / _pix_to_face); the
                               silhouette term INSIDE the loop (jrr_engine_set_silhouette) is built for 224 and 256, the sizes the reference
                               instantiates.  0 = 224, or 256 with JRR_FLAG_SIL_256 */
  JRR_FLAG_SUPPORT_TILES = 128,/* with JRR_FLAG_KEEP_VERTS: once jrr_j_support_info has reported that the regressor's support fits,
                               the iterations of jrr_refine_run* whose loss reads the JOINTS only (no silhouette term) run their three
                               skinning kernels on the 32-vertex tiles that hold a support entry -- every other tile multiplies
                               its vertices by a zero block of the regressor and receives a zero vertex adjoint: exact zeros in
                               the joints and in every gradient.  Same numbers as without the flag up to the order of the sums
                               over the tiles; until jrr_j_support_info is called (and after a jrr_engine_set_j_regressor from
                               outside) the iterations run all 216 tiles.  v_posed / vertices of the other tiles are NOT
                               produced by those iterations (jrr_find_joints_forward and jrr_smpl_vertices* always are dense). */
  JRR_FLAG_BLEND_BF16X3 = 256  /* SIDE MODE, not the reference's arithmetic (the reference computes in fp32 and so does every engine
                               without this flag): the all-tiles blend-basis adjoint (jrr_refine_run*, jrr_find_joints_backward, jrr_smpl_vertices_backward) runs as a split-bf16 product
                               -- operands taken as bf16 hi + lo, three bf16 matrix instructions per exact-fp32 eight, fp32
                               accumulation; relative error of the product ~ 3e-5.  Reserves 18.6 MB for the split basis.  Every
                               other kernel, and the support-tile iterations, are unchanged.  bench.py reports it as a separately
                               labelled block (`bf16x3_mode`); it is never the headline. */
};

#define JRR_FLAG_SIL_SIZE(size) ((((size) / 32) & 15) << 16)

typedef struct jrr_model jrr_model_t;   /* device-resident, re-laid-out SMPL constants */
typedef struct jrr_engine jrr_engine_t; /* per-batch plan: workspace carve-up + launch geometry */

const char* jrr_last_error(void);
int jrr_version(void);

/* ---- SMPL model ---------------------------------------------------------------------------
 * Replaces SMPL("SPIN/data/smpl", batch_size=1).to(device) (scripts/optimize.py:96-99;
 * wrapper scripts/smpl.py:61-85).  Host arrays in smplx layout:
 *   v_template (6890,3)  shapedirs (6890,3,10)  posedirs (207,20670)
 *   J_regressor (24,6890)  lbs_weights (6890,24)  parents (24)
 * Uploads tile-major copies of the blend basis / skinning weights and the folded rest-joint
 * regressor (J_template, J_shapedirs).  Synchronous.                                         */
int jrr_model_create(const float* v_template_host, const float* shapedirs_host,
                     const float* posedirs_host, const float* J_regressor_host,
                     const float* lbs_weights_host, const int32_t* parents_host,
                     jrr_model_t** out);
void jrr_model_destroy(jrr_model_t* m);
/* The same with the model's device memory owned by the CALLER (the caller owns every buffer):
 * buffer_dev = jrr_model_bytes() bytes, 256-byte aligned, it must outlive the model; NULL = the library allocates its own
 * (what jrr_model_create does).  The buffer also holds the face lists of jrr_model_set_faces.                              */
size_t jrr_model_bytes(void);
int jrr_model_create_in(const float* v_template_host, const float* shapedirs_host, const float* posedirs_host,
                        const float* J_regressor_host, const float* lbs_weights_host, const int32_t* parents_host,
                        void* buffer_dev, size_t buffer_bytes, jrr_model_t** out);
/* ... with a HINT for the internal vertex order: hint_vertices_host[n_hint] (file indices; duplicates ignored) are the vertices the
 * caller's H36M regressor will read -- the columns where it is positive.  They are stored first (ceil(n / 32) tiles instead of
 * up to one tile per vertex), which is what the iterations of JRR_FLAG_SUPPORT_TILES run; everything visible at the API keeps the
 * file's vertex order.  The hint is dropped (jrr_model_info out[29] = 0) when one of those tiles would see more than 16 joints.
 * A regressor with OTHER positive columns works as before, on more tiles.  No reference counterpart: the reference multiplies
 * all 6890 vertices by the regressor (scripts/utils.py:87-92).                                                              */
int jrr_model_create_hinted(const float* v_template_host, const float* shapedirs_host, const float* posedirs_host,
                            const float* J_regressor_host, const float* lbs_weights_host, const int32_t* parents_host,
                            const int32_t* hint_vertices_host, int n_hint, void* buffer_dev, size_t buffer_bytes, jrr_model_t** out);
/* What the LBS kernels will run for this body (measurement / diagnostics; the reference has no counterpart):
 * out[0] = joint slots per 32-vertex tile and pass of the joint-sparse kernels (8 or 12; 0 = the dense kernels),
 * out[1] = WIDE tiles (more joints than that: each runs a second pass -- it costs itself, not the model),
 * out[2] = most joints of any tile, out[3] = 1 when the vertices are stored in the library's own joint-sorted order
 * (invisible at the API), out[4 + k] = number of tiles with k joints, k = 0 .. 24, out[29] = vertices stored first on the
 * caller's hint (jrr_model_create_hinted; 0 = no hint or hint dropped).                                                  */
int jrr_model_info(const jrr_model_t* m, int32_t* out, int n);
/* triangle list of the mesh (SMPL `f`, 13776 x 3 int32; the reference reads it from data/body_model/smpl_uv.obj,
 * scripts/mesh_renderer.py:40-41); needed by the silhouette renderer only.  Synchronous.                  */
int jrr_model_set_faces(jrr_model_t* m, const int32_t* faces_host, int n_faces);

/* ---- engine -------------------------------------------------------------------------------
 * `batch` = poses on this device; `batch_norm` = divisor batch of the MSE means
 * (== batch single-GPU; == global batch under data parallelism so that a sharded run equals
 * the single-process run).  The caller owns `workspace_dev`: jrr_engine_workspace_bytes bytes,
 * 256-byte aligned and ZERO-FILLED at creation (padding rows / columns of several sections are operands of the matrix
 * kernels and are never written).
 * `model` may be NULL for an engine that serves the discriminators only (flags within
 * JRR_FLAG_POSE_DISC | JRR_FLAG_SHAPE_DISC): Discriminator / Shape_Discriminator modules need no body model. */
size_t jrr_engine_workspace_bytes(int batch, int flags);
int jrr_engine_create(const jrr_model_t* model, int batch, int batch_norm, void* workspace_dev,
                      size_t workspace_bytes, int flags, jrr_engine_t** out);
void jrr_engine_destroy(jrr_engine_t* e);
int jrr_engine_set_batch_norm(jrr_engine_t* e, int batch_norm);
/* Folded joint regression for jrr_refine_run (needs JRR_FLAG_FOLDED): the pose-independent contraction
 * H = sum_v Jn W D (1224 x 218) is rebuilt at every jrr_engine_set_j_regressor, and each iteration evaluates
 * joints = A . (H F) instead of skinning 6890 vertices -- the same function of (theta, beta, J) up to fp32
 * rounding, ~25x fewer FLOP, no vertices.  A separate mode with its own denominator; never the default.   */
int jrr_engine_set_folded(jrr_engine_t* e, int enabled, void* stream);
/* J*mask -> ReLU -> row-normalise (scripts/utils.py:87-92), into the engine's tile-major
 * copies.  J_dev: (17,6890) row-major raw parameter; mask_dev may be NULL.                   */
int jrr_engine_set_j_regressor(jrr_engine_t* e, const float* J_dev, const float* mask_dev, void* stream);

/* Pose-discriminator weights (scripts/discriminator.py:7-30) as ONE flat fp32 vector in
 * state_dict order: conv_operations.{0,2}.{weight,bias}, linears.{0..23}.{weight,bias},
 * linear_operations.{0,2,4}.{weight,bias} (1 840 153 floats).  Shape discriminator
 * (discriminator.py:57-68): shape_operations.{0,2,4}.{weight,bias} (171 floats).            */
int jrr_engine_set_pose_disc(jrr_engine_t* e, const float* params_dev, void* stream);
int jrr_engine_set_shape_disc(jrr_engine_t* e, const float* params_dev, void* stream);

/* ---- operator-level entry points (autograd.Function backends) ------------------------------ */

/* rot6d_to_rotmat, scripts/utils.py:190-204: x (n,6) -> R (n,3,3); and its adjoint.          */
int jrr_rot6d_forward(const float* x6d_dev, float* R_dev, int n, void* stream);
int jrr_rot6d_backward(const float* x6d_dev, const float* dR_dev, float* dx6d_dev, int n, void* stream);

/* Axis-angle -> rotation matrix, smplx 0.1.26 lbs.batch_rodrigues: the pose2rot=True branch of the SMPL operator
 * (smplx.SMPL.forward default; the reference's wrapper inherits it, scripts/smpl.py:61-85, base class :7-9).
 * aa (n,3) -> R (n,3,3) with theta = |aa + 1e-8|, R = I + sin(theta) K + (1-cos(theta)) K^2; and its adjoint
 * dR (n,3,3) -> daa (n,3), finite at aa = 0.                                                               */
int jrr_rodrigues_forward(const float* aa_dev, float* R_dev, int n, void* stream);
int jrr_rodrigues_backward(const float* aa_dev, const float* dR_dev, float* daa_dev, int n, void* stream);

/* find_joints, scripts/utils.py:85-103 (SMPL forward + J_regressor contraction).
 * Exactly one of x6d_dev (B,24,6) / R_dev (B,24,3,3) is non-NULL.  joints_dev (B,17,3).
 * verts_dev (B,6890,3) may be NULL (return_verts=False); non-NULL needs JRR_FLAG_KEEP_VERTS (or _SILHOUETTE):
 * the kernel stores the vertices coordinate-major and a transposing pass produces the reference's layout.  */
int jrr_find_joints_forward(jrr_engine_t* e, const float* x6d_dev, const float* R_dev,
                            const float* betas_dev, float* joints_dev, float* verts_dev, void* stream);
/* Adjoint of the call above for the SAME inputs (must follow it): djoints (B,17,3) ->
 * dx6d (B,24,6) or dR (B,24,3,3), dbetas (B,10), dJ (17,6890) w.r.t. the RAW J_regressor
 * (through the normalisation + ReLU + mask).  Any output pointer may be NULL.                 */
int jrr_find_joints_backward(jrr_engine_t* e, const float* x6d_dev, const float* R_dev,
                             const float* betas_dev, const float* djoints_dev, float* dx6d_dev,
                             float* dR_dev, float* dbetas_dev, float* dJ_dev, void* stream);

/* find_joints (scripts/utils.py:85-103) of the poses of the J step that PRECEDED -- jrr_j_regressor_grad[_support] followed by
 * jrr_j_step_apply[_support] on the same x6d / betas buffers, nothing else in between -- with the stepped regressor, re-regressed from
 * that step's stored vertices instead of a second SMPL forward: the joints the driver evaluates after the step
 * (scripts/optimize.py:317-321).  joints_dev (B,17,3).  A mismatch the engine can detect returns JRR_ERR_STATE.          */
int jrr_find_joints_after_j_step(jrr_engine_t* e, const float* x6d_dev, const float* betas_dev, float* joints_dev, void* stream);

/* The `joints` field of the SMPL operator's output (scripts/smpl.py:69-84: smplx's 24 posed joints J_transformed = G_j[:3, 3] of the
 * kinematic chain head the list the wrapper re-maps).  Must follow a forward on this engine (jrr_find_joints_forward,
 * jrr_refine_run, ...) with the SAME betas: reads the stored skinning transforms.  joints24_dev (B,24,3).
 * (dead on the hot path: every caller of the reference reads `.vertices` only).        */
int jrr_smpl_posed_joints(jrr_engine_t* e, const float* betas_dev, float* joints24_dev, void* stream);
/* Its adjoint (smplx's `joints` are differentiable, scripts/smpl.py:69-84): djoints24_dev (B,24,3) -> the gradients w.r.t. the forward's
 * inputs through the kinematic chain and the rest joints J(beta).  Exactly one of x6d_dev (B,24,6) / R_dev (B,24,3,3) names the rotation
 * input of that forward; outputs dx6d_dev (B,24,6) or dR_dev (B,24,3,3), and dbetas_dev (B,10), nullable.  Must follow the forward
 * like jrr_smpl_posed_joints.                                                                                          */
int jrr_smpl_posed_joints_backward(jrr_engine_t* e, const float* x6d_dev, const float* R_dev, const float* betas_dev,
                                   const float* djoints24_dev, float* dx6d_dev, float* dR_dev, float* dbetas_dev, void* stream);

/* SMPL operator on its own: smpl(global_orient, body_pose, betas, pose2rot=False).vertices
 * (call sites scripts/utils.py:94-95, scripts/optimize.py:78-79, scripts/renderer.py:32-33).
 * Forward = jrr_find_joints_forward with verts_dev != NULL.  This is the adjoint w.r.t. the
 * vertices for the SAME inputs (must follow that forward): dverts (B,6890,3) -> dx6d / dR, dbetas.
 * Needs JRR_FLAG_KEEP_VERTS (the padded vertex buffer doubles as the transposed adjoint).        */
int jrr_smpl_vertices_backward(jrr_engine_t* e, const float* x6d_dev, const float* R_dev,
                               const float* betas_dev, const float* dverts_dev, float* dx6d_dev,
                               float* dR_dev, float* dbetas_dev, void* stream);

/* move_pelvis + MSELoss, scripts/utils.py:106-114 + scripts/optimize.py:238-239:
 * sqerr_dev[b] = sum_{i,c} (joints[b,i,c]-joints[b,0,c] - gt_mm[b,i,c]/1000)^2 ;
 * djoints = d(weight * mean)/d joints with mean over batch_norm*51.                            */
int jrr_joint_loss(const float* joints_dev, const float* gt_centred_mm_dev, float weight,
                   int batch, int batch_norm, float* sqerr_dev, float* djoints_dev, void* stream);

/* Discriminator.forward, scripts/discriminator.py:32-54: x (B,24,6) -> out (B,25) sigmoid.    */
int jrr_pose_disc_forward(jrr_engine_t* e, const float* x6d_dev, float* out_dev, void* stream);
/* d[ weight * mean((D(x)-target)^2) ] / dx for the forward just run (optimize.py:246-247).    */
int jrr_pose_disc_backward_input(jrr_engine_t* e, const float* x6d_dev, float weight, float target,
                                 float* dx6d_dev, void* stream);
/* vector-Jacobian product of Discriminator.forward w.r.t. its input for an arbitrary upstream
 * gradient gout (B,25) (autograd backward of the module); must follow jrr_pose_disc_forward.   */
int jrr_pose_disc_vjp_input(jrr_engine_t* e, const float* x6d_dev, const float* gout_dev,
                            float* dx6d_dev, void* stream);
/* weight gradients of  mean((D(x)-target)^2)  (mean over batch_norm*25) accumulated (+=) into a
 * flat vector laid out like the parameter vector (one term of scripts/optimize.py:276-284);
 * sqerr_dev (B, nullable) receives sum_k (D(x)[b,k]-target)^2.                                   */
int jrr_pose_disc_backward_params(jrr_engine_t* e, const float* x6d_dev, float target,
                                  float* dparams_dev, float* sqerr_dev, void* stream);
/* vector-Jacobian product of Discriminator.forward w.r.t. the WEIGHTS for an arbitrary upstream gradient gout (B,25),
 * accumulated (+=) into a flat vector laid out like the parameter vector: what `loss.backward()` leaves in the
 * module's .grad for any loss of the discriminator output (scripts/optimize.py:276-284 through the nn.Module).  */
int jrr_pose_disc_vjp_params(jrr_engine_t* e, const float* x6d_dev, const float* gout_dev, float* dparams_dev,
                             void* stream);
/* the same for Shape_Discriminator: gout (B), 171 parameters (float atomics) */
int jrr_shape_disc_vjp_params(jrr_engine_t* e, const float* betas_dev, const float* gout_dev, float* dparams_dev,
                              void* stream);
/* the same for Shape_Discriminator (scripts/optimize.py:286-293): betas (B,10), 171 parameters */
int jrr_shape_disc_backward_params(jrr_engine_t* e, const float* betas_dev, float target,
                                   float* dparams_dev, float* sqerr_dev, void* stream);

/* Shape_Discriminator.forward, scripts/discriminator.py:70-74: betas (B,10) -> out (B) sigmoid scores; and the
 * vector-Jacobian product w.r.t. the input for an upstream gradient gout (B) (autograd backward of the module).
 * Stateless apart from the parameters (jrr_engine_set_shape_disc): the vjp recomputes the 171-parameter forward. */
int jrr_shape_disc_forward(jrr_engine_t* e, const float* betas_dev, float* out_dev, void* stream);
int jrr_shape_disc_vjp_input(jrr_engine_t* e, const float* betas_dev, const float* gout_dev, float* dbetas_dev,
                             void* stream);

/* torch.optim.Adam single-tensor update (defaults used at scripts/optimize.py:116-126,201):
 * step_dev holds the 1-based step count of THIS update.                                        */
int jrr_adam_step(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, size_t n,
                  const int32_t* step_dev, float lr, float beta1, float beta2, float eps, void* stream);

/* evaluate, scripts/utils.py:117-145 + scripts/eval_utils.py:7-58 (row f3): per-pose mean joint error and
 * Procrustes-aligned mean joint error in METRES (pred in m, target in mm, both pelvis-centred inside);
 * MPJPE / PA-MPJPE in mm = 1000 * mean over poses.                                                    */
int jrr_evaluate(const float* pred_j3d_dev, const float* target_j3d_mm_dev, float* err_dev, float* err_pa_dev,
                 int batch, void* stream);

/* ---- 2-D reprojection --------------------------------------------
 * return_2d_joints core, scripts/renderer.py:35-49 (pytorch3d 0.3.0 PerspectiveCameras, R = I,
 * T = cam, focal 5000/224 NDC, 224x224): joints (B,17,3), cam (B,3) -> screen xy (B,17,2).       */
int jrr_project_joints(const float* joints_dev, const float* cam_dev, float* j2d_dev, int batch, void* stream);
/* Enable (gt_j2d_dev != NULL) / disable the 2-D term  mean((gt_j2d - joints_2d)^2)/100  of the inner
 * loop (scripts/optimize.py:231-233,252); the camera translation then joins the Adam parameters
 * (optimize.py:201-202): cam (B,3) in place, cam_m / cam_v (B,3) its Adam state.                   */
int jrr_engine_set_reprojection(jrr_engine_t* e, const float* gt_j2d_dev, float* cam_dev, float* cam_m_dev,
                                float* cam_v_dev);
/* Camera pre-fit, scripts/optimize.py:187-199: n_steps Adam(lr) steps on cam against the 2-D joints
 * of the current pose.  The joints do not depend on the camera, so SMPL runs ONCE and the n_steps
 * run inside one kernel.  sq2d_dev (B, nullable): squared 2-D error at the last evaluation.        */
int jrr_camera_prefit(jrr_engine_t* e, const float* x6d_dev, const float* betas_dev, const float* gt_j2d_dev,
                      float* cam_dev, int n_steps, float lr, float* sq2d_dev, void* stream);

/* ---- soft silhouette (BASELINE configs[4]) ---------------------------
 * render_mesh(...) = Mesh_Renderer(224)(batch, verts*[-2,-2,2])[:, 3], scripts/optimize.py:77-85 +
 * scripts/mesh_renderer.py:23-79 (pytorch3d 0.3.0 rasteriser, blur_radius 0, 1 face per pixel,
 * SoftSilhouetteShader sigma 1e-4).  verts (B,6890,3), cam (B,3) -> alpha (B,224,224).
 * Image size: 224 (JRR_SIL_SIZE: what the reference's loop instantiates, scripts/optimize.py:110 `Mesh_Renderer(image_size=224)`)
 * or -- engines created with JRR_FLAG_SIL_256 -- 256, the constructor's own default (scripts/mesh_renderer.py:25); the camera's focal
 * length is 5000 / size (mesh_renderer.py:52-53).  The kernels are instantiated for these two sizes (the pixel-index packing of the
 * covered-pixel lists holds 16 bits: at most 256 x 256); the host mirror's Mesh_Renderer raises NotImplementedError for others. */
int jrr_silhouette_forward(jrr_engine_t* e, const float* verts_dev, const float* cam_dev, float* alpha_dev,
                           void* stream);
/* adjoint for the SAME inputs (must follow the forward): galpha (B,224,224) -> dverts (B,6890,3), dcam (B,3);
 * float atomics: summation order (last bits) varies between runs.                                    */
int jrr_silhouette_backward(jrr_engine_t* e, const float* galpha_dev, float* dverts_dev, float* dcam_dev,
                            void* stream);
/* pix_to_face of the most recent rasterisation on this engine (jrr_silhouette_forward, jrr_silhouette_loss_grad or the
 * last silhouette iteration of jrr_refine_run): the pytorch3d rasteriser's Fragments.pix_to_face at faces_per_pixel = 1
 * (scripts/mesh_renderer.py:34-38,59-63): (B,224,224) int32, -1 = background, else the index of the nearest face.      */
int jrr_silhouette_pix_to_face(jrr_engine_t* e, int32_t* pix_to_face_dev, void* stream);
/* Enable (mask_dev != NULL, (B,224,224)) / disable the term 100 * mean((silhouette - mask)^2) of the inner
 * loop (scripts/optimize.py:234-237,252); shares the camera parameter with jrr_engine_set_reprojection.
 * The engine caches sum(mask^2) per pose at the next jrr_refine_run: call this again after changing the mask's contents. */
int jrr_engine_set_silhouette(jrr_engine_t* e, const float* mask_dev, float* cam_dev, float* cam_m_dev,
                              float* cam_v_dev);

/* The silhouette term as the fused inner loop evaluates it (projection from the engine's internal vertex layout, nearest
 * face per pixel, loss and adjoint in one kernel), exposed as an operator: SMPL forward of (x6d, betas), then
 * sqsil_dev[b] = sum_pixels (silhouette - mask)^2 and the gradient of 100 * mean((silhouette - mask)^2) (mean over
 * batch_norm * 224 * 224; scripts/optimize.py:234-237,252) w.r.t. the SMPL vertices, dverts_dev (B,6890,3), and the
 * camera, dcam_dev (B,3).  mask_dev (B,224,224).  Outputs nullable.  Needs JRR_FLAG_SILHOUETTE | JRR_FLAG_KEEP_VERTS.
 * Bitwise reproducible (fixed-point accumulation), unlike jrr_silhouette_backward.                                   */
int jrr_silhouette_loss_grad(jrr_engine_t* e, const float* x6d_dev, const float* betas_dev, const float* cam_dev,
                             const float* mask_dev, float* sqsil_dev, float* dverts_dev, float* dcam_dev, void* stream);

/* ---- fused inner loop ---------------------------------------------------------------------
 * n_iters iterations of scripts/optimize.py:220-265 restricted to the engine's loss terms:
 * rot6d->R, SMPL, J-regress, pelvis-centre, MSE x10000 [+ pose-D x10] [+ shape-D x10],
 * analytic backward to (pose6d, betas), Adam(lr) in place.  x6d (B,24,6) holds orient (joint 0)
 * and pose (joints 1..23); adam_m/adam_v (B,154) = [144 pose | 10 betas]; step_dev counts
 * completed Adam steps (0 before the first).  sqerr_dev (B) receives the per-pose squared
 * joint error of the LAST iteration's forward (may be NULL).                                  */
int jrr_refine_run(jrr_engine_t* e, float* x6d_dev, float* betas_dev, const float* gt_centred_mm_dev,
                   float* adam_m_dev, float* adam_v_dev, int32_t* step_dev, float lr, int n_iters,
                   float* sqerr_dev, void* stream);

/* Adversarial loss values of the LAST iteration of the last jrr_refine_run, for the reference's log record
 * (scripts/optimize.py:246-250,323-337 `pose_discriminated_loss`, `shape_discriminated_loss`):
 * pose_disc_sq_dev[b] = sum_k (D(x_b)[k] - 1)^2 over the 25 outputs, shape_disc_sq_dev[b] = (SD(beta_b) - 1)^2.
 * Either pointer may be NULL; a non-NULL one needs its term active in the engine.                           */
int jrr_refine_aux_losses(jrr_engine_t* e, float* pose_disc_sq_dev, float* shape_disc_sq_dev, void* stream);

/* J step, scripts/optimize.py:300-312: gradient of mean((move_pelvis(joints)-gt/1000)^2) w.r.t.
 * the raw J_regressor for the current (detached) poses; dJ_dev (17,6890).  sqerr_dev (B, nullable): per-pose squared
 * joint error; joints_dev (B,17,3, nullable): the joints of this forward, i.e. of the regressor BEFORE its step (what
 * utils.evaluate reads at scripts/optimize.py:314-315).  The product behind it runs over the POSITIVE entries of
 * J*mask only (dJ is exactly zero elsewhere: relu'); rows with more than 128 of them switch to the dense product by
 * themselves (device-side decision, no synchronisation).                                                         */
int jrr_j_regressor_grad(jrr_engine_t* e, const float* x6d_dev, const float* betas_dev,
                         const float* gt_centred_mm_dev, float* dJ_dev, float* sqerr_dev, float* joints_dev, void* stream);

/* The second half of the J step in one call: step_dev += 1, torch.optim.Adam(lr) on the raw regressor J_dev (17,6890)
 * in place with the (all-reduced) gradient dJ_dev and the state m_dev / v_dev, then J*mask -> ReLU -> row-normalise into
 * the engine's layouts (= jrr_adam_step + jrr_engine_set_j_regressor).  Under data parallelism the J step is
 * jrr_j_regressor_grad -> ONE RCCL all-reduce of dJ -> jrr_j_step_apply (scripts/optimize.py:300-312).  mask_dev nullable. */
int jrr_j_step_apply(jrr_engine_t* e, float* J_dev, const float* dJ_dev, float* m_dev, float* v_dev, int32_t* step_dev,
                     float lr, const float* mask_dev, void* stream);

/* The J step's all-reduce restricted to the regressor's SUPPORT (data parallelism; scripts/optimize.py:300-312 under sharding).
 * dJ w.r.t. the raw regressor is exactly zero wherever J*mask <= 0 (ReLU'), so the ranks only need to exchange its values on the
 * positive entries: [17][128] floats (8 704 bytes) instead of 17 x 6890 (468 520 bytes).  Every rank holds the same regressor,
 * hence the same device-side support lists (ascending vertex order).
 *   jrr_j_support_info            positive entries per row (counts_host[17], nullable) and fits_host = 1 when every row has at
 *                                 most 128 of them.  SYNCHRONOUS (waits for `stream`); call once after jrr_engine_set_j_regressor:
 *                                 J steps only ever shrink the support (an entry at <= 0 gets no gradient), so the answer holds
 *                                 until the next jrr_engine_set_j_regressor from outside or a J step with ANOTHER mask pointer (a
 *                                 mask whose contents change in place must be re-announced through jrr_engine_set_j_regressor).
 *                                 Once the engine has been told that the
 *                                 support fits, EVERY J step on it (jrr_j_regressor_grad, jrr_refine_run_j_steps, the forward reuse)
 *                                 enqueues the support-restricted products only; before, the dense products are enqueued beside them
 *                                 and a device flag picks (no host knowledge needed, ~14 us of idle launches per J step).
 *   jrr_j_regressor_grad_support  = jrr_j_regressor_grad, gradient delivered as dJs_dev [17][128] (0 behind a row's count)
 *   jrr_j_step_apply_support      = jrr_j_step_apply with the (all-reduced) dJs_dev: the dense gradient is rebuilt on the device
 *                                 (zero outside the support, as the dense path has it) and torch's Adam runs over the whole
 *                                 (17,6890) parameter as before -- entries that left the support keep coasting on their momentum.
 *                                 Its forward (and that of the in-call steps of jrr_refine_run_j_steps) keeps the vertices of
 *                                 the 32-vertex tiles that hold a support entry only -- what the gradient product and the reusing
 *                                 iteration read -- not all 6890 x 3 x B of them: a jrr_find_joints_backward(dJ) afterwards
 *                                 needs its own jrr_find_joints_forward, and a jrr_engine_set_j_regressor from outside between
 *                                 this call and jrr_refine_run_after_j_step drops the cached forward (JRR_ERR_STATE there).
 * Both return JRR_ERR_STATE unless jrr_j_support_info has reported fits = 1 for the current regressor (fall back to the
 * dense pair).  Same results as the dense pair bit for bit on this rank; across ranks only the all-reduce's own summation
 * order can differ (none with two ranks).                                                                                  */
int jrr_j_support_info(jrr_engine_t* e, int32_t* counts_host, int32_t* fits_host, void* stream);
/* JRR_FLAG_SUPPORT_TILES: returns 1 when the next joint-loss iteration of jrr_refine_run* will run on the support's tiles only
 * (*n_tiles_host = their number, nullable), 0 when it will run all 216 (flag absent, jrr_j_support_info not asked or fits = 0,
 * a silhouette term set, folded mode on, a model without the joint-sparse kernels).  No reference counterpart: the reference
 * multiplies all 6890 vertices by the (17,6890) regressor, zeros included (scripts/utils.py:87-92).                          */
int jrr_engine_support_tiles(const jrr_engine_t* e, int32_t* n_tiles_host);
/* ... and returns 1 when those iterations run per support VERTEX (*n_vertices_host = the vertices the regressor reads, nullable): the
 * support then has at most 64 vertices and ONE launch per iteration takes a 32-pose group through chain forward, the SMPL forward
 * of the support vertices, the joint loss [+ the 2-D term] (scripts/utils.py:87-114, scripts/optimize.py:231-233), its backward, the
 * chain adjoint and Adam (scripts/optimize.py:220-265) -- preceded by the discriminator's four GEMM launches when it is on.  0: the tile lists above (or
 * all tiles) run as separate launches.  Same numbers up to the order of the sums.  JRR_SUPPORT_FUSED=0 in the environment of
 * jrr_j_support_info keeps the tile lists (verification).  No reference counterpart.                                              */
int jrr_engine_support_vertices(const jrr_engine_t* e, int32_t* n_vertices_host);
int jrr_j_regressor_grad_support(jrr_engine_t* e, const float* x6d_dev, const float* betas_dev, const float* gt_centred_mm_dev,
                                 float* dJs_dev, float* sqerr_dev, float* joints_dev, void* stream);
int jrr_j_step_apply_support(jrr_engine_t* e, float* J_dev, const float* dJs_dev, float* m_dev, float* v_dev, int32_t* step_dev,
                             float lr, const float* mask_dev, void* stream);

/* Forward reuse across the J step (needs JRR_FLAG_KEEP_VERTS).  jrr_j_regressor_grad evaluates SMPL on the current
 * poses; the inner iteration that follows evaluates it on the SAME poses (only the regressor has changed in between,
 * scripts/optimize.py:300-312 then :220-229).  jrr_refine_run_after_j_step is jrr_refine_run whose FIRST iteration
 * re-regresses its joints from the J step's stored vertices with the new regressor instead of repeating the forward --
 * the same arithmetic up to the summation order of the regressor product.  Reuse is explicit per call: by calling this
 * entry point the caller states that the previous calls on this engine were jrr_j_regressor_grad on the same x6d / betas
 * buffers (optionally followed by jrr_j_step_apply / jrr_engine_set_j_regressor) and that nothing has written those
 * buffers since.  What the engine can check it checks: any other entry point drops the cached forward, and the call then
 * returns JRR_ERR_STATE (also when the pointers differ) instead of differentiating through a stale forward.           */
int jrr_refine_run_after_j_step(jrr_engine_t* e, float* x6d_dev, float* betas_dev, const float* gt_centred_mm_dev,
                                float* adam_m_dev, float* adam_v_dev, int32_t* step_dev, float lr, int n_iters,
                                float* sqerr_dev, void* stream);

/* The inner loop WITH its J steps in one call, for a single process (no collective between the two halves of a J step):
 * n_iters iterations of jrr_refine_run; after every j_every-th one the J step of scripts/optimize.py:300-312
 * (jrr_j_regressor_grad into engine scratch, jrr_j_step_apply with J_dev / J_m_dev / J_v_dev / J_step_dev / j_lr /
 * mask_dev), the next iteration reusing its forward.  j_sqerr_dev (B, nullable): per-pose squared joint error of the
 * last J step.  after_j_step bit 0: as jrr_refine_run_after_j_step for the first iteration; bit 1 (value 2): NO forward reuse
 * inside the call either -- every iteration repeats its SMPL forward (the reference draws a new batch after each J step,
 * scripts/optimize.py:144-148, so nothing is shared there: what bench.py's headline times).  BASELINE configs[3]'s
 * "J_regressor step each iteration" at world size 1 is j_every = 1.  Needs JRR_FLAG_KEEP_VERTS.                       */
int jrr_refine_run_j_steps(jrr_engine_t* e, float* x6d_dev, float* betas_dev, const float* gt_centred_mm_dev,
                           float* adam_m_dev, float* adam_v_dev, int32_t* step_dev, float lr, int n_iters,
                           float* sqerr_dev, int j_every, float* J_dev, float* J_m_dev, float* J_v_dev,
                           int32_t* J_step_dev, float j_lr, const float* mask_dev, float* j_sqerr_dev, int after_j_step,
                           void* stream);

/* Loss history of the inner loop (scripts/optimize.py:255-261 prints the five weighted terms when i % 10 == 0): while
 * hist_dev != NULL, every `every`-th iteration run by jrr_refine_run* (counted from this call, first one included)
 * appends one record of 5 floats {loss_j2d/100, silhouette_loss*100, joint_loss*10000, pose_discriminated_loss*10,
 * shape_discriminated_loss*10} (inactive terms 0) to hist_dev, up to capacity_records.  Each value is this engine's
 * share of the global mean (local sum / batch_norm denominators): under data parallelism the ranks' records add up.
 * jrr_engine_loss_history_count returns the number of records written so far.  hist_dev == NULL disables.            */
int jrr_engine_set_loss_history(jrr_engine_t* e, float* hist_dev, int capacity_records, int every);
int jrr_engine_loss_history_count(const jrr_engine_t* e);

/* launch geometry: {B, BP, batch_norm, nvc, nvcb, nsplit, nsplitJ, flags, joint_sparse}; joint_sparse = 8 or 12: the LBS kernels
 * multiply each 32-vertex tile by its own joints only, that many slots per pass (exact: the skipped terms are zeros; a tile with
 * more joints runs a second pass, jrr_model_info), 0 = dense kernels (a tile with more than 16 joints, or JRR_DENSE_SKINNING=1 in
 * the environment of jrr_model_create) */
int jrr_engine_info(const jrr_engine_t* e, int32_t* out, int n);

/* Per-kernel timing of jrr_refine_run with HIP events recorded on the launch stream.
 * While enabled, every launch group of every iteration is bracketed by an event pair.
 * jrr_engine_profile_read synchronises on the recorded events and writes the MEAN duration in
 * milliseconds per launch of each class into ms_host[JRR_PROF_CLASSES] and the number of samples
 * into counts_host, then clears the recorded events.  Classes:
 *   0 chain forward (+ the pose discriminator's per-joint MLP, same launch)  1 k_lbs_fwd  2 k_joints_loss  3 k_lbs_bwd
 *   4 blend adjoint  5 pose discriminator: the four wide-layer products  6 k_shape_disc
 *   7 dF slab sum (+ per-joint MLP adjoint, same launch) and k_chain_bwd (+Adam)  8 silhouette (fwd+bwd) */
enum { JRR_PROF_CLASSES = 9 };
int jrr_engine_set_profiling(jrr_engine_t* e, int enabled);
int jrr_engine_profile_read(jrr_engine_t* e, float* ms_host, int32_t* counts_host);
/* Shader-clock probe of the dominant kernel (k_lbs_fwd), recorded while profiling is on by workgroup 0 / wave 0
 * of the last launch inside jrr_refine_run: out[0] = shader clocks the wave was resident (s_memtime),
 * out[1] = MFMA instructions it issued, out[2] = waves resident per SIMD, out[3] = issue clocks per MFMA,
 * out[4] = the same interval in ns (s_memrealtime).  MFMA-pipe occupancy = out[1]*out[2]*out[3]/out[0];
 * sustained shader clock = out[0]/out[4] GHz.  out_host holds 5 values.  Synchronous (device -> host copy).
 * (48 of the 423 instructions per tile of the joint-sparse kernel are the 33-clock four-block form v_mfma_f32_16x16x1_4b_f32:
 * the occupancy formula, which prices every instruction at out[3] = 64 clocks, reads ~5 % high for it.)               */
int jrr_engine_probe_read(jrr_engine_t* e, int64_t* out_host);

#ifdef __cplusplus
}
#endif
#endif /* JRR_H */
