// Shared constants, layouts and MFMA helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include "../../include/jrr.h"

namespace jrr {

constexpr int V = 6890;        // SMPL vertices
constexpr int VT = 216;        // vertex tiles of 32
constexpr int VP = VT * 32;    // 6912 padded vertices
constexpr int NJ = 24;         // SMPL joints
constexpr int NH = 17;         // H36M joints
constexpr int NHP = 18;        // H36M joints padded to the MFMA K granule (2)
constexpr int NB = 10;         // betas
constexpr int KF = 218;        // blend features: 207 pose + 10 shape + 1 template
constexpr int KFP = 224;       // padded to 7 tiles of 32
constexpr int KCH = 32;        // feature rows per staged chunk of the blend basis
constexpr int NKCH = KFP / KCH;  // 7
constexpr int BT = 32;         // poses per wave tile
constexpr int BG = 128;        // poses per forward workgroup (4 waves)
constexpr int FOLD_MJ = 512;   // (H36M joint, SMPL joint) pairs 17*24 = 408, padded
constexpr int FOLD_M = 1280;   // (i, j, c) triples 1224, padded to 10 x 128
constexpr int NPARAM = 154;    // 144 pose6d + 10 betas per pose
constexpr int KJS_MAX = 12;    // joint SLOTS per vertex tile and pass the joint-sparse skinning kernels are built for (8 and 12)
constexpr int KJS_TILE_MAX = 16;   // most joints one tile may have under the joint-sparse kernels (two passes; the backward
                                  // kernel's dA products run over 16-row joint windows); one tile above it: dense kernels

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact fp32.
//   A operand: lane l holds A[m = l&31][k = l>>5]
//   B operand: lane l holds B[k = l>>5][n = l&31]
//   C/D:       reg r of lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// (verified on hardware by tools/probe/mfma_probe.hip)
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// XCD-aware bijective remap of a linear block id: blocks with equal (id % 8) share an XCD's L2,
// so give each XCD a contiguous range of logical work (guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int id, int n) {
  int xcd = id & 7, q = n >> 3, r = n & 7;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (id >> 3);
}

// LDS-DMA (global_load_lds_dwordx4): one wave-instruction copies 64 lanes x 16 B from per-lane global
// addresses into LDS [dst, dst + 1 KB) -- the LDS address is wave-uniform base + lane*16 (verified by
// tools/probe/dma_probe.hip).  No VGPR round trip; completion is tracked by vmcnt.
#define JRR_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define JRR_GLB(p) ((const __attribute__((address_space(1))) void*)(p))

// Row pointer of a [rows][BP] array for a WAVE-UNIFORM row, kept in an SGPR pair: the access is then
// `global_* v_lane_offset, v_data, s[row]` (scalar address arithmetic) instead of five VALU instructions, two of
// them quarter-rate 64-bit multiplies, per element.  The per-lane part (4 * half rows + pose column) is a
// loop-invariant 32-bit offset.
__device__ __forceinline__ float* urow(float* base, size_t row, int BP) {
  float* p = base + row * (size_t)BP;
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ const float* urow(const float* base, size_t row, int BP) {
  const float* p = base + row * (size_t)BP;
  asm volatile("" : "+s"(p));
  return p;
}
// "Row-quad" layout of the per-vertex arrays that only travel between the LBS kernels (v_posed):
//     element (row v, pose b) of plane c lives at  ((c * VP/4 + v/4) * BP + b) * 4 + v % 4.
// Registers 4g .. 4g+3 of a lane of the accumulator layout are the rows 8g + 4 half + {0,1,2,3} = ONE row quad
// (index 2g + half) of that lane's pose, i.e. 16 contiguous bytes: a tile moves as four dwordx4 per lane (a wave
// instruction = two contiguous 512-byte runs) instead of sixteen dword accesses.
__device__ __forceinline__ f32x4* quad_ptr(float* base, size_t plane_quad0, int g, int BP, unsigned lane_q) {
  float* p = base + ((plane_quad0 + 2 * g) * (size_t)BP) * 4;
  asm volatile("" : "+s"(p));
  return reinterpret_cast<f32x4*>(p) + lane_q;
}
__device__ __forceinline__ const f32x4* quad_ptr(const float* base, size_t plane_quad0, int g, int BP, unsigned lane_q) {
  const float* p = base + ((plane_quad0 + 2 * g) * (size_t)BP) * 4;
  asm volatile("" : "+s"(p));
  return reinterpret_cast<const f32x4*>(p) + lane_q;
}
// uniform part of acc_row(q, half) = (q & 3) + 8 (q >> 2) + 4 half
__device__ __forceinline__ constexpr int acc_row_u(int q) { return (q & 3) + 8 * (q >> 2); }

// Compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1.  The stage loop of k_lbs_fwd needs its index as
// a CONSTANT in every stage (trip counts, register indices, which stores a barrier may leave in flight); relying on the
// unroller for that worked for three of the four template variants and silently left the fourth one rolled.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Workgroup barrier that waits for this wave's LDS operations and for all but its N youngest vector-memory operations.
// `__syncthreads()` drains vmcnt to 0, i.e. it also waits for every global STORE the wave has issued (write latency
// 1-2 us under load) and for register prefetches that are not needed yet.  LDS-DMA, loads and stores retire in issue
// order, so when the copies that must have landed are OLDER than N later operations, waiting for vmcnt(N) is enough;
// LDS-DMA and stores stay in flight across s_barrier (MI355X_MICROARCH.md, "Co-residence costs").
template <int N>
__device__ __forceinline__ void barrier_keep_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// torch.optim.Adam, single-tensor formula: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps),
// bias corrections evaluated in double like torch's Python scalars
struct AdamScalars { float step_size, bc2_sqrt, beta1, beta2, eps; };
__device__ __forceinline__ AdamScalars adam_scalars(int step, float lr, float beta1, float beta2, float eps) {
  AdamScalars s;
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  s.step_size = (float)((double)lr / bc1);
  s.bc2_sqrt = (float)sqrt(bc2);
  s.beta1 = beta1; s.beta2 = beta2; s.eps = eps;
  return s;
}
__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, const AdamScalars& s) {
  m = m * s.beta1 + (1.f - s.beta1) * g;
  v = v * s.beta2 + (1.f - s.beta2) * g * g;
  float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
  return p - s.step_size * (m / denom);
}

struct Parents {
  int p[NJ];       // parent joint (p[0] = -1), parents precede children
  int depth[NJ];   // tree depth of each joint (root = 0)
  int maxd;        // deepest level
  // children of joint j, ascending: child[child_off[j] .. child_off[j + 1])
  unsigned char child_off[NJ + 1];
  unsigned char child[NJ];
};

// ---- device-resident SMPL model ------------------------------------------------------------
struct Model {
  float* Dk;    // [VT][KFP/4][3][32][4]  blend basis in K-quads inside a vertex tile (a K chunk of a tile = one contiguous 12 KB DMA)
  float* Dn;    // [3][VP][KFP]      blend basis, feature-contiguous (folded-regressor tables)
  float* Dq;    // [3][VP/4][KFP][4] blend basis in vertex quads (A operand of the blend adjoint, k_blend_adjoint)
  float* Wjv;   // [VT][24][32]      skinning weights W^T tile  (lane = vertex)
  float* Wvj;   // [VT][32][32]      skinning weights tile [vertex][joint padded to 32] (lane = joint)
  // Joint-sparse skinning: every SMPL vertex has <= 4 influences, and a tile of 32 consecutive vertices usually few
  // joints in total.  kjs = 8 or 12 = the joint SLOTS a tile's skinning products run over per pass (0: the dense kernels
  // run: some tile has more than KJS_TILE_MAX joints, or JRR_DENSE_SKINNING=1).  Per tile: the ascending list of its joints
  // (NJ slots, padded with joint 0 and zero weights), the W^T rows of exactly those, and the joint count.  PER-TILE CLASSES:
  // a tile with more than kjs joints ("wide") runs a second pass over slots kjs .. 2 kjs - 1 in the forward kernel and
  // extra K steps of the T recompute in the backward kernel -- it costs itself, not the model.
  float* Wc;    // [VT][NJ][32] (+ one tile of slack)
  int* jl;      // [VT][NJ]
  int* tnj;     // [VT] joints of the tile
  int kjs;
  int wide_tiles, most_joints, permuted;      // tiles above kjs joints; most joints of any tile; internal order != file order
  int hint_applied;                           // vertices stored first on the caller's hint (jrr_model_create_hinted)
  int tile_hist[NJ + 1];                      // tiles by joint count (jrr_model_info)
  // Joint WINDOWS of the backward kernel's dA products (kjs > 0).  dA_{r,c}[j][pose] = sum_v W[v][j] (...) only has
  // non-zero rows for the joints that skin the tile; it is accumulated over the tiles of a vertex chunk, so the rows must
  // mean the same joints from tile to tile: consecutive tiles are grouped (greedily, at model upload) into SEGMENTS whose
  // joint union is <= 16, the product runs over the segment's 16-row window (v_mfma_f32_16x16x1_4b_f32: half the issue
  // time of 32 padded joint rows), and a workgroup adds its accumulators into its dA slab when the segment changes.
  float* W16;   // [VT][16][36]      W[v][window joint n] of the tile's segment, [n][v] with rows padded to 36 floats
  int* segid;   // [VT]              segment of each tile
  int* segj;    // [VT][16]          the window of the tile's segment: joint of row n, -1 = unused row
  int bwd16;    // kjs > 0: run the symmetric 16-pose backward kernel k_lbs_bwd16 (0 with JRR_BWD16=0 in the environment
                // of jrr_model_create: the role kernel k_lbs_bwd<., role_kjs>)
  int role_kjs; // joint slots of the role kernel's T recompute: kjs when no tile is wide, else 0 (its dense form)
  // Internal vertex order.  Nothing inside the LBS path depends on WHICH vertex sits in which row (every consumer sums
  // over vertices), so jrr_model_create may store the vertices in an order that makes the tiles joint-coherent (sorted
  // by their influencing joints) when the file order does not fit kjs.  p2v / v2p are NULL for the identity; otherwise
  // p2v[row] = vertex of that row (-1: padding), v2p[vertex] = row.  They are applied where vertex identity is visible:
  // J_regressor columns in / dJ out, the (B,6890,3) vertices out / their adjoint in, and the face indices of the fused
  // rasteriser.
  int* p2v;     // [VP] or NULL
  int* v2p;     // [V] or NULL
  float* Jt;    // [24][3]           rest joints of the template
  float* JS;    // [24][3][10]       rest-joint shape directions
  Parents parents;
  int* faces;   // [nfaces][3] (device) or NULL; vertex indices of the caller's mesh
  int* faces_int;   // the same faces in internal row indices (only when p2v != NULL; fused rasteriser)
  unsigned* faces_pk;      // the rasteriser's table: one 8-byte record per face, {i0 | i1 << 13, i2} (vertex indices < 8192)
  unsigned* faces_int_pk;  // ... in internal row indices
  int nfaces;
};

inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

}  // namespace jrr

struct jrr_model {
  jrr::Model d;
  void* base;      // the model buffer (jrr_model_bytes): the caller's (jrr_model_create_in) or the library's own
  bool owns_base;
  int* faces_area; // tail of the buffer: faces [MAX_FACES][3], faces in internal row indices [MAX_FACES][3], the two packed tables [MAX_FACES][2] each
  int* v2p_host;   // host copy of Model::v2p (jrr_model_set_faces), NULL for the identity order
};

// error plumbing (api.hip)
void jrr_set_error(const char* fmt, ...);
#define JRR_HIP(expr)                                                                 \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      jrr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return JRR_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)
