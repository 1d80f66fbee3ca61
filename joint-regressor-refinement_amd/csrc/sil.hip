// Soft-silhouette renderer of the posed mesh (BASELINE config 5):
//   render_mesh      /root/reference/scripts/optimize.py:77-85   (flip x,y; x2; alpha channel)
//   Mesh_Renderer    /root/reference/scripts/mesh_renderer.py:23-79
//       pytorch3d 0.3.0 PerspectiveCameras(T=cam, focal=5000/224) -> MeshRasterizer(224, blur_radius=0,
//       faces_per_pixel=1) -> SoftSilhouetteShader(sigma=1e-4): alpha = sigmoid(d / sigma) on covered
//       pixels, d = squared NDC distance from the pixel centre to the nearest edge of the NEAREST face.
// pytorch3d is absent from the image: restated from the published algorithm (oracle/silhouette_port.py),
// parity unpinned.
//
// Kernels (one mesh = 6890 vertices / 13776 faces of ~1 pixel each at 224x224):
//   k_sil_project  per (pose, vertex): world -> (x_ndc, y_ndc, view depth Z)
//   k_sil_raster   per pose (one 1024-thread workgroup, ~155 KB LDS): projected vertices resident in LDS, face
//                  indices in registers; the pixel box of the projected mesh is swept in strips of 8960 pixels: each thread tests the pixel centres
//                  inside its faces' bounding boxes and keeps the nearest face per pixel with a 64-bit atomicMin on
//                  an LDS z-buffer keyed (depth bits << 32 | face index).  Background pixels are finished per
//                  strip; covered pixels (8-9 %) go to a per-pose list (pixel << 14 | face) and are resolved densely
//                  at the end: alpha, squared error against the target mask and -- <true>, the fused inner loop --
//                  the adjoint of the loss term, accumulated per vertex in NDC space in the z-buffer's LDS
//   k_sil_bwd      per pose: the same adjoint for an arbitrary upstream gradient (standalone API), from the list
// Measured costs that shaped this (MI355X): LDS float atomics retire ~1 lane per 2.5 clocks (so: 4 per covered
// pixel, NDC space, not 6 in world space); the face sweep is VALU-bound (max-over-lanes bounding box trips).
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

// Image size S: 224 (scripts/optimize.py:110 Mesh_Renderer(image_size=224)) or 256 (the reference constructor's default,
// scripts/mesh_renderer.py:25) -- a template parameter of the kernels; the focal length follows it (5000 / S, mesh_renderer.py:52-53).
constexpr int SIL_MAX = 256;
constexpr int SIL_ZPIX = 40 * 224;       // z-buffer capacity in pixels (8960 * 8 B = 70 KB); strips cover the mesh's pixel box
constexpr int SIL_RT = 1024;             // threads of the raster workgroup (one workgroup per pose and per CU: LDS-bound)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
constexpr int SIL_FPT = 14;              // faces per thread, kept in registers (14 * 1024 >= 13776)
constexpr int SIL_FBITS = 14;             // covered-pixel list entry = pixel << 14 | face (faces < 16384, pixels < 65536)
constexpr int SIL_EB = 6;                 // list entries fetched together per thread in the resolve
static_assert((3 * V + 2) % 2 == 0, "z-buffer alignment");
static_assert(V <= 8192 && SIL_MAX <= 256, "face records: 13-bit vertex indices, 8-bit pixel rows");
static_assert(V * 8 <= SIL_ZPIX * 8, "adjoint accumulators must fit the z-buffer");
constexpr int SIL_VPAD = 3 * V + 2;      // floats of the LDS vertex arrays, padded so the u64 z-buffer is 8-byte aligned
// BlendParams sigma = 1e-4 (mesh_renderer.py:28); only its reciprocal is used
constexpr float SIL_ISIGMA = 1e4f;
constexpr float SIL_EPS = 1e-8f;

struct alignas(16) NdcV { float x, y, z, pad; };

__global__ void k_sil_project(const float* __restrict__ verts, int ldv, const float* __restrict__ cam,
                              NdcV* __restrict__ ndc, int B, float SIL_F) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over B*V
  if (idx >= B * V) return;
  const int b = idx / V, v = idx % V;
  const float* p = verts + (size_t)b * ldv + v * 3;
  const float X = -2.f * p[0] + cam[(size_t)b * 3], Y = -2.f * p[1] + cam[(size_t)b * 3 + 1];
  const float Z = 2.f * p[2] + cam[(size_t)b * 3 + 2];
  NdcV o;
  o.x = SIL_F * X / Z; o.y = SIL_F * Y / Z; o.z = Z; o.pad = 0.f;
  ndc[idx] = o;
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}
// pixel centre 1 - fl32((2 xi + 1)/224).  The fp32 quotient is formed as a rounded f64 product: (2 xi + 1)/224 is
// never within 2^-53 (relative) of an fp32 rounding boundary, so this equals the IEEE fp32 division bit for bit
// (checked for all 224 indices in tests/test_host_logic.py) at 3 instructions instead of 12.
// (S = 256: 1 / S and the product are exact, the statement holds trivially)
template <int S>
__device__ __forceinline__ float pix_x(int xi) { return 1.f - (float)((double)(2 * xi + 1) * (1.0 / S)); }

// squared distance from p to segment a-b, and the clamped parameter t
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by, float& t) {
  const float dx = bx - ax, dy = by - ay;
  const float l2 = dx * dx + dy * dy;
  t = (l2 <= SIL_EPS) ? 0.f : fminf(fmaxf(((px - ax) * dx + (py - ay) * dy) / l2, 0.f), 1.f);
  const float qx = ax + t * dx, qy = ay + t * dy;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

// nearest edge (first minimum) of the triangle (x[k], y[k]): returns the squared distance; the edge runs from
// corner ka to corner (ka + 1) % 3 and tt is the clamped parameter of the closest point on it
__device__ __forceinline__ float sil_nearest_edge(float px, float py, const float x[3], const float y[3], int& ka, float& tt) {
  float t1, t2;
  float dist = seg_dist2(px, py, x[0], y[0], x[1], y[1], tt);
  const float d1 = seg_dist2(px, py, x[1], y[1], x[2], y[2], t1);
  const float d2 = seg_dist2(px, py, x[2], y[2], x[0], y[0], t2);
  ka = 0;
  if (d1 < dist) { dist = d1; tt = t1; ka = 1; }
  if (d2 < dist) { dist = d2; tt = t2; ka = 2; }
  return dist;
}
// alpha = sigmoid(dist / sigma) (sigma = 1e-4: 1/sigma = 1e4 is exact in fp32)
__device__ __forceinline__ float sil_alpha(float dist) { return 1.f / (1.f + expf(-dist * SIL_ISIGMA)); }

// One workgroup per pose.  The pose's projected vertices (3 x 6890 floats = 81 KB) are loaded ONCE into LDS and
// each thread keeps the vertex indices of its <= 14 faces (and their pixel-row ranges) in registers, so the only
// global traffic of the face passes is zero: the mesh's pixel box is swept in strips of <= 8960 pixels (LDS z-buffer 70 KB), and a
// face is only touched in the strip(s) its rows fall into.
//
// ADJ (the fused inner loop): the adjoint of scale * sum((alpha - mask)^2)/2... i.e. g_alpha = scale * (alpha - mask)
// is taken in the same kernel: after the last strip the z-buffer's LDS holds the vertex-adjoint accumulators,
// written out pose-major.
// ADJ also selects the vertex interface: <false> (stand-alone forward) reads the projected records `ndc` written by
// k_sil_project; <true> (fused loop) reads the pose's vertices straight from the LBS kernels' row-quad buffer
// VQ [3][VP/4][BP][4] (16-byte pieces, one per vertex quad and plane), projects them itself, and at the end OVERWRITES the
// same pieces with the vertex adjoint -- the layout k_lbs_bwd<2> consumes.  No pose-major copies of the vertices or of
// their adjoint exist (two transposes of 0.35 ms and 0.8 GB of buffers gone).  A 128-byte line of VQ holds 8 consecutive
// poses, so consecutive poses are given to the SAME XCD (blockIdx % 8 selects the XCD) and meet in its L2.
// the lane index, recomputed from the hardware lane count on an operand the compiler cannot see through: neither hoisted out of a
// loop nor kept alive across one
__device__ __forceinline__ int fresh_lane() {
  unsigned z = 0u;
  asm volatile("" : "+v"(z));
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

template <bool ADJ, int SIL>
__device__ __forceinline__ void sil_raster_pose(const int b, const int wave_s, const NdcV* __restrict__ ndc, const uint2* __restrict__ faces,
                                                int nfaces, const float* __restrict__ mask,
                                                unsigned* __restrict__ cover, int* __restrict__ ncover,
                                                float* __restrict__ alpha_out, float* __restrict__ sqsil,
                                                float scale, float* __restrict__ VQ, int BP,
                                                const float* __restrict__ cam, int B,
                                                float* __restrict__ gcam, int accumulate_cam,
                                                const float* __restrict__ smask, const float* __restrict__ VPM) {
  constexpr float SIL_F = 5000.f / (float)SIL;        // NDC focal length (mesh_renderer.py:52-53)
  // every per-thread quantity below derives from `tix`, a laundered copy of the thread index: loop-invariant address arithmetic and
  // loads (the face indices!) then cannot be hoisted out of the persistent kernel's pose loop, where they would cost ~140 spilled registers
  // (formed from the hardware lane count and the wave index in a scalar register, so that no copy of threadIdx.x has to stay alive
  // across the poses of the persistent kernel)
  int tix = wave_s * 64 + fresh_lane();
  extern __shared__ unsigned long long smem64[];      // 8-byte aligned whatever static LDS precedes it
  float* vx = reinterpret_cast<float*>(smem64);       // [V]; the vertex arrays first: ds offsets stay < 64 KB
  unsigned long long* zb = smem64 + SIL_VPAD / 2;     // [SIL_ZPIX]
  float* vy = vx + V;
  float* vz = vx + 2 * V;
  __shared__ float red[4 * (SIL_RT / 64)];   // wave partials of the four end-of-pose sums
  __shared__ float pxt[SIL_MAX];      // pixel centres
  __shared__ int ncov;
  if (ADJ && b >= B) {                       // padded pose (whole workgroup): its adjoint is zero
    f32x4* P4 = reinterpret_cast<f32x4*>(VQ);
    for (int q = tix; q < 3 * (VP / 4); q += SIL_RT) P4[(size_t)q * BP + b] = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  // (the fused kernel is PERSISTENT: this function runs once per pose of the workgroup's list.  The face indices below do not depend
  // on the pose -- laundering the pointer keeps the compiler from hoisting their loads out of the pose loop)
  asm volatile("" : "+s"(faces));
  const NdcV* vb = ndc + (size_t)b * V;
  unsigned* lst = cover + (size_t)b * SIL * SIL;
  __shared__ float bbp[4][SIL_RT / 64];   // per-wave partial bounding box of the projected vertices
  if (tix == 0) ncov = 0;
  if (tix < SIL) pxt[tix] = pix_x<SIL>(tix);
  float bxn = 3e38f, bxx = -3e38f, byn = 3e38f, byx = -3e38f;
  // after the set-up a face is TWO registers: fr0 = i0 | i1 << 13, fr1 = i2 | first row << 13 | last row << 21 (vertex indices < 8192,
  // pixel rows < 256) -- 28 registers for the 14 faces instead of 56, which is what lets the persistent kernel stay inside 128
  unsigned fr0[SIL_FPT], fr1[SIL_FPT];
  f32x4* VQ4 = reinterpret_cast<f32x4*>(VQ);
  float tcam[3] = {0.f, 0.f, 0.f};
  if (ADJ) {
    tcam[0] = cam[(size_t)b * 3]; tcam[1] = cam[(size_t)b * 3 + 1]; tcam[2] = cam[(size_t)b * 3 + 2];
    // VPM (round 5): the forward kernel left this pose's vertices contiguous, [3][VP] -- three coalesced 16-byte loads per thread
    // and quad instead of three pieces out of three different cache lines (the set-up phase was 25.6 of a pose's 80 us)
    const f32x4* PM4 = VPM ? reinterpret_cast<const f32x4*>(VPM) + (size_t)b * 3 * (VP / 4) : nullptr;
    auto project4 = [&](int q, const f32x4& t0, const f32x4& t1, const f32x4& t2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = 4 * q + u;
        if (v < V) {      // scripts/optimize.py:80-82 flip / scale + mesh_renderer.py:52-57 camera (k_sil_project's arithmetic)
          const float X = -2.f * t0[u] + tcam[0], Y = -2.f * t1[u] + tcam[1], Z = 2.f * t2[u] + tcam[2];
          const float xn = SIL_F * X / Z, yn = SIL_F * Y / Z;
          vx[v] = xn; vy[v] = yn; vz[v] = Z;
          bxn = fminf(bxn, xn); bxx = fmaxf(bxx, xn); byn = fminf(byn, yn); byx = fmaxf(byx, yn);
        }
      }
    };
    static_assert(VP / 4 <= 2 * SIL_RT, "two quads per thread cover the pose");
    // both quads of a thread are requested before the first is projected: one round trip to the vertices, not two
    const int q0 = tix, q1 = tix + SIL_RT;
    const bool h1 = q1 < VP / 4;
    const int q1c = h1 ? q1 : q0;
    auto ldq = [&](int plane, int q) { return PM4 ? PM4[plane * (VP / 4) + q] : VQ4[((size_t)plane * (VP / 4) + q) * BP + b]; };
    const f32x4 t0 = ldq(0, q0), t1 = ldq(1, q0), t2 = ldq(2, q0);
    const f32x4 u0 = ldq(0, q1c), u1 = ldq(1, q1c), u2 = ldq(2, q1c);
    project4(q0, t0, t1, t2);
    if (h1) project4(q1, u0, u1, u2);
  } else {
    for (int v = tix; v < V; v += SIL_RT) {
      const NdcV p = vb[v];
      vx[v] = p.x; vy[v] = p.y; vz[v] = p.z;
      bxn = fminf(bxn, p.x); bxx = fmaxf(bxx, p.x); byn = fminf(byn, p.y); byx = fmaxf(byx, p.y);
    }
  }
  if (alpha_out)                      // stand-alone forward: background alpha; covered pixels are overwritten in pass 2
    for (int i = tix; i < SIL * SIL; i += SIL_RT) alpha_out[(size_t)b * SIL * SIL + i] = 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    bxn = fminf(bxn, __shfl_xor(bxn, o)); bxx = fmaxf(bxx, __shfl_xor(bxx, o));
    byn = fminf(byn, __shfl_xor(byn, o)); byx = fmaxf(byx, __shfl_xor(byx, o));
  }
  if ((tix & 63) == 0) {
    const int w = tix >> 6;
    bbp[0][w] = bxn; bbp[1][w] = bxx; bbp[2][w] = byn; bbp[3][w] = byx;
  }
  __syncthreads();
  float err = 0.f;
  // The z-buffer only ever holds the pixel bounding box of the projected mesh (a person covers 8-9 % of the crop and
  // its box ~23 %): strips of as many box rows as fit 70 KB -- 1.8 strips per pose on the synthetic batches instead of the
  // 6 a full-width sweep needs, and every face-sweep / clear / background pass is per strip.  Pixels outside the box
  // are background by construction: their squared error is sum(mask^2) over the image (per pose, computed once when the
  // mask is set) minus the covered pixels' share.
  int bx0, bx1, by0, by1;
  {
    float xn = 3e38f, xx = -3e38f, yn = 3e38f, yx = -3e38f;
#pragma unroll
    for (int w = 0; w < SIL_RT / 64; ++w) {
      xn = fminf(xn, bbp[0][w]); xx = fmaxf(xx, bbp[1][w]); yn = fminf(yn, bbp[2][w]); yx = fmaxf(yx, bbp[3][w]);
    }
    // pixel centres inside [min, max] (same formula as the per-face boxes, which are subsets); +-inf / NaN clamp to the image
    const float fx0 = (SIL * (1.f - xx) - 1.f) * 0.5f - 1e-3f, fx1 = (SIL * (1.f - xn) - 1.f) * 0.5f + 1e-3f;
    const float fy0 = (SIL * (1.f - yx) - 1.f) * 0.5f - 1e-3f, fy1 = (SIL * (1.f - yn) - 1.f) * 0.5f + 1e-3f;
    bx0 = (fx0 > 0.f) ? (int)ceilf(fminf(fx0, (float)SIL)) : 0;
    bx1 = (fx1 < (float)(SIL - 1)) ? (int)floorf(fmaxf(fx1, -1.f)) : SIL - 1;
    by0 = (fy0 > 0.f) ? (int)ceilf(fminf(fy0, (float)SIL)) : 0;
    by1 = (fy1 < (float)(SIL - 1)) ? (int)floorf(fmaxf(fy1, -1.f)) : SIL - 1;
  }
  const int bw = bx1 - bx0 + 1;
  const int rows_per = (bw > 0) ? SIL_ZPIX / bw : SIL;
  // the first strip's z-buffer is cleared HERE, under the face records' index loads (which wait on the L2 with nothing else to do),
  // not behind them
  if (bw > 0) {
    const int npx0 = (min(by0 + rows_per, by1 + 1) - by0) * bw;
    for (int i = tix; i < npx0; i += SIL_RT) zb[i] = ~0ull;
  }
  // this thread's faces: vertex indices -> pixel-row range of the face -> the two-register record
  {
    uint2 fi[SIL_FPT];         // {i0 | i1 << 13, i2}: one 8-byte load per face
#pragma unroll
    for (int u = 0; u < SIL_FPT; ++u) {
      const int f = min(tix + u * SIL_RT, nfaces - 1);      // (a slot past the last face reads the last face; it is marked empty below)
      fi[u] = faces[f];
    }
#pragma unroll
    for (int u = 0; u < SIL_FPT; ++u) {
      const int f = tix + u * SIL_RT;
      const float y0v = vy[fi[u].x & 8191u], y1v = vy[fi[u].x >> 13], y2v = vy[fi[u].y];
      const float ymax = fmaxf(y0v, fmaxf(y1v, y2v)), ymin = fminf(y0v, fminf(y1v, y2v));
      // pixel centres inside the bounding box: yf = 1 - (2 yi + 1)/H in [ymin, ymax]  <=>
      // yi in [ceil((H (1 - ymax) - 1)/2), floor((H (1 - ymin) - 1)/2)]; 1e-3 px of slack, the inside test is exact.
      int ylo = (int)ceilf((SIL * (1.f - ymax) - 1.f) * 0.5f - 1e-3f), yhi = (int)floorf((SIL * (1.f - ymin) - 1.f) * 0.5f + 1e-3f);
      ylo = max(ylo, 0); yhi = min(yhi, SIL - 1);
      if (f >= nfaces || !(ymax == ymax) || !(ymin == ymin) || ylo > yhi) { ylo = 1; yhi = 0; }     // empty range
      fr0[u] = fi[u].x;
      fr1[u] = fi[u].y | ((unsigned)ylo << 13) | ((unsigned)yhi << 21);
    }
  }
  for (int y0 = by0; y0 <= by1 && bw > 0; y0 += rows_per) {
    const int y1 = min(y0 + rows_per, by1 + 1);                        // rows [y0, y1) of the box columns [bx0, bx1]
    const int npx = (y1 - y0) * bw;
    if (y0 != by0)
      for (int i = tix; i < npx; i += SIL_RT) zb[i] = ~0ull;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SIL_FPT; ++u) {
      unsigned r0 = fr0[u], r1 = fr1[u];
      asm volatile("" : "+v"(r0), "+v"(r1));     // unpacked HERE, per strip: hoisted out of the strip loop the fields would be registers again
      const int ylo = max((int)((r1 >> 13) & 255u), y0), yhi = min((int)(r1 >> 21), y1 - 1);
      if (ylo > yhi) continue;                                         // no pixel centre of this face in the strip
      const int f = tix + u * SIL_RT;
      const int i0 = (int)(r0 & 8191u), i1 = (int)(r0 >> 13), i2 = (int)(r1 & 8191u);
      const float ax = vx[i0], ay = vy[i0], az = vz[i0];
      const float bx = vx[i1], by = vy[i1], bz = vz[i1];
      const float cx = vx[i2], cy = vy[i2], cz = vz[i2];
      const float xmax = fmaxf(ax, fmaxf(bx, cx)), xmin = fminf(ax, fminf(bx, cx));
      int xlo = (int)ceilf((SIL * (1.f - xmax) - 1.f) * 0.5f - 1e-3f), xhi = (int)floorf((SIL * (1.f - xmin) - 1.f) * 0.5f + 1e-3f);
      xlo = max(xlo, bx0); xhi = min(xhi, bx1);
      if (xlo > xhi) continue;
      const float area = edge_fn(cx, cy, ax, ay, bx, by);
      if (!(fabsf(area) > SIL_EPS)) continue;                    // also rejects NaN
      if (fmaxf(az, fmaxf(bz, cz)) < 0.f) continue;              // behind the camera
      const float inv = 1.f / area;
      // Barycentric coordinates as affine functions of the pixel centre RELATIVE TO CORNER a (differences of the size of
      // the face: no cancellation): w0 = 1 - w1 - w2 at a, w1 = edge(p; c, a) / area, w2 = edge(p; a, b) / area with
      // edge(p; u, v) = (p - u) x (v - u).  10 instructions per pixel instead of 21 -- the pixel loop is the sweep's cost.
      const float a1x = (ay - cy) * inv, a1y = -(ax - cx) * inv;         // d w1 / d (px, py)
      const float a2x = (by - ay) * inv, a2y = -(bx - ax) * inv;         // d w2 / d (px, py)
      const float dzb = bz - az, dzc = cz - az;
      // One flat loop over the bounding box in blocks of 2 x 2 pixel centres (x fastest).  Stamps and no-op variants (round 5) put
      // 25 of the sweep's 32 us in this loop and showed it bound by instruction ISSUE (four waves per SIMD, ~2.6 clocks per
      // instruction), not by latency, the LDS gathers or the atomics: what counts is instructions per trip x trips of the
      // longest lane.  2 x 2 blocks cut the trips from 2027 to 692 per pose; per block the two columns go through packed
      // fp32 arithmetic (v_pk_fma_f32: one instruction for both), the three edge tests are one v_min3 + compare, and a block
      // hanging over the box's edge REPEATS its last column / row instead of testing bounds (atomicMin is idempotent).
      // Per pixel the arithmetic is what it was one pixel at a time (dy * a1y shared by a row, nothing incremental).
      // The loop variables are BYTE offsets into pxt (4 x the pixel index): no shifts for the table reads, one shift-add for a
      // z-buffer address, one multiply-add for a row's base.
      {
        const char* const pxb = reinterpret_cast<const char*>(pxt);
        // (the z-buffer is addressed through its 32-bit LDS offset: + row * bw * 8 + column * 8, the row term from a 24-bit
        // multiply -- the compiler's own address arithmetic used the quarter-rate 32-bit multiply)
        const unsigned zrow = (unsigned)(size_t)(lds_u64*)zb - (unsigned)(bx0 * 8) - (unsigned)(y0 * bw * 8);
        const unsigned long long keyf = (unsigned long long)(unsigned)f;
        const int xlo4 = xlo * 4, xhi4 = xhi * 4, yhi4 = yhi * 4, bw2 = bw * 2;
        for (int xi = xlo4, yi = ylo * 4; yi <= yhi4;) {
          const int xj = min(xi + 4, xhi4), yj = min(yi + 4, yhi4);
          const f32x2 DX = f32x2{*reinterpret_cast<const float*>(pxb + xi), *reinterpret_cast<const float*>(pxb + xj)} - ax;
          const float dyr[2] = {*reinterpret_cast<const float*>(pxb + yi) - ay, *reinterpret_cast<const float*>(pxb + yj) - ay};
          unsigned zr[2];
          asm("v_mul_u32_u24 %0, %1, %2" : "=v"(zr[0]) : "s"(bw2), "v"(yi));
          asm("v_mul_u32_u24 %0, %1, %2" : "=v"(zr[1]) : "s"(bw2), "v"(yj));
          zr[0] += zrow; zr[1] += zrow;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            // edge(p; c, a) = (p - c) x (a - c); with p - c = (p - a) + (a - c) the constant term vanishes: = (p - a) x (a - c)
            const float t1 = dyr[j] * a1y, t2 = dyr[j] * a2y;
            const f32x2 W1 = __builtin_elementwise_fma(DX, f32x2{a1x, a1x}, f32x2{t1, t1});
            const f32x2 W2 = __builtin_elementwise_fma(DX, f32x2{a2x, a2x}, f32x2{t2, t2});
            const f32x2 W0 = (1.f - W1) - W2;
            const f32x2 PZ = __builtin_elementwise_fma(W1, f32x2{dzb, dzb}, __builtin_elementwise_fma(W2, f32x2{dzc, dzc}, f32x2{az, az}));
#pragma unroll
            for (int i = 0; i < 2; ++i)
              // (a NaN weight makes pz NaN, so the min's NaN-dropping cannot let a pixel through)
              if (__builtin_fminf(__builtin_fminf(W0[i], W1[i]), W2[i]) > 0.f && PZ[i] >= 0.f)
                __hip_atomic_fetch_min((lds_u64*)(size_t)(zr[j] + 2u * (unsigned)(i ? xj : xi)), ((unsigned long long)__float_as_uint(PZ[i]) << 32) | keyf,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          const bool wrap = xi + 8 > xhi4;
          xi = wrap ? xlo4 : xi + 8;
          yi += wrap ? 8 : 0;
        }
      }
    }
    __syncthreads();
    // resolve, pass 1: the covered pixels (8-9 % of the image) are appended to the pose's list (pixel << 14 | face) and
    // handled densely in pass 2 instead of under divergence here; one LDS atomic per wave reserves the list slots.
    // One box row per wave and step, lanes over the columns.
    {
      const int wave = tix >> 6, lane = tix & 63;
      for (int r = wave; r < y1 - y0; r += SIL_RT / 64) {
        for (int x0 = 0; x0 < bw; x0 += 64) {
          const int x = x0 + lane;
          const unsigned long long key = (x < bw) ? zb[r * bw + x] : ~0ull;
          const bool cov = key != ~0ull;
          const unsigned long long bal = __ballot(cov);
          if (bal) {                                                       // wave-uniform
            const int leader = __ffsll((long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&ncov, __popcll(bal));
            base = __shfl(base, leader);
            if (cov) lst[base + __popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned)((y0 + r) * SIL + bx0 + x) << SIL_FBITS) | (unsigned)(key & 0xffffffffu);
          }
        }
      }
    }
    __syncthreads();                                                   // strip resolved before the z-buffer is reused
  }
  // resolve, pass 2: alpha of the covered pixels from the edge distances of their winning faces
  const int n = ncov;
  if (tix == 0) ncover[b] = n;
  // ADJ: the adjoint is accumulated per vertex in NDC space, (G_x, G_y) = sum of d loss / d (x_ndc, y_ndc), in the
  // z-buffer's LDS (2 x 6890 floats); the chain through x_ndc = f X / Z, y_ndc = f Y / Z is linear in (G_x, G_y)
  // with per-VERTEX coefficients and is applied once per vertex at write-out:
  //   gX = f G_x / Z, gY = f G_y / Z, gZ = -(x G_x + y G_y) / Z;  d/d verts = (-2 gX, -2 gY, 2 gZ), d/d cam = (gX, gY, gZ)
  // LDS atomics retire about one lane per 2.5 clocks, so their NUMBER is what counts: (G_x, G_y) of a vertex are kept as two
  // 32-bit fixed-point integers in ONE 64-bit word, value = (ix << 32) + iy as a signed 64-bit sum, added with one
  // ds_add_u64 -- 2 atomics per covered pixel (6 in world space, 4 with float pairs; 10.5 of the pass's 19.7 us were
  // atomics).  Integer addition is associative: the adjoint is bitwise reproducible, which the float version was not.
  // Quantum q = scale * 2^-17.  One contribution is at most 86 scale (|c| <= 2 * 1e4 scale * |alpha - mask| * max_x
  // e^-x sqrt(x) * 0.01) = 2^23.4 q, so ~190 maximal contributions of one sign on a vertex stay below 2^31 q (a carry from
  // the low word into the high one would corrupt BOTH components silently).  Only pixels within ~2 px of an edge
  // contribute noticeably (alpha (1 - alpha) decays like e^(-d / sigma), sqrt(sigma) = 1.1 px), about 0.6 L maximal
  // contributions for an edge L px long, shared by its two end points: with six edges per vertex the margin holds up to
  // edges of ~100 px -- a whole-frame close-up of a mesh of this resolution (tested: tests/test_gpu_round3.py, camera at a
  // third of the distance).  Rounding error per contribution q / 2 = 4e-6 scale: 4e-5 .. 4e-8 of the contributions that
  // matter.  (Round 2 used 2^-19: 4 x finer and 4 x less head-room, ~47 contributions.)
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(zb);
  const float fq = scale * (1.f / 131072.f), fqi = (scale > 0.f) ? 131072.f / scale : 0.f;
  auto pack2 = [&](float gx, float gy) {
    const long long ix = (long long)__float2int_rn(gx * fqi), iy = (long long)__float2int_rn(gy * fqi);
    return (unsigned long long)((ix << 32) + iy);
  };
  if (ADJ) {
    for (int i = tix; i < V; i += SIL_RT) acc[i] = 0ull;
    __syncthreads();
  }
  for (int e0 = tix; e0 < n; e0 += SIL_RT * SIL_EB) {
    unsigned ent[SIL_EB];
    int id[SIL_EB][3];
    float tg[SIL_EB];
#pragma unroll
    for (int u = 0; u < SIL_EB; ++u) ent[u] = (e0 + u * SIL_RT < n) ? lst[e0 + u * SIL_RT] : 0u;
#pragma unroll
    for (int u = 0; u < SIL_EB; ++u) {
      const int f = (int)(ent[u] & ((1u << SIL_FBITS) - 1));
      const uint2 fp = faces[f];
      id[u][0] = (int)(fp.x & 8191u); id[u][1] = (int)(fp.x >> 13); id[u][2] = (int)fp.y;
      tg[u] = mask ? mask[(size_t)b * SIL * SIL + (ent[u] >> SIL_FBITS)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SIL_EB; ++u) {
      if (e0 + u * SIL_RT >= n) continue;
      const int pix = (int)(ent[u] >> SIL_FBITS);
      const float px = pxt[pix % SIL], py = pxt[pix / SIL];
      const float x[3] = {vx[id[u][0]], vx[id[u][1]], vx[id[u][2]]}, y[3] = {vy[id[u][0]], vy[id[u][1]], vy[id[u][2]]};
      int ka;
      float tt;
      const float dist = sil_nearest_edge(px, py, x, y, ka, tt);
      const float al = sil_alpha(dist);
      if (alpha_out) alpha_out[(size_t)b * SIL * SIL + pix] = al;
      if (mask) { const float dm = al - tg[u]; err += dm * dm - tg[u] * tg[u]; }   // (the pixel's background share is in smask)
      if (ADJ) {
        const float gd = scale * (al - tg[u]) * al * (1.f - al) * SIL_ISIGMA;          // d loss / d dist
        if (gd == 0.f) continue;
        // end points of edge ka (selects, not indexed registers)
        const int ida = ka == 0 ? id[u][0] : ka == 1 ? id[u][1] : id[u][2], idb = ka == 0 ? id[u][1] : ka == 1 ? id[u][2] : id[u][0];
        const float xa = ka == 0 ? x[0] : ka == 1 ? x[1] : x[2], xb = ka == 0 ? x[1] : ka == 1 ? x[2] : x[0];
        const float ya = ka == 0 ? y[0] : ka == 1 ? y[1] : y[2], yb = ka == 0 ? y[1] : ka == 1 ? y[2] : y[0];
        // dist = |p - q|^2, q = a + t (b - a): d/da = -2 (1-t) r, d/db = -2 t r (t clamped: the same formulas)
        const float rx = px - (xa + tt * (xb - xa)), ry = py - (ya + tt * (yb - ya));
        const float ca = -2.f * (1.f - tt) * gd, cb = -2.f * tt * gd;
        if (ca != 0.f) atomicAdd(&acc[ida], pack2(ca * rx, ca * ry));
        if (cb != 0.f) atomicAdd(&acc[idb], pack2(cb * rx, cb * ry));
      }
    }
  }
  // (a FRESH laundered thread index for the write-out and the reductions: nothing derived from `tix` has to survive the resolve pass)
  int tq = wave_s * 64 + fresh_lane();
  float gc[3] = {0.f, 0.f, 0.f};
  if (ADJ) {
    __syncthreads();
    for (int q = tq; q < VP / 4; q += SIL_RT) {       // the pose's pieces of VQ now take the vertex adjoint
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0, o2 = o0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = 4 * q + u;
        if (v < V) {
          const long long sum = (long long)acc[v];
          const int iy = (int)(unsigned)(sum & 0xffffffffll);                  // low word, sign-extended
          const int ix = (int)((sum - (long long)iy) >> 32);
          const float Gx = (float)ix * fq, Gy = (float)iy * fq;
          float g[3] = {0.f, 0.f, 0.f};
          if (Gx != 0.f || Gy != 0.f) {
            const float iz = 1.f / vz[v];
            g[0] = SIL_F * iz * Gx; g[1] = SIL_F * iz * Gy; g[2] = -(vx[v] * Gx + vy[v] * Gy) * iz;
          }
          o0[u] = -2.f * g[0]; o1[u] = -2.f * g[1]; o2[u] = 2.f * g[2];
#pragma unroll
          for (int c = 0; c < 3; ++c) gc[c] += g[c];
        }
      }
      VQ4[(size_t)q * BP + b] = o0; VQ4[((size_t)(VP / 4) + q) * BP + b] = o1; VQ4[((size_t)2 * (VP / 4) + q) * BP + b] = o2;
    }
  }
  // the camera adjoint's three sums and the squared error: wave butterflies, one LDS row per quantity, rows added in wave order by
  // one thread each (deterministic).  ONE barrier -- the shared-memory tree this replaces cost a pose twelve.
  // (`red` is next written at the next pose's tail, many barriers after these reads)
  if ((ADJ && gcam) || sqsil) {
    const float rv[4] = {gc[0], gc[1], gc[2], err};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < 3 && !(ADJ && gcam)) continue;
      float w = rv[c];
      // (shuffle addresses from the FRESH lane index: shared with the bounding-box reduction of the set-up they would be six
      // registers alive across the whole pose)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) w += __int_as_float(__builtin_amdgcn_ds_bpermute(((tq & 63) ^ o) << 2, __float_as_int(w)));
      if ((tq & 63) == 0) red[c * (SIL_RT / 64) + (tq >> 6)] = w;
    }
    __syncthreads();
    if (tq < 4) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < SIL_RT / 64; ++w) t += red[tq * (SIL_RT / 64) + w];
      if (tq < 3) {
        if (ADJ && gcam) {
          if (accumulate_cam) gcam[(size_t)b * 3 + tq] += t;
          else gcam[(size_t)b * 3 + tq] = t;
        }
      } else if (sqsil) sqsil[b] = t + (smask ? smask[b] : 0.f);
    }
  }
}

// The kernels.  <false> (stand-alone forward): one workgroup per pose.  <true> (fused loop): PERSISTENT -- one workgroup per CU walks
// its poses (round 5: in-kernel stamps put a pose at 67 us where the launch averaged 83 us per pose and CU: the turnover of a
// 1024-thread / 155 KB workgroup is not free).  XCD x (workgroups with blockIdx % 8 == x) takes the poses [x per, (x + 1) per) and
// its workgroups take consecutive poses in every step: the 8 poses whose adjoint pieces share a 128-byte line of VQ are written by
// one XCD at about the same time and meet in its L2.
template <bool ADJ, int SIL>
__global__ __launch_bounds__(SIL_RT) void k_sil_raster(const NdcV* __restrict__ ndc, const uint2* __restrict__ faces,
                                                       int nfaces, const float* __restrict__ mask,
                                                       unsigned* __restrict__ cover, int* __restrict__ ncover,
                                                       float* __restrict__ alpha_out, float* __restrict__ sqsil,
                                                       float scale, float* __restrict__ VQ, int BP,
                                                       const float* __restrict__ cam, int B,
                                                       float* __restrict__ gcam, int accumulate_cam,
                                                       const float* __restrict__ smask, const float* __restrict__ VPM) {
  const int wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // the wave's index, in a scalar register for the whole kernel
  if (!ADJ) {
    sil_raster_pose<ADJ, SIL>(blockIdx.x, wave_s, ndc, faces, nfaces, mask, cover, ncover, alpha_out, sqsil, scale, VQ, BP, cam, B, gcam, accumulate_cam, smask, VPM);
    return;
  }
  const int per = BP >> 3, nw = gridDim.x >> 3, xcd = blockIdx.x & 7;
  for (int i = blockIdx.x >> 3; i < per; i += nw) {
    sil_raster_pose<ADJ, SIL>(xcd * per + i, wave_s, ndc, faces, nfaces, mask, cover, ncover, alpha_out, sqsil, scale, VQ, BP, cam, B, gcam, accumulate_cam, smask, VPM);
    __syncthreads();                          // the pose's last reads of LDS before the next pose's first writes
  }
}

// adjoint for an arbitrary upstream gradient: g_alpha = galpha[pixel] if given, else scale * (alpha - mask[pixel]).
// One workgroup per pose walks the pose's covered-pixel list (dense lanes) and accumulates the NDC-space vertex
// adjoints (G_x, G_y) in LDS (2 x 6890 floats, ds_add_f32), exactly as k_sil_raster<true> does; the projection's
// chain rule is applied per vertex at write-out (plain coalesced stores of all 6890 x 3 floats: no global
// atomics, no zero-fill of the output).  LDS float adds and the list order: the summation order, hence the last
// bits, varies between runs.
constexpr int SIL_BT = 1024;
constexpr int SIL_PB = 3;                // list entries fetched together per thread
template <int SIL>
__global__ __launch_bounds__(SIL_BT) void k_sil_bwd(const NdcV* __restrict__ ndc, const int* __restrict__ faces,
                                                    const unsigned* __restrict__ cover, const int* __restrict__ ncover,
                                                    const float* __restrict__ mask,
                                                    const float* __restrict__ galpha, float scale,
                                                    float* __restrict__ dverts, int ldv, float* __restrict__ gcam,
                                                    int accumulate_cam) {
  constexpr float SIL_F = 5000.f / (float)SIL;
  extern __shared__ float acc[];                 // [V*2]
  __shared__ float gcs[3];
  const int b = blockIdx.x;
  const NdcV* vb = ndc + (size_t)b * V;
  const unsigned* lst = cover + (size_t)b * SIL * SIL;
  const int n = ncover[b];
  for (int i = threadIdx.x; i < V * 2; i += SIL_BT) acc[i] = 0.f;
  if (threadIdx.x < 3) gcs[threadIdx.x] = 0.f;
  __syncthreads();
  // SIL_PB entries per thread are fetched together: the dependent chain entry -> face indices -> vertices costs
  // three memory round trips per BATCH rather than per pixel.
  for (int e0 = threadIdx.x; e0 < n; e0 += SIL_BT * SIL_PB) {
    unsigned ent[SIL_PB];
    int id[SIL_PB][3];
    float tgt[SIL_PB];
    NdcV vv[SIL_PB][3];
#pragma unroll
    for (int u = 0; u < SIL_PB; ++u) ent[u] = (e0 + u * SIL_BT < n) ? lst[e0 + u * SIL_BT] : 0u;
#pragma unroll
    for (int u = 0; u < SIL_PB; ++u) {
      const int fc = (int)(ent[u] & ((1u << SIL_FBITS) - 1));
#pragma unroll
      for (int k = 0; k < 3; ++k) id[u][k] = faces[fc * 3 + k];
      const size_t o = (size_t)b * SIL * SIL + (ent[u] >> SIL_FBITS);
      tgt[u] = galpha ? galpha[o] : mask[o];
    }
#pragma unroll
    for (int u = 0; u < SIL_PB; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) vv[u][k] = vb[id[u][k]];
#pragma unroll
    for (int u = 0; u < SIL_PB; ++u) {
      if (e0 + u * SIL_BT >= n) continue;
      const int pix = (int)(ent[u] >> SIL_FBITS);
      const float px = pix_x<SIL>(pix % SIL), py = pix_x<SIL>(pix / SIL);
      const float x[3] = {vv[u][0].x, vv[u][1].x, vv[u][2].x}, y[3] = {vv[u][0].y, vv[u][1].y, vv[u][2].y};
      int ka;
      float tt;
      const float dist = sil_nearest_edge(px, py, x, y, ka, tt);
      const float al = sil_alpha(dist);
      const float ga = galpha ? tgt[u] : scale * (al - tgt[u]);
      const float gd = ga * al * (1.f - al) * SIL_ISIGMA;          // d loss / d dist
      if (gd == 0.f) continue;
      const int ida = ka == 0 ? id[u][0] : ka == 1 ? id[u][1] : id[u][2], idb = ka == 0 ? id[u][1] : ka == 1 ? id[u][2] : id[u][0];
      const float xa = ka == 0 ? x[0] : ka == 1 ? x[1] : x[2], xb = ka == 0 ? x[1] : ka == 1 ? x[2] : x[0];
      const float ya = ka == 0 ? y[0] : ka == 1 ? y[1] : y[2], yb = ka == 0 ? y[1] : ka == 1 ? y[2] : y[0];
      const float rx = px - (xa + tt * (xb - xa)), ry = py - (ya + tt * (yb - ya));
      const float ca = -2.f * (1.f - tt) * gd, cb = -2.f * tt * gd;
      if (ca != 0.f) { atomicAdd(&acc[ida * 2], ca * rx); atomicAdd(&acc[ida * 2 + 1], ca * ry); }
      if (cb != 0.f) { atomicAdd(&acc[idb * 2], cb * rx); atomicAdd(&acc[idb * 2 + 1], cb * ry); }
    }
  }
  __syncthreads();
  float* dv = dverts + (size_t)b * ldv;
  float gc[3] = {0.f, 0.f, 0.f};
  for (int v = threadIdx.x; v < V; v += SIL_BT) {
    const float Gx = acc[v * 2], Gy = acc[v * 2 + 1];
    float g[3] = {0.f, 0.f, 0.f};
    if (Gx != 0.f || Gy != 0.f) {
      const NdcV p = vb[v];
      const float iz = 1.f / p.z;
      g[0] = SIL_F * iz * Gx; g[1] = SIL_F * iz * Gy; g[2] = -(p.x * Gx + p.y * Gy) * iz;
    }
    dv[v * 3] = -2.f * g[0]; dv[v * 3 + 1] = -2.f * g[1]; dv[v * 3 + 2] = 2.f * g[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) gc[c] += g[c];
  }
  if (gcam) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float w = gc[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
      if ((threadIdx.x & 63) == 0) atomicAdd(&gcs[c], w);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      if (accumulate_cam) gcam[(size_t)b * 3 + threadIdx.x] += gcs[threadIdx.x];
      else gcam[(size_t)b * 3 + threadIdx.x] = gcs[threadIdx.x];
    }
  }
}

// per-pose sum(mask^2) over the image (the squared error of an all-background rendering), once per mask
__global__ __launch_bounds__(256) void k_mask_sq(const float* __restrict__ mask, float* __restrict__ smask, int SIL) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < SIL * SIL; i += 256) { const float m = mask[(size_t)b * SIL * SIL + i]; acc = fmaf(m, m, acc); }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) smask[b] = red[0];
}
int launch_mask_sq(const float* mask, float* smask, int B, hipStream_t s, int S) {
  hipLaunchKernelGGL(k_mask_sq, dim3(B), dim3(256), 0, s, mask, smask, S);
  return 0;
}

// pix_to_face of the last rasterisation (pytorch3d Fragments.pix_to_face, faces_per_pixel = 1): -1 = background, else the
// winning face, from the per-pose covered-pixel lists
__global__ void k_sil_pix_to_face(const unsigned* __restrict__ cover, const int* __restrict__ ncover, int* __restrict__ p2f, int SIL) {
  const int b = blockIdx.x;
  int* out = p2f + (size_t)b * SIL * SIL;
  for (int i = threadIdx.x; i < SIL * SIL; i += blockDim.x) out[i] = -1;
  __syncthreads();
  const unsigned* lst = cover + (size_t)b * SIL * SIL;
  const int n = ncover[b];
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const unsigned e = lst[i];
    out[e >> SIL_FBITS] = (int)(e & ((1u << SIL_FBITS) - 1));
  }
}
int launch_sil_pix_to_face(const unsigned* cover, const int* ncover, int* p2f, int B, hipStream_t s, int S) {
  hipLaunchKernelGGL(k_sil_pix_to_face, dim3(B), dim3(1024), 0, s, cover, ncover, p2f, S);
  return 0;
}

static bool g_sil_attr = false;
constexpr int SIL_LDS_BYTES = SIL_VPAD * 4 + SIL_ZPIX * 8;
// Image sizes.  The in-loop silhouette term (ADJ) is built for the two sizes the reference instantiates: 224 (scripts/optimize.py:110) and 256
// (the constructor's default, scripts/mesh_renderer.py:25).  The stand-alone forward / backward pair behind `Mesh_Renderer(image_size)` takes
// every multiple of 32 up to 256 (round 6): the image size is a template parameter of the strip arithmetic and of the pixel-centre table.
#define JRR_SIL_SIZES(X) X(32) X(64) X(96) X(128) X(160) X(192) X(224) X(256)
static void sil_attrs() {
  if (g_sil_attr) return;
#define JRR_SIL_ATTR(S_)                                                                                                           \
  (void)hipFuncSetAttribute((const void*)k_sil_raster<false, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, SIL_LDS_BYTES);       \
  (void)hipFuncSetAttribute((const void*)k_sil_bwd<S_>, hipFuncAttributeMaxDynamicSharedMemorySize, V * 2 * 4);
  JRR_SIL_SIZES(JRR_SIL_ATTR)
#undef JRR_SIL_ATTR
  (void)hipFuncSetAttribute((const void*)k_sil_raster<true, 224>, hipFuncAttributeMaxDynamicSharedMemorySize, SIL_LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)k_sil_raster<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, SIL_LDS_BYTES);
  g_sil_attr = true;
}
static int sil_size_ok(int S, bool in_loop = false) {
  if (in_loop ? (S == 224 || S == 256) : (S >= 32 && S <= 256 && S % 32 == 0)) return 0;
  if (in_loop) jrr_set_error("silhouette term inside the loop: image size %d (built for 224 and 256, the sizes the reference instantiates)", S);
  else jrr_set_error("silhouette: image size %d (the stand-alone renderer takes the multiples of 32 up to 256)", S);
  return JRR_ERR_ARG;
}
int launch_sil_project(const float* verts, int ldv, const float* cam, float* ndc, int B, hipStream_t s, int S) {
  hipLaunchKernelGGL(k_sil_project, dim3((B * V + 255) / 256), dim3(256), 0, s, verts, ldv, cam, (NdcV*)ndc, B, 5000.f / (float)S);
  return 0;
}
int launch_sil_raster(const float* ndc, const unsigned* faces_pk, int nfaces, unsigned* cover, int* ncover, float* alpha, int B,
                      hipStream_t s, int S) {
  sil_attrs();
  if (sil_size_ok(S)) return JRR_ERR_ARG;
  if (nfaces > SIL_FPT * SIL_RT) { jrr_set_error("silhouette: at most %d faces supported", SIL_FPT * SIL_RT); return JRR_ERR_ARG; }
  switch (S) {
#define JRR_SIL_FWD(S_)                                                                                                            \
  case S_:                                                                                                                         \
    hipLaunchKernelGGL((k_sil_raster<false, S_>), dim3(B), dim3(SIL_RT), SIL_LDS_BYTES, s, (const NdcV*)ndc,                        \
                       reinterpret_cast<const uint2*>(faces_pk), nfaces, nullptr, cover, ncover, alpha, nullptr, 0.f, nullptr, 0, nullptr, \
                       B, nullptr, 0, nullptr, nullptr);                                                                            \
    break;
    JRR_SIL_SIZES(JRR_SIL_FWD)
#undef JRR_SIL_FWD
  }
  return 0;
}
// fused loop: project the pose's vertices from the row-quad buffer VQ [3][VP/4][BP][4], rasterise, squared error against
// mask, and the adjoint of scale/2 * sum((alpha - mask)^2) written back over the same pieces of VQ; gcam: overwrite or
// accumulate
int launch_sil_raster_adj(float* VQ, int BP, const float* cam, const unsigned* faces_pk, int nfaces, const float* mask, const float* smask,
                          unsigned* cover, int* ncover, float* sqsil, float scale, float* gcam, int accumulate_cam, int B,
                          hipStream_t s, int S, const float* VPM) {
  sil_attrs();
  if (sil_size_ok(S, true)) return JRR_ERR_ARG;
  if (nfaces > SIL_FPT * SIL_RT) { jrr_set_error("silhouette: at most %d faces supported", SIL_FPT * SIL_RT); return JRR_ERR_ARG; }
  const int grid = BP < 256 ? BP : 256;    // persistent: one workgroup per CU (a multiple of 8: BP is a multiple of 128); padded poses zero their pieces
  if (S == 224)
    hipLaunchKernelGGL((k_sil_raster<true, 224>), dim3(grid), dim3(SIL_RT), SIL_LDS_BYTES, s, (const NdcV*)nullptr, reinterpret_cast<const uint2*>(faces_pk), nfaces, mask, cover,
                       ncover, nullptr, sqsil, scale, VQ, BP, cam, B, gcam, accumulate_cam, smask, VPM);
  else
    hipLaunchKernelGGL((k_sil_raster<true, 256>), dim3(grid), dim3(SIL_RT), SIL_LDS_BYTES, s, (const NdcV*)nullptr, reinterpret_cast<const uint2*>(faces_pk), nfaces, mask, cover,
                       ncover, nullptr, sqsil, scale, VQ, BP, cam, B, gcam, accumulate_cam, smask, VPM);
  return 0;
}
// writes ALL of dverts[b][0 .. 6890*3) (no zero-fill needed); gcam: overwrite or accumulate
int launch_sil_bwd(const float* ndc, const int* faces, const unsigned* cover, const int* ncover, const float* mask,
                   const float* galpha, float scale, float* dverts, int ldv, float* gcam, int accumulate_cam, int B,
                   hipStream_t s, int S) {
  sil_attrs();
  if (sil_size_ok(S)) return JRR_ERR_ARG;
  switch (S) {
#define JRR_SIL_BWD(S_)                                                                                                            \
  case S_:                                                                                                                         \
    hipLaunchKernelGGL(k_sil_bwd<S_>, dim3(B), dim3(SIL_BT), V * 2 * 4, s, (const NdcV*)ndc, faces, cover, ncover, mask, galpha, scale, \
                       dverts, ldv, gcam, accumulate_cam);                                                                          \
    break;
    JRR_SIL_SIZES(JRR_SIL_BWD)
#undef JRR_SIL_BWD
  }
  return 0;
}

}  // namespace jrr
