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
//   k_sil_raster   per (pose, 75-row strip): every thread walks faces (4 fetched together), tests the <= few
//                  pixels of each face's bounding box and keeps the nearest face per pixel with a 64-bit
//                  atomicMin on an LDS z-buffer keyed (depth bits << 32 | face index); then resolves alpha,
//                  writes pix_to_face, and reduces the squared error against the target mask
//   k_sil_bwd      per pose: adjoint of alpha -> the two end points of the nearest edge (NDC) -> world
//                  vertices, accumulated in LDS (81 KB per mesh) and written once, and the camera translation
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

constexpr int SIL = 224;                 // image size (scripts/optimize.py:110 Mesh_Renderer(image_size=224))
constexpr int SIL_STRIP = 75;            // rows per LDS z-buffer strip (75*224*8 B = 131 KB; 3 strips: 75+75+74)
constexpr int SIL_NSTRIP = 3;
constexpr int SIL_RT = 1024;             // threads of the raster workgroup (one workgroup per CU: LDS-bound)
constexpr int SIL_FB = 4;                // faces fetched together per thread (gather latency amortised)
constexpr float SIL_F = 5000.f / 224.f;  // NDC focal length
constexpr float SIL_SIGMA = 1e-4f;
constexpr float SIL_EPS = 1e-8f;

struct alignas(16) NdcV { float x, y, z, pad; };

__global__ void k_sil_project(const float* __restrict__ verts, int ldv, const float* __restrict__ cam,
                              NdcV* __restrict__ ndc, int B) {
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
__device__ __forceinline__ float pix_x(int xi) { return 1.f - (2.f * xi + 1.f) / SIL; }

// squared distance from p to segment a-b, and the clamped parameter t
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by, float& t) {
  const float dx = bx - ax, dy = by - ay;
  const float l2 = dx * dx + dy * dy;
  t = (l2 <= SIL_EPS) ? 0.f : fminf(fmaxf(((px - ax) * dx + (py - ay) * dy) / l2, 0.f), 1.f);
  const float qx = ax + t * dx, qy = ay + t * dy;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

__global__ __launch_bounds__(SIL_RT) void k_sil_raster(const NdcV* __restrict__ ndc, const int* __restrict__ faces,
                                                       int nfaces, const float* __restrict__ mask,
                                                       int* __restrict__ p2f, float* __restrict__ alpha_out,
                                                       float* __restrict__ sqsil) {
  extern __shared__ unsigned long long zb[];     // [SIL_STRIP][SIL]
  __shared__ float red[SIL_RT];
  const int b = blockIdx.x / SIL_NSTRIP, strip = blockIdx.x % SIL_NSTRIP;
  const int y0 = strip * SIL_STRIP, y1 = min(y0 + SIL_STRIP, SIL);   // rows [y0, y1)
  const NdcV* vb = ndc + (size_t)b * V;
  for (int i = threadIdx.x; i < SIL_STRIP * SIL; i += SIL_RT) zb[i] = ~0ull;
  __syncthreads();
  for (int f0 = threadIdx.x; f0 < nfaces; f0 += SIL_RT * SIL_FB) {
    // fetch SIL_FB faces at once: all index loads, then all vertex gathers, are in flight together
    int fi[SIL_FB][3];
    NdcV fv[SIL_FB][3];
#pragma unroll
    for (int u = 0; u < SIL_FB; ++u) {
      const int f = min(f0 + u * SIL_RT, nfaces - 1);
#pragma unroll
      for (int k = 0; k < 3; ++k) fi[u][k] = faces[f * 3 + k];
    }
#pragma unroll
    for (int u = 0; u < SIL_FB; ++u)
#pragma unroll
      for (int k = 0; k < 3; ++k) fv[u][k] = vb[fi[u][k]];
#pragma unroll
    for (int u = 0; u < SIL_FB; ++u) {
      const int f = f0 + u * SIL_RT;
      if (f >= nfaces) continue;
      const NdcV a = fv[u][0], bb = fv[u][1], c = fv[u][2];
      const float ymax = fmaxf(a.y, fmaxf(bb.y, c.y)), ymin = fminf(a.y, fminf(bb.y, c.y));
      // pixel centres inside the bounding box: yf = 1 - (2 yi + 1)/H in [ymin, ymax]  <=>
      // yi in [ceil((H (1 - ymax) - 1)/2), floor((H (1 - ymin) - 1)/2)]; 1e-3 px of slack, the inside test is exact.
      // Most faces are smaller than a pixel and contain no centre at all.
      int ylo = (int)ceilf((SIL * (1.f - ymax) - 1.f) * 0.5f - 1e-3f), yhi = (int)floorf((SIL * (1.f - ymin) - 1.f) * 0.5f + 1e-3f);
      ylo = max(ylo, y0); yhi = min(yhi, y1 - 1);
      if (ylo > yhi) continue;                                     // not in this strip / no pixel centre
      const float xmax = fmaxf(a.x, fmaxf(bb.x, c.x)), xmin = fminf(a.x, fminf(bb.x, c.x));
      int xlo = (int)ceilf((SIL * (1.f - xmax) - 1.f) * 0.5f - 1e-3f), xhi = (int)floorf((SIL * (1.f - xmin) - 1.f) * 0.5f + 1e-3f);
      xlo = max(xlo, 0); xhi = min(xhi, SIL - 1);
      if (xlo > xhi) continue;
      const float area = edge_fn(c.x, c.y, a.x, a.y, bb.x, bb.y);
      if (!(fabsf(area) > SIL_EPS)) continue;                    // also rejects NaN
      if (fmaxf(a.z, fmaxf(bb.z, c.z)) < 0.f) continue;          // behind the camera
      const float inv = 1.f / area;
      for (int yi = ylo; yi <= yhi; ++yi) {
        const float py = pix_x(yi);
        for (int xi = xlo; xi <= xhi; ++xi) {
          const float px = pix_x(xi);
          const float w0 = edge_fn(px, py, bb.x, bb.y, c.x, c.y) * inv;
          const float w1 = edge_fn(px, py, c.x, c.y, a.x, a.y) * inv;
          const float w2 = edge_fn(px, py, a.x, a.y, bb.x, bb.y) * inv;
          if (!(w0 > 0.f && w1 > 0.f && w2 > 0.f)) continue;
          const float pz = w0 * a.z + w1 * bb.z + w2 * c.z;
          if (!(pz >= 0.f)) continue;
          const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)f;
          atomicMin(&zb[(yi - y0) * SIL + xi], key);
        }
      }
    }
  }
  __syncthreads();
  float err = 0.f;
  for (int i = threadIdx.x; i < (y1 - y0) * SIL; i += SIL_RT) {
    const int yi = y0 + i / SIL, xi = i % SIL;
    const unsigned long long key = zb[i];
    float al = 0.f;
    int f = -1;
    if (key != ~0ull) {
      f = (int)(key & 0xffffffffu);
      const NdcV a = vb[faces[f * 3]], bb = vb[faces[f * 3 + 1]], c = vb[faces[f * 3 + 2]];
      const float px = pix_x(xi), py = pix_x(yi);
      float t;
      const float d = fminf(fminf(seg_dist2(px, py, a.x, a.y, bb.x, bb.y, t), seg_dist2(px, py, bb.x, bb.y, c.x, c.y, t)),
                            seg_dist2(px, py, c.x, c.y, a.x, a.y, t));
      al = 1.f / (1.f + expf(-d / SIL_SIGMA));
    }
    const size_t o = ((size_t)b * SIL + yi) * SIL + xi;
    p2f[o] = f;
    if (alpha_out) alpha_out[o] = al;
    if (mask) { const float dm = al - mask[o]; err += dm * dm; }
  }
  if (sqsil) {
    red[threadIdx.x] = err;
    __syncthreads();
    for (int s = SIL_RT / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) sqsil[blockIdx.x] = red[0];
  }
}

// adjoint: g_alpha = galpha[pixel] if given, else scale * (alpha - mask[pixel]).
// One workgroup per pose: the vertex adjoints of the whole mesh (6890 x 3 floats = 81 KB) are accumulated in
// LDS (ds_add_f32) and written out once, pose-major, with plain coalesced stores -- no global atomics, no
// zero-fill of the output.  (LDS float adds: the summation order, hence the last bits, varies between runs.)
constexpr int SIL_BT = 1024;
__global__ __launch_bounds__(SIL_BT) void k_sil_bwd(const NdcV* __restrict__ ndc, const int* __restrict__ faces,
                                                    const int* __restrict__ p2f, const float* __restrict__ mask,
                                                    const float* __restrict__ galpha, float scale,
                                                    float* __restrict__ dverts, int ldv, float* __restrict__ gcam,
                                                    int accumulate_cam) {
  extern __shared__ float acc[];                 // [V*3] + 3 camera sums
  const int b = blockIdx.x;
  const NdcV* vb = ndc + (size_t)b * V;
  for (int i = threadIdx.x; i < V * 3 + 3; i += SIL_BT) acc[i] = 0.f;
  __syncthreads();
  float gc[3] = {0.f, 0.f, 0.f};
  for (int pix = threadIdx.x; pix < SIL * SIL; pix += SIL_BT) {
    const size_t o = (size_t)b * SIL * SIL + pix;
    const int f = p2f[o];
    if (f < 0) continue;
    const int yi = pix / SIL, xi = pix % SIL;
    const int id[3] = {faces[f * 3], faces[f * 3 + 1], faces[f * 3 + 2]};
    const NdcV vv[3] = {vb[id[0]], vb[id[1]], vb[id[2]]};
    const float px = pix_x(xi), py = pix_x(yi);
    float t0, t1, t2;
    const float d0 = seg_dist2(px, py, vv[0].x, vv[0].y, vv[1].x, vv[1].y, t0);
    const float d1 = seg_dist2(px, py, vv[1].x, vv[1].y, vv[2].x, vv[2].y, t1);
    const float d2 = seg_dist2(px, py, vv[2].x, vv[2].y, vv[0].x, vv[0].y, t2);
    // nearest edge (first minimum), its end points A -> B
    NdcV A = vv[0], Bv = vv[1];
    int ida = id[0], idb = id[1];
    float dist = d0, tt = t0;
    if (d1 < dist) { dist = d1; tt = t1; A = vv[1]; Bv = vv[2]; ida = id[1]; idb = id[2]; }
    if (d2 < dist) { dist = d2; tt = t2; A = vv[2]; Bv = vv[0]; ida = id[2]; idb = id[0]; }
    const float al = 1.f / (1.f + expf(-dist / SIL_SIGMA));
    const float ga = galpha ? galpha[o] : scale * (al - mask[o]);
    const float gd = ga * al * (1.f - al) / SIL_SIGMA;          // d loss / d dist
    if (gd == 0.f) continue;
    const float rx = px - (A.x + tt * (Bv.x - A.x)), ry = py - (A.y + tt * (Bv.y - A.y));
    // dist = |p - q|^2, q = a + t (b - a): d/da = -2 (1-t) r, d/db = -2 t r (t clamped: the same formulas)
    const float ca = -2.f * (1.f - tt) * gd, cb = -2.f * tt * gd;
    {   // end point A
      const float gx = ca * rx, gy = ca * ry;
      const float Z = A.z, X = A.x * Z / SIL_F, Y = A.y * Z / SIL_F;
      const float gX = SIL_F / Z * gx, gY = SIL_F / Z * gy, gZ = -SIL_F * (X * gx + Y * gy) / (Z * Z);
      atomicAdd(&acc[ida * 3], -2.f * gX); atomicAdd(&acc[ida * 3 + 1], -2.f * gY); atomicAdd(&acc[ida * 3 + 2], 2.f * gZ);
      gc[0] += gX; gc[1] += gY; gc[2] += gZ;
    }
    {   // end point B
      const float gx = cb * rx, gy = cb * ry;
      const float Z = Bv.z, X = Bv.x * Z / SIL_F, Y = Bv.y * Z / SIL_F;
      const float gX = SIL_F / Z * gx, gY = SIL_F / Z * gy, gZ = -SIL_F * (X * gx + Y * gy) / (Z * Z);
      atomicAdd(&acc[idb * 3], -2.f * gX); atomicAdd(&acc[idb * 3 + 1], -2.f * gY); atomicAdd(&acc[idb * 3 + 2], 2.f * gZ);
      gc[0] += gX; gc[1] += gY; gc[2] += gZ;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (gc[c] != 0.f) atomicAdd(&acc[V * 3 + c], gc[c]);
  __syncthreads();
  float* dv = dverts + (size_t)b * ldv;
  for (int i = threadIdx.x; i < V * 3; i += SIL_BT) dv[i] = acc[i];
  if (gcam && threadIdx.x < 3) {
    if (accumulate_cam) gcam[(size_t)b * 3 + threadIdx.x] += acc[V * 3 + threadIdx.x];
    else gcam[(size_t)b * 3 + threadIdx.x] = acc[V * 3 + threadIdx.x];
  }
}

// per-pose sum of the strip partials
__global__ void k_sil_sum(const float* __restrict__ sqsil, float* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int s = 0; s < SIL_NSTRIP; ++s) acc += sqsil[(size_t)b * SIL_NSTRIP + s];
  out[b] = acc;
}

static bool g_sil_attr = false;
static void sil_attrs() {
  if (g_sil_attr) return;
  (void)hipFuncSetAttribute((const void*)k_sil_raster, hipFuncAttributeMaxDynamicSharedMemorySize, SIL_STRIP * SIL * 8);
  (void)hipFuncSetAttribute((const void*)k_sil_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (V * 3 + 3) * 4);
  g_sil_attr = true;
}
int launch_sil_project(const float* verts, int ldv, const float* cam, float* ndc, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_sil_project, dim3((B * V + 255) / 256), dim3(256), 0, s, verts, ldv, cam, (NdcV*)ndc, B);
  return 0;
}
int launch_sil_raster(const float* ndc, const int* faces, int nfaces, const float* mask, int* p2f, float* alpha,
                      float* sqsil_strips, float* sqsil, int B, hipStream_t s) {
  sil_attrs();
  hipLaunchKernelGGL(k_sil_raster, dim3(B * SIL_NSTRIP), dim3(SIL_RT), SIL_STRIP * SIL * 8, s, (const NdcV*)ndc, faces, nfaces,
                     mask, p2f, alpha, sqsil_strips);
  if (sqsil && sqsil_strips) hipLaunchKernelGGL(k_sil_sum, dim3((B + 255) / 256), dim3(256), 0, s, sqsil_strips, sqsil, B);
  return 0;
}
// writes ALL of dverts[b][0 .. 6890*3) (no zero-fill needed); gcam: overwrite or accumulate
int launch_sil_bwd(const float* ndc, const int* faces, const int* p2f, const float* mask, const float* galpha, float scale,
                   float* dverts, int ldv, float* gcam, int accumulate_cam, int B, hipStream_t s) {
  sil_attrs();
  hipLaunchKernelGGL(k_sil_bwd, dim3(B), dim3(SIL_BT), (V * 3 + 3) * 4, s, (const NdcV*)ndc, faces, p2f, mask, galpha, scale,
                     dverts, ldv, gcam, accumulate_cam);
  return 0;
}

}  // namespace jrr
