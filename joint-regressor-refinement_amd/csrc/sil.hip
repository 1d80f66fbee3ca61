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
//   k_sil_raster   per (pose, 56-row strip): every thread walks faces, tests the <= few pixels of each face's
//                  bounding box and keeps the nearest face per pixel with a 64-bit atomicMin on an LDS
//                  z-buffer keyed (depth bits << 32 | face index); then resolves alpha, writes pix_to_face,
//                  and reduces the squared error against the target mask
//   k_sil_bwd      per pixel: adjoint of alpha -> the two end points of the nearest edge (NDC) -> world
//                  vertices (float atomics into a pose-major buffer) and the camera translation
#include "jrr_common.h"
#include "kernels.h"

namespace jrr {

constexpr int SIL = 224;                 // image size (scripts/optimize.py:110 Mesh_Renderer(image_size=224))
constexpr int SIL_STRIP = 56;            // rows per LDS z-buffer strip (56*224*8 B = 98 KB)
constexpr int SIL_NSTRIP = SIL / SIL_STRIP;
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

__global__ __launch_bounds__(512) void k_sil_raster(const NdcV* __restrict__ ndc, const int* __restrict__ faces,
                                                    int nfaces, const float* __restrict__ mask,
                                                    int* __restrict__ p2f, float* __restrict__ alpha_out,
                                                    float* __restrict__ sqsil) {
  extern __shared__ unsigned long long zb[];     // [SIL_STRIP][SIL]
  __shared__ float red[512];
  const int b = blockIdx.x / SIL_NSTRIP, strip = blockIdx.x % SIL_NSTRIP;
  const int y0 = strip * SIL_STRIP;
  const NdcV* vb = ndc + (size_t)b * V;
  for (int i = threadIdx.x; i < SIL_STRIP * SIL; i += blockDim.x) zb[i] = ~0ull;
  __syncthreads();
  for (int f = threadIdx.x; f < nfaces; f += blockDim.x) {
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    const NdcV a = vb[i0], bb = vb[i1], c = vb[i2];
    const float area = edge_fn(c.x, c.y, a.x, a.y, bb.x, bb.y);
    if (!(fabsf(area) > SIL_EPS)) continue;                    // also rejects NaN
    if (fmaxf(a.z, fmaxf(bb.z, c.z)) < 0.f) continue;          // behind the camera
    const float xmax = fmaxf(a.x, fmaxf(bb.x, c.x)), xmin = fminf(a.x, fminf(bb.x, c.x));
    const float ymax = fmaxf(a.y, fmaxf(bb.y, c.y)), ymin = fminf(a.y, fminf(bb.y, c.y));
    // xf = 1 - (2 xi + 1)/W  =>  xi = (W (1 - xf) - 1) / 2 ; one pixel of slack, exact test below
    int xlo = (int)floorf((SIL * (1.f - xmax) - 1.f) * 0.5f), xhi = (int)ceilf((SIL * (1.f - xmin) - 1.f) * 0.5f);
    int ylo = (int)floorf((SIL * (1.f - ymax) - 1.f) * 0.5f), yhi = (int)ceilf((SIL * (1.f - ymin) - 1.f) * 0.5f);
    xlo = max(xlo, 0); xhi = min(xhi, SIL - 1);
    ylo = max(ylo, y0); yhi = min(yhi, y0 + SIL_STRIP - 1);
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
  __syncthreads();
  float err = 0.f;
  for (int i = threadIdx.x; i < SIL_STRIP * SIL; i += blockDim.x) {
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
    for (int s = 256; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) sqsil[blockIdx.x] = red[0];
  }
}

// adjoint: g_alpha = galpha[pixel] if given, else scale * (alpha - mask[pixel])
__global__ __launch_bounds__(256) void k_sil_bwd(const NdcV* __restrict__ ndc, const int* __restrict__ faces,
                                                 const int* __restrict__ p2f, const float* __restrict__ mask,
                                                 const float* __restrict__ galpha, float scale,
                                                 float* __restrict__ dverts, int ldv, float* __restrict__ gcam) {
  __shared__ float red[3][256];
  const int b = blockIdx.x / (SIL * SIL / 256 / 7), chunk = blockIdx.x % (SIL * SIL / 256 / 7);   // 28 blocks x 7 px per thread
  const NdcV* vb = ndc + (size_t)b * V;
  float gc[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < 7; ++k) {
    const int pix = (chunk * 7 + k) * 256 + threadIdx.x;
    const size_t o = (size_t)b * SIL * SIL + pix;
    const int f = p2f[o];
    if (f < 0) continue;
    const int yi = pix / SIL, xi = pix % SIL;
    const int id[3] = {faces[f * 3], faces[f * 3 + 1], faces[f * 3 + 2]};
    const NdcV vv[3] = {vb[id[0]], vb[id[1]], vb[id[2]]};
    const float px = pix_x(xi), py = pix_x(yi);
    float t[3], d[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) d[e] = seg_dist2(px, py, vv[e].x, vv[e].y, vv[(e + 1) % 3].x, vv[(e + 1) % 3].y, t[e]);
    int e = 0;
    if (d[1] < d[e]) e = 1;
    if (d[2] < d[e]) e = 2;
    const float dist = d[e];
    const float al = 1.f / (1.f + expf(-dist / SIL_SIGMA));
    const float ga = galpha ? galpha[o] : scale * (al - mask[o]);
    const float gd = ga * al * (1.f - al) / SIL_SIGMA;          // d alpha / d dist
    if (gd == 0.f) continue;
    const NdcV A = vv[e], Bv = vv[(e + 1) % 3];
    const float tt = t[e];
    const float rx = px - (A.x + tt * (Bv.x - A.x)), ry = py - (A.y + tt * (Bv.y - A.y));
    // dist = |p - q|^2, q = a + t (b - a): d/da = -2 (1-t) r, d/db = -2 t r (t clamped: the same formulas)
    const float ca = -2.f * (1.f - tt) * gd, cb = -2.f * tt * gd;
    const int ida = id[e], idb = id[(e + 1) % 3];
    const float g2[2][2] = {{ca * rx, ca * ry}, {cb * rx, cb * ry}};
    const NdcV ve[2] = {A, Bv};
    const int ide[2] = {ida, idb};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (g2[q][0] == 0.f && g2[q][1] == 0.f) continue;
      // x_ndc = f X / Z, y_ndc = f Y / Z ; X = x_ndc Z / f
      const float Z = ve[q].z, X = ve[q].x * Z / SIL_F, Y = ve[q].y * Z / SIL_F;
      const float gX = SIL_F / Z * g2[q][0], gY = SIL_F / Z * g2[q][1];
      const float gZ = -SIL_F * (X * g2[q][0] + Y * g2[q][1]) / (Z * Z);
      float* dv = dverts + (size_t)b * ldv + ide[q] * 3;
      atomicAdd(dv, -2.f * gX);              // world vertex = (-X/2.., ) : X = -2x + tx etc.
      atomicAdd(dv + 1, -2.f * gY);
      atomicAdd(dv + 2, 2.f * gZ);
      gc[0] += gX; gc[1] += gY; gc[2] += gZ;
    }
  }
  if (gcam) {
#pragma unroll
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = gc[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + s];
      }
      __syncthreads();
    }
    if (threadIdx.x < 3) atomicAdd(&gcam[(size_t)b * 3 + threadIdx.x], red[threadIdx.x][0]);
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
int launch_sil_project(const float* verts, int ldv, const float* cam, float* ndc, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_sil_project, dim3((B * V + 255) / 256), dim3(256), 0, s, verts, ldv, cam, (NdcV*)ndc, B);
  return 0;
}
int launch_sil_raster(const float* ndc, const int* faces, int nfaces, const float* mask, int* p2f, float* alpha,
                      float* sqsil_strips, float* sqsil, int B, hipStream_t s) {
  const int lds = SIL_STRIP * SIL * 8;
  if (!g_sil_attr) {
    (void)hipFuncSetAttribute((const void*)k_sil_raster, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    g_sil_attr = true;
  }
  hipLaunchKernelGGL(k_sil_raster, dim3(B * SIL_NSTRIP), dim3(512), lds, s, (const NdcV*)ndc, faces, nfaces, mask, p2f, alpha,
                     sqsil_strips);
  if (sqsil && sqsil_strips) hipLaunchKernelGGL(k_sil_sum, dim3((B + 255) / 256), dim3(256), 0, s, sqsil_strips, sqsil, B);
  return 0;
}
int launch_sil_bwd(const float* ndc, const int* faces, const int* p2f, const float* mask, const float* galpha, float scale,
                   float* dverts, int ldv, float* gcam, int B, hipStream_t s) {
  hipLaunchKernelGGL(k_sil_bwd, dim3(B * (SIL * SIL / 256 / 7)), dim3(256), 0, s, (const NdcV*)ndc, faces, p2f, mask, galpha,
                     scale, dverts, ldv, gcam);
  return 0;
}

}  // namespace jrr
