// Perspective projection of the regressed joints and its adjoint (scripts/renderer.py:35-49 with pytorch3d 0.3.0 PerspectiveCameras,
// R = I, T = cam, focal 5000/224 in NDC, principal point 0, 224x224 screen):
//   X = -2x + tx, Y = -2y + ty, Z = 2z + tz ;  x_ndc = f X / Z ;  x_screen = (W-1)/2 (1 - x_ndc)  (same for y).
// Shared by k_joints_loss (prep.hip) and the support-vertex iteration (supk.h).
#pragma once
#include "jrr_common.h"

namespace jrr {

constexpr float PROJ_F = 5000.f / 224.f;
constexpr float PROJ_HALF = (224.f - 1.f) * 0.5f;

__device__ __forceinline__ void project_point(const float p[3], const float t[3], float& xs, float& ys, float& invZ,
                                              float& X, float& Y) {
  X = -2.f * p[0] + t[0];
  Y = -2.f * p[1] + t[1];
  const float Z = 2.f * p[2] + t[2];
  invZ = 1.f / Z;
  xs = PROJ_HALF * (1.f - PROJ_F * X * invZ);
  ys = PROJ_HALF * (1.f - PROJ_F * Y * invZ);
}

// adjoint of the 2-D squared error of one joint: g2 = dL/d(xs,ys) -> accumulates dL/dp (3) and dL/dt (3)
__device__ __forceinline__ void project_point_bwd(float gxs, float gys, float invZ, float X, float Y, float gp[3],
                                                  float gt[3]) {
  const float gxn = -PROJ_HALF * gxs, gyn = -PROJ_HALF * gys;       // d/dx_ndc
  const float gX = gxn * PROJ_F * invZ, gY = gyn * PROJ_F * invZ;
  const float gZ = -(gxn * X + gyn * Y) * PROJ_F * invZ * invZ;
  gp[0] += -2.f * gX; gp[1] += -2.f * gY; gp[2] += 2.f * gZ;
  gt[0] += gX; gt[1] += gY; gt[2] += gZ;
}

struct Reproj {
  const float* gt_j2d;   // (B,17,2) or NULL (term disabled)
  const float* cam;      // (B,3)
  float* gcam;           // (B,3) out: dL/dcam
  float* sq2d;           // (B) out: sum of squared 2-D errors (nullable)
  float scale2d;         // 2*weight/(batch_norm*34)
};

}  // namespace jrr
