"""CPU restatement of the soft-silhouette renderer on the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned**: the arithmetic lives in
`pytorch3d==0.3.0` (reference requirements.txt:10), which is absent from the reference tree and from this
image.  Restated from the published pytorch3d 0.3.0 algorithm and the reference's call sites:

  render_mesh            /root/reference/scripts/optimize.py:77-85   flip x,y; x2; alpha channel [:, 3]
  Mesh_Renderer          /root/reference/scripts/mesh_renderer.py:23-79
      PerspectiveCameras(T=batch['cam'], focal_length=5000/image_size, principal_point=0)   (:52-57)
      MeshRasterizer(RasterizationSettings(image_size, blur_radius=0.0, faces_per_pixel=1)) (:34-38,59-63)
      SoftSilhouetteShader(BlendParams(sigma=1e-4, gamma=1e-4))                              (:28,66)

pytorch3d 0.3.0 semantics restated:
  * NDC: x_ndc = f X / Z, y_ndc = f Y / Z with (X,Y,Z) = verts + T (R = I); +x_ndc is LEFT, +y_ndc is UP.
  * pixel (row yi, col xi) has its centre at xf = 1 - (2 xi + 1)/W, yf = 1 - (2 yi + 1)/H.
  * a face covers a pixel iff all three barycentric coordinates of the centre are > 0 (blur_radius = 0),
    its interpolated view depth pz = sum_k bary_k z_k is >= 0 and |signed area| > 1e-8; the face with the
    smallest pz wins (faces_per_pixel = 1; ties -> lowest face index).
  * dists = -(squared euclidean distance from the centre to the nearest EDGE of the winning face) inside it;
    alpha = sigmoid(-dists / sigma) on covered pixels, 0 elsewhere (sigmoid_alpha_blend with K = 1).
  * gradients flow through dists to the NDC xy of the winning face's vertices only.
"""
from __future__ import annotations

import numpy as np
import torch

SIGMA = 1e-4
K_EPS = 1e-8


def project_mesh(verts: torch.Tensor, cam: torch.Tensor, image_size: int = 224) -> torch.Tensor:
    """optimize.py:80-82 + mesh_renderer.py:52-57: (B,V,3), (B,3) -> (B,V,3) = (x_ndc, y_ndc, view depth)."""
    f = 5000.0 / image_size
    X = -2 * verts[..., 0] + cam[:, None, 0]
    Y = -2 * verts[..., 1] + cam[:, None, 1]
    Z = 2 * verts[..., 2] + cam[:, None, 2]
    return torch.stack([f * X / Z, f * Y / Z, Z], dim=-1)


def _edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def rasterize_nearest(ndc: np.ndarray, faces: np.ndarray, H: int, W: int) -> np.ndarray:
    """pix_to_face (H,W) int32 (-1 = background) for one mesh, brute force over each face's bounding box."""
    x, y, z = ndc[:, 0].astype(np.float32), ndc[:, 1].astype(np.float32), ndc[:, 2].astype(np.float32)
    fx, fy, fz = x[faces], y[faces], z[faces]                       # (F,3)
    area = _edge(fx[:, 2], fy[:, 2], fx[:, 0], fy[:, 0], fx[:, 1], fy[:, 1])
    ok = (np.abs(area) > K_EPS) & (fz.max(1) >= 0) & np.isfinite(fx).all(1) & np.isfinite(fy).all(1)
    # pixel index ranges covered by the bbox: xf = 1 - (2 xi + 1)/W  =>  xi = (W (1 - xf) - 1)/2
    xi_lo = np.ceil((W * (1 - fx.max(1)) - 1) / 2 - 1e-6).astype(np.int64)
    xi_hi = np.floor((W * (1 - fx.min(1)) - 1) / 2 + 1e-6).astype(np.int64)
    yi_lo = np.ceil((H * (1 - fy.max(1)) - 1) / 2 - 1e-6).astype(np.int64)
    yi_hi = np.floor((H * (1 - fy.min(1)) - 1) / 2 + 1e-6).astype(np.int64)
    xi_lo, xi_hi = np.clip(xi_lo, 0, W), np.clip(xi_hi, -1, W - 1)
    yi_lo, yi_hi = np.clip(yi_lo, 0, H), np.clip(yi_hi, -1, H - 1)
    nx, ny = np.maximum(xi_hi - xi_lo + 1, 0), np.maximum(yi_hi - yi_lo + 1, 0)
    cnt = np.where(ok, nx * ny, 0)
    fid = np.repeat(np.arange(len(faces)), cnt)
    if len(fid) == 0:
        return -np.ones((H, W), dtype=np.int32)
    start = np.cumsum(cnt) - cnt
    k = np.arange(len(fid)) - np.repeat(start, cnt)
    xi = xi_lo[fid] + k % nx[fid]
    yi = yi_lo[fid] + k // nx[fid]
    px = (1 - (2 * xi + 1) / W).astype(np.float32)
    py = (1 - (2 * yi + 1) / H).astype(np.float32)
    a = area[fid]
    w0 = _edge(px, py, fx[fid, 1], fy[fid, 1], fx[fid, 2], fy[fid, 2]) / a
    w1 = _edge(px, py, fx[fid, 2], fy[fid, 2], fx[fid, 0], fy[fid, 0]) / a
    w2 = _edge(px, py, fx[fid, 0], fy[fid, 0], fx[fid, 1], fy[fid, 1]) / a
    pz = w0 * fz[fid, 0] + w1 * fz[fid, 1] + w2 * fz[fid, 2]
    hit = (w0 > 0) & (w1 > 0) & (w2 > 0) & (pz >= 0)
    fid, xi, yi, pz = fid[hit], xi[hit], yi[hit], pz[hit]
    order = np.lexsort((fid, pz))                 # nearest first, ties -> lowest face index
    pix = yi[order] * W + xi[order]
    first = np.unique(pix, return_index=True)[1]
    out = -np.ones(H * W, dtype=np.int32)
    out[pix[first]] = fid[order][first]
    return out.reshape(H, W)


def _seg_dist2(px, py, ax, ay, bx, by):
    """squared distance from (px,py) to segment a-b (pytorch3d PointLineDistanceForward)"""
    dx, dy = bx - ax, by - ay
    l2 = dx * dx + dy * dy
    t = ((px - ax) * dx + (py - ay) * dy) / torch.clamp(l2, min=K_EPS)
    t = torch.where(l2 <= K_EPS, torch.zeros_like(t), torch.clamp(t, 0, 1))
    qx, qy = ax + t * dx, ay + t * dy
    return (px - qx) ** 2 + (py - qy) ** 2


def soft_silhouette(verts: torch.Tensor, faces: np.ndarray, cam: torch.Tensor, image_size: int = 224,
                    sigma: float = SIGMA, return_pix_to_face: bool = False):
    """render_mesh(...)[:, 3].unsqueeze(1): (B,V,3), (F,3), (B,3) -> alpha (B,1,H,W); differentiable w.r.t.
    verts and cam through the edge distances of the winning faces."""
    B = verts.shape[0]
    H = W = image_size
    ndc = project_mesh(verts, cam, image_size)
    ft = torch.as_tensor(np.asarray(faces), dtype=torch.long)
    out, p2fs = [], []
    for b in range(B):
        p2f = rasterize_nearest(ndc[b].detach().cpu().numpy(), np.asarray(faces), H, W)
        p2fs.append(p2f)
        alpha = torch.zeros(H * W, dtype=verts.dtype)
        pix = np.nonzero(p2f.reshape(-1) >= 0)[0]
        if len(pix):
            f = ft[torch.as_tensor(p2f.reshape(-1)[pix], dtype=torch.long)]          # (P,3)
            yi, xi = pix // W, pix % W
            px = torch.as_tensor(1 - (2 * xi + 1) / W, dtype=verts.dtype)
            py = torch.as_tensor(1 - (2 * yi + 1) / H, dtype=verts.dtype)
            vx, vy = ndc[b, :, 0], ndc[b, :, 1]
            d01 = _seg_dist2(px, py, vx[f[:, 0]], vy[f[:, 0]], vx[f[:, 1]], vy[f[:, 1]])
            d12 = _seg_dist2(px, py, vx[f[:, 1]], vy[f[:, 1]], vx[f[:, 2]], vy[f[:, 2]])
            d20 = _seg_dist2(px, py, vx[f[:, 2]], vy[f[:, 2]], vx[f[:, 0]], vy[f[:, 0]])
            dist = torch.minimum(torch.minimum(d01, d12), d20)
            alpha = alpha.index_put((torch.as_tensor(pix),), torch.sigmoid(dist / sigma))
        out.append(alpha.view(1, H, W))
    img = torch.stack(out, 0)
    return (img, np.stack(p2fs)) if return_pix_to_face else img


def silhouette_loss(verts, faces, cam, mask, batch_norm=None, image_size: int = 224):
    """scripts/optimize.py:237: MSELoss(img, batch['mask_rcnn']) (mean over B*1*H*W)."""
    img = soft_silhouette(verts, faces, cam, image_size)
    nb = verts.shape[0] if batch_norm is None else batch_norm
    return ((img - mask) ** 2).sum() / (nb * image_size * image_size), img
