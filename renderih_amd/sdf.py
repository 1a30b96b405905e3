"""Interior distance field of closed meshes on a voxel grid and the penetration loss built on it -- drop-in for the
reference's `sdf` extension (pose_data_optimize/sdf/sdf/sdf.py:8-35 `SDFFunction` / `SDF` / `sdf`, sdf_loss.py:7-103
`SDFLoss`), whose CUDA kernel is the only native code of the reference.  The voxeliser is the HIP kernel
csrc/rih_sdf.hip; like the reference's it has no gradient (phi is used as a constant field that other meshes' vertices
sample).  STATUS: harness-verified, not yet run on a GPU.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops import check


def sdf(faces, vertices, grid_size=32):
    """faces [F,3] int32, vertices [B,V,3] fp32 inside [-1,1]^3 (GPU) -> phi [B,G,G,G] indexed [b][z][y][x]."""
    ops._chk(vertices)
    ops._chk(faces, dtype=torch.int32)
    vertices, faces = vertices.contiguous(), faces.contiguous()
    B, V, _ = vertices.shape
    phi = torch.empty((B, grid_size, grid_size, grid_size), device=vertices.device, dtype=torch.float32)
    check(ops._L().rih_sdf(phi.data_ptr(), faces.data_ptr(), vertices.data_ptr(), B, faces.shape[0], V, grid_size,
                           ops._stream()), 'rih_sdf')
    return phi


class SDF(nn.Module):
    def forward(self, faces, vertices, grid_size=32):
        with torch.no_grad():
            return sdf(faces, vertices.detach(), grid_size)


class SDFLoss(nn.Module):
    """sdf_loss.py:7-103: every mesh ("person" there, hand here) is voxelised in its own padded bounding cube; the vertices of
    the other meshes sample that field (trilinear `grid_sample`); positive samples = penetration depth."""

    def __init__(self, faces, grid_size=32, robustifier=None):
        super().__init__()
        self.sdf = SDF()
        self.register_buffer('faces', torch.as_tensor(np.asarray(faces).astype(np.int32)))
        self.grid_size, self.robustifier = grid_size, robustifier

    def forward(self, vertices, translation, scale_factor=0.2):
        n = vertices.shape[0]
        vertices = vertices + translation.unsqueeze(dim=1)
        loss = torch.tensor(0., device=vertices.device)
        if n == 1:
            return loss
        with torch.no_grad():
            lo, hi = vertices.min(dim=1)[0], vertices.max(dim=1)[0]                          # [n,3] bounding boxes
            apart = ((lo[:, None] > hi[None]) | (lo[None] > hi[:, None])).any(-1)             # [n,n] boxes i, j disjoint
            apart = apart | torch.eye(n, dtype=torch.bool, device=vertices.device)
            isolated = (apart.sum(1) - 1) > 0                                                 # sic: ANY disjoint partner
            keep = ~isolated
        if keep.sum() == 0:
            return loss
        vertices = vertices[keep].contiguous()
        lo, hi = lo[keep], hi[keep]
        center = ((lo + hi) / 2).unsqueeze(1)
        scale = ((1 + scale_factor) * 0.5 * (hi - lo).max(dim=-1)[0])[:, None, None]
        with torch.no_grad():
            phi = self.sdf(self.faces, (vertices - center) / scale, self.grid_size)
        m = vertices.shape[0]
        for i in range(m):
            w = torch.ones(m, 1, device=vertices.device)
            w[i, 0] = 0.
            local = ((vertices - center[i].unsqueeze(0)) / scale[i].unsqueeze(0)).view(1, -1, 1, 1, 3)
            val = nn.functional.grid_sample(phi[i][None, None], local, align_corners=False).view(m, -1)
            cur = w * val
            if self.robustifier:
                frac = (cur / self.robustifier) ** 2
                cur = frac / (frac + 1)
            loss = loss + cur.sum() / m ** 2
        return loss
