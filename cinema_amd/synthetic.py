"""Deterministic synthetic subjects for benchmarks and tests: per subject
an ellipsoid "brain" in a cubic grid, labels 1..9 as concentric shells split by octant, two smooth
intensity modalities with noise, affine diag(spacing) + translation.  Built with torch ops on the
target device (no dataset exists offline)."""
from __future__ import annotations

from typing import List, Tuple

import torch


def make_subject(shape: Tuple[int, int, int], spacing: float, n_modalities: int, n_labels: int, seed: int, device):
    """Returns (intens (V, n_modalities) fp32 min-max normalised over the foreground, labels (V,) uint8,
    affine (4,4) float64) with V = prod(shape), C-order voxel index."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    X, Y, Z = shape
    ext = torch.tensor([X, Y, Z], dtype=torch.float64) * spacing
    semi = (torch.rand(3, generator=g, dtype=torch.float64) * 15.0 + 45.0) / 155.0 * ext      # U(45,60) mm at bbox 155
    centre = ext / 2 + (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 0.04 * ext
    ax = [torch.arange(n, device=device, dtype=torch.float32) * spacing for n in (X, Y, Z)]
    dx = ((ax[0] - float(centre[0])) / float(semi[0])).view(X, 1, 1)
    dy = ((ax[1] - float(centre[1])) / float(semi[1])).view(1, Y, 1)
    dz = ((ax[2] - float(centre[2])) / float(semi[2])).view(1, 1, Z)
    r2 = dx * dx + dy * dy + dz * dz
    inside = r2 < 1.0
    shell = torch.clamp((r2.sqrt() * 3).floor(), max=2).to(torch.int32)             # 3 concentric shells
    octant = ((dx > 0).to(torch.int32) + (dy > 0).to(torch.int32) + (dz > 0).to(torch.int32)) % 3
    labels = torch.where(inside, 1 + (shell * 3 + octant) % (n_labels - 1), torch.zeros_like(shell)).to(torch.uint8)
    gd = torch.Generator(device=device).manual_seed(seed + 7919)
    mods = []
    for m in range(n_modalities):
        smooth = 0.5 + 0.35 * torch.cos(3.0 * dx + m) * torch.cos(2.0 * dy - m) + 0.15 * torch.sin(4.0 * dz + 0.5 * m)
        noisy = smooth + 0.05 * torch.randn(shape, generator=gd, device=device)
        mods.append(torch.clamp(noisy, min=0.0))
    fg = labels.reshape(-1) > 0
    cols = []
    for m in mods:
        v = m.reshape(-1)
        lo, hi = v[fg].min(), v[fg].max()
        cols.append((v - lo) / (hi - lo))
    intens = torch.stack(cols, dim=1).contiguous()
    affine = torch.eye(4, dtype=torch.float64)
    affine[0, 0] = affine[1, 1] = affine[2, 2] = spacing
    affine[:3, 3] = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 10.0
    return intens, labels.reshape(-1).contiguous(), affine


def foreground_lists(labels: torch.Tensor, shape, affine: torch.Tensor, world_bbox) -> Tuple[torch.Tensor, torch.Tensor]:
    """Foreground voxel list of one subject as the reference builds it (data_loading/dataset.py:87-96):
    C-order argwhere(seg > 0); world = A ijk + t, minus the float64 centre of mass, / (bbox/2), cast to f32.
    Returns (coords (n,3) fp32, voxel (n,) int64 linear voxel index)."""
    X, Y, Z = shape
    vox = torch.nonzero(labels > 0, as_tuple=False).reshape(-1)          # ascending = C-order
    i = torch.div(vox, Y * Z, rounding_mode="floor")
    j = torch.div(vox - i * (Y * Z), Z, rounding_mode="floor")
    k = vox - i * (Y * Z) - j * Z
    ijk = torch.stack((i, j, k), dim=1).to(torch.float64)
    A = affine.to(labels.device)
    world = ijk @ A[:3, :3].T + A[:3, 3]
    world = world - world.mean(dim=0)
    coords = world / (torch.tensor(world_bbox, dtype=torch.float64, device=labels.device) / 2.0)
    return coords.to(torch.float32), vox
