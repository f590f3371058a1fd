"""Post-hoc masking of a reconstructed volume (behaviour of the reference's
`INR_Decoder.mask_reconstruction` / `connected_components`, models/inr_decoder.py:87-107) on the GPU.

What the reference does, on the host through scipy: foreground = hard label > 0 and != 'BG'; its 6-connected components
are labelled; the label that occurs most often in the central patch (10 % of each axis, background counted as label 0,
ties to the smaller label) is kept; the kept component is widened by a Gaussian (sigma 1 voxel) thresholded at 0.001;
every channel of the reconstruction is multiplied by that mask.  Here the labelling is `cinema_cc_label` (label = 1 +
smallest linear index of the component, which orders components like scipy's raster-scan numbering, so the same
component wins a tie) and the rest are device tensor ops; nothing leaves the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib
from .dataset import _blur_axis


def component_labels(mask: torch.Tensor, sweeps_per_call: int = 8) -> torch.Tensor:
    """int32 labels of the 6-connected components of a 3-d boolean mask (0 outside)."""
    lib = _lib.load()
    if not mask.is_cuda or mask.dim() != 3:
        raise ValueError("component_labels: need a 3-d mask on the GPU")
    m8 = mask.to(torch.uint8).contiguous()
    X, Y, Z = m8.shape
    labels = torch.empty((X, Y, Z), dtype=torch.int32, device=mask.device)
    changed = torch.zeros(1, dtype=torch.int32, device=mask.device)
    stream = C.c_void_p(torch.cuda.current_stream(mask.device).cuda_stream)
    init = 1
    for _ in range(100000):
        _lib.check(lib.cinema_cc_label(_lib.ptr(m8), X, Y, Z, _lib.ptr(labels), _lib.ptr(changed), sweeps_per_call, init, stream),
                   "cinema_cc_label")
        init = 0
        if int(changed.item()) == 0:
            return labels
    raise RuntimeError("component_labels: label propagation did not settle")


def central_component_mask(seg_hard: torch.Tensor, label_names: Sequence[str], bg_label_str: str = "BG") -> torch.Tensor:
    """(X, Y, Z) float mask of the widened central component of the foreground of a hard segmentation (X, Y, Z)."""
    bg = list(label_names).index(bg_label_str)
    fg = (seg_hard > 0) & (seg_hard != bg)
    labels = component_labels(fg)
    dims = list(fg.shape)
    half = [int((d * 0.1) // 2) for d in dims]
    mid = [d // 2 for d in dims]
    patch = labels[mid[0] - half[0]:mid[0] + half[0], mid[1] - half[1]:mid[1] + half[1], mid[2] - half[2]:mid[2] + half[2]]
    if patch.numel() == 0:
        raise ValueError("central_component_mask: the central patch (10 % of every axis) is empty for a volume of shape "
                         f"{tuple(dims)} (the reference fails here too)")
    values, counts = torch.unique(patch, return_counts=True)          # ascending labels: argmax takes the smallest on a tie
    chosen = values[torch.argmax(counts)]
    keep = (fg & (labels == chosen)).to(torch.float64)
    t = torch.arange(-4, 5, dtype=torch.float64, device=keep.device)   # sigma 1, truncated at 4 sigma (scipy's default)
    w = torch.exp(-0.5 * t * t)
    w = w / w.sum()
    for axis in range(3):
        keep = _blur_axis(keep, axis, w).to(torch.float32).to(torch.float64)   # scipy keeps float32 between the passes
    return (keep > 0.001).to(torch.float32)


def mask_reconstruction(recs: torch.Tensor, sr_dims: int, label_names: Sequence[str]) -> torch.Tensor:
    """recs (X, Y, Z, sr_dims + 1 + n_cls) as `INR_Decoder.inference` assembles it; channel sr_dims is the hard label."""
    mask = central_component_mask(recs[..., sr_dims], label_names)
    return recs * mask.unsqueeze(-1)
