"""Post-hoc brain masking of a reconstructed volume (reference: models/inr_decoder.py:87-107).

Out of the hot-path scope: sequential connected-component labelling
with scipy on the host, exactly where the reference runs it.  Kept so that
`INR_Decoder.inference` honours `mask_reconstruction: True`.
"""
import numpy as np
import scipy.ndimage as ndi
import torch


def largest_central_component(seg: torch.Tensor, label_names, bg_label_str="BG") -> torch.Tensor:
    bg = label_names.index(bg_label_str)
    fg = ((seg > 0) & (seg != bg)).detach().cpu().numpy()
    shp = np.array(fg.shape[:-1])
    half = ((shp * 0.1) // 2).astype(int)
    mid = shp // 2
    comp, _ = ndi.label(fg)
    patch = comp[mid[0] - half[0]:mid[0] + half[0], mid[1] - half[1]:mid[1] + half[1], mid[2] - half[2]:mid[2] + half[2]]
    keep = np.argmax(np.bincount(patch.flatten()))
    fg = fg * (comp == keep)
    soft = ndi.gaussian_filter(fg.astype(np.float32), sigma=1.0) > 0.001
    return torch.from_numpy(soft.astype(np.uint8)).to(seg.device, torch.float)


def mask_reconstruction(recs: torch.Tensor, sr_dims: int, label_names) -> torch.Tensor:
    seg_hard = recs[..., sr_dims:sr_dims + 1]
    return recs * largest_central_component(seg_hard, label_names).expand_as(recs)
