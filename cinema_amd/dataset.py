"""Ground-truth side of the hot path: NIfTI volumes -> resident, GPU-side foreground lists.

Reference behaviour: data_loading/dataset.py:26-47 (`__getitem__`, `collate_fn`), :49-81 (modality loading and checks),
:83-98 (`load_coords_and_values`), :126-135 (condition normalisation).  The reference rebuilds every subject's (coords, values) list on
16 CPU DataLoader workers every epoch and ships it over PCIe; here each subject is read ONCE, its foreground list is
extracted on the GPU (compaction, float64 affine, centre of mass, per-subject normalisation) and stays in HBM as a
compact table (per foreground voxel: coords 12 B, intensities 4*sr_dims B, label 1 B) instead of whole volumes.
Subject curation (tsv filtering, stratified sampling, torchio augmentation) is out of scope; `rows` are the already
selected dataframe rows as dicts.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import nifti


def read_subject_volumes(args: dict, row: dict):
    """One subject's modalities as float64 arrays + their common affine (reference behaviour: data_loading/dataset.py:49-81).

    Every key of args['dataset']['modalities'] must be present in `row` with a non-empty path; all volumes must agree in
    shape and (to 1e-6) in affine, else ValueError.  dHCP files stored as (x, y, z, 1) are read as 3-d; unlike the
    reference the file on disk is left as it is."""
    names = list(args["dataset"]["modalities"])
    absent = [k for k in names if not row.get(k)]
    if absent:
        raise ValueError(f"subject row has no file for {absent}")
    vols, affines = [], []
    for k in names:
        img = nifti.load(row[k])
        data = img.get_fdata()
        vols.append(data.reshape(data.shape[:3]) if data.ndim == 4 and data.shape[3] == 1 else data)
        affines.append(np.asarray(img.affine, dtype=np.float64))
    if any(v.shape != vols[0].shape for v in vols):
        raise ValueError(f"modalities differ in shape: { {k: v.shape for k, v in zip(names, vols)} }")
    if any(not np.allclose(a, affines[0], atol=1e-6) for a in affines[1:]):
        raise ValueError(f"modalities differ in affine: {names}")
    return vols, affines[0]


def _blur_axis(x: torch.Tensor, axis: int, w: torch.Tensor) -> torch.Tensor:
    """1-d correlation along `axis` with edge-mirrored borders (d c b a | a b c d | d c b a)."""
    r = (w.numel() - 1) // 2
    x = x.movedim(axis, -1)
    L = x.shape[-1]
    if L < r:
        raise ValueError(f"volume axis of {L} voxels is shorter than the halo kernel radius {r}")
    xp = torch.cat((x[..., :r].flip(-1), x, x[..., L - r:].flip(-1)), dim=-1)
    out = torch.zeros_like(x)
    for k in range(w.numel()):
        out += w[k] * xp[..., k:k + L]
    return out.movedim(-1, axis)


def premask_on_device(vols: Sequence[torch.Tensor], label_names: Sequence[str], halo_sigma: float = 1.5,
                      bg_label: str = "BG") -> List[torch.Tensor]:
    """args['mask_reconstruction'] at load time, on the device the volumes live on: intensities are zeroed outside
    seg > 0, and the segmentation gains a background shell -- every voxel outside the foreground where a Gaussian blur
    (sigma 1.5 voxels, truncated at 4 sigma, as scipy.ndimage.gaussian_filter) of the foreground mask exceeds 0.001 gets
    the index of 'BG' in label_names.  Reference behaviour: utils.py:385-397 called from dataset.py:75-79."""
    seg = vols[-1]
    fg = seg > 0
    r = int(4.0 * halo_sigma + 0.5)
    t = torch.arange(-r, r + 1, dtype=torch.float64, device=seg.device)
    w = torch.exp(-0.5 * (t / halo_sigma) ** 2)
    w = w / w.sum()
    blur = fg.to(torch.float64)
    for axis in range(3):
        blur = _blur_axis(blur, axis, w).to(torch.float32).to(torch.float64)     # scipy keeps float32 between the passes
    shell = (blur > 0.001) & ~fg
    new_seg = torch.where(shell, torch.full_like(seg, float(list(label_names).index(bg_label))), seg * fg)
    return [v * fg for v in vols[:-1]] + [new_seg]


def condition_row(args: dict, row: dict, normalize: bool = True) -> torch.Tensor:
    """The enabled conditions of a subject, in config order, mapped to cond_scale * [-1, 1] over the configured
    [min, max] (reference behaviour: data_loading/dataset.py:126-135; `row` is not modified)."""
    ds = args["dataset"]
    vals = []
    for key, enabled in ds["conditions"].items():
        if not enabled:
            continue
        v = float(row[key])
        if normalize:
            lo, hi = ds["constraints"][key]["min"], ds["constraints"][key]["max"]
            v = (2.0 * (v - lo) / (hi - lo) - 1.0) * args["atlas_gen"]["cond_scale"]
        vals.append(v)
    return torch.tensor(vals, dtype=torch.float32)


def foreground_on_device(volumes: Sequence, affine, world_bbox, norm_type: str, device, normalize: bool = True
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """`load_coords_and_values` (dataset.py:83-98) + `normalize_intensities` (utils.py:604-622) on the GPU.

    volumes: per-modality arrays of equal shape (numpy or tensors, any real dtype; they are the `get_fdata()` float64
    arrays of the reference), the LAST is the segmentation.  Returns
      coords (n,3) fp32, values (n,M) fp32 (intensity columns normalised, last column = label), voxel (n,) int64
    in C-order argwhere order -- the order, the labels and the normalised values are bit-identical to the reference's
    float64 arithmetic; the coordinates are the same float64 formula (A ijk + t, minus the foreground mean, / (bbox/2),
    one cast to fp32) with the device's summation order in the mean, i.e. equal to within one fp32 ulp."""
    vols = [torch.as_tensor(np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v).to(device) for v in volumes]
    shape = tuple(vols[-1].shape)
    if len(shape) != 3 or any(tuple(v.shape) != shape for v in vols):
        raise ValueError(f"foreground_on_device: need equal 3-d volumes, got {[tuple(v.shape) for v in vols]}")
    X, Y, Z = shape
    vox = torch.nonzero(vols[-1].reshape(-1) > 0, as_tuple=False).reshape(-1)          # ascending linear index = C order
    if vox.numel() == 0:
        raise ValueError("foreground_on_device: the segmentation has no foreground voxel")
    values = torch.stack([v.reshape(-1)[vox].to(torch.float64) for v in vols], dim=1)
    del vols
    i = torch.div(vox, Y * Z, rounding_mode="floor")
    j = torch.div(vox - i * (Y * Z), Z, rounding_mode="floor")
    k = vox - i * (Y * Z) - j * Z
    A = torch.as_tensor(np.asarray(affine, dtype=np.float64)).to(device)
    ijk = torch.stack((i, j, k), dim=1).to(torch.float64)
    world = ijk @ A[:3, :3].T + A[:3, 3]
    coords = world - world.mean(dim=0)
    if normalize:
        coords = coords / (torch.as_tensor(np.asarray(world_bbox, dtype=np.float64)).to(device) / 2.0)
        lo, hi = float(coords.min()), float(coords.max())
        if lo < -1.0 or hi > 1.0:       # utils.py:353-367
            raise AssertionError(f"normalised coordinates span [{lo}, {hi}]: the subject leaves world_bbox")
        values = torch.clamp(values, min=0.0)
        inten = values[:, :-1]
        if norm_type == "minmax":
            mn, mx = inten.min(dim=0)[0], inten.max(dim=0)[0]
            inten = (inten - mn) / (mx - mn)
        elif norm_type == "zscore":
            inten = (inten - inten.mean(dim=0)) / (inten.std(dim=0, unbiased=False) + 1e-5)
        values = torch.cat((inten, values[:, -1:]), dim=1)
    return coords.to(torch.float32), values.to(torch.float32), vox


def subject_store_from_rows(args: dict, rows: Sequence[dict], device, subject_ids: Optional[Sequence[int]] = None):
    """Read every row's modalities once and keep only the foreground tables resident (compact SubjectStore: per-voxel
    intensities / labels indexed by a global row number, voxels_per_subject = 0).  `subject_ids` maps rows to this
    rank's local subject slots (default 0..len(rows)-1).  Also returns the (S, cond_dims) condition table."""
    from .trainer import SubjectStore
    keys = args["dataset"]["modalities"]
    sr_dims = len(keys) - 1
    cs, its, lbs, ss, conds = [], [], [], [], []
    for slot, row in enumerate(rows):
        vols, affine = read_subject_volumes(args, row)
        if args.get("mask_reconstruction", False):
            vols = premask_on_device([torch.as_tensor(np.ascontiguousarray(v)).to(device) for v in vols],
                                     args["dataset"]["label_names"])
        c, v, _ = foreground_on_device(vols, affine, args["dataset"]["world_bbox"], args["dataset"]["normalize_values"], device)
        cs.append(c)
        its.append(v[:, :sr_dims].contiguous())
        lbs.append(v[:, -1].to(torch.uint8))                   # target[:, -1].long() of the reference (utils.py:94)
        sid = slot if subject_ids is None else int(subject_ids[slot])
        ss.append(torch.full((c.shape[0],), sid, dtype=torch.int32, device=device))
        conds.append(condition_row(args, row))
    n = sum(int(c.shape[0]) for c in cs)
    cond = torch.stack(conds).to(device) if conds and conds[0].numel() else None
    store = SubjectStore(torch.cat(its), torch.cat(lbs), 0, torch.cat(cs), torch.arange(n, device=device), torch.cat(ss), cond,
                         n_subjects=len(rows))
    return store
