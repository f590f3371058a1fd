"""Hot-path subset of the reference's `utils.py`: Criterion, the affine helpers and the atlas
grid / condition helpers, with the same names, argument meaning, dict keys and quirks.

Criterion's data terms and their gradient run in the K3 HIP kernel; embed2affine / euler2rot
run in the K1 HIP kernel.  The grid / condition helpers are host-side set-up code (a few
scalars or one arange per axis) and stay in torch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .functions import LossFunction


class Criterion(nn.Module):
    """Reference: utils.py:77-104.  Returns {'seg','sr','trafo','total','tf_rot','tf_trans','tf_scale','tf'}.

    Kept quirks: `seg` is already multiplied by seg_weight and `total` multiplies it again
    (utils.py:95,103); `tf_scale` only for 9-wide tf rows (utils.py:100); tfs=None raises KeyError.
    Only `total` carries gradient to the network output (the reference's callers only ever
    backpropagate `total`, build_atlas.py:83,87)."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.tf_weight = args["optimizer"]["tf_weight"]
        self.n_classes = args["inr_decoder"]["out_dim"][-1] - 1
        self.sr_dims = sum(args["inr_decoder"]["out_dim"][:-1])
        self.loss_metric = "mse" if args["optimizer"]["loss_metric"] == "mse" else "l1"
        cw = args["dataset"]["class_weights"]
        self.ce_weights = torch.tensor(cw, dtype=torch.float32, device=args["device"]) if cw is not None else None

    def forward(self, output, target, tfs, sr_weight=1.0, seg_weight=1.0):
        dev = output.device
        zero = torch.tensor(0.0, device=dev)
        total_data, sr_mean, ce_mean = LossFunction.apply(output, target, self.sr_dims, self.loss_metric, self.ce_weights,
                                                          float(sr_weight), float(seg_weight) ** 2 if seg_weight > 0 else 0.0)
        loss = {"seg": zero, "sr": sr_weight * sr_mean, "trafo": zero, "total": 0.0}
        if seg_weight > 0:
            loss["seg"] = seg_weight * ce_mean
        if tfs is not None:
            loss["tf_rot"] = torch.mean(tfs[..., :3] ** 2)
            loss["tf_trans"] = torch.mean(tfs[..., 3:6] ** 2)
            loss["tf_scale"] = torch.mean(tfs[..., 6:9] ** 2) if tfs.shape[-1] == 9 else zero
            loss["tf"] = loss["tf_rot"] + loss["tf_trans"] + loss["tf_scale"]
        loss["total"] = total_data + self.tf_weight * loss["tf"]
        return loss


def embed2affine(embed):
    """(…, 6|9|12) -> (R (…,3,3), t (…,3)).  Reference: utils.py:126-145; runs in the K1 kernel."""
    if embed.device.type != "cuda":
        raise _lib.CinemaHipError("cinema_amd runs on an MI355X only; there is no CPU fallback")
    lib = _lib.load()
    e = embed.detach().float().contiguous()
    rows = e.numel() // e.shape[-1]
    aff = torch.empty(rows, 12, device=e.device)
    stream = C.c_void_p(torch.cuda.current_stream(e.device).cuda_stream)
    _lib.check(lib.cinema_affine_from_embedding(_lib.ptr(e), rows, e.shape[-1], _lib.ptr(aff), stream),
               "cinema_affine_from_embedding")
    lead = tuple(embed.shape[:-1])
    return aff[:, :9].reshape(lead + (3, 3)), aff[:, 9:].reshape(lead + (3,))


def euler2rot(theta):
    """Z-X-Z Euler angles (…,3) -> (…,3,3).  Reference: utils.py:148-166."""
    pad = torch.zeros(theta.shape[:-1] + (3,), device=theta.device, dtype=theta.dtype)
    return embed2affine(torch.cat((theta, pad), dim=-1))[0]


def normalize_condition(args, condition_key, condition_values, cond_scale=None):
    """Reference: utils.py:199-205."""
    c_scale = args["atlas_gen"]["cond_scale"] if cond_scale is None else cond_scale
    lo = args["dataset"]["constraints"][condition_key]["min"]
    hi = args["dataset"]["constraints"][condition_key]["max"]
    return 2 * ((condition_values - lo) / (hi - lo) - 0.5) * c_scale


def denormalize_conditions(args, cond_key, values):
    """Reference: utils.py:625-639."""
    lo = args["dataset"]["constraints"][cond_key]["min"]
    hi = args["dataset"]["constraints"][cond_key]["max"]
    return (values / args["atlas_gen"]["cond_scale"] + 1) / 2 * (hi - lo) + lo


def generate_combinations(args_data, conditions, keys=None, idx=0, current=None, results=None):
    """Cartesian product of condition values, first key slowest.  Reference: utils.py:208-232."""
    if conditions is None:
        return [[]]
    combos = [[]]
    for key in (list(conditions.keys()) if keys is None else keys):
        spec = conditions[key]
        step = []
        for prefix in combos:
            for v in spec["values"]:
                nv = v * args_data["atlas_gen"]["cond_scale"] if spec["normed_values"] else normalize_condition(args_data, key, v)
                step.append(prefix + [nv])
        combos = step
    return combos


def generate_world_grid(args, normed=True, device="cpu"):
    """Dense evaluation grid, x-major ('ij') flattening.  Reference: utils.py:235-250."""
    bbox = args["dataset"]["world_bbox"]
    sp = args["atlas_gen"]["spacing"]
    axes = []
    for b, s in zip(bbox, sp):
        ax = torch.arange(0, b, s, device=device)
        axes.append(ax / b * 2 - 1 if normed else ax)
    shape = [int(a.shape[0]) for a in axes]
    grid = torch.stack(torch.meshgrid(axes[0], axes[1], axes[2], indexing="ij"), dim=-1).reshape(-1, 3)
    affine = torch.diag(torch.tensor([sp[0], sp[1], sp[2], 1.0], device=device))
    return grid, shape, affine


def normalize_intensities(values, norm_type):
    """Reference: utils.py:604-622 (numpy, float64; last column is the label)."""
    values = np.clip(values, 0, None)
    inten = values[..., :-1]
    if norm_type == "minmax":
        lo, hi = inten.min(axis=0), inten.max(axis=0)
        inten = (inten - lo) / (hi - lo)
    elif norm_type == "zscore":
        inten = (inten - inten.mean(axis=0)) / (inten.std(axis=0) + 1e-5)
    values[..., :-1] = inten
    return values


def to_device(x, device="cuda"):
    """Reference: utils.py:188-196 (the reference ignores nested `device`; this one honours it)."""
    if isinstance(x, torch.Tensor):
        return x.to(device, non_blocking=True)
    if isinstance(x, (list, tuple)):
        return type(x)(to_device(t, device) for t in x)
    if isinstance(x, dict):
        return {k: to_device(v, device) for k, v in x.items()}
    return x


# ----------------------------------------------------------------------------------------------
# volume I/O on either side of the hot path.  nibabel is replaced by cinema_amd.nifti.
# ----------------------------------------------------------------------------------------------


def harmonize_labels(subject_seg, dataset):
    """Reference: utils.py:169-185 (in place, sequential relabelling; order matters)."""
    if "dhcp_neo" in dataset:
        subject_seg[subject_seg == 4] = 0
        subject_seg[subject_seg == 8] = 6
        subject_seg[subject_seg == 9] = 7
    elif "dhcp_fetal" in dataset or "marsfet" in dataset:
        subject_seg[subject_seg == 4] = 0
        subject_seg[subject_seg == 5] = 4
        for v in (6, 7, 8, 9):
            subject_seg[subject_seg == v] = 3
    else:
        raise NotImplementedError("Atlas Type not implemented.")
    return subject_seg


def typecheck_img_affine(img, affine):
    """Reference: utils.py:290-299."""
    if isinstance(img, torch.Tensor):
        img = img.detach().cpu().numpy()
    if isinstance(affine, torch.Tensor):
        affine = affine.detach().cpu().numpy()
    if img.dtype == "int64" or img.dtype == "int32":
        img = img.astype(np.int16)
    elif img.dtype == "float64":
        img = img.astype(np.float32)
    return img, affine


def save_img(img, affine, output_path, filename):
    """Reference: utils.py:341-350 (int64/int32 -> int16, float64 -> float32, directories created)."""
    import os
    from . import nifti
    img, affine = typecheck_img_affine(img, affine)
    full_path = os.path.join(output_path, filename)
    os.makedirs(os.path.dirname(full_path), exist_ok=True)
    nifti.save(nifti.Nifti1Image(img, affine), full_path)
    print(f"Saved {filename} to {output_path}")


def save_atlas(args, atlas, affine, temp_steps, condition_vectors, epoch):
    """Reference: utils.py:253-271.  atlas [x, y, z, modalities, conditions, t]; one 4-D file per (modality, condition)."""
    if isinstance(atlas, torch.Tensor):
        atlas = atlas.detach().cpu().numpy()
    if isinstance(affine, torch.Tensor):
        affine = affine.detach().cpu().numpy()
    n_conds = atlas.shape[4]
    mod_labels = list(args["dataset"]["modalities"])
    if args["save_certainty_maps"]:
        mod_labels = mod_labels + [f"CertaintyMaps/{label}" for label in args["dataset"]["label_names"]]
    for c in range(n_conds):
        for i in range(len(mod_labels)):
            filename = f"{mod_labels[i]}_ga={temp_steps[0]}-{temp_steps[-1]}_cond={c}_ep={epoch}.nii.gz"
            save_img(atlas[..., i, c, :], affine=affine, output_path=args["output_dir"], filename=filename)
    print("Atlas saved to {}".format(args["output_dir"]))


def save_subject(args, img, affine, sub_row_df=None, sub_name=None, epoch=0, split="train"):
    """Reference: utils.py:302-338 without the epoch-0 ANTs registration of the acquired scans to the reconstruction
    (ANTs is out of scope; the reconstructed modalities are written exactly as the reference does)."""
    if sub_row_df is not None:
        sub_name = sub_row_df["subject_id"]
    elif sub_name is None:
        print("No subject name provided, assigning random subject name.")
        sub_name = "subject_" + str(np.random.randint(100000))
    img, affine = typecheck_img_affine(img, affine)
    modalities = args["dataset"]["modalities"]
    for i, mod in enumerate(modalities):
        if i == len(modalities) - 1 and "Seg" in mod:
            img_mod = img[..., i].astype(np.int16)
        else:
            img_mod = img[..., i].astype(np.float32)
        save_img(img_mod, affine, args["output_dir"], f"{split}/{mod}_{sub_name}_ep={epoch}.nii.gz")


def assert_correct_coord_normalization(coords, min_val=-1.0, max_val=1.0):
    """Reference: utils.py:353-367 (numpy array or tensor of shape (n, 3))."""
    lo, hi = coords.min(axis=0) if isinstance(coords, np.ndarray) else coords.min(dim=0)[0], \
        coords.max(axis=0) if isinstance(coords, np.ndarray) else coords.max(dim=0)[0]
    for a, name in enumerate("xyz"):
        assert float(lo[a]) >= min_val, f"min_{name} = {float(lo[a])} is less than min_val = {min_val}"
        assert float(hi[a]) <= max_val, f"max_{name} = {float(hi[a])} is greater than max_val = {max_val}"


def squeeze_modalities(modality, path, write2disk=False):
    """Reference: utils.py:369-382: dHCP volumes stored as (x, y, z, 1) are squeezed (and rewritten in place on disk)."""
    from . import nifti
    if modality.ndim == 4:
        mod_nii = nifti.Nifti1Image(np.squeeze(modality.get_fdata()), modality.affine)
        if write2disk:
            mod_nii.to_filename(path)
        return mod_nii
    return modality


def add_background_halo(label_names, seg_nii, halo_width=1.5, background_label_str="BG"):
    """Reference: utils.py:385-391 (scipy Gaussian on the host, as in the reference; runs once per subject at load)."""
    import scipy.ndimage as ndi
    from . import nifti
    seg = seg_nii.get_fdata()
    bg_label = label_names.index(background_label_str)
    mask_bg = (seg > 0).astype(np.float32)
    mask_bg = (ndi.gaussian_filter(mask_bg, sigma=halo_width) > 0.001).astype(np.uint8) * bg_label
    mask_bg[seg > 0] = seg[seg > 0]
    return nifti.Nifti1Image(mask_bg, seg_nii.affine)


def mask_nifti(nii, mask):
    """Reference: utils.py:394-397."""
    from . import nifti
    data = nii.get_fdata()
    data *= mask
    return nifti.Nifti1Image(data, nii.affine)
