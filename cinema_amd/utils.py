"""Hot-path subset of the reference's `utils.py`: Criterion, the affine helpers and the atlas
grid / condition helpers, with the same names, argument meaning, dict keys and quirks.

Criterion's data terms and their gradient run in the K3 HIP kernel; embed2affine / euler2rot
run in the K1 HIP kernel.  The grid / condition helpers are host-side set-up code (a few
scalars or one arange per axis) and stay in torch.
"""
from __future__ import annotations

import ctypes as C
import os

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


class WorldGrid:
    """The dense evaluation grid as its three axis vectors: `INR_Decoder.inference` makes each chunk's coordinates on the
    GPU from these (cinema_grid_coords) instead of holding the (G, 3) meshgrid (357 MB at 310^3).  `shape` and `len()`
    are those of the (G, 3) tensor it stands for; `materialize()` builds that tensor."""

    def __init__(self, axes):
        self.axes = [a.to(torch.float32).contiguous() for a in axes]
        self.dims = [int(a.shape[0]) for a in self.axes]
        self.shape = (self.dims[0] * self.dims[1] * self.dims[2], 3)
        self.device = self.axes[0].device

    def __len__(self):
        return self.shape[0]

    def materialize(self):
        return torch.stack(torch.meshgrid(*self.axes, indexing="ij"), dim=-1).reshape(-1, 3)


def generate_world_grid(args, normed=True, device="cpu", lazy=False):
    """Dense evaluation grid, x-major ('ij') flattening.  Reference: utils.py:235-250.
    lazy=True returns a WorldGrid (axis vectors only) in place of the (G, 3) tensor."""
    bbox = args["dataset"]["world_bbox"]
    sp = args["atlas_gen"]["spacing"]
    axes = []
    for b, s in zip(bbox, sp):
        ax = torch.arange(0, b, s, device=device)
        axes.append(ax / b * 2 - 1 if normed else ax)
    shape = [int(a.shape[0]) for a in axes]
    grid = WorldGrid(axes)
    affine = torch.diag(torch.tensor([sp[0], sp[1], sp[2], 1.0], device=device))
    return (grid if lazy else grid.materialize()), shape, affine


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
# atlas writer.  Behaviour kept from the reference's writer (SURVEY.md section 5 "Output
# naming"; utils.py:253-271, 341-350): one 4-d (x, y, z, t) file per channel and condition vector, named
# {channel}_ga={t0}-{tN}_cond={c}_ep={epoch}.nii.gz, certainty maps under CertaintyMaps/{label}; 64/32-bit integers are
# stored as int16 and float64 as float32.  nibabel is replaced by cinema_amd.nifti.
# ----------------------------------------------------------------------------------------------
_STORED_AS = {np.dtype(np.int64): np.int16, np.dtype(np.int32): np.int16, np.dtype(np.float64): np.float32}


def _on_host(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def save_img(img, affine, output_path, filename):
    """Write one volume to output_path/filename (directories created); returns the path written."""
    from . import nifti
    data = _on_host(img)
    data = data.astype(_STORED_AS.get(data.dtype, data.dtype), copy=False)
    target = os.path.join(output_path, filename)
    os.makedirs(os.path.dirname(target), exist_ok=True)
    nifti.save(nifti.Nifti1Image(data, _on_host(affine)), target)
    return target


def atlas_channel_names(args):
    """Channel axis of an atlas array: the modalities, then (save_certainty_maps) one certainty map per label."""
    names = list(args["dataset"]["modalities"])
    if args["save_certainty_maps"]:
        names += ["CertaintyMaps/" + label for label in args["dataset"]["label_names"]]
    return names


def save_atlas(args, atlas, affine, temp_steps, condition_vectors, epoch):
    """atlas [x, y, z, channel, condition, t] -> one 4-d file per (condition, channel); returns the paths."""
    vols = _on_host(atlas)
    span = f"{temp_steps[0]}-{temp_steps[-1]}"
    return [save_img(vols[:, :, :, ch, c, :], affine, args["output_dir"], f"{name}_ga={span}_cond={c}_ep={epoch}.nii.gz")
            for c in range(vols.shape[4]) for ch, name in enumerate(atlas_channel_names(args))]
