"""CPU oracle for the CINeMA conditional-INR hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional torch/numpy on the CPU, the algorithm of the
reference's hot path (affine registration + trilinear latent interpolation -> SIREN stack
-> loss, plus the atlas helpers).  It exists to CHECK the HIP kernels; nothing in the
product package (`cinema_amd/`) imports it.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import this module.

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned by vectors generated in the build container by importing the reference's
own Python modules (`tests/golden/make_golden.py`, outputs committed under
`tests/golden/*.npz`).  `tests/test_oracle_golden.py` checks every function below against
those vectors.  Third-party arithmetic (torch ops, nibabel's `apply_affine`) is restated
from its documented formula; see DESIGN.md "Oracle".

Every function cites the reference file:line it follows (paths relative to the reference
repository root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# geometry: utils.py:126-166 (embed2affine / euler2rot), models/inr_decoder.py:64-71
# --------------------------------------------------------------------------------------


def rotation_zxz(theta: torch.Tensor) -> torch.Tensor:
    """Z-X-Z Euler angles -> rotation matrices (..., 3, 3).  utils.py:148-166.

    R = Rz(t1) @ Rx(t2) @ Rz(t3); written here as an explicit matrix product so the
    closed form of the reference is re-derived rather than transcribed.
    """
    t1, t2, t3 = theta[..., 0], theta[..., 1], theta[..., 2]
    zero, one = torch.zeros_like(t1), torch.ones_like(t1)

    def rz(a):
        c, s = torch.cos(a), torch.sin(a)
        return torch.stack([torch.stack([c, -s, zero], -1),
                            torch.stack([s, c, zero], -1),
                            torch.stack([zero, zero, one], -1)], -2)

    def rx(a):
        c, s = torch.cos(a), torch.sin(a)
        return torch.stack([torch.stack([one, zero, zero], -1),
                            torch.stack([zero, c, -s], -1),
                            torch.stack([zero, s, c], -1)], -2)

    return rz(t1) @ rx(t2) @ rz(t3)


def affine_from_embedding(embed: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """6/9/12-DoF embedding -> (A (...,3,3), t (...,3)).  utils.py:126-145.

    cols 0:3 Euler angles, 3:6 translation, 6:9 log-free scale (1+s) applied on the right,
    9:12 shear, applied as three unit-diagonal matrices Sx, Sy, Sz on the right (only when
    the embedding has exactly 12 entries).
    """
    A = rotation_zxz(embed[..., 0:3])
    t = embed[..., 3:6]
    d = embed.shape[-1]
    if d >= 9:
        A = A * (1.0 + embed[..., 6:9]).unsqueeze(-2)  # right-multiply by diag(1+s)
    if d == 12:
        hx, hy, hz = embed[..., 9], embed[..., 10], embed[..., 11]
        eye = torch.eye(3, dtype=embed.dtype).expand(embed.shape[:-1] + (3, 3))
        Sx, Sy, Sz = eye.clone(), eye.clone(), eye.clone()
        Sx[..., 0, 1], Sx[..., 0, 2] = hy, hz
        Sy[..., 1, 0], Sy[..., 1, 2] = hx, hz
        Sz[..., 2, 0], Sz[..., 2, 1] = hx, hy
        A = A @ Sx @ Sy @ Sz
    return A, t


def register_coords(coords: torch.Tensor, tfs: torch.Tensor) -> torch.Tensor:
    """c' = A c + t per sample.  models/inr_decoder.py:64-71 (inverse=False branch)."""
    A, t = affine_from_embedding(tfs)
    return (A @ coords.unsqueeze(-1)).squeeze(-1) + t


# --------------------------------------------------------------------------------------
# trilinear interpolation of the latent grid: models/inr_decoder.py:73-85
# (F.grid_sample 5-D, mode='bilinear', align_corners=True, padding_mode='border')
# --------------------------------------------------------------------------------------


def _axis_nodes(x: torch.Tensor, size: int):
    """Per-axis source index + weights with grid_sample's border/align_corners rule.

    u = (x+1)/2*(size-1), clipped to [0,size-1]; the gradient is cut where the clip is
    active (torch zeroes it for u<=0 and u>=size-1).
    """
    u = (x + 1.0) * 0.5 * (size - 1)
    inside = (u > 0) & (u < (size - 1))
    u = torch.where(inside, u, u.detach().clamp(0, size - 1))
    i0 = torch.floor(u.detach())
    w1 = u - i0
    w0 = 1.0 - w1
    i0 = i0.long()
    i1 = i0 + 1
    ok1 = i1 <= size - 1
    i1 = torch.where(ok1, i1, i0)
    w1 = torch.where(ok1, w1, torch.zeros_like(w1))
    return i0, i1, w0, w1


def trilinear_latent(coords: torch.Tensor, latents: torch.Tensor, subject: torch.Tensor,
                     condition_vecs: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Interpolate subject `subject[n]`'s latent grid at coords[n].

    latents (S, C, lx, ly, lz); coords (N,3) in [-1,1]: column 0 walks the LAST grid axis
    (lz), column 1 ly, column 2 lx (grid_sample's x,y,z = W,H,D convention).
    Equivalent to models/inr_decoder.py:33 ([idcs_df] gather) + :80-84, without
    materialising the per-sample grids.
    """
    S, C, lx, ly, lz = latents.shape
    s = subject.long()
    kz0, kz1, wz0, wz1 = _axis_nodes(coords[:, 0], lz)
    ky0, ky1, wy0, wy1 = _axis_nodes(coords[:, 1], ly)
    kx0, kx1, wx0, wx1 = _axis_nodes(coords[:, 2], lx)
    flat = latents.reshape(S, C, lx * ly * lz)
    out = 0.0
    for ix, wx in ((kx0, wx0), (kx1, wx1)):
        for iy, wy in ((ky0, wy0), (ky1, wy1)):
            for iz, wz in ((kz0, wz0), (kz1, wz1)):
                node = (ix * ly + iy) * lz + iz
                vals = flat[s, :, node]  # (N, C)
                out = out + vals * (wx * wy * wz).unsqueeze(-1)
    if condition_vecs is not None:
        out = torch.cat((out, condition_vecs), dim=-1)
    return out


# --------------------------------------------------------------------------------------
# SIREN stack: models/siren.py:6-61
# --------------------------------------------------------------------------------------


def siren_param_names(num_hidden_layers: int, modulated_layers: Sequence[int]) -> List[str]:
    """state_dict keys of the reference's Siren, in registration order (siren.py:44-57)."""
    names = ["final_linear.weight", "final_linear.bias"]  # registered before `net` (siren.py:50-57)
    for i in range(num_hidden_layers + 1):
        names += [f"net.{i}.linear.weight", f"net.{i}.linear.bias"]
        if i in modulated_layers:
            names += [f"net.{i}.linear_lats.weight", f"net.{i}.linear_lats.bias"]
    return names


def init_siren_params(in_size: int, lat_size: int, out_size: int, hidden: int, num_hidden_layers: int,
                      first_omega: float, hidden_omega: float, modulated_layers: Sequence[int],
                      generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """SIREN initialisation by the rule of siren.py:17-24,50-54 (own RNG stream: this is
    NOT bit-identical to constructing the reference module; parity tests copy weights)."""
    g = generator

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    p: Dict[str, torch.Tensor] = {}
    for i in range(num_hidden_layers + 1):
        fan_in = in_size if i == 0 else hidden
        om = first_omega if i == 0 else hidden_omega
        bound = 1.0 / fan_in if i == 0 else math.sqrt(6.0 / fan_in) / om
        p[f"net.{i}.linear.weight"] = uni((hidden, fan_in), bound)
        p[f"net.{i}.linear.bias"] = uni((hidden,), 1.0 / math.sqrt(fan_in))
        if i in modulated_layers and lat_size > 0:
            p[f"net.{i}.linear_lats.weight"] = uni((2 * hidden, lat_size), 1.0 / math.sqrt(lat_size))
            p[f"net.{i}.linear_lats.bias"] = uni((2 * hidden,), 1.0 / math.sqrt(lat_size))
    p["final_linear.weight"] = uni((out_size, hidden + lat_size), math.sqrt(6.0 / hidden) / hidden_omega)
    p["final_linear.bias"] = uni((out_size,), 1.0 / math.sqrt(hidden + lat_size))
    return p


def siren_forward(params: Dict[str, torch.Tensor], coords: torch.Tensor, latent: torch.Tensor,
                  num_hidden_layers: int, first_omega: float, hidden_omega: float) -> torch.Tensor:
    """x_{l+1} = sin(w z) or sin(w z s + t) with [s|t] = latent @ Wl^T + bl (siren.py:26-36);
    out = [x_L | latent] @ Wf^T + bf (siren.py:59-61)."""
    x = coords
    for i in range(num_hidden_layers + 1):
        om = first_omega if i == 0 else hidden_omega
        W, b = params[f"net.{i}.linear.weight"], params[f"net.{i}.linear.bias"]
        z = x @ W.t() + b
        key = f"net.{i}.linear_lats.weight"
        if key in params:
            H = W.shape[0]
            st = latent @ params[key].t() + params[f"net.{i}.linear_lats.bias"]
            x = torch.sin(om * z * st[:, :H] + st[:, H:])
        else:
            x = torch.sin(om * z)
    return torch.cat((x, latent), dim=-1) @ params["final_linear.weight"].t() + params["final_linear.bias"]


def decoder_forward(params: Dict[str, torch.Tensor], coords: torch.Tensor, latents: torch.Tensor,
                    condition_vecs: Optional[torch.Tensor], tfs: Optional[torch.Tensor], subject: torch.Tensor,
                    num_hidden_layers: int, first_omega: float, hidden_omega: float) -> torch.Tensor:
    """INR_Decoder.forward with an Identity modulator (models/inr_decoder.py:24-36).
    `params` uses the Siren key names (without the `sr_net.` prefix)."""
    c = register_coords(coords, tfs) if tfs is not None else coords
    lat = trilinear_latent(c, latents, subject, condition_vecs)
    return siren_forward(params, c, lat, num_hidden_layers, first_omega, hidden_omega)


# --------------------------------------------------------------------------------------
# loss: utils.py:77-104
# --------------------------------------------------------------------------------------


def criterion(output: torch.Tensor, target: torch.Tensor, tfs: Optional[torch.Tensor], sr_dims: int,
              loss_metric: str, class_weights: Optional[torch.Tensor], tf_weight: float,
              sr_weight: float = 1.0, seg_weight: float = 1.0) -> Dict[str, torch.Tensor]:
    """Dict of losses with the reference's keys and its quirks: `seg` already carries
    seg_weight and `total` multiplies by it again (utils.py:95,103); `tf_scale` only for
    9-wide tf rows (utils.py:100); tfs=None raises KeyError like the reference (:103)."""
    pred, tgt = output[..., :sr_dims], target[..., :sr_dims]
    diff = pred - tgt
    sr = (diff * diff).mean() if loss_metric == "mse" else diff.abs().mean()
    loss = {"seg": torch.tensor(0.0), "sr": sr_weight * sr, "trafo": torch.tensor(0.0), "total": 0.0}
    if seg_weight > 0:
        logits = output[..., sr_dims:]
        y = target[..., -1].to(torch.int64)
        logp = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        nll = -logp.gather(-1, y.unsqueeze(-1)).squeeze(-1)
        if class_weights is None:
            seg = nll.mean()
        else:
            w = class_weights[y]
            seg = (w * nll).sum() / w.sum()
        loss["seg"] = seg_weight * seg
    if tfs is not None:
        loss["tf_rot"] = (tfs[..., 0:3] ** 2).mean()
        loss["tf_trans"] = (tfs[..., 3:6] ** 2).mean()
        loss["tf_scale"] = (tfs[..., 6:9] ** 2).mean() if tfs.shape[-1] == 9 else torch.tensor(0.0)
        loss["tf"] = loss["tf_rot"] + loss["tf_trans"] + loss["tf_scale"]
    loss["total"] = loss["sr"] + seg_weight * loss["seg"] + tf_weight * loss["tf"]
    return loss


# --------------------------------------------------------------------------------------
# GT sampling: data_loading/dataset.py:83-98, utils.py:604-622
# --------------------------------------------------------------------------------------


def normalize_intensities_np(values: np.ndarray, norm_type: str) -> np.ndarray:
    """clip >= 0, then per-column min-max or z-score (eps 1e-5) on all but the last column
    (the label).  utils.py:604-622.  float64 in, float64 out."""
    v = np.maximum(values, 0.0)
    inten = v[..., :-1]
    if norm_type == "minmax":
        lo, hi = inten.min(axis=0), inten.max(axis=0)
        inten = (inten - lo) / (hi - lo)
    elif norm_type == "zscore":
        inten = (inten - inten.mean(axis=0)) / (inten.std(axis=0) + 1e-5)
    v = v.copy()
    v[..., :-1] = inten
    return v


def foreground_coords_and_values(volumes: Sequence[np.ndarray], affine: np.ndarray, world_bbox: Sequence[float],
                                 norm_type: str = "minmax"):
    """Foreground voxel list of one subject.  data_loading/dataset.py:83-98.

    volumes: per-modality float64 arrays of equal shape, the LAST one is the segmentation.
    Returns (coords f32 (n,3), values f32 (n,M), voxel ijk int64 (n,3)).  Voxel order is
    C-order argwhere of seg>0; world = ijk @ A[:3,:3]^T + A[:3,3] in float64 (nibabel's
    apply_affine formula), minus the float64 mean, divided by world_bbox/2; one cast to
    float32 at the end (dataset.py:30-31).
    """
    seg = volumes[-1]
    ijk = np.argwhere(seg > 0)
    vals = np.stack([v[ijk[:, 0], ijk[:, 1], ijk[:, 2]] for v in volumes], axis=-1).astype(np.float64)
    A = np.asarray(affine, dtype=np.float64)
    world = ijk.astype(np.float64) @ A[:3, :3].T + A[:3, 3]
    world = world - world.mean(axis=0)
    coords = world / (np.asarray(world_bbox, dtype=np.float64) / 2.0)
    if not (coords.min() >= -1.0 and coords.max() <= 1.0):
        raise AssertionError("coordinates leave [-1, 1] (utils.py:353-367)")
    vals = normalize_intensities_np(vals, norm_type)
    return coords.astype(np.float32), vals.astype(np.float32), ijk


# --------------------------------------------------------------------------------------
# atlas helpers: utils.py:199-250, build_atlas.py:190-219, models/inr_decoder.py:50-57
# --------------------------------------------------------------------------------------


def world_grid(world_bbox: Sequence[float], spacing: Sequence[float], normed: bool = True):
    """Dense grid, x-major flattening ('ij'), x_i = i*spacing (/bbox*2-1).  utils.py:235-250."""
    axes = []
    for b, sp in zip(world_bbox, spacing):
        a = torch.arange(0, b, sp)
        if normed:
            a = a / b * 2 - 1
        axes.append(a)
    shape = [len(a) for a in axes]
    gx, gy, gz = torch.meshgrid(axes[0], axes[1], axes[2], indexing="ij")
    coords = torch.stack((gx, gy, gz), dim=-1).reshape(-1, 3)
    affine = torch.diag(torch.tensor([spacing[0], spacing[1], spacing[2], 1.0]))
    return coords, shape, affine, axes


def normalize_condition_value(value, c_min: float, c_max: float, cond_scale: float):
    """2*((v-min)/(max-min)-0.5)*cond_scale.  utils.py:199-205."""
    return 2 * ((value - c_min) / (c_max - c_min) - 0.5) * cond_scale


def condition_combinations(conditions: Optional[dict], constraints: dict, cond_scale: float) -> List[List[float]]:
    """Cartesian product of the (normalised) condition values, first key slowest.
    utils.py:208-232."""
    if conditions is None:
        return [[]]
    combos: List[List[float]] = [[]]
    for key, spec in conditions.items():
        nxt = []
        for prefix in combos:
            for v in spec["values"]:
                if spec["normed_values"]:
                    nv = v * cond_scale
                else:
                    nv = normalize_condition_value(v, constraints[key]["min"], constraints[key]["max"], cond_scale)
                nxt.append(prefix + [nv])
        combos = nxt
    return combos


def mean_latent(latents: torch.Tensor, condition_values: torch.Tensor, condition_mean: float, c_min: float,
                c_max: float, gaussian_span: float, cond_scale: float, n_max: int = 100) -> torch.Tensor:
    """Gaussian-kernel regression of the subject latents.  build_atlas.py:190-219."""
    sigma = 0.5 * gaussian_span * (2.0 / (c_max - c_min)) * cond_scale
    w = torch.exp(-(condition_values - condition_mean) ** 2 / (2 * sigma ** 2))
    n_max = min(n_max, len(w))
    order = torch.argsort(w, descending=True)
    w = w.clone()
    w[order[n_max:]] = 0
    w = w / w.sum()
    return (latents * w.view(-1, 1, 1, 1, 1)).sum(dim=0, keepdim=True)


def inference_postprocess(output: torch.Tensor, sr_dims: int, img_shape: Sequence[int]) -> torch.Tensor:
    """clamp + per-modality whole-volume min-max, argmax, softmax, channel concat.
    models/inr_decoder.py:50-57."""
    imgs = output[..., :sr_dims].clamp(0, 1)
    lo = imgs.min(dim=0, keepdim=True)[0]
    hi = imgs.max(dim=0, keepdim=True)[0]
    imgs = ((imgs - lo) / (hi - lo)).reshape(list(img_shape) + [-1])
    logits = output[..., sr_dims:]
    hard = torch.argmax(logits, dim=-1).reshape(list(img_shape) + [-1])
    soft = torch.softmax(logits, dim=-1).reshape(list(img_shape) + [-1])
    return torch.cat((imgs, hard, soft), dim=-1)


# --------------------------------------------------------------------------------------
# one whole training step on the CPU (the bench's cpu_baseline "port" leg)
# --------------------------------------------------------------------------------------


def training_step(params: Dict[str, torch.Tensor], latents: torch.Tensor, tf_table: torch.Tensor,
                  coords: torch.Tensor, target: torch.Tensor, subject: torch.Tensor,
                  condition_vecs: Optional[torch.Tensor], cfg: dict):
    """fwd + loss + bwd exactly as build_atlas.py:69-88 (without the optimiser step).
    Leaves .grad on every tensor that requires grad; returns the loss dict."""
    tfs = tf_table[subject.long()]
    out = decoder_forward(params, coords, latents, condition_vecs, tfs, subject,
                          cfg["num_hidden_layers"], cfg["omega"][0], cfg["omega"][1])
    loss = criterion(out, target, tfs, cfg["sr_dims"], cfg["loss_metric"], cfg.get("class_weights"),
                     cfg["tf_weight"], seg_weight=cfg.get("seg_weight", 1.0))
    loss["total"].backward()
    return loss, out
