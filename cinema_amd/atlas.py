"""Atlas generation drivers of the reference's AtlasBuilder (build_atlas.py:128-219) above INR_Decoder.inference.

`mean_latent` is get_mean_latent (build_atlas.py:190-219): a Gaussian-weighted average of the subjects' latent grids
around a condition value, clipped to the n_max closest subjects.  `generate_subject_from_latent` and
`generate_atlas` mirror build_atlas.py:128-137 and 159-188.  With a process group the subjects' latents stay sharded:
ranks exchange only the scalar weights (S floats) and the partial weighted sums (C*lx*ly*lz floats), and the
independent (time point x condition) volumes are split round-robin over ranks.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import parallel as par
from .utils import generate_combinations, generate_world_grid, normalize_condition, save_atlas


def gaussian_sigma(args: dict, condition_key: str) -> float:
    """build_atlas.py:204-207."""
    c = args["dataset"]["constraints"][condition_key]
    c_ratio = 2 / (c["max"] - c["min"])
    return 0.5 * args["atlas_gen"]["gaussian_span"] * c_ratio * args["atlas_gen"]["cond_scale"]


def mean_latent(args: dict, latents: torch.Tensor, condition_values: torch.Tensor, condition_key: str, condition_mean,
                n_max: int = 100, process_group=None) -> torch.Tensor:
    """latents (S_local, C, lx, ly, lz) and condition_values (S_local,) of THIS rank's subjects -> (1, C, lx, ly, lz),
    identical on every rank.  Ties in the top-n_max selection follow torch.argsort on the rank-order concatenation,
    i.e. on the reference's dataframe order when subjects are sharded with parallel.shard_subjects."""
    if condition_values.shape[0] != latents.shape[0]:
        raise AssertionError("Condition values (all entries from the dataframe) and latents must have the same length!")
    sigma = gaussian_sigma(args, condition_key)
    rank, world = par.world_info(process_group)
    cv = condition_values.to(latents.device, torch.float32)
    w_local = torch.exp(-(cv - condition_mean) ** 2 / (2 * (sigma ** 2)))
    w_all = par.gather_rows(w_local, process_group)
    n_keep = min(n_max, w_all.shape[0])
    w_all = w_all.clone()
    w_all[torch.argsort(w_all, descending=True)[n_keep:]] = 0
    total = torch.sum(w_all)
    if world > 1:
        sizes = par.gather_rows(torch.tensor([w_local.shape[0]], device=latents.device), process_group)
        lo = int(sizes[:rank].sum().item())
        w_mine = w_all[lo:lo + w_local.shape[0]]
    else:
        w_mine = w_all
    w_mine = (w_mine / total)[:, None, None, None, None]
    part = torch.sum(latents.detach() * w_mine, dim=0, keepdim=True)
    if world > 1:
        part = par.allreduce_sum(part, process_group)
    return part


def generate_subject_from_latent(decoder, args: dict, latent_vec, condition_vector, transformation=None, device=None):
    """build_atlas.py:128-137 (autocast is replaced by the decoder's own precision switch)."""
    device = device or args["device"]
    grid_coords, grid_shape, _ = generate_world_grid(args, device=device, lazy=True)
    with torch.no_grad():
        return decoder.inference(grid_coords, latent_vec, condition_vector, grid_shape, transformation)


def atlas_jobs(args: dict) -> Tuple[List[float], List[List[float]], List[Tuple[int, int]]]:
    temp_steps = list(args["atlas_gen"]["temporal_values"])
    condition_vectors = generate_combinations(args, args["atlas_gen"]["conditions"])
    jobs = [(ti, ci) for ti in range(len(temp_steps)) for ci in range(len(condition_vectors))]
    return temp_steps, condition_vectors, jobs


def generate_atlas(decoder, args: dict, latents: torch.Tensor, scan_age_values: torch.Tensor, n_max: int = 100,
                   process_group=None, device=None, epoch: Optional[int] = None):
    """build_atlas.py:159-188.  Returns (atlas, affine, temp_steps, condition_vectors) with atlas
    [x, y, z, modalities, conditions, t] on the CPU when run on one rank; with a process group every rank returns
    {(ti, ci): volume} for the jobs it owns instead of the stacked tensor.  With `epoch` given the atlas is also written
    like utils.save_atlas (utils.py:253-271) does -- under a process group the volumes are first brought to rank 0
    (parallel.collect_jobs), which writes the files."""
    device = device or args["device"]
    rank, world = par.world_info(process_group)
    decoder.eval()
    grid_coords, grid_shape, affine = generate_world_grid(args, device=device, lazy=True)
    temp_steps, condition_vectors, jobs = atlas_jobs(args)
    mine = set(par.shard_jobs(len(jobs), rank, world))
    volumes: Dict[Tuple[int, int], torch.Tensor] = {}
    with torch.no_grad():
        for ti, temp_step in enumerate(temp_steps):
            temp_normed = normalize_condition(args, "scan_age", temp_step)
            ml = mean_latent(args, latents, scan_age_values, "scan_age", temp_normed, n_max=n_max,
                             process_group=process_group)            # collective: every rank, every time point
            for ci, c_v in enumerate(condition_vectors):
                if ti * len(condition_vectors) + ci not in mine:
                    continue
                c_v = torch.tensor(c_v, dtype=torch.float32).to(device)
                values_p = decoder.inference(grid_coords, ml, c_v, grid_shape, None)
                seg = values_p[:, :, :, -1]
                seg[seg == 4] = 0                                       # build_atlas.py:178-180
                values_p[:, :, :, -1] = seg
                volumes[(ti, ci)] = values_p.detach().cpu()
    n_c = len(condition_vectors)
    if world > 1:
        if epoch is not None:
            flat = par.collect_jobs({ti * n_c + ci: v for (ti, ci), v in volumes.items()}, len(jobs), process_group)
            if rank == 0:
                save_atlas(args, _stack_atlas(flat, len(temp_steps), n_c), affine, temp_steps, condition_vectors, epoch)
        return volumes, affine, temp_steps, condition_vectors
    atlas = _stack_atlas([volumes[j] for j in jobs], len(temp_steps), n_c)
    if epoch is not None:
        save_atlas(args, atlas, affine, temp_steps, condition_vectors, epoch)
    return atlas, affine, temp_steps, condition_vectors


def _stack_atlas(flat: List[torch.Tensor], n_t: int, n_c: int) -> torch.Tensor:
    """[t-major list of (x, y, z, m)] -> [x, y, z, m, conditions, t] (build_atlas.py:185-186)."""
    return torch.stack([torch.stack([flat[ti * n_c + ci] for ci in range(n_c)], dim=-1) for ti in range(n_t)], dim=-1)
