"""World-size-2 `gloo` tests (CPU) of the multi-GPU restatement of the hot path:
subjects sharded, MLP replicated, one averaged all-reduce of the flat gradient per step, atlas weights and
partial sums exchanged for the mean latent.  The per-rank arithmetic here is the CPU oracle -- the collective
plumbing (cinema_amd/parallel.py, cinema_amd/atlas.py) is what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(__file__))
from conftest import load_golden, rel_err  # noqa: E402

from cinema_amd import parallel as par  # noqa: E402

WORLD = 2
HID, LAYERS, MOD, LAT, OUT = 32, 2, [1], [8, 2, 2, 2], 5      # out = 1 modality + 4 classes
S_TOTAL, N_PER_RANK = 5, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(fn, *extra):
    port = _free_port()
    mp.spawn(_entry, args=(fn, port, extra), nprocs=WORLD, join=True)


def _entry(rank, fn, port, extra):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        fn(rank, *extra)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ partitioning (single process)
def test_shard_subjects_is_a_balanced_partition():
    for n in (0, 1, 5, 64, 423):
        for world in (1, 2, 3, 8):
            shards = [par.shard_subjects(n, r, world) for r in range(world)]
            assert sum(shards, []) == list(range(n))
            sizes = [len(s) for s in shards]
            assert max(sizes) - min(sizes) <= 1
            for r, sh in enumerate(shards):
                assert all(par.owner_of(s, n, world) == r for s in sh)
    with pytest.raises(ValueError):
        par.shard_subjects(4, 2, 2)
    jobs = [par.shard_jobs(7, r, 3) for r in range(3)]
    assert sorted(sum(jobs, [])) == list(range(7))


def test_single_process_helpers_are_identity():
    t = torch.arange(6.0)
    assert par.allreduce_mean_(t.clone()) is None
    assert par.agree_steps(7) == 7
    assert torch.equal(par.gather_rows(t), t)
    assert par.world_info() == (0, 1)


# ------------------------------------------------------------------ gradient all-reduce == union batch
def _problem():
    from oracle import cinema_oracle as O
    g = torch.Generator().manual_seed(11)
    lat_size = LAT[0]
    params = O.init_siren_params(3, lat_size, OUT, HID, LAYERS, 30, 30, MOD, g)
    latents = torch.randn(S_TOTAL, *LAT, generator=g) * 0.3
    tfs = (torch.rand(S_TOTAL, 6, generator=g) - 0.5) * 0.1
    batches = []
    for r in range(WORLD):
        own = torch.tensor(par.shard_subjects(S_TOTAL, r, WORLD))
        subj = own[torch.randint(0, len(own), (N_PER_RANK,), generator=g)]
        coords = torch.rand(N_PER_RANK, 3, generator=g) * 1.6 - 0.8
        target = torch.cat((torch.rand(N_PER_RANK, 1, generator=g),
                            torch.randint(0, 4, (N_PER_RANK, 1), generator=g).float()), 1)
        batches.append((coords, target, subj))
    cfg = dict(num_hidden_layers=LAYERS, omega=[30, 30], sr_dims=1, loss_metric="mse", class_weights=None, tf_weight=0.0)
    return O, params, latents, tfs, batches, cfg


def _flat(params, names, grad=False):
    return torch.cat([(params[k].grad if grad else params[k]).reshape(-1) for k in names])


def _grad_worker(rank):
    O, params, latents, tfs, batches, cfg = _problem()
    names = O.siren_param_names(LAYERS, MOD)
    own = par.shard_subjects(S_TOTAL, rank, WORLD)
    # this rank: its own subjects' tables only, local subject ids
    p_loc = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lat_loc = latents[own].clone().requires_grad_(True)
    tf_loc = tfs[own].clone().requires_grad_(True)
    coords, target, subj = batches[rank]
    O.training_step(p_loc, lat_loc, tf_loc, coords, target, subj - own[0], None, cfg)
    flat_grad = _flat(p_loc, names, grad=True).clone()
    par.allreduce_mean_(flat_grad)
    # union batch on one device (what the reference would do with a W-times larger sample_size)
    p_u = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lat_u, tf_u = latents.clone().requires_grad_(True), tfs.clone().requires_grad_(True)
    cu = torch.cat([b[0] for b in batches]); tu = torch.cat([b[1] for b in batches]); su = torch.cat([b[2] for b in batches])
    O.training_step(p_u, lat_u, tf_u, cu, tu, su, None, cfg)
    assert rel_err(flat_grad.numpy(), _flat(p_u, names, grad=True).numpy()) < 1e-5
    # sharded tables: the local gradient is WORLD x the union-batch gradient of the owned rows (the local mean divides
    # by n, the union mean by WORLD*n) -- a per-table constant that AdamW's normalisation removes
    assert rel_err(lat_loc.grad.numpy(), WORLD * lat_u.grad[own].numpy()) < 1e-5
    assert rel_err(tf_loc.grad.numpy(), WORLD * tf_u.grad[own].numpy()) < 1e-5
    # replicas stay bit-identical after the exchange
    both = par.gather_rows(flat_grad[None])
    assert torch.equal(both[0], both[1])


def test_flat_gradient_allreduce_equals_union_batch():
    _run(_grad_worker)


def _ragged_weighted_worker(rank):
    """Non-uniform class weights and ragged n (rank 1 runs a shorter chunk): each rank normalises its means by the union
    batch's counts / world (par.share_batch_normaliser_), then the rank-average of the gradients is the union batch's."""
    O, params, latents, tfs, batches, cfg = _problem()
    names = O.siren_param_names(LAYERS, MOD)
    cw = torch.tensor([0.0, 1.0, 2.5, 0.5])
    keep = [N_PER_RANK, N_PER_RANK - 17]
    batches = [tuple(t[:keep[r]] for t in b) for r, b in enumerate(batches)]
    for b in batches:                      # class 0 has weight 0, as the background class of the configs
        b[1][:, 1].clamp_(min=0)
    own = par.shard_subjects(S_TOTAL, rank, WORLD)
    p_loc = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lat_loc, tf_loc = latents[own].clone().requires_grad_(True), tfs[own].clone().requires_grad_(True)
    coords, target, subj = batches[rank]
    y = target[:, 1].long()
    norm = par.share_batch_normaliser_(torch.tensor([float(len(y)), float(cw[y].sum())])).wait()
    n_union, w_union = sum(keep), sum(float(cw[b[1][:, 1].long()].sum()) for b in batches)
    assert torch.allclose(norm, torch.tensor([n_union / WORLD, w_union / WORLD]))
    local = subj - own[0]
    out = O.decoder_forward(p_loc, coords, lat_loc, None, tf_loc[local.long()], local, cfg["num_hidden_layers"], *cfg["omega"])
    nll = torch.nn.functional.cross_entropy(out[:, 1:], y, reduction="none")
    loss = ((out[:, :1] - target[:, :1]) ** 2).sum() / (norm[0] * 1) + (cw[y] * nll).sum() / norm[1]
    loss.backward()
    flat_grad = _flat(p_loc, names, grad=True).clone()
    par.allreduce_mean_(flat_grad)
    p_u = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lat_u, tf_u = latents.clone().requires_grad_(True), tfs.clone().requires_grad_(True)
    cu = torch.cat([b[0] for b in batches]); tu = torch.cat([b[1] for b in batches]); su = torch.cat([b[2] for b in batches])
    O.training_step(p_u, lat_u, tf_u, cu, tu, su, None, dict(cfg, class_weights=cw))
    assert rel_err(flat_grad.numpy(), _flat(p_u, names, grad=True).numpy()) < 1e-5
    assert rel_err(lat_loc.grad.numpy(), WORLD * lat_u.grad[own].numpy()) < 1e-5


def test_ragged_weighted_batches_allreduce_equals_union_batch():
    _run(_ragged_weighted_worker)


# ------------------------------------------------------------------ helpers with ragged shards
def _helpers_worker(rank):
    assert par.world_info() == (rank, WORLD)
    assert par.agree_steps(10 + 3 * rank) == 10
    local = torch.arange(3 + 2 * rank, dtype=torch.float32)[:, None] + 100 * rank
    full = par.gather_rows(local)
    assert torch.equal(full, torch.cat([torch.arange(3.0)[:, None], torch.arange(5.0)[:, None] + 100]))
    t = torch.full((4,), float(rank))
    par.broadcast_(t, 0)
    assert torch.equal(t, torch.zeros(4))


def test_agree_steps_gather_broadcast():
    _run(_helpers_worker)


# ------------------------------------------------------------------ distributed mean latent == golden (reference)
def _mean_latent_worker(rank):
    from cinema_amd import atlas
    from cinema_amd.utils import normalize_condition
    g = load_golden("g9_atlas_helpers.npz")
    args = {"atlas_gen": {"cond_scale": 0.05, "gaussian_span": 1.0},
            "dataset": {"constraints": {"scan_age": {"min": 37, "max": 46}}}}
    lat, cv = torch.from_numpy(g["ml_lat"]), torch.from_numpy(g["ml_cv"])
    own = list(range(0, 7)) if rank == 0 else list(range(7, 16))     # ragged split on purpose: 7 / 9
    for i in range(2):
        mu = normalize_condition(args, "scan_age", float(g[f"ml{i}_age"]))
        m = atlas.mean_latent(args, lat[own], cv[own], "scan_age", mu, n_max=int(g[f"ml{i}_nmax"]),
                              process_group=None)
        assert m.shape == (1, *lat.shape[1:])
        assert rel_err(m.numpy(), g[f"ml{i}_out"]) < 1e-6


def test_distributed_mean_latent_matches_reference_golden():
    _run(_mean_latent_worker)


def test_mean_latent_single_process_matches_golden():
    from cinema_amd import atlas
    from cinema_amd.utils import normalize_condition
    g = load_golden("g9_atlas_helpers.npz")
    args = {"atlas_gen": {"cond_scale": 0.05, "gaussian_span": 1.0},
            "dataset": {"constraints": {"scan_age": {"min": 37, "max": 46}}}}
    lat, cv = torch.from_numpy(g["ml_lat"]), torch.from_numpy(g["ml_cv"])
    for i in range(2):
        mu = normalize_condition(args, "scan_age", float(g[f"ml{i}_age"]))
        m = atlas.mean_latent(args, lat, cv, "scan_age", mu, n_max=int(g[f"ml{i}_nmax"]))
        assert rel_err(m.numpy(), g[f"ml{i}_out"]) < 1e-6
    with pytest.raises(AssertionError):
        atlas.mean_latent(args, lat, cv[:3], "scan_age", 0.0)


def _collect_worker(rank):
    from cinema_amd import parallel as par
    n_jobs = 5
    vol = lambda j: torch.full((2, 3, 4, 2), float(j)) + torch.arange(2.0)
    mine = {j: vol(j) for j in par.shard_jobs(n_jobs, rank, WORLD)}
    got = par.collect_jobs(mine, n_jobs)
    if rank == 0:
        assert len(got) == n_jobs
        for j in range(n_jobs):
            assert torch.equal(got[j], vol(j)), j
    else:
        assert got is None


def test_collect_jobs_brings_round_robin_volumes_to_rank_0_in_order():
    _run(_collect_worker)
