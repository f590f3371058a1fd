"""Two ranks of the data-parallel trainer on ONE GPU (gloo moves the gradient): replicas stay bit-identical.
Round 1 saw both ranks sit inside gloo's all_reduce of a DEVICE tensor; since then gloo only ever sees host tensors
(cinema_amd/parallel.py stages through pinned memory after a stream synchronise).  The workers run under a deadline
and a stall FAILS the test (the driver runs pytest -x without a per-test timeout, so the deadline is ours)."""
import os
import socket
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _trainer_args():
    return {
        "inr_decoder": {"in_dim": 3, "out_dim": [2, 10], "latent_dim": [32, 3, 3, 3], "cond_dims": 0, "hidden_size": 128,
                        "num_hidden_layers": 3, "modulated_layers": [1, 3], "omega": [30, 30], "cnn_kernel_size": 0,
                        "tf_dim": 6, "precision": "fp32"},
        "optimizer": {"tf_weight": 0.0, "loss_metric": "mse", "seg_weight": 1.0, "lr_inr": 1e-4, "inr_weight_decay": 0.0,
                      "lr_latent": 5e-4, "latent_weight_decay": 0.0, "lr_tf": 7.5e-3, "tf_weight_decay": 0.0,
                      "scheduler": {"type": "cosine", "eta_min": 2.5e-5}},
        # non-uniform class weights: the cross-entropy denominators of the two ranks differ
        "dataset": {"class_weights": [0.0, 1.0, 2.5, 0.5, 1.0, 3.0, 1.0, 0.25, 1.0, 2.0], "world_bbox": [155, 155, 155],
                    "label_names": []},
        "epochs": {"train": 5, "val": 5}, "mask_reconstruction": False, "device": DEV,
    }


def _dp_worker(rank, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)     # both ranks share the one GPU: gloo moves the gradient
    try:
        from cinema_amd.trainer import HotPathTrainer, SubjectStore
        torch.manual_seed(100 + rank)                              # different MLP init per rank: rank 0's must win
        store = SubjectStore.synthetic(2, (20, 20, 20), 155.0 / 20, [155, 155, 155], 2, 10, seed=50 + rank, device=DEV)
        store.shuffle(torch.Generator(device=DEV).manual_seed(7 + rank))
        tr = HotPathTrainer(_trainer_args(), store, DEV, keep_grads=True)
        flat0 = tr.eng.flat.detach().cpu().clone()
        n = 512 - 130 * rank                                       # ragged: rank 1 runs a shorter chunk
        base = store.subj[:n].long() * store.V + store.voxel[:n]
        batch = {"coords": store.coords[:n].cpu(), "subj": store.subj[:n].cpu(), "lat": tr.latents.detach().cpu().clone(),
                 "tfs": tr.tfs.detach().cpu().clone(),
                 "target": torch.cat((store.intens[base], store.labels[base].float().unsqueeze(1)), 1).cpu()}
        tr.step(0, n)
        torch.cuda.synchronize()
        out = {"flat0": flat0, "flat1": tr.eng.flat.detach().cpu(), "grad": tr.eng.flat_grad.detach().cpu(),
               "loss": tr.loss_dict()["total"], "batch": batch,
               "views0": {k: v.clone() for k, v in tr.eng.param_views(flat0).items()},
               "gviews": {k: v.clone() for k, v in tr.eng.param_views(tr.eng.flat_grad.detach().cpu()).items()}}
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_step_on_one_gpu(tmp_path):
    """Two processes, subjects sharded, MLP replicated, non-uniform class weights, ragged n: after one step (weight
    gradients first, all-reduce, then the d-latent GEMM) both replicas hold identical weights, started from rank 0's
    initialisation, and the all-reduced gradient is the union batch's gradient (CPU oracle)."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.start_processes(_dp_worker, args=(port, str(tmp_path)), nprocs=2, join=False, start_method="spawn")
    deadline = time.time() + 90.0
    done = False
    while not done and time.time() < deadline:
        done = ctx.join(timeout=2.0)          # raises if a worker failed
    if not done:
        for pr in ctx.processes:
            if pr.is_alive():
                pr.kill()
        for pr in ctx.processes:
            pr.join(10)
        pytest.fail("two ranks sharing one GPU did not finish one trainer step in 90 s")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert torch.equal(r0["flat0"], r1["flat0"])                   # broadcast at construction
    assert torch.equal(r0["grad"], r1["grad"])                     # the averaged gradient is the same tensor on both
    assert torch.equal(r0["flat1"], r1["flat1"])                   # replicas stay bit-identical after AdamW
    assert not torch.equal(r0["flat0"], r0["flat1"])
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"]) and r0["loss"] != r1["loss"]   # different shards
    # the averaged gradient IS the union batch's: the CPU oracle on both ranks' samples at once (class-weighted
    # cross-entropy over the union, ragged n), rank 0's initial weights
    from conftest import rel_err
    from oracle import cinema_oracle as O
    b0, b1 = r0["batch"], r1["batch"]
    S0 = b0["lat"].shape[0]
    params = {k: v.clone().requires_grad_(True) for k, v in r0["views0"].items()}
    lat = torch.cat((b0["lat"], b1["lat"])).requires_grad_(True)
    tfs = torch.cat((b0["tfs"], b1["tfs"])).requires_grad_(True)
    cfg = dict(num_hidden_layers=3, omega=[30, 30], sr_dims=2, loss_metric="mse", tf_weight=0.0,
               class_weights=torch.tensor(_trainer_args()["dataset"]["class_weights"]))
    O.training_step(params, lat, tfs, torch.cat((b0["coords"], b1["coords"])), torch.cat((b0["target"], b1["target"])),
                    torch.cat((b0["subj"], b1["subj"] + S0)), None, cfg)
    bad = {k: rel_err(r0["gviews"][k].numpy(), p.grad.numpy()) for k, p in params.items()}
    bad = {k: v for k, v in bad.items() if not v < 1e-4}
    assert not bad, bad
