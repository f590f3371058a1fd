"""GPU tests of the layers above the decoder: atlas generation (mean latent -> dense inference -> post-processing,
build_atlas.py:159-219) and the data-parallel trainer step with two ranks sharing the one GPU of the test box."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from conftest import rel_err  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def atlas_args(precision):
    return {
        "inr_decoder": {"in_dim": 3, "out_dim": [2, 5], "latent_dim": [16, 3, 3, 3], "cond_dims": 1, "hidden_size": 64,
                        "num_hidden_layers": 2, "modulated_layers": [1, 2], "omega": [30, 30], "cnn_kernel_size": 0,
                        "tf_dim": 6, "precision": precision},
        "dataset": {"world_bbox": [60, 60, 60], "class_weights": None, "label_names": [],
                    "constraints": {"scan_age": {"min": 37, "max": 46}, "birth_age": {"min": 25, "max": 42}}},
        "atlas_gen": {"spacing": [5.0, 5.0, 5.0], "temporal_values": [38.0, 43.5], "cond_scale": 0.05, "gaussian_span": 1.0,
                      "conditions": {"birth_age": {"values": [28, 40], "normed_values": False}}},
        "mask_reconstruction": False, "device": DEV,
    }


def test_generate_atlas_matches_oracle():
    """[x, y, z, modalities, conditions, t] atlas against mean_latent + decoder + post-processing of the CPU oracle."""
    from oracle import cinema_oracle as O
    from cinema_amd import atlas
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import generate_combinations, normalize_condition
    args = atlas_args("fp32")
    torch.manual_seed(3)
    dec = INR_Decoder(args, DEV).to(DEV)
    g = torch.Generator().manual_seed(4)
    S = 12
    lat = torch.randn(S, 16, 3, 3, 3, generator=g) * 0.3
    ages = torch.rand(S, generator=g) * 9 + 37
    cv = normalize_condition(args, "scan_age", ages)
    vol, affine, tsteps, conds = atlas.generate_atlas(dec, args, lat.to(DEV), cv.to(DEV), n_max=5)
    gs = 12                                                       # 60 mm / 5 mm
    assert vol.shape == (gs, gs, gs, 2 + 1 + 5, 2, 2) and len(conds) == 2 and tsteps == [38.0, 43.5]
    params = {k[len("sr_net."):]: v.detach().cpu() for k, v in dec.state_dict().items()}
    coords, shape, aff_o, _ = O.world_grid([60, 60, 60], [5.0] * 3)
    assert np.array_equal(torch.as_tensor(affine).cpu().numpy(), aff_o.numpy())
    n = coords.shape[0]
    for ti, age in enumerate(tsteps):
        mu = O.normalize_condition_value(age, 37, 46, 0.05)
        ml = O.mean_latent(lat, cv, mu, 37, 46, 1.0, 0.05, n_max=5)
        for ci, c in enumerate(generate_combinations(args, args["atlas_gen"]["conditions"])):
            cond = torch.tensor(c, dtype=torch.float32).expand(n, -1)
            raw = O.decoder_forward(params, coords, ml, cond, None, torch.zeros(n, dtype=torch.long), 2, 30, 30)
            ref = O.inference_postprocess(raw, 2, shape)
            ref[..., -1][ref[..., -1] == 4] = 0                    # build_atlas.py:178-180
            got = vol[..., ci, ti]
            assert (got[..., 2] != ref[..., 2]).float().mean() < 2e-3          # argmax ties at fp32 noise level
            assert np.abs(np.nan_to_num(got[..., :2].numpy()) - np.nan_to_num(ref[..., :2].numpy())).max() < 2e-4
            assert np.abs(got[..., 3:].numpy() - ref[..., 3:].numpy()).max() < 2e-4


def test_generate_atlas_writes_the_reference_file_set(tmp_path):
    """epoch given -> utils.save_atlas layout (utils.py:253-271): one 4-d .nii.gz (x, y, z, t) per modality / certainty
    map and condition, holding exactly the returned atlas, affine diag(spacing, 1)."""
    from cinema_amd import atlas, nifti
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import normalize_condition
    args = atlas_args("bf16")
    args["dataset"].update(modalities=["T2w", "T1w", "Seg"], label_names=["BG", "a", "b", "c", "d"])
    args.update(save_certainty_maps=True, output_dir=str(tmp_path))
    torch.manual_seed(3)
    dec = INR_Decoder(args, DEV).to(DEV)
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(6, 16, 3, 3, 3, generator=g) * 0.3
    cv = normalize_condition(args, "scan_age", torch.rand(6, generator=g) * 9 + 37)
    vol, affine, tsteps, conds = atlas.generate_atlas(dec, args, lat.to(DEV), cv.to(DEV), n_max=5, epoch=3)
    names = ["T2w", "T1w", "Seg"] + [f"CertaintyMaps/{n}" for n in args["dataset"]["label_names"]]
    for c in range(len(conds)):
        for i, nme in enumerate(names):
            img = nifti.load(os.path.join(str(tmp_path), f"{nme}_ga=38.0-43.5_cond={c}_ep=3.nii.gz"))
            assert img.shape == (12, 12, 12, 2) and img.get_data_dtype() == np.float32
            assert np.array_equal(img.get_fdata(), vol[..., i, c, :].numpy().astype(np.float64), equal_nan=True)
            assert np.array_equal(np.diag(img.affine), [5.0, 5.0, 5.0, 1.0])


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
        "dataset": {"class_weights": None, "world_bbox": [155, 155, 155], "label_names": []},
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
        n = 512
        tr.step(0, n)
        torch.cuda.synchronize()
        out = {"flat0": flat0, "flat1": tr.eng.flat.detach().cpu(), "grad": tr.eng.flat_grad.detach().cpu(),
               "loss": tr.loss_dict()["total"]}
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_step_on_one_gpu(tmp_path):
    """Two processes, subjects sharded, MLP replicated: after one step both replicas hold identical weights, started
    from rank 0's initialisation, and the all-reduced gradient is the mean of the two local gradients."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_dp_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert torch.equal(r0["flat0"], r1["flat0"])                   # broadcast at construction
    assert torch.equal(r0["grad"], r1["grad"])                     # the averaged gradient is the same tensor on both
    assert torch.equal(r0["flat1"], r1["flat1"])                   # replicas stay bit-identical after AdamW
    assert not torch.equal(r0["flat0"], r0["flat1"])
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"]) and r0["loss"] != r1["loss"]   # different shards
    # the numerical identity of the averaged gradient with the union-batch gradient is covered on the CPU by
    # tests/test_dp_gloo.py (same helper, gloo backend)
