"""GPU-side foreground extraction (cinema_amd/dataset.py) against the golden g8 vectors (made by running the reference's
own normalize_intensities, tests/golden/make_golden.py) and an end-to-end pass NIfTI files -> compact resident tables ->
trainer step against the CPU oracle, with subjects of different shapes."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from test_gpu_trainer import base_args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _assert_within_one_ulp(a, ref):
    """fp32 results of the same float64 formula whose only freedom is the summation order of the centre of mass: the
    float64 values agree to ~1e-15, so the fp32 casts are equal or neighbours (absolute floor for values next to 0)."""
    tol = np.maximum(np.spacing(np.abs(ref)), np.float32(2.0 ** -40))
    bad = np.abs(a.astype(np.float64) - ref.astype(np.float64)) > tol
    assert not bad.any(), f"{bad.sum()} coordinates differ by more than one fp32 ulp"


@pytest.mark.parametrize("norm", ["minmax", "zscore"])
def test_g8_foreground_on_device(norm):
    from cinema_amd.dataset import foreground_on_device
    g = load_golden("g8_sampling.npz")
    coords, vals, vox = foreground_on_device([g["t2"], g["t1"], g["seg"]], g["affine"], g["bbox"], norm, DEV)
    X, Y, Z = g["seg"].shape
    ijk = g["ijk"]
    assert np.array_equal(vox.cpu().numpy(), (ijk[:, 0] * Y + ijk[:, 1]) * Z + ijk[:, 2])          # voxel order: bit-exact
    v, ref = vals.cpu().numpy(), g[f"values_{norm}"]
    assert np.array_equal(v[:, -1], ref[:, -1])                                                     # labels: bit-exact
    if norm == "minmax":      # (v - lo) / (hi - lo) in float64: same operations, bit-exact
        assert np.array_equal(v.view(np.uint32), ref.view(np.uint32))
    else:                     # mean / std reductions sum in a different order: float64 noise, then one cast to fp32
        assert np.abs(v - ref).max() <= 2.0 ** -22 * np.abs(ref).max()
    _assert_within_one_ulp(coords.cpu().numpy(), g["coords"])


def test_foreground_rejects_out_of_bbox_and_empty():
    from cinema_amd.dataset import foreground_on_device
    seg = np.zeros((6, 6, 6))
    with pytest.raises(ValueError):
        foreground_on_device([seg, seg], np.eye(4), [10, 10, 10], "minmax", DEV)
    seg[0, 0, 0] = seg[5, 5, 5] = 1
    with pytest.raises(AssertionError):
        foreground_on_device([seg + 1, seg], np.eye(4), [4, 4, 4], "minmax", DEV)                   # spans +-2.5 of a 4 mm box


def _write_subject(tmp, name, shape, seed):
    from cinema_amd import nifti
    rng = np.random.default_rng(seed)
    ii, jj, kk = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    c = (np.array(shape) - 1) / 2.0
    r = ((ii - c[0]) / (0.42 * shape[0])) ** 2 + ((jj - c[1]) / (0.38 * shape[1])) ** 2 + ((kk - c[2]) / (0.40 * shape[2])) ** 2
    seg = np.where(r < 1.0, 1 + (ii + 2 * jj + 3 * kk) % 9, 0).astype(np.uint8)
    t2 = ((rng.random(shape) * 800.0 + 100 * np.cos(ii / 3.0)) * (r < 1.2)).astype(np.float32)
    t1 = ((rng.random(shape) * 300.0 + 50 * np.sin(jj / 2.0)) * (r < 1.2)).astype(np.float32)
    sp = 120.0 / max(shape)
    aff = np.diag([sp, sp, sp, 1.0])
    aff[:3, 3] = rng.random(3) * 10 - 5
    row = {"subject_id": name}
    for key, vol in (("T2w", t2), ("T1w", t1), ("SegEM9", seg)):
        row[key] = os.path.join(tmp, f"{name}_{key}.nii.gz")
        nifti.save(nifti.Nifti1Image(vol, aff), row[key])
    return row, (t2, t1, seg), nifti.load(row["SegEM9"]).affine         # the affine as stored (fp32 srow_* fields)


def test_nifti_to_trainer_step_matches_oracle(tmp_path):
    """Subjects of different shapes written as .nii.gz -> subject_store_from_rows -> one trainer step: the targets K3
    gathers from the compact tables equal the oracle's foreground_coords_and_values, and the step's gradients equal the
    oracle's on those samples."""
    from oracle import cinema_oracle as O
    from cinema_amd.dataset import subject_store_from_rows
    from cinema_amd.trainer import HotPathTrainer
    args = base_args([48, 3, 3, 3], 128, 3, [1, 3], [2, 10], 0, "fp32")
    args["dataset"].update(modalities=["T2w", "T1w", "SegEM9"], normalize_values="minmax", conditions={}, constraints={})
    args["atlas_gen"] = {"cond_scale": 0.05}
    shapes = [(20, 22, 18), (16, 16, 24), (25, 19, 21)]
    rows, raw = [], []
    for s, shp in enumerate(shapes):
        row, vols, aff = _write_subject(str(tmp_path), f"sub{s}", shp, 40 + s)
        rows.append(row)
        raw.append((vols, aff))
    store = subject_store_from_rows(args, rows, DEV)
    assert store.n_subjects == 3 and store.V == 0
    # the resident tables are the oracle's lists, subject after subject
    oc, ov = [], []
    for vols, aff in raw:
        c, v, _ = O.foreground_coords_and_values([x.astype(np.float64) for x in vols], aff, [155, 155, 155], "minmax")
        oc.append(c)
        ov.append(v)
    oc, ov = np.concatenate(oc), np.concatenate(ov)
    assert len(store) == oc.shape[0]
    assert np.array_equal(store.intens.cpu().numpy().view(np.uint32), ov[:, :2].view(np.uint32))
    assert np.array_equal(store.labels.cpu().numpy(), ov[:, 2].astype(np.uint8))
    _assert_within_one_ulp(store.coords.cpu().numpy(), oc)
    torch.manual_seed(5)
    tr = HotPathTrainer(args, store, DEV, keep_grads=True)
    store.shuffle(torch.Generator(device=DEV).manual_seed(2), group_chunk=900)
    n = 900
    lat0, tf0, flat0 = tr.latents.detach().cpu().clone(), tr.tfs.detach().cpu().clone(), tr.eng.flat.detach().cpu().clone()
    coords, voxel, subj = store.coords[:n].cpu(), store.voxel[:n].cpu(), store.subj[:n].cpu()
    target = torch.cat((store.intens.cpu()[voxel], store.labels.cpu()[voxel].float().unsqueeze(1)), 1)
    params = {k: v.clone().requires_grad_(True) for k, v in tr.eng.param_views(flat0).items()}
    lat_o, tf_o = lat0.clone().requires_grad_(True), tf0.clone().requires_grad_(True)
    cfg = dict(num_hidden_layers=3, omega=[30, 30], sr_dims=2, loss_metric="mse",
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=0.0)
    loss_o, _ = O.training_step(params, lat_o, tf_o, coords, target, subj, None, cfg)
    tr.step(0, n)
    got = tr.loss_dict()
    assert abs(got["total"] - float(loss_o["total"])) < 1e-4 * abs(float(loss_o["total"]))
    errs = {"glat": rel_err(tr.lat_grad.cpu().numpy(), lat_o.grad.numpy())}
    gviews = tr.eng.param_views(tr.eng.flat_grad.cpu())
    for k, p in params.items():
        errs["g_" + k] = rel_err(gviews[k].numpy(), p.grad.numpy())
    bad = {k: v for k, v in errs.items() if not v < 1e-4}
    assert not bad, bad
