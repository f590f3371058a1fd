"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python
modules (imported from /root/reference, never copied) on seeded inputs.

Run in the build container only:  python tests/golden/make_golden.py
The reference is absent on the GPU box; tests read the committed .npz files.

Third-party modules the reference's `utils.py` imports at top level but never touches on
the hot path (ants, nibabel, wandb, skimage, torchio) are absent here and are registered
as empty placeholders so that `import utils` succeeds.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("CINEMA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    for name in ["ants", "nibabel", "wandb", "torchio", "skimage", "skimage.metrics"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchio"].Transform = object
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = None
    sys.modules["skimage.metrics"].structural_similarity = None
    sys.path.insert(0, REF)
    import utils as ref_utils  # noqa
    from models import inr_decoder as ref_dec  # noqa
    from models import siren as ref_siren  # noqa
    return ref_utils, ref_dec, ref_siren


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def make_args(latent_dim, hidden, layers, mod, out_dim, cond_dims, tf_dim=6, loss_metric="mse",
              class_weights=None, tf_weight=0.0):
    return {
        "inr_decoder": {"in_dim": 3, "out_dim": out_dim, "latent_dim": latent_dim, "cond_dims": cond_dims,
                        "hidden_size": hidden, "num_hidden_layers": layers, "modulated_layers": mod,
                        "omega": [30, 30], "cnn_kernel_size": 0, "tf_dim": tf_dim},
        "optimizer": {"tf_weight": tf_weight, "loss_metric": loss_metric},
        "dataset": {"class_weights": class_weights, "label_names": ["ClearLabel", "CSF", "cGM", "WM", "BG"]},
        "mask_reconstruction": False, "device": "cpu",
    }


def g1_affine(ru):
    g = torch.Generator().manual_seed(1)
    out = {}
    for d in (6, 9, 12):
        e = (torch.rand(17, d, generator=g) - 0.5)
        R, t = ru.embed2affine(e)
        out[f"embed{d}"], out[f"R{d}"], out[f"t{d}"] = npy(e), npy(R), npy(t)
    R, _ = ru.embed2affine(torch.tensor([[0.1, 0.2, 0.3, 0, 0, 0]]))
    out["R_known"] = npy(R)
    np.savez(os.path.join(OUT, "g1_affine.npz"), **out)


def g2_transform(rd):
    g = torch.Generator().manual_seed(2)
    out = {}
    for d in (6, 9, 12):
        tfs = ((torch.rand(33, d, generator=g) - 0.5) * 0.4).requires_grad_(True)
        c = torch.rand(33, 3, generator=g) * 2 - 1
        y = rd.INR_Decoder.transform(c, tfs)
        gy = torch.randn(33, 3, generator=g)
        (y * gy).sum().backward()
        out[f"tfs{d}"], out[f"coords{d}"], out[f"out{d}"] = npy(tfs), npy(c), npy(y)
        out[f"gout{d}"], out[f"gtfs{d}"] = npy(gy), npy(tfs.grad)
    np.savez(os.path.join(OUT, "g2_transform.npz"), **out)


def g3_interp(rd):
    g = torch.Generator().manual_seed(3)
    out = {}
    cases = {"g333": (8, 3, 3, 3), "g111": (8, 1, 1, 1), "g234": (8, 2, 3, 4)}
    for name, (C, lx, ly, lz) in cases.items():
        S, N = 5, 64
        lat = torch.randn(S, C, lx, ly, lz, generator=g).requires_grad_(True)
        c = torch.rand(N, 3, generator=g) * 2.6 - 1.3  # inside, on and outside [-1,1]
        c[0] = torch.tensor([-1.0, 1.0, 0.0])
        c[1] = torch.tensor([1.0, -1.0, 1.0])
        c[2] = torch.tensor([0.0, 0.0, 0.0])
        c.requires_grad_(True)
        idx = torch.randint(0, S, (N,), generator=g, dtype=torch.int32)
        cond = torch.randn(N, 2, generator=g)
        o = rd.INR_Decoder.spatial_interpolation(c, lat[idx], cond)
        go = torch.randn(o.shape, generator=g)
        (o * go).sum().backward()
        for k, v in dict(lat=lat, coords=c, idx=idx, cond=cond, out=o, gout=go, glat=lat.grad, gcoords=c.grad).items():
            out[f"{name}_{k}"] = npy(v)
    # N == 1: .squeeze() drops the batch dimension (inr_decoder.py:82)
    lat = torch.randn(1, 8, 3, 3, 3, generator=g)
    o1 = rd.INR_Decoder.spatial_interpolation(torch.zeros(1, 3), lat, None)
    out["n1_shape"] = np.array(o1.shape)
    np.savez(os.path.join(OUT, "g3_interp.npz"), **out)


def g4_sinelayer(rs):
    g = torch.Generator().manual_seed(4)
    torch.manual_seed(4)
    out = {}
    for tag, lat_feat in (("mod", 24), ("plain", 0)):
        layer = rs.SineLayer(40, lat_feat, 48, is_first=False, omega=30)
        x = (torch.rand(50, 40, generator=g) * 2 - 1).requires_grad_(True)
        lat = torch.randn(50, max(lat_feat, 1), generator=g).requires_grad_(True)
        y, _ = layer((x, lat))
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out[f"{tag}_x"], out[f"{tag}_lat"], out[f"{tag}_y"], out[f"{tag}_gy"] = npy(x), npy(lat), npy(y), npy(gy)
        out[f"{tag}_gx"] = npy(x.grad)
        for n, p in layer.named_parameters():
            out[f"{tag}_p_{n}"] = npy(p)
            out[f"{tag}_g_{n}"] = npy(p.grad)
        if lat_feat:
            out[f"{tag}_glat"] = npy(lat.grad)
    np.savez(os.path.join(OUT, "g4_sinelayer.npz"), **out)


def g5_decoder(ru, rd):
    """Full INR_Decoder fwd + all grads: cfg-T, cfg-D dims at small N/S, a cut-down cfg-V (odd lat), cfg-S (modulated layer 0)."""
    cfgs = {
        "cfgT": dict(latent_dim=[32, 1, 1, 1], hidden=128, layers=3, mod=[1, 3], out_dim=[1, 10], cond=0, S=4, N=96),
        "cfgD": dict(latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], out_dim=[2, 10], cond=0, S=8, N=160),
        "cfgV": dict(latent_dim=[65, 3, 3, 3], hidden=128, layers=5, mod=[1, 3, 5], out_dim=[1, 10], cond=1, S=8, N=96),  # lat = 66: K not a multiple of the MFMA K
        "cfgS": dict(latent_dim=[16, 2, 3, 4], hidden=64, layers=2, mod=[0, 2], out_dim=[2, 5], cond=2, S=3, N=70),
    }
    for name, c in cfgs.items():
        torch.manual_seed(5)
        g = torch.Generator().manual_seed(50)
        args = make_args(c["latent_dim"], c["hidden"], c["layers"], c["mod"], c["out_dim"], c["cond"],
                         loss_metric="mse", class_weights=[0.0] + [1.0] * (c["out_dim"][-1] - 1), tf_weight=0.1)
        dec = rd.INR_Decoder(args, "cpu")
        crit = ru.Criterion(args)
        S, N = c["S"], c["N"]
        lat = (torch.randn(S, *c["latent_dim"], generator=g) * 0.01).requires_grad_(True)
        tft = ((torch.rand(S, 6, generator=g) - 0.5) * 0.1).requires_grad_(True)
        coords = torch.rand(N, 3, generator=g) * 1.8 - 0.9
        idx = torch.randint(0, S, (N,), generator=g, dtype=torch.int32)
        cond = (torch.randn(N, c["cond"], generator=g) * 0.05).requires_grad_(True)
        M = len(c["out_dim"]) - 1 + sum(c["out_dim"][:-1]) - (len(c["out_dim"]) - 1)
        sr = sum(c["out_dim"][:-1])
        target = torch.cat((torch.rand(N, sr, generator=g),
                            torch.randint(1, c["out_dim"][-1], (N, 1), generator=g).float()), dim=1)
        tfs = tft[idx.long()]
        out = dec(coords, lat, cond, tfs, idcs_df=idx.long())
        loss = crit(out, target, tfs, seg_weight=1.0)
        loss["total"].backward()
        d = dict(coords=coords, idx=idx, cond=cond, target=target, lat=lat, tft=tft, out=out,
                 glat=lat.grad, gtft=tft.grad, gcond=cond.grad,
                 loss_total=loss["total"], loss_sr=loss["sr"], loss_seg=loss["seg"], loss_tf=loss["tf"])
        d = {k: npy(v) for k, v in d.items()}
        for n, p in dec.named_parameters():
            d["p_" + n] = npy(p)
            d["g_" + n] = npy(p.grad)
        d["state_keys"] = np.array(list(dec.state_dict().keys()))
        np.savez_compressed(os.path.join(OUT, f"g5_decoder_{name}.npz"), **d)


def g6_criterion(ru):
    g = torch.Generator().manual_seed(6)
    out = {}
    N = 77
    case = 0
    for metric in ("mse", "l1"):
        for segw in (0.0, 1.0, 0.5):
            for cw in (None, [0.0, 2.0, 3.0, 2.0, 1.5, 1.0, 1.0, 1.0, 1.0, 1.0]):
                for tfd in (6, 9):
                    args = make_args([8, 1, 1, 1], 16, 1, [1], [2, 10], 0, loss_metric=metric, class_weights=cw,
                                     tf_weight=0.3)
                    crit = ru.Criterion(args)
                    o = torch.randn(N, 12, generator=g).requires_grad_(True)
                    tgt = torch.cat((torch.rand(N, 2, generator=g),
                                     torch.randint(1, 10, (N, 1), generator=g).float()), dim=1)
                    tfs = ((torch.rand(N, tfd, generator=g) - 0.5) * 0.2).requires_grad_(True)
                    loss = crit(o, tgt, tfs, seg_weight=segw)
                    loss["total"].backward()
                    pre = f"c{case}_"
                    out[pre + "meta"] = np.array([metric, str(segw), "none" if cw is None else "w", str(tfd)])
                    out[pre + "cw"] = np.array(cw if cw is not None else [], dtype=np.float32)
                    out[pre + "out"], out[pre + "target"], out[pre + "tfs"] = npy(o), npy(tgt), npy(tfs)
                    out[pre + "gout"], out[pre + "gtfs"] = npy(o.grad), npy(tfs.grad)
                    for k in ("seg", "sr", "total", "tf_rot", "tf_trans", "tf_scale", "tf"):
                        out[pre + "loss_" + k] = npy(torch.as_tensor(loss[k]))
                    case += 1
    out["n_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT, "g6_criterion.npz"), **out)


def g7_adamw(ru, rd):
    """10-step AdamW + cosine trajectory with the reference's three param groups
    (build_atlas.py:432-459; lr from config_atlas.yaml:57-68)."""
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(70)
    args = make_args([8, 2, 2, 2], 32, 2, [1], [1, 4], 0, loss_metric="l1", class_weights=[0.0, 1.0, 1.0, 1.0])
    dec = rd.INR_Decoder(args, "cpu")
    crit = ru.Criterion(args)
    S, N = 3, 40
    lat = torch.nn.Parameter(torch.randn(S, 8, 2, 2, 2, generator=g) * 0.01)
    tft = torch.nn.Parameter(torch.zeros(S, 6))
    coords = torch.rand(N, 3, generator=g) * 1.6 - 0.8
    idx = torch.randint(0, S, (N,), generator=g)
    target = torch.cat((torch.rand(N, 1, generator=g), torch.randint(1, 4, (N, 1), generator=g).float()), dim=1)
    init = {"p_" + n: npy(p).copy() for n, p in dec.named_parameters()}
    init.update(lat0=npy(lat).copy(), tft0=npy(tft).copy(), coords=npy(coords), idx=npy(idx), target=npy(target))
    opt = torch.optim.AdamW([
        {"params": lat, "lr": 5e-4, "weight_decay": 0.0},
        {"params": tft, "lr": 7.5e-3, "weight_decay": 0.0},
        {"params": dec.parameters(), "lr": 1e-4, "weight_decay": 0.0}])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=5, eta_min=2.5e-5)
    losses = []
    for step in range(10):
        opt.zero_grad()
        tfs = tft[idx]
        out = dec(coords, lat, torch.zeros(N, 0), tfs, idcs_df=idx)
        loss = crit(out, target, tfs, seg_weight=1.0)
        loss["total"].backward()
        opt.step()
        losses.append(loss["total"].item())
        if step % 2 == 1:
            sched.step()  # "epoch" of two steps
    init.update(losses=np.array(losses), lat10=npy(lat), tft10=npy(tft))
    for n, p in dec.named_parameters():
        init["q_" + n] = npy(p)
    np.savez_compressed(os.path.join(OUT, "g7_adamw.npz"), **init)


def g8_sampling(ru):
    """Foreground extraction on a 16^3 phantom: the reference's normalize_intensities is run
    as is; apply_affine (nibabel, absent) is restated by its documented formula
    pts @ A[:3,:3].T + A[:3,3] (dataset.py:87-97)."""
    rng = np.random.default_rng(8)
    shp = (16, 14, 12)
    ii, jj, kk = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
    r = ((ii - 7.5) / 6.0) ** 2 + ((jj - 6.5) / 5.0) ** 2 + ((kk - 5.5) / 4.5) ** 2
    seg = np.where(r < 1.0, 1 + (ii + 2 * jj + kk) % 4, 0).astype(np.float64)
    t2 = (rng.random(shp) * 900.0 - 50.0) * (r < 1.3)
    t1 = (rng.random(shp) * 300.0) * (r < 1.3)
    affine = np.array([[0.8, 0.05, 0.0, -6.3], [-0.04, 0.8, 0.02, -5.1], [0.0, -0.03, 0.8, -4.4], [0, 0, 0, 1.0]])
    bbox = np.array([20.0, 20.0, 20.0])
    c_nz = np.argwhere(seg > 0)
    values = np.stack([m[c_nz[:, 0], c_nz[:, 1], c_nz[:, 2]].flatten() for m in (t2, t1, seg)], axis=-1)
    world = c_nz @ affine[:3, :3].T + affine[:3, 3]
    coords = (world - np.mean(world, axis=0)) / (bbox / 2)
    out = dict(t2=t2, t1=t1, seg=seg, affine=affine, bbox=bbox, ijk=c_nz)
    out["values_minmax"] = ru.normalize_intensities(values.copy(), "minmax").astype(np.float32)
    out["values_zscore"] = ru.normalize_intensities(values.copy(), "zscore").astype(np.float32)
    out["coords"] = torch.tensor(coords, dtype=torch.float32).numpy()
    np.savez_compressed(os.path.join(OUT, "g8_sampling.npz"), **out)


def g9_atlas_helpers(ru):
    out = {}
    for tag, sp in (("s05", 0.5), ("s15", 1.5)):
        args = {"dataset": {"world_bbox": [155, 155, 155]}, "atlas_gen": {"spacing": [sp, sp, sp]}}
        coords, shape, affine = ru.generate_world_grid(args)
        out[f"{tag}_shape"] = np.array(shape)
        out[f"{tag}_affine"] = npy(affine)
        n = shape[0]
        out[f"{tag}_axis"] = npy(coords.reshape(n, n, n, 3)[:, 0, 0, 0])
        sel = torch.tensor([0, 1, n, n * n, n * n + n + 1, coords.shape[0] // 2, coords.shape[0] - 1])
        out[f"{tag}_sel"], out[f"{tag}_selcoords"] = npy(sel), npy(coords[sel])
    args = {"atlas_gen": {"cond_scale": 0.05},
            "dataset": {"constraints": {"scan_age": {"min": 37, "max": 46}, "birth_age": {"min": 25, "max": 42},
                                        "lv": {"min": 0.0085, "max": 0.061}}}}
    vals = torch.tensor([37.0, 39.0, 41.5, 46.0])
    out["normcond"] = npy(ru.normalize_condition(args, "scan_age", vals))
    conds = {"birth_age": {"values": [26, 31, 36], "normed_values": False},
             "lv": {"values": [-1.25, 0.0, 1.25], "normed_values": True}}
    out["combos"] = np.array(ru.generate_combinations(args, conds), dtype=np.float64)
    out["combos_none"] = np.array(ru.generate_combinations(args, None), dtype=np.float64)
    # get_mean_latent: the reference method is bound to AtlasBuilder (needs datasets); call it on a stand-in self
    sys.modules.setdefault("data_loading", types.ModuleType("data_loading"))
    g = torch.Generator().manual_seed(9)
    S = 16
    lat = torch.randn(S, 4, 3, 3, 3, generator=g)
    ages = torch.rand(S, generator=g) * 9 + 37
    cv = ru.normalize_condition(args, "scan_age", ages)
    import importlib
    for name in ["pandas", "tqdm", "sklearn"]:
        importlib.import_module(name)
    ds_mod = types.ModuleType("data_loading.dataset")
    ds_mod.Data = object
    sys.modules["data_loading.dataset"] = ds_mod
    import build_atlas as rb

    class _DS:
        def get_condition_values(self, key, normed=True, device="cpu"):
            return cv, None

    holder = types.SimpleNamespace(args={**args, "atlas_gen": {"cond_scale": 0.05, "gaussian_span": 1.0}},
                                   latents={"train": lat}, datasets={"train": _DS()}, device="cpu")
    for i, (age, nmax) in enumerate(((40.0, 4), (44.0, 100))):
        mu = ru.normalize_condition(args, "scan_age", age)
        m = rb.AtlasBuilder.get_mean_latent(holder, "scan_age", mu, n_max=nmax)
        out[f"ml{i}_age"], out[f"ml{i}_nmax"], out[f"ml{i}_out"] = np.array(age), np.array(nmax), npy(m)
    out["ml_lat"], out["ml_ages"], out["ml_cv"] = npy(lat), npy(ages), npy(cv)
    np.savez_compressed(os.path.join(OUT, "g9_atlas_helpers.npz"), **out)


def g10_inference(rd):
    torch.manual_seed(10)
    g = torch.Generator().manual_seed(100)
    args = make_args([16, 3, 3, 3], 64, 2, [1, 2], [2, 5], 1)
    dec = rd.INR_Decoder(args, "cpu")
    n = 12
    ax = torch.arange(0, n, 1.0) / n * 2 - 1
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    coords = torch.stack((gx, gy, gz), -1).reshape(-1, 3)
    lat = torch.randn(1, 16, 3, 3, 3, generator=g) * 0.3
    cond = torch.tensor([0.03])
    tf = (torch.rand(1, 6, generator=g) - 0.5) * 0.1
    with torch.no_grad():
        vol = dec.inference(coords, lat, cond, [n, n, n], None, step_size=500)
        vol_tf = dec.inference(coords, lat, cond, [n, n, n], tf, step_size=500)
        raw = dec(coords, lat, cond.expand(coords.shape[0], -1), idcs_df=torch.zeros(coords.shape[0], dtype=torch.long))
    d = dict(coords=coords, lat=lat, cond=cond, tf=tf, vol=vol, vol_tf=vol_tf, raw=raw)
    d = {k: npy(v) for k, v in d.items()}
    for nme, p in dec.named_parameters():
        d["p_" + nme] = npy(p)
    np.savez_compressed(os.path.join(OUT, "g10_inference.npz"), **d)


def main():
    torch.set_num_threads(4)
    ru, rd, rs = import_reference()
    g1_affine(ru)
    g2_transform(rd)
    g3_interp(rd)
    g4_sinelayer(rs)
    g5_decoder(ru, rd)
    g6_criterion(ru)
    g7_adamw(ru, rd)
    g8_sampling(ru)
    g9_atlas_helpers(ru)
    g10_inference(rd)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
