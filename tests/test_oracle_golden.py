"""Pin the CPU oracle (oracle/cinema_oracle.py) to the golden vectors generated from the
reference's own modules (tests/golden/make_golden.py).  CPU only.

Tolerances: bit-exact for integer/index work (voxel order, labels); <= 1e-6 max-norm
relative error for fp32 arithmetic restated with a different operation order (1e-5 for
gradients that sum many terms)."""
import numpy as np
import torch

from conftest import load_golden, rel_err
from oracle import cinema_oracle as O

T = torch.from_numpy


def params_from_golden(g, prefix="p_sr_net."):
    return {k[len(prefix):]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith(prefix)}


def test_g1_affine():
    g = load_golden("g1_affine.npz")
    for d in (6, 9, 12):
        A, t = O.affine_from_embedding(T(g[f"embed{d}"]))
        assert rel_err(A.numpy(), g[f"R{d}"]) < 1e-6
        assert np.array_equal(t.numpy(), g[f"t{d}"])
    A, _ = O.affine_from_embedding(torch.tensor([[0.1, 0.2, 0.3, 0, 0, 0]]))
    assert rel_err(A.numpy(), g["R_known"]) < 1e-6
    # SURVEY 8c known answer
    known = np.array([[0.9216, -0.3875, 0.0198], [0.3836, 0.9021, -0.1977], [0.0587, 0.1898, 0.9801]])
    assert np.abs(A[0].numpy() - known).max() < 5e-5


def test_g2_transform_and_grads():
    g = load_golden("g2_transform.npz")
    for d in (6, 9, 12):
        tfs = T(g[f"tfs{d}"]).clone().requires_grad_(True)
        y = O.register_coords(T(g[f"coords{d}"]), tfs)
        assert rel_err(y.detach().numpy(), g[f"out{d}"]) < 1e-6
        (y * T(g[f"gout{d}"])).sum().backward()
        assert rel_err(tfs.grad.numpy(), g[f"gtfs{d}"]) < 1e-5


def test_g3_trilinear_interp():
    g = load_golden("g3_interp.npz")
    for name in ("g333", "g111", "g234"):
        lat = T(g[f"{name}_lat"]).clone().requires_grad_(True)
        c = T(g[f"{name}_coords"]).clone().requires_grad_(True)
        o = O.trilinear_latent(c, lat, T(g[f"{name}_idx"]), T(g[f"{name}_cond"]))
        assert rel_err(o.detach().numpy(), g[f"{name}_out"]) < 1e-6, name
        (o * T(g[f"{name}_gout"])).sum().backward()
        assert rel_err(lat.grad.numpy(), g[f"{name}_glat"]) < 1e-5, name
        gc = c.grad if c.grad is not None else torch.zeros_like(c)
        assert np.abs(gc.numpy() - g[f"{name}_gcoords"]).max() <= 1e-5 * max(1.0, np.abs(g[f"{name}_gcoords"]).max()), name
    assert tuple(g["n1_shape"]) == (8,)  # the reference's N==1 squeeze quirk (documented, not reproduced)


def test_g4_sine_layer():
    g = load_golden("g4_sinelayer.npz")
    for tag in ("mod", "plain"):
        x = T(g[f"{tag}_x"]).clone().requires_grad_(True)
        lat = T(g[f"{tag}_lat"]).clone().requires_grad_(True)
        W = T(g[f"{tag}_p_linear.weight"]).clone().requires_grad_(True)
        b = T(g[f"{tag}_p_linear.bias"]).clone().requires_grad_(True)
        z = x @ W.t() + b
        if tag == "mod":
            Wl = T(g[f"{tag}_p_linear_lats.weight"]).clone().requires_grad_(True)
            bl = T(g[f"{tag}_p_linear_lats.bias"]).clone().requires_grad_(True)
            st = lat @ Wl.t() + bl
            y = torch.sin(30 * z * st[:, :48] + st[:, 48:])
        else:
            y = torch.sin(30 * z)
        assert rel_err(y.detach().numpy(), g[f"{tag}_y"]) < 1e-5
        (y * T(g[f"{tag}_gy"])).sum().backward()
        assert rel_err(x.grad.numpy(), g[f"{tag}_gx"]) < 1e-4
        assert rel_err(W.grad.numpy(), g[f"{tag}_g_linear.weight"]) < 1e-4


def _decoder_case(name, tol_out=2e-5, tol_grad=2e-4):
    g = load_golden(f"g5_decoder_{name}.npz")
    params = params_from_golden(g)
    nl = max(int(k.split(".")[1]) for k in params if k.startswith("net."))
    lat = T(g["lat"]).clone().requires_grad_(True)
    tft = T(g["tft"]).clone().requires_grad_(True)
    cond = T(g["cond"]).clone().requires_grad_(True)
    idx = T(g["idx"])
    sr = g["target"].shape[1] - 1
    ncls = g["out"].shape[1] - sr
    cfg = dict(num_hidden_layers=nl, omega=[30, 30], sr_dims=sr, loss_metric="mse",
               class_weights=torch.tensor([0.0] + [1.0] * (ncls - 1)), tf_weight=0.1)
    loss, out = O.training_step(params, lat, tft, T(g["coords"]), T(g["target"]), idx,
                                cond if cond.shape[1] > 0 else cond, cfg)
    assert rel_err(out.detach().numpy(), g["out"]) < tol_out
    assert abs(float(loss["total"].detach()) - float(g["loss_total"])) < 1e-5 * abs(float(g["loss_total"]))
    assert abs(float(loss["sr"].detach()) - float(g["loss_sr"])) < 1e-5 * abs(float(g["loss_sr"]))
    assert abs(float(loss["seg"].detach()) - float(g["loss_seg"])) < 1e-5 * abs(float(g["loss_seg"]))
    assert rel_err(lat.grad.numpy(), g["glat"]) < tol_grad
    assert rel_err(tft.grad.numpy(), g["gtft"]) < tol_grad
    if cond.shape[1] > 0:
        assert rel_err(cond.grad.numpy(), g["gcond"]) < tol_grad
    for k, p in params.items():
        assert rel_err(p.grad.numpy(), g["g_sr_net." + k]) < tol_grad, k
    # the state_dict key list the drop-in must reproduce
    assert list(g["state_keys"]) == ["sr_net." + k for k in O.siren_param_names(nl, [int(k.split(".")[1]) for k in params if k.endswith("linear_lats.weight")])]


def test_g5_decoder_cfgT():
    _decoder_case("cfgT")


def test_g5_decoder_cfgS_modulated_first_layer():
    _decoder_case("cfgS")


def test_g5_decoder_cfgV_odd_latent_width():
    _decoder_case("cfgV")


def test_g5_decoder_cfgD():
    _decoder_case("cfgD")


def test_g6_criterion_all_cases():
    g = load_golden("g6_criterion.npz")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        metric, segw, cwtag, tfd = g[pre + "meta"]
        cw = T(g[pre + "cw"]) if cwtag == "w" else None
        o = T(g[pre + "out"]).clone().requires_grad_(True)
        tfs = T(g[pre + "tfs"]).clone().requires_grad_(True)
        loss = O.criterion(o, T(g[pre + "target"]), tfs, 2, str(metric), cw, 0.3, seg_weight=float(segw))
        loss["total"].backward()
        for k in ("seg", "sr", "total", "tf_rot", "tf_trans", "tf_scale", "tf"):
            ref = float(g[pre + "loss_" + k])
            assert abs(float(loss[k].detach()) - ref) <= 2e-6 * max(abs(ref), 1e-3), (c, k)
        assert rel_err(o.grad.numpy(), g[pre + "gout"]) < 1e-5, c
        assert rel_err(tfs.grad.numpy(), g[pre + "gtfs"]) < 1e-5, c


def test_g8_foreground_sampling_bit_exact():
    g = load_golden("g8_sampling.npz")
    for norm in ("minmax", "zscore"):
        coords, vals, ijk = O.foreground_coords_and_values([g["t2"], g["t1"], g["seg"]], g["affine"], g["bbox"], norm)
        assert np.array_equal(ijk, g["ijk"])                       # voxel order: bit-exact
        assert np.array_equal(vals[:, -1], g[f"values_{norm}"][:, -1])  # labels: bit-exact
        assert np.array_equal(coords.view(np.uint32), g["coords"].view(np.uint32))  # f32 bit patterns
        assert np.array_equal(vals.view(np.uint32), g[f"values_{norm}"].view(np.uint32))


def test_g9_world_grid_conditions_mean_latent():
    g = load_golden("g9_atlas_helpers.npz")
    for tag, sp in (("s05", 0.5), ("s15", 1.5)):
        coords, shape, affine, axes = O.world_grid([155, 155, 155], [sp] * 3)
        assert shape == list(g[f"{tag}_shape"])
        assert np.array_equal(axes[0].numpy(), g[f"{tag}_axis"])
        assert np.array_equal(coords[T(g[f"{tag}_sel"])].numpy(), g[f"{tag}_selcoords"])
        assert np.array_equal(affine.numpy(), g[f"{tag}_affine"])
    v = O.normalize_condition_value(torch.tensor([37.0, 39.0, 41.5, 46.0]), 37, 46, 0.05)
    assert np.allclose(v.numpy(), g["normcond"], rtol=1e-6, atol=0)
    cons = {"birth_age": {"min": 25, "max": 42}, "lv": {"min": 0.0085, "max": 0.061}}
    conds = {"birth_age": {"values": [26, 31, 36], "normed_values": False},
             "lv": {"values": [-1.25, 0.0, 1.25], "normed_values": True}}
    assert np.allclose(np.array(O.condition_combinations(conds, cons, 0.05)), g["combos"], rtol=1e-12)
    assert O.condition_combinations(None, cons, 0.05) == [[]]
    for i in range(2):
        mu = O.normalize_condition_value(float(g[f"ml{i}_age"]), 37, 46, 0.05)
        m = O.mean_latent(T(g["ml_lat"]), T(g["ml_cv"]), mu, 37, 46, 1.0, 0.05, n_max=int(g[f"ml{i}_nmax"]))
        assert rel_err(m.numpy(), g[f"ml{i}_out"]) < 1e-6


def test_g10_inference_postprocess():
    g = load_golden("g10_inference.npz")
    params = {k[len("p_sr_net."):]: T(g[k]) for k in g.files if k.startswith("p_sr_net.")}
    coords, lat, cond = T(g["coords"]), T(g["lat"]), T(g["cond"])
    n = coords.shape[0]
    raw = O.decoder_forward(params, coords, lat, cond.expand(n, -1), None, torch.zeros(n, dtype=torch.long), 2, 30, 30)
    assert rel_err(raw.numpy(), g["raw"]) < 2e-5
    vol = O.inference_postprocess(T(g["raw"]), 2, [12, 12, 12])
    assert vol.shape == g["vol"].shape
    assert np.array_equal(vol[..., 2].numpy(), g["vol"][..., 2])          # hard labels exact
    # a modality that clamps to a constant gives 0/0 = NaN in the reference (inr_decoder.py:52): keep it
    assert np.array_equal(np.isnan(vol.numpy()), np.isnan(g["vol"]))
    assert rel_err(np.nan_to_num(vol.numpy()), np.nan_to_num(g["vol"])) < 1e-6
    raw_tf = O.decoder_forward(params, coords, lat, cond.expand(n, -1), T(g["tf"]).expand(n, -1),
                               torch.zeros(n, dtype=torch.long), 2, 30, 30)
    vol_tf = O.inference_postprocess(raw_tf, 2, [12, 12, 12])
    assert (vol_tf[..., 2].numpy() != g["vol_tf"][..., 2]).mean() < 1e-3  # argmax ties at fp32 noise level
    assert np.abs(np.nan_to_num(vol_tf[..., :2].numpy()) - np.nan_to_num(g["vol_tf"][..., :2])).max() < 1e-4
