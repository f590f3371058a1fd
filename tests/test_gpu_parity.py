"""GPU parity tests: the HIP path (through the C ABI, via the drop-in modules) against
 (a) the golden vectors generated from the reference's own modules, and
 (b) the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star): max-norm relative error <= 1e-4 in fp32 mode and
<= 2e-2 in bf16-MFMA mode on outputs, losses and gradients; labels / indices bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import L2_TOL, both_errs, failing, load_golden, rel_err

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-4, "bf16": 2e-2}
T = torch.from_numpy


def dev():
    return torch.device("cuda:0")


def make_args(latent_dim, hidden, layers, mod, out_dim, cond_dims, precision, loss_metric="mse", class_weights=None,
              tf_weight=0.0):
    return {
        "inr_decoder": {"in_dim": 3, "out_dim": out_dim, "latent_dim": latent_dim, "cond_dims": cond_dims,
                        "hidden_size": hidden, "num_hidden_layers": layers, "modulated_layers": mod,
                        "omega": [30, 30], "cnn_kernel_size": 0, "tf_dim": 6, "precision": precision},
        "optimizer": {"tf_weight": tf_weight, "loss_metric": loss_metric},
        "dataset": {"class_weights": class_weights, "label_names": ["ClearLabel", "CSF", "cGM", "WM", "BG"]},
        "mask_reconstruction": False, "device": "cuda:0",
    }


def test_library_loads_and_reports_version():
    from cinema_amd import _lib
    assert _lib.load().cinema_version() >= 100


def test_g1_embed2affine():
    from cinema_amd.utils import embed2affine, euler2rot
    g = load_golden("g1_affine.npz")
    for d in (6, 9, 12):
        R, t = embed2affine(T(g[f"embed{d}"]).to(dev()))
        assert rel_err(R.cpu().numpy(), g[f"R{d}"]) < 1e-5
        assert np.array_equal(t.cpu().numpy(), g[f"t{d}"])
    R = euler2rot(torch.tensor([[0.1, 0.2, 0.3]], device=dev()))
    assert rel_err(R.cpu().numpy(), g["R_known"][:, :]) < 1e-5


def test_g2_transform_fwd_bwd():
    from cinema_amd.models.inr_decoder import INR_Decoder
    g = load_golden("g2_transform.npz")
    for d in (6, 9, 12):
        tfs = T(g[f"tfs{d}"]).to(dev()).requires_grad_(True)
        y = INR_Decoder.transform(T(g[f"coords{d}"]).to(dev()), tfs)
        assert rel_err(y.detach().cpu().numpy(), g[f"out{d}"]) < 1e-5
        (y * T(g[f"gout{d}"]).to(dev())).sum().backward()
        assert rel_err(tfs.grad.cpu().numpy(), g[f"gtfs{d}"]) < 1e-4, d


@pytest.mark.parametrize("name", ["g333", "g111", "g234"])
def test_g3_spatial_interpolation_fwd_bwd(name):
    from cinema_amd.models.inr_decoder import INR_Decoder
    g = load_golden("g3_interp.npz")
    lat_tab = T(g[f"{name}_lat"])
    idx = T(g[f"{name}_idx"]).long()
    lat = lat_tab[idx].to(dev()).requires_grad_(True)          # per-sample grids, as the reference passes them
    c = T(g[f"{name}_coords"]).to(dev()).requires_grad_(True)
    cond = T(g[f"{name}_cond"]).to(dev())
    o = INR_Decoder.spatial_interpolation(c, lat, cond)
    assert rel_err(o.detach().cpu().numpy(), g[f"{name}_out"]) < 1e-5
    (o * T(g[f"{name}_gout"]).to(dev())).sum().backward()
    glat = torch.zeros_like(lat_tab).index_add_(0, idx, lat.grad.cpu())
    assert rel_err(glat.numpy(), g[f"{name}_glat"]) < 1e-5
    ref_gc = g[f"{name}_gcoords"]
    assert np.abs(c.grad.cpu().numpy() - ref_gc).max() <= 1e-5 * max(1.0, np.abs(ref_gc).max())


def _load_decoder(g, latent_dim, hidden, layers, mod, out_dim, cond, precision, tf_weight=0.1):
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import Criterion
    ncls = out_dim[-1]
    args = make_args(latent_dim, hidden, layers, mod, out_dim, cond, precision, loss_metric="mse",
                     class_weights=[0.0] + [1.0] * (ncls - 1), tf_weight=tf_weight)
    dec = INR_Decoder(args, "cuda:0").to(dev())
    sd = {k[2:]: T(g[k]) for k in g.files if k.startswith("p_")}
    assert list(dec.state_dict().keys()) == list(g["state_keys"])
    dec.load_state_dict(sd)
    return dec, Criterion(args)


CFGS = {
    "cfgT": dict(latent_dim=[32, 1, 1, 1], hidden=128, layers=3, mod=[1, 3], out_dim=[1, 10], cond=0),
    "cfgS": dict(latent_dim=[16, 2, 3, 4], hidden=64, layers=2, mod=[0, 2], out_dim=[2, 5], cond=2),
    "cfgV": dict(latent_dim=[65, 3, 3, 3], hidden=128, layers=5, mod=[1, 3, 5], out_dim=[1, 10], cond=1),
    "cfgD": dict(latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], out_dim=[2, 10], cond=0),
}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["cfgT", "cfgS", "cfgV", "cfgD"])
def test_g5_decoder_golden(name, precision):
    """Full INR_Decoder forward + Criterion + backward against the reference's own outputs."""
    c = CFGS[name]
    g = load_golden(f"g5_decoder_{name}.npz")
    dec, crit = _load_decoder(g, c["latent_dim"], c["hidden"], c["layers"], c["mod"], c["out_dim"], c["cond"], precision)
    tol = TOL[precision]
    lat = T(g["lat"]).to(dev()).requires_grad_(True)
    tft = T(g["tft"]).to(dev()).requires_grad_(True)
    cond = T(g["cond"]).to(dev()).requires_grad_(True)
    coords, target = T(g["coords"]).to(dev()), T(g["target"]).to(dev())
    idx = T(g["idx"]).to(dev())
    tfs = tft[idx.long()]
    out = dec(coords, lat, cond, tfs, idcs_df=idx)
    loss = crit(out, target, tfs, seg_weight=1.0)
    loss["total"].backward()
    errs = both_errs("out", out.detach().cpu().numpy(), g["out"])
    for k in ("total", "sr", "seg", "tf"):
        errs["loss_" + k] = abs(float(loss[k]) - float(g["loss_" + k])) / max(abs(float(g["loss_" + k])), 1e-12)
    errs.update(both_errs("glat", lat.grad.cpu().numpy(), g["glat"]))
    errs.update(both_errs("gtft", tft.grad.cpu().numpy(), g["gtft"]))
    if c["cond"] > 0:
        errs.update(both_errs("gcond", cond.grad.cpu().numpy(), g["gcond"]))
    for n, p in dec.named_parameters():
        errs.update(both_errs("g_" + n, p.grad.cpu().numpy(), g["g_" + n]))
    bad = failing(errs, tol, L2_TOL[precision])
    print(f"{name}/{precision}: worst max-norm {max(v for k, v in errs.items() if not k.endswith('/l2')):.3g}, "
          f"worst L2 {max(v for k, v in errs.items() if k.endswith('/l2')):.3g}")
    assert not bad, f"{name}/{precision}: rel-err above {tol} (max-norm) / {L2_TOL[precision]} (L2): {bad}\nall: {errs}"


def test_g6_criterion_golden():
    from cinema_amd.utils import Criterion
    g = load_golden("g6_criterion.npz")
    for c in range(int(g["n_cases"])):
        pre = f"c{c}_"
        metric, segw, cwtag, tfd = g[pre + "meta"]
        cw = list(g[pre + "cw"]) if cwtag == "w" else None
        args = make_args([8, 1, 1, 1], 16, 1, [1], [2, 10], 0, "fp32", loss_metric=str(metric), class_weights=cw, tf_weight=0.3)
        crit = Criterion(args)
        o = T(g[pre + "out"]).to(dev()).requires_grad_(True)
        tfs = T(g[pre + "tfs"]).to(dev()).requires_grad_(True)
        loss = crit(o, T(g[pre + "target"]).to(dev()), tfs, seg_weight=float(segw))
        assert set(loss.keys()) == {"seg", "sr", "trafo", "total", "tf_rot", "tf_trans", "tf_scale", "tf"}
        loss["total"].backward()
        for k in ("seg", "sr", "total", "tf_rot", "tf_trans", "tf_scale", "tf"):
            ref = float(g[pre + "loss_" + k])
            assert abs(float(loss[k]) - ref) <= 1e-5 * max(abs(ref), 1e-3), (c, k, float(loss[k]), ref)
        assert rel_err(o.grad.cpu().numpy(), g[pre + "gout"]) < 1e-4, c
        assert rel_err(tfs.grad.cpu().numpy(), g[pre + "gtfs"]) < 1e-4, c
    # tfs=None raises KeyError like the reference (utils.py:103)
    with pytest.raises(KeyError):
        crit(o, T(g[pre + "target"]).to(dev()), None)


def test_out_of_range_label_is_not_silent():
    """A segmentation label outside [0, n_cls) aborts the reference's CrossEntropyLoss; the K3 kernel marks the loss and
    that sample's gradient row with NaN instead of clamping the label quietly."""
    from cinema_amd.utils import Criterion
    args = make_args([8, 1, 1, 1], 16, 1, [1], [2, 10], 0, "fp32", loss_metric="mse", class_weights=None, tf_weight=0.0)
    crit = Criterion(args)
    o = torch.randn(40, 12, device=dev(), requires_grad=True)
    target = torch.cat((torch.rand(40, 2), torch.randint(1, 10, (40, 1)).float()), 1).to(dev())
    tfs = torch.zeros(40, 6, device=dev(), requires_grad=True)
    good = crit(o, target, tfs, seg_weight=1.0)
    assert bool(torch.isfinite(good["total"]))
    target[7, 2] = 10.0
    bad = crit(o, target, tfs, seg_weight=1.0)
    assert bool(torch.isnan(bad["seg"])) and bool(torch.isfinite(bad["sr"]))
    bad["total"].backward()
    assert bool(torch.isnan(o.grad[7, 2:]).all()) and bool(torch.isfinite(o.grad[8]).all())


def _oracle_case(N, S, latent_dim, hidden, layers, mod, out_dim, cond, seed, precision, loss_metric="mse", tf_table=False):
    from oracle import cinema_oracle as O
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import Criterion
    gen = torch.Generator().manual_seed(seed)
    ncls = out_dim[-1]
    sr = sum(out_dim[:-1])
    cw = [0.0] + [1.0 + 0.25 * (i % 3) for i in range(ncls - 1)]
    args = make_args(latent_dim, hidden, layers, mod, out_dim, cond, precision, loss_metric=loss_metric, class_weights=cw,
                     tf_weight=0.05)
    torch.manual_seed(seed)
    dec = INR_Decoder(args, "cuda:0").to(dev())
    crit = Criterion(args)
    lat = (torch.randn(S, *latent_dim, generator=gen) * 0.05)
    tft = (torch.rand(S, 6, generator=gen) - 0.5) * 0.1
    coords = torch.rand(N, 3, generator=gen) * 1.9 - 0.95
    idx = torch.randint(0, S, (N,), generator=gen, dtype=torch.int32)
    condv = torch.randn(N, cond, generator=gen) * 0.05
    target = torch.cat((torch.rand(N, sr, generator=gen), torch.randint(1, ncls, (N, 1), generator=gen).float()), 1)
    # oracle on CPU
    params = {k[len("sr_net."):]: v.detach().cpu().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
    lat_o, tft_o, cond_o = (x.clone().requires_grad_(True) for x in (lat, tft, condv))
    cfg = dict(num_hidden_layers=layers, omega=[30, 30], sr_dims=sr, loss_metric=loss_metric,
               class_weights=torch.tensor(cw), tf_weight=0.05)
    loss_o, out_o = O.training_step(params, lat_o, tft_o, coords, target, idx, cond_o, cfg)
    # HIP path
    lat_g, tft_g, cond_g = (x.to(dev()).requires_grad_(True) for x in (lat, tft, condv))
    idx_g = idx.to(dev())
    if tf_table:
        out = dec(coords.to(dev()), lat_g, cond_g, tft_g, idcs_df=idx_g, tf_table=True)
        tfs_for_loss = tft_g[idx_g.long()]
    else:
        tfs_for_loss = tft_g[idx_g.long()]
        out = dec(coords.to(dev()), lat_g, cond_g, tfs_for_loss, idcs_df=idx_g)
    loss = crit(out, target.to(dev()), tfs_for_loss, seg_weight=1.0)
    loss["total"].backward()
    errs = both_errs("out", out.detach().cpu().numpy(), out_o.detach().numpy())
    errs["loss"] = abs(float(loss["total"]) - float(loss_o["total"])) / abs(float(loss_o["total"]))
    errs.update(both_errs("glat", lat_g.grad.cpu().numpy(), lat_o.grad.numpy()))
    errs.update(both_errs("gtft", tft_g.grad.cpu().numpy(), tft_o.grad.numpy()))
    if cond:
        errs.update(both_errs("gcond", cond_g.grad.cpu().numpy(), cond_o.grad.numpy()))
    for n, p in dec.named_parameters():
        errs.update(both_errs("g_" + n, p.grad.cpu().numpy(), params[n[len("sr_net."):]].grad.numpy()))
    print("worst max-norm %.3g, worst L2 %.3g" % (max(v for k, v in errs.items() if not k.endswith("/l2")),
                                                  max(v for k, v in errs.items() if k.endswith("/l2"))))
    return errs


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_oracle_parity_ragged_batch(precision):
    """N not a multiple of the 128-row tile, many subjects some of which get no sample, L1 loss."""
    errs = _oracle_case(N=1000, S=37, latent_dim=[48, 3, 3, 3], hidden=256, layers=4, mod=[1, 3], out_dim=[2, 10], cond=1,
                        seed=11, precision=precision, loss_metric="l1")
    bad = failing(errs, TOL[precision], L2_TOL[precision])
    assert not bad, f"{bad}\nall: {errs}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_oracle_parity_dhcp_shape_tf_table(precision):
    """cfg-D dimensions (latent 256x3^3, hidden 512, out 2+10) at N = 4096, S = 16; tf table mode."""
    errs = _oracle_case(N=4096, S=16, latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], out_dim=[2, 10],
                        cond=0, seed=12, precision=precision, tf_table=True)
    bad = failing(errs, TOL[precision], L2_TOL[precision])
    assert not bad, f"{bad}\nall: {errs}"


def test_oracle_parity_wide_ragged_256_tiles():
    """bf16, hidden 320 (HP = 320: a 256-row tile and a 64-row remainder), latent 64 + 1 condition (LP = 128, K = 65),
    N = 5000 >= 4096: the 256 x 256 weight-gradient tiles with their row / column masks and the slab reduction
    (hidden > 256 and <= 512 also takes the activation-stationary chains)."""
    errs = _oracle_case(N=5000, S=9, latent_dim=[64, 2, 2, 2], hidden=320, layers=3, mod=[1, 2], out_dim=[1, 6], cond=1,
                        seed=23, precision="bf16", tf_table=True)
    bad = failing(errs, TOL["bf16"], L2_TOL["bf16"])
    assert not bad, f"{bad}\nall: {errs}"


def test_oracle_parity_wide_latent_257():
    """bf16, latent 256 + 1 condition = 257 (LP = 320: the 256-row d-latent tile plus a 64-row remainder tile whose upper
    waves idle), hidden 256, N = 4200."""
    errs = _oracle_case(N=4200, S=5, latent_dim=[256, 1, 1, 1], hidden=256, layers=2, mod=[1, 2], out_dim=[1, 6], cond=1,
                        seed=29, precision="bf16", tf_table=True)
    bad = failing(errs, TOL["bf16"], L2_TOL["bf16"])
    assert not bad, f"{bad}\nall: {errs}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_oracle_parity_cfgV_full_shape(precision):
    """BASELINE config 5 (cfg-V) at its real shape through the module API: latent 256 x 3^3 + 1 condition (lat 257,
    LP 320), hidden 512, 5 hidden layers modulated 1/3/5, out 1 + 10, per-sample condition rows with gradient."""
    errs = _oracle_case(N=5000, S=12, latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], out_dim=[1, 10], cond=1,
                        seed=47, precision=precision, tf_table=True)
    bad = failing(errs, TOL[precision], L2_TOL[precision])
    assert not bad, f"{bad}\nall: {errs}"


@pytest.mark.parametrize("n", [16400, 25000])
def test_oracle_parity_mixed_chain_tiles(n):
    """bf16, batch between one and two chip fillings of 64-sample workgroups (256 CUs x 64 = 16 384): the chain kernels
    then run ~256 workgroups of 64 samples followed by 48-sample ones in one launch (fused_launch_plan).  Ragged N, the
    bench's batch size and the smallest batch that takes the plan."""
    errs = _oracle_case(N=n, S=11, latent_dim=[32, 3, 3, 3], hidden=128, layers=3, mod=[1, 3], out_dim=[2, 10], cond=0,
                        seed=31, precision="bf16", tf_table=True)
    bad = failing(errs, TOL["bf16"], L2_TOL["bf16"])
    assert not bad, f"{bad}\nall: {errs}"


@pytest.mark.parametrize("n,hidden", [(16500, 512), (20500, 512), (20500, 256)])
def test_oracle_parity_wide_backward_tile_counts(n, hidden):
    """bf16, hidden 256 / 512: the wide backward chain (fused_wide.hip) with workgroups of 5 + 4 and of 6 + 5 sample tiles
    on a 256-CU chip (the bench shape below runs 7 + 6, the small cases above 4), ragged N."""
    errs = _oracle_case(N=n, S=7, latent_dim=[32, 3, 3, 3], hidden=hidden, layers=3, mod=[1, 3], out_dim=[2, 10], cond=0,
                        seed=37, precision="bf16", tf_table=True)
    bad = failing(errs, TOL["bf16"], L2_TOL["bf16"])
    assert not bad, f"{bad}\nall: {errs}"


def test_oracle_parity_full_bench_shape():
    """The configuration the bench line is quoted on, at full size (BASELINE.json cfg-D: latent 256 x 3^3, hidden 512,
    5 hidden layers, modulated 1/3/5, out 2 + 10, 64 subjects, n_samples = 25 000), bf16, against the CPU oracle:
    mixed 64 / 48-sample chain tiles, 256 x 256 weight-gradient tiles with 7 sample splits, the wide d-latent GEMM."""
    errs = _oracle_case(N=25000, S=64, latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], out_dim=[2, 10],
                        cond=0, seed=41, precision="bf16", tf_table=True)
    bad = failing(errs, TOL["bf16"], L2_TOL["bf16"])
    assert not bad, f"{bad}\nall: {errs}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_oracle_parity_hidden_1024_reference_default_width(precision):
    """hidden_size 1024, the reference's default (configs/config_atlas.yaml:78): above FUSED_MAX_HP, so the
    layer-by-layer kernels (128 x 128 tiles) run.  Parity against the CPU oracle at N = 3000, then one timing line at the
    bench batch size (N = 25 000) so this width has a measured number (printed; `pytest -s` or the captured log)."""
    shape = dict(latent_dim=[256, 3, 3, 3], hidden=1024, layers=5, mod=[1, 3, 5], out_dim=[2, 10], cond=0)
    errs = _oracle_case(N=3000, S=8, seed=43, precision=precision, tf_table=True, **shape)
    bad = failing(errs, TOL[precision], L2_TOL[precision])
    assert not bad, f"{bad}\nall: {errs}"
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import Criterion
    n, S = 25000, 64
    args = make_args(shape["latent_dim"], 1024, 5, [1, 3, 5], [2, 10], 0, precision, loss_metric="mse",
                     class_weights=[0.0] + [1.0] * 9, tf_weight=0.0)
    torch.manual_seed(3)
    dec, crit = INR_Decoder(args, "cuda:0").to(dev()), Criterion(args)
    lat = (torch.randn(S, 256, 3, 3, 3, device=dev()) * 0.05).requires_grad_(True)
    tft = ((torch.rand(S, 6, device=dev()) - 0.5) * 0.1).requires_grad_(True)
    coords = torch.rand(n, 3, device=dev()) * 1.9 - 0.95
    idx = torch.randint(0, S, (n,), device=dev(), dtype=torch.int32)
    target = torch.cat((torch.rand(n, 2, device=dev()), torch.randint(1, 10, (n, 1), device=dev()).float()), 1)

    def step():
        out = dec(coords, lat, torch.zeros(n, 0, device=dev()), tft, idcs_df=idx, tf_table=True)
        crit(out, target, tft[idx.long()], seg_weight=1.0)["total"].backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    flops = 2 * 3 * (3 * 1024 + 5 * 1024 * 1024 + 3 * 256 * 2048 + (1024 + 256) * 12) * n
    print(f"hidden 1024 / {precision}: fwd + loss + bwd through the module API {ms:.3f} ms at N = {n} "
          f"({n / ms * 1e3 / 1e6:.2f} M samples/s, {flops / ms / 1e9:.0f} TFLOP/s algorithmic)")
    assert ms < 50.0


def test_oracle_parity_large_batch_grouping_path():
    """N above the single-launch grouping limit (65536): the histogram / scan / fill launches and many sample tiles."""
    errs = _oracle_case(N=70001, S=7, latent_dim=[16, 2, 2, 2], hidden=64, layers=2, mod=[1], out_dim=[1, 4], cond=0,
                        seed=19, precision="fp32", tf_table=True)
    bad = failing(errs, 2e-4, 1e-4)     # sums over 10^4 samples per subject in fp32
    assert not bad, f"{bad}\nall: {errs}"


def test_tiny_batch_and_single_sample():
    """N = 1 and N = 3 (far below one tile): shapes stay (N, out) and values match the oracle."""
    for n in (1, 3):
        errs = _oracle_case(N=n, S=2, latent_dim=[32, 1, 1, 1], hidden=128, layers=3, mod=[1, 3], out_dim=[1, 10], cond=0,
                            seed=13 + n, precision="fp32")
        bad = {k: v for k, v in errs.items() if not (v < 1e-4)}
        assert not bad, f"N={n}: {bad}"


def test_empty_batch():
    from cinema_amd.models.inr_decoder import INR_Decoder
    args = make_args([32, 1, 1, 1], 128, 3, [1, 3], [1, 10], 0, "bf16")
    dec = INR_Decoder(args, "cuda:0").to(dev())
    out = dec(torch.zeros(0, 3, device=dev()), torch.zeros(2, 32, 1, 1, 1, device=dev()), torch.zeros(0, 0, device=dev()),
              None, idcs_df=torch.zeros(0, dtype=torch.int32, device=dev()))
    assert out.shape == (0, 11)


def test_cpu_tensors_fail_loudly():
    from cinema_amd import _lib
    from cinema_amd.models.inr_decoder import INR_Decoder
    args = make_args([32, 1, 1, 1], 128, 3, [1, 3], [1, 10], 0, "bf16")
    dec = INR_Decoder(args, "cpu")
    with pytest.raises(_lib.CinemaHipError):
        dec(torch.zeros(4, 3), torch.zeros(2, 32, 1, 1, 1), torch.zeros(4, 0), None, idcs_df=torch.zeros(4, dtype=torch.int32))


def test_g10_inference_volume():
    from cinema_amd.models.inr_decoder import INR_Decoder
    g = load_golden("g10_inference.npz")
    args = make_args([16, 3, 3, 3], 64, 2, [1, 2], [2, 5], 1, "fp32")
    dec = INR_Decoder(args, "cuda:0").to(dev())
    dec.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("p_")})
    coords, lat, cond, tf = (T(g[k]).to(dev()) for k in ("coords", "lat", "cond", "tf"))
    dec.min_inference_chunk = 1          # keep step_size = 500 as the chunk: four chunks, the last one ragged
    vol = dec.inference(coords, lat, cond, [12, 12, 12], None, step_size=500).cpu().numpy()
    ref = g["vol"]
    assert vol.shape == ref.shape
    assert np.array_equal(np.isnan(vol), np.isnan(ref))
    # hard labels: identical except where the two top logits tie at fp32 noise level
    assert (vol[..., 2] != ref[..., 2]).mean() < 1e-3
    assert np.abs(np.nan_to_num(vol[..., 3:]) - np.nan_to_num(ref[..., 3:])).max() < 1e-4      # soft labels
    assert np.abs(np.nan_to_num(vol[..., :2]) - np.nan_to_num(ref[..., :2])).max() < 1e-3      # min-max scaled intensities
    vol_tf = dec.inference(coords, lat, cond, [12, 12, 12], tf, step_size=500).cpu().numpy()
    assert np.abs(np.nan_to_num(vol_tf[..., 3:]) - np.nan_to_num(g["vol_tf"][..., 3:])).max() < 1e-4


@pytest.mark.parametrize("cond_dims", [0, 1])
def test_inference_nodes_path_matches_per_sample_path(cond_dims, monkeypatch):
    """Dense inference on the bf16 activation-stationary stack takes the nodes path (FiLM modulation interpolated from the
    volume's node projections); CINEMA_NODES=0 runs the per-sample forward.  Same volume within bf16 tolerance, from a
    WorldGrid (coordinates made per chunk on the GPU) and from the explicit (G, 3) tensor, several ragged chunks, with a
    transformation row."""
    from cinema_amd import engine as E
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import generate_world_grid
    args = make_args([64, 3, 3, 3], 256, 4, [1, 3], [2, 6], cond_dims, "bf16")
    args["atlas_gen"] = {"spacing": [155.0 / 21, 155.0 / 19, 155.0 / 23]}
    args["dataset"]["world_bbox"] = [155, 155, 155]
    torch.manual_seed(5)
    dec = INR_Decoder(args, "cuda:0").to(dev())
    dec.min_inference_chunk = 1
    lat = torch.randn(1, 64, 3, 3, 3, device=dev()) * 0.1
    cond = torch.full((cond_dims,), 0.03, device=dev())
    tf = (torch.rand(6, device=dev()) - 0.5) * 0.1
    grid, shape, _ = generate_world_grid(args, device=dev(), lazy=True)
    assert len(grid) == shape[0] * shape[1] * shape[2] and torch.equal(grid.materialize(), generate_world_grid(args, device=dev())[0])
    assert E.nodes_inference_eligible(dec.sr_net.hot_state(dev()).engine, (1, 64, 3, 3, 3))
    vol_n = dec.inference(grid, lat, cond, shape, tf, step_size=2000)
    vol_t = dec.inference(grid.materialize(), lat, cond, shape, tf, step_size=2000)
    # coordinates made on the GPU == the meshgrid's (a clamped-flat channel is 0/0 = NaN as in the reference)
    assert torch.equal(torch.isnan(vol_n), torch.isnan(vol_t)) and torch.equal(torch.nan_to_num(vol_n), torch.nan_to_num(vol_t))
    monkeypatch.setenv("CINEMA_NODES", "0")
    vol_p = dec.inference(grid, lat, cond, shape, tf, step_size=2000)
    assert vol_n.shape == (*shape, 2 + 1 + 6)
    soft = (vol_n[..., 3:] - vol_p[..., 3:]).abs().max()
    assert torch.equal(torch.isnan(vol_n), torch.isnan(vol_p))
    inten = (torch.nan_to_num(vol_n[..., :2]) - torch.nan_to_num(vol_p[..., :2])).abs().max()
    assert float(soft) < 2e-2 and float(inten) < 2e-2, (float(soft), float(inten))
    flips = float((vol_n[..., 2] != vol_p[..., 2]).float().mean())            # argmax flips only where two classes tie
    print(f"[nodes-vs-per-sample cond={cond_dims}] soft {float(soft):.2e} inten {float(inten):.2e} argmax flips {flips:.2e}", flush=True)
    assert flips < FLIP_BOUND


FLIP_BOUND = 2e-3       # measured: no flip at all on these grids (9 - 10 k voxels); one or two ties may flip on another seed


def test_inference_full_310_volume_sampled_against_oracle():
    """BASELINE config 4 at FULL size: one 310^3 volume (29.8 M coordinates) through INR_Decoder.inference on the cfg-D
    stack (bf16, nodes path, WorldGrid coordinates made on the GPU), and 8 192 randomly chosen grid points of it against
    the CPU oracle on those coordinates: soft labels, the clamped intensities before the whole-volume min-max (undone
    with the min / max the call leaves in `last_minmax`), hard labels wherever the oracle's top two logits are apart."""
    from oracle import cinema_oracle as O
    from cinema_amd import engine as E
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import generate_world_grid
    args = make_args([256, 3, 3, 3], 512, 5, [1, 3, 5], [2, 10], 0, "bf16")
    args["atlas_gen"] = {"spacing": [0.5, 0.5, 0.5]}
    args["dataset"]["world_bbox"] = [155, 155, 155]
    torch.manual_seed(11)
    dec = INR_Decoder(args, "cuda:0").to(dev())
    # a fresh SIREN answers with near-zero outputs: give the heads a range in which clamp, min-max and softmax matter
    sd = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    sd["sr_net.final_linear.weight"] *= 12.0
    sd["sr_net.final_linear.weight"][2:] *= 8.0          # class logits a few units apart
    sd["sr_net.final_linear.bias"][:2] = 0.5
    dec.load_state_dict(sd)
    lat = torch.randn(1, 256, 3, 3, 3, device=dev()) * 0.1
    tf = (torch.rand(6, device=dev()) - 0.5) * 0.1
    grid, shape, _ = generate_world_grid(args, device=dev(), lazy=True)
    assert shape == [310, 310, 310]
    assert E.nodes_inference_eligible(dec.sr_net.hot_state(dev()).engine, (1, 256, 3, 3, 3))
    vol = dec.inference(grid, lat, torch.zeros(0, device=dev()), shape, tf)        # default step_size, default chunk floor
    assert vol.shape == (310, 310, 310, 2 + 1 + 10)
    G = 310 ** 3
    pick = torch.randint(0, G, (8192,), generator=torch.Generator().manual_seed(3))
    ix, iy, iz = pick // (310 * 310), (pick // 310) % 310, pick % 310
    ax = [a.cpu() for a in grid.axes]
    coords = torch.stack((ax[0][ix], ax[1][iy], ax[2][iz]), 1)
    got = vol.reshape(G, -1)[pick.to(dev())].cpu()
    mm = dec.last_minmax.cpu()
    params = {k[len("sr_net."):]: v.detach().cpu() for k, v in dec.state_dict().items() if k.startswith("sr_net.")}
    out = O.decoder_forward(params, coords, lat.cpu(), None, tf.cpu().reshape(1, 6).expand(8192, 6), torch.zeros(8192, dtype=torch.int64),
                            5, 30, 30)
    inten_o = out[:, :2].clamp(0, 1)
    inten_g = got[:, :2] * (mm[2:] - mm[:2]) + mm[:2]
    soft_o = torch.softmax(out[:, 2:], -1)
    e_int, e_soft = float((inten_g - inten_o).abs().max()), float((got[:, 3:] - soft_o).abs().max())
    top2 = out[:, 2:].topk(2, dim=-1).values
    margin = top2[:, 0] - top2[:, 1]
    hard_o = out[:, 2:].argmax(-1).float()
    flip = got[:, 2] != hard_o
    print(f"[310^3 vs oracle] oracle intensities in [{float(out[:, :2].min()):.2f}, {float(out[:, :2].max()):.2f}], largest soft label "
          f"{float(soft_o.max()):.2f}, min/max of the volume {mm.tolist()}", flush=True)
    print(f"[310^3 vs oracle] intensities {e_int:.2e} soft labels {e_soft:.2e} argmax flips {float(flip.float().mean()):.2e}, "
          f"largest oracle margin at a flip {float(margin[flip].max()) if bool(flip.any()) else 0.0:.2e}", flush=True)
    # measured on MI355X: intensities 6.8e-4, soft labels 7.5e-4, 0.16 % argmax flips, all where the oracle's two top
    # logits are < 4e-3 apart (logits of a few units) -- bounds = that with a margin of 4-5 x
    assert e_int < 3e-3 and e_soft < 3e-3, (e_int, e_soft)
    assert not bool((flip & (margin > 2e-2)).any())
    assert float(flip.float().mean()) < 8e-3


def test_state_dict_roundtrip_and_deepcopy():
    import copy
    from cinema_amd.models.inr_decoder import INR_Decoder
    args = make_args([32, 1, 1, 1], 128, 3, [1, 3], [1, 10], 0, "bf16")
    torch.manual_seed(0)
    dec = INR_Decoder(args, "cuda:0").to(dev())
    x = (torch.rand(50, 3, device=dev()) * 2 - 1, torch.randn(2, 32, 1, 1, 1, device=dev()) * 0.1)
    idx = torch.randint(0, 2, (50,), device=dev(), dtype=torch.int32)
    y0 = dec(x[0], x[1], torch.zeros(50, 0, device=dev()), None, idcs_df=idx)
    dec2 = copy.deepcopy(dec)
    y1 = dec2(x[0], x[1], torch.zeros(50, 0, device=dev()), None, idcs_df=idx)
    assert torch.equal(y0, y1)
    sd = {k: v.cpu() for k, v in dec.state_dict().items()}
    dec3 = INR_Decoder(args, "cuda:0").to(dev())
    dec3.load_state_dict(sd)
    y2 = dec3(x[0], x[1], torch.zeros(50, 0, device=dev()), None, idcs_df=idx)
    assert torch.equal(y0, y2)


def test_inference_forward_keeps_nothing_and_matches():
    """no_grad forward (cinema_siren_forward with saved == NULL on the activation-stationary path) is bit-identical to
    the training forward, leaves the saved buffer untouched, and the layer-by-layer (fp32) path refuses saved == NULL."""
    import ctypes as C
    from cinema_amd import _lib
    from cinema_amd._lib import ptr
    c = CFGS["cfgD"]
    g = load_golden("g5_decoder_cfgD.npz")
    dec, _ = _load_decoder(g, c["latent_dim"], c["hidden"], c["layers"], c["mod"], c["out_dim"], c["cond"], "bf16")
    lat = T(g["lat"]).to(dev())
    coords, idx = T(g["coords"]).to(dev()), T(g["idx"]).to(dev())
    cond = T(g["cond"]).to(dev())
    tfs = T(g["tft"]).to(dev())[idx.long()]
    y_train = dec(coords, lat.clone().requires_grad_(True), cond, tfs, idcs_df=idx).detach().clone()
    eng = dec.sr_net.hot_state(dev()).engine
    assert eng.can_skip_saved()
    eng.ws.saved.fill_(7.0)
    with torch.no_grad():
        y_inf = dec(coords, lat, cond, tfs, idcs_df=idx)
    assert torch.equal(y_inf, y_train)
    assert bool((eng.ws.saved == 7.0).all()), "inference forward wrote saved activations"
    assert rel_err(y_inf.cpu().numpy(), g["out"]) < TOL["bf16"]

    dec32, _ = _load_decoder(g, c["latent_dim"], c["hidden"], c["layers"], c["mod"], c["out_dim"], c["cond"], "fp32")
    with torch.no_grad():
        y32 = dec32(coords, lat, cond, tfs, idcs_df=idx)          # falls back to keeping the activations
    assert rel_err(y32.cpu().numpy(), g["out"]) < TOL["fp32"]
    e32 = dec32.sr_net.hot_state(dev()).engine
    assert not e32.can_skip_saved()
    n = coords.shape[0]
    rc = e32.lib.cinema_siren_forward(C.byref(e32.desc), ptr(e32.flat), ptr(e32.packed), ptr(e32.ws.coords_t), ptr(e32.ws.lat),
                                      n, (n + 127) // 128 * 128, None, ptr(e32.ws.out), e32.stream())
    assert rc != 0 and b"saved" in e32.lib.cinema_last_error()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("S,C,grid,k", [(5, 48, (3, 3, 3), 3), (64, 256, (3, 3, 3), 3), (3, 16, (2, 3, 4), 5), (2, 32, (1, 1, 1), 3),
                                       (4, 24, (3, 3, 3), 2), (3, 16, (3, 4, 2), 4)])
def test_conv_modulator_matches_torch_conv3d(S, C, grid, k, precision):
    """Modulator with cnn_kernel_size > 0 (models/inr_decoder.py:109-123): forward and all three
    gradients of Conv3d(C, C, k, padding='same') over the latent table against torch's conv3d in float64 on the host
    (the reference's own op), incl. the bench's table size, a non-cubic grid with a kernel wider than the grid, 1^3, and even
    kernel sizes (torch pads (k - 1) / 2 in front and the rest behind)."""
    import torch.nn.functional as F
    from cinema_amd.models.inr_decoder import Modulator
    torch.manual_seed(S * 1000 + C)
    mod = Modulator([C, *grid], kernel_size=k, precision=precision).to(dev())
    x = (torch.randn(S, C, *grid) * 0.1).to(dev()).requires_grad_(True)
    gy = torch.randn(S, C, *grid).to(dev())
    y = mod(x)
    (y * gy).sum().backward()
    xr = x.detach().cpu().double().requires_grad_(True)
    wr, br = (p.detach().cpu().double().requires_grad_(True) for p in (mod.conv.weight, mod.conv.bias))
    yr = F.conv3d(xr, wr, br, padding="same")
    (yr * gy.cpu().double()).sum().backward()
    errs = both_errs("y", y.detach().cpu().numpy(), yr.detach().numpy())
    errs.update(both_errs("gx", x.grad.cpu().numpy(), xr.grad.numpy()))
    errs.update(both_errs("gw", mod.conv.weight.grad.cpu().numpy(), wr.grad.numpy()))
    errs.update(both_errs("gb", mod.conv.bias.grad.cpu().numpy(), br.grad.numpy()))
    bad = failing(errs, TOL[precision], L2_TOL[precision])
    assert not bad, f"{bad}\nall: {errs}"


def test_decoder_with_conv_modulator_matches_oracle():
    """INR_Decoder with cnn_kernel_size = 3 end to end (conv over the table, then the usual path) against the CPU oracle
    fed with torch's conv3d of the table; the conv parameters get gradients through the HIP backward."""
    import torch.nn.functional as F
    from oracle import cinema_oracle as O
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import Criterion
    args = make_args([32, 3, 3, 3], 128, 3, [1, 3], [2, 6], 0, "fp32", loss_metric="mse", class_weights=None, tf_weight=0.0)
    args["inr_decoder"]["cnn_kernel_size"] = 3
    torch.manual_seed(9)
    dec, crit = INR_Decoder(args, "cuda:0").to(dev()), Criterion(args)
    gen = torch.Generator().manual_seed(10)
    S, N = 6, 900
    lat = torch.randn(S, 32, 3, 3, 3, generator=gen) * 0.1
    tft = (torch.rand(S, 6, generator=gen) - 0.5) * 0.1
    coords = torch.rand(N, 3, generator=gen) * 1.9 - 0.95
    idx = torch.randint(0, S, (N,), generator=gen, dtype=torch.int32)
    target = torch.cat((torch.rand(N, 2, generator=gen), torch.randint(1, 6, (N, 1), generator=gen).float()), 1)
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    params = {k[len("sr_net."):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("sr_net.")}
    cw, cb = sd["modulator.conv.weight"].clone().requires_grad_(True), sd["modulator.conv.bias"].clone().requires_grad_(True)
    lat_o, tft_o = lat.clone().requires_grad_(True), tft.clone().requires_grad_(True)
    cfg = dict(num_hidden_layers=3, omega=[30, 30], sr_dims=2, loss_metric="mse", class_weights=None, tf_weight=0.0)
    loss_o, out_o = O.training_step(params, F.conv3d(lat_o, cw, cb, padding="same"), tft_o, coords, target, idx, None, cfg)
    lat_g, tft_g = lat.to(dev()).requires_grad_(True), tft.to(dev()).requires_grad_(True)
    out = dec(coords.to(dev()), lat_g, torch.zeros(N, 0, device=dev()), tft_g, idcs_df=idx.to(dev()), tf_table=True)
    crit(out, target.to(dev()), tft_g[idx.to(dev()).long()], seg_weight=1.0)["total"].backward()
    errs = both_errs("out", out.detach().cpu().numpy(), out_o.detach().numpy())
    errs.update(both_errs("glat", lat_g.grad.cpu().numpy(), lat_o.grad.numpy()))
    errs.update(both_errs("g_conv_w", dec.modulator.conv.weight.grad.cpu().numpy(), cw.grad.numpy()))
    errs.update(both_errs("g_conv_b", dec.modulator.conv.bias.grad.cpu().numpy(), cb.grad.numpy()))
    errs.update(both_errs("g_final_w", dec.sr_net.final_linear.weight.grad.cpu().numpy(), params["final_linear.weight"].grad.numpy()))
    bad = failing(errs, TOL["fp32"], L2_TOL["fp32"])
    assert not bad, f"{bad}\nall: {errs}"


def test_inference_memory_follows_the_path(monkeypatch):
    """The default 2^19-row chunk floor applies only where a forward keeps nothing (bf16 chain kernels): there the
    saved-activation tensor is never allocated; the fp32 stack writes its activations and keeps step_size as its memory
    bound.  A workspace grown by an inference call does not stay with an engine that already had one."""
    from cinema_amd.models.inr_decoder import INR_Decoder
    n_vox = 9 * 9 * 9
    coords = torch.rand(n_vox, 3, device=dev()) * 2 - 1
    for prec, hidden in (("bf16", 128), ("fp32", 64)):
        args = make_args([16, 3, 3, 3], hidden, 2, [1, 2], [2, 5], 0, prec)
        torch.manual_seed(3)
        dec = INR_Decoder(args, "cuda:0").to(dev())
        lat = torch.randn(1, 16, 3, 3, 3, device=dev()) * 0.1
        eng = dec.sr_net.hot_state(dev()).engine
        assert eng.ws is None
        vol = dec.inference(coords, lat, torch.zeros(0, device=dev()), [9, 9, 9], None, step_size=200)
        assert vol.shape == (9, 9, 9, 2 + 1 + 5)
        if prec == "bf16":
            assert eng.can_skip_saved() and eng.ws._saved is None            # nothing kept, nothing allocated
            assert eng.ws.npad_cap >= n_vox                                   # whole volume in one chunk (floor 2^19, capped at n)
        else:
            assert not eng.can_skip_saved() and eng.ws._saved is not None
            assert eng.ws.npad_cap == 256                                     # step_size = 200 rows, padded to the row alignment
        # an engine that already has a (training-sized) workspace gets it back after a larger inference chunk
        if prec == "bf16":
            eng.ws = None
            small = eng.workspace(128)
            dec.inference(coords, lat, torch.zeros(0, device=dev()), [9, 9, 9], None, step_size=200)
            assert eng.ws is small


def test_transform_inverse_undoes_transform():
    """INR_Decoder.transform(inverse=True) (models/inr_decoder.py:67-69: R^-1, -R^-1 t) against the oracle's affine and as the
    inverse of the forward transform, for rigid, scaling and shearing embeddings."""
    from oracle import cinema_oracle as O
    from cinema_amd.models.inr_decoder import INR_Decoder
    g = torch.Generator().manual_seed(4)
    for w in (6, 9, 12):
        tf = (torch.rand(300, w, generator=g) - 0.5) * 0.4
        c = torch.rand(300, 3, generator=g) * 2 - 1
        fwd = INR_Decoder.transform(c.to(dev()), tf.to(dev()))
        back = INR_Decoder.transform(fwd, tf.to(dev()), inverse=True).cpu()
        assert float((back - c).abs().max()) < 1e-5
        R, t = O.affine_from_embedding(tf)
        Ri = torch.linalg.inv(R)
        ref = torch.einsum("nxy,ny->nx", Ri, c) - torch.einsum("nij,nj->ni", Ri, t)
        got = INR_Decoder.transform(c.to(dev()), tf.to(dev()), inverse=True).cpu()
        assert float((got - ref).abs().max()) < 1e-5
