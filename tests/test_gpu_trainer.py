"""GPU tests of the training loop pieces above the decoder: fused AdamW, the autograd-free trainer
step (pack -> K1 -> SIREN -> K3 with volume sampling -> backward -> optimiser) and label sampling."""
import numpy as np
import pytest
import torch

from conftest import L2_TOL, both_errs, failing, load_golden, rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def base_args(latent_dim, hidden, layers, mod, out_dim, cond, precision, loss_metric="mse", tf_weight=0.0, cw="default", tf_dim=6):
    ncls = out_dim[-1]
    return {
        "inr_decoder": {"in_dim": 3, "out_dim": out_dim, "latent_dim": latent_dim, "cond_dims": cond, "hidden_size": hidden,
                        "num_hidden_layers": layers, "modulated_layers": mod, "omega": [30, 30], "cnn_kernel_size": 0,
                        "tf_dim": tf_dim, "precision": precision},
        "optimizer": {"tf_weight": tf_weight, "loss_metric": loss_metric, "seg_weight": 1.0, "lr_inr": 1e-4,
                      "inr_weight_decay": 0.0, "lr_latent": 5e-4, "latent_weight_decay": 0.0, "lr_tf": 7.5e-3,
                      "tf_weight_decay": 0.0, "scheduler": {"type": "cosine", "eta_min": 2.5e-5}},
        "dataset": {"class_weights": ([0.0] + [1.0] * (ncls - 1)) if cw == "default" else cw, "world_bbox": [155, 155, 155],
                    "label_names": []},
        "epochs": {"train": 5, "val": 5}, "mask_reconstruction": False, "device": DEV,
    }


def test_hip_adamw_matches_torch_adamw():
    from cinema_amd.optim import HipAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(1000003,), (37, 6), (5, 16, 3, 3, 3)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    a = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    b = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    lrs, wds = [1e-3, 7.5e-3, 5e-4], [0.0, 0.01, 0.1]
    oa = HipAdamW([{"params": [p], "lr": lr, "weight_decay": wd} for p, lr, wd in zip(a, lrs, wds)])
    ob = torch.optim.AdamW([{"params": [p], "lr": lr, "weight_decay": wd} for p, lr, wd in zip(b, lrs, wds)])
    sa = torch.optim.lr_scheduler.CosineAnnealingLR(oa, T_max=4, eta_min=1e-5)
    sb = torch.optim.lr_scheduler.CosineAnnealingLR(ob, T_max=4, eta_min=1e-5)
    for it in range(6):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(DEV) * (0.1 + it)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
    for pa, pb in zip(a, b):
        assert rel_err(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 2e-6


def test_g7_adamw_trajectory_golden():
    """10 optimisation steps with the reference's three parameter groups + cosine schedule
    (build_atlas.py:432-459): drop-in modules in fp32 mode + HipAdamW against the reference's trajectory."""
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.optim import HipAdamW
    from cinema_amd.utils import Criterion
    g = load_golden("g7_adamw.npz")
    args = base_args([8, 2, 2, 2], 32, 2, [1], [1, 4], 0, "fp32", loss_metric="l1")
    dec = INR_Decoder(args, DEV).to(DEV)
    dec.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("p_")})
    crit = Criterion(args)
    lat = torch.nn.Parameter(T(g["lat0"]).to(DEV))
    tft = torch.nn.Parameter(T(g["tft0"]).to(DEV))
    coords, idx, target = T(g["coords"]).to(DEV), T(g["idx"]).to(DEV), T(g["target"]).to(DEV)
    dec(coords, lat, torch.zeros(40, 0, device=DEV), tft[idx], idcs_df=idx)   # re-homes the parameters before the optimiser sees them
    opt = HipAdamW([{"params": [lat], "lr": 5e-4, "weight_decay": 0.0}, {"params": [tft], "lr": 7.5e-3, "weight_decay": 0.0},
                    {"params": list(dec.parameters()), "lr": 1e-4, "weight_decay": 0.0}])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=5, eta_min=2.5e-5)
    losses = []
    for step in range(10):
        opt.zero_grad()
        tfs = tft[idx]
        out = dec(coords, lat, torch.zeros(40, 0, device=DEV), tfs, idcs_df=idx)
        loss = crit(out, target, tfs, seg_weight=1.0)
        loss["total"].backward()
        for p in [lat, tft] + list(dec.parameters()):
            p.grad = p.grad.contiguous()
        opt.step()
        losses.append(float(loss["total"]))
        if step % 2 == 1:
            sched.step()
    assert np.allclose(losses, g["losses"], rtol=2e-4), (losses, g["losses"].tolist())
    # Adam normalises the update, so tiny gradient differences flip early steps of near-zero-gradient entries:
    # compare the trajectories in units of the total distance travelled
    for name, p, ref0, ref in (("lat", lat, g["lat0"], g["lat10"]), ("tft", tft, g["tft0"], g["tft10"])):
        move = np.abs(ref - ref0).max()
        assert np.abs(p.detach().cpu().numpy() - ref).max() < 2e-2 * move, name
    for n, p in dec.named_parameters():
        move = np.abs(g["q_" + n] - g["p_" + n]).max()
        assert np.abs(p.detach().cpu().numpy() - g["q_" + n]).max() < 2e-2 * max(move, 1e-12), n


@pytest.mark.parametrize("precision,grouped", [("fp32", False), ("bf16", False), ("fp32", True), ("bf16", True)])
def test_trainer_step_matches_oracle(precision, grouped):
    """The autograd-free step: gradients it leaves behind == CPU oracle on the same samples (targets gathered from
    the resident volumes: labels bit-exact), and the parameters move as torch.optim.AdamW would move them."""
    _trainer_step_vs_oracle(precision, grouped, S=5, side=24, latent_dim=[48, 3, 3, 3], hidden=128, layers=3, mod=[1, 3], n=700)


def test_trainer_step_matches_oracle_full_bench_shape():
    """The same at the size the bench line is quoted on (cfg-D: 64 subjects, latent 256 x 3^3, hidden 512, 5 hidden
    layers modulated 1/3/5, n_samples 25 000, bf16, batch grouped by subject): the grouped K1 kernels over 64 subjects,
    the 64 / 48-sample chain tiles, the side-stream preparation with the class-weight normaliser at full batch size."""
    _trainer_step_vs_oracle("bf16", True, S=64, side=28, latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], n=25000)


@pytest.mark.parametrize("shape", ["small", "bench"])
def test_trainer_step_nodes_path_matches_oracle(shape, monkeypatch):
    """The opt-in nodes path of the bf16 chain kernels (CINEMA_NODES=1: FiLM modulation interpolated from per-subject
    node projections instead of the per-sample linear_lats GEMM) against the same CPU oracle, same tolerance."""
    monkeypatch.setenv("CINEMA_NODES", "1")
    if shape == "small":
        _trainer_step_vs_oracle("bf16", True, S=5, side=24, latent_dim=[48, 3, 3, 3], hidden=128, layers=3, mod=[1, 3], n=700, nodes=True)
    else:
        _trainer_step_vs_oracle("bf16", True, S=64, side=28, latent_dim=[256, 3, 3, 3], hidden=512, layers=5, mod=[1, 3, 5], n=25000, nodes=True)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_trainer_step_l1_scaling_tfs_and_tf_regulariser_match_oracle(precision):
    """The reference's default loss_metric 'l1' (config_atlas.yaml:64), 9-wide tf rows (rigid + scaling, :73) and a
    non-zero tf_weight (utils.py:97-103): loss dictionary with the reference's eight keys, gradients incl. the
    regulariser's part of d tfs, one AdamW step."""
    _trainer_step_vs_oracle(precision, True, S=5, side=24, latent_dim=[48, 3, 3, 3], hidden=128, layers=3, mod=[1, 3], n=700,
                            loss_metric="l1", tf_weight=0.1, tf_dim=9)


def _trainer_step_vs_oracle(precision, grouped, S, side, latent_dim, hidden, layers, mod, n, nodes=False, loss_metric="mse",
                            tf_weight=0.0, tf_dim=6):
    from oracle import cinema_oracle as O
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    tol = 1e-4 if precision == "fp32" else 2e-2
    args = base_args(latent_dim, hidden, layers, mod, [2, 10], 0, precision, tf_weight=tf_weight, loss_metric=loss_metric, tf_dim=tf_dim)
    torch.manual_seed(21)
    store = SubjectStore.synthetic(S, (side, side, side), 155.0 / side, [155, 155, 155], 2, 10, seed=7, device=DEV)
    assert len(store) >= 2 * n
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=n if grouped else None)
    if grouped:      # every step-sized chunk is sorted by subject -> the grouped K1 entry points are taken
        assert bool((store.subj[:n][1:] >= store.subj[:n][:-1]).all()) and bool((store.subj[n:2 * n][1:] >= store.subj[n:2 * n][:-1]).all())
    tr = HotPathTrainer(args, store, DEV, keep_grads=True)
    with torch.no_grad():
        tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(5)) - 0.5) * 0.05)
    assert tr.tfs.shape[1] == tf_dim
    lat0, tf0, flat0 = tr.latents.detach().cpu().clone(), tr.tfs.detach().cpu().clone(), tr.eng.flat.detach().cpu().clone()
    coords, voxel, subj = store.coords[:n].cpu(), store.voxel[:n].cpu(), store.subj[:n].cpu()
    base = subj.long() * store.V + voxel
    target = torch.cat((store.intens.cpu()[base], store.labels.cpu()[base].float().unsqueeze(1)), 1)
    params = {k: v.clone().requires_grad_(True) for k, v in tr.eng.param_views(flat0).items()}
    lat_o, tf_o = lat0.clone().requires_grad_(True), tf0.clone().requires_grad_(True)
    cfg = dict(num_hidden_layers=layers, omega=[30, 30], sr_dims=2, loss_metric=loss_metric,
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=tf_weight)
    loss_o, _ = O.training_step(params, lat_o, tf_o, coords, target, subj, None, cfg)
    # K3 sampled exactly these targets (bit-exact labels and intensities)
    tgt_dev = torch.empty(n, 3, device=DEV)
    from cinema_amd import engine as E
    assert E.nodes_eligible(tr.eng, tuple(tr.latents.shape)) == nodes
    tr.step(0, n)
    sums = torch.zeros(4, device=DEV)
    dout = torch.empty(E.round_up(n, 128), 12, device=DEV)
    E.sample_loss(None, tr.eng.ws.out, n, 2, 10, loss_metric, tr.class_w, 1.0, 1.0, dout, sums, voxel=store.voxel[:n],
                  subj=store.subj[:n], intens=store.intens, labels=store.labels, voxels_per_subject=store.V, target_out=tgt_dev)
    assert torch.equal(tgt_dev.cpu(), target)
    got = tr.loss_dict()
    # the reference's eight keys (utils.py:88-104); sr / seg belong to the parameters before the step, the tf terms are
    # evaluated on the table after it (loss_dict's docstring) -- compared with the criterion on exactly those rows
    assert set(got) == {"seg", "sr", "trafo", "total", "tf_rot", "tf_trans", "tf_scale", "tf"}
    for k in ("sr", "seg"):
        assert abs(got[k] - float(loss_o[k])) < tol * abs(float(loss_o[k])), (k, got[k], float(loss_o[k]))
    tf_now = O.criterion(torch.zeros(n, 12), target, tr.tfs.detach().cpu()[subj.long()], 2, loss_metric, None, tf_weight)
    for k in ("tf_rot", "tf_trans", "tf_scale", "tf"):
        assert abs(got[k] - float(tf_now[k])) <= 1e-5 * abs(float(tf_now[k])) + 1e-12, (k, got[k], float(tf_now[k]))
    if tf_dim == 9:
        assert got["tf_scale"] > 0
    want_total = float(loss_o["sr"]) + 1.0 * float(loss_o["seg"]) + tf_weight * float(tf_now["tf"])
    assert abs(got["total"] - want_total) < tol * abs(want_total)
    errs = both_errs("glat", tr.lat_grad.cpu().numpy(), lat_o.grad.numpy())
    errs.update(both_errs("gtf", tr.tf_grad.cpu().numpy(), tf_o.grad.numpy()))
    gviews = tr.eng.param_views(tr.eng.flat_grad.cpu())
    for k, p in params.items():
        errs.update(both_errs("g_" + k, gviews[k].numpy(), p.grad.numpy()))
    bad = failing(errs, tol, L2_TOL[precision])
    assert not bad, bad
    # one AdamW step from the trainer's own gradients
    ref_p = [torch.nn.Parameter(lat0.clone()), torch.nn.Parameter(tf0.clone()), torch.nn.Parameter(flat0.clone())]
    ref_p[0].grad, ref_p[1].grad, ref_p[2].grad = tr.lat_grad.cpu(), tr.tf_grad.cpu(), tr.eng.flat_grad.cpu()
    ref_opt = torch.optim.AdamW([{"params": [ref_p[0]], "lr": 5e-4, "weight_decay": 0.0},
                                 {"params": [ref_p[1]], "lr": 7.5e-3, "weight_decay": 0.0},
                                 {"params": [ref_p[2]], "lr": 1e-4, "weight_decay": 0.0}])
    ref_opt.step()
    assert rel_err(tr.latents.detach().cpu().numpy(), ref_p[0].detach().numpy()) < 1e-5
    assert rel_err(tr.tfs.detach().cpu().numpy(), ref_p[1].detach().numpy()) < 1e-5
    assert rel_err(tr.eng.flat.detach().cpu().numpy(), ref_p[2].detach().numpy()) < 1e-5


@pytest.mark.parametrize("precision,as_rank_of_two", [("fp32", False), ("bf16", False), ("bf16", True)])
def test_clearing_step_equals_keeping_step(precision, as_rank_of_two, monkeypatch):
    """Default trainer (per-step clearing + class-weight normaliser on the side stream through cinema_step_prepare, K3
    through the _prepared entry, backward without its memsets, AdamW that zeroes the gradients it consumed) against
    keep_grads=True (every call clears for itself) over several steps on the same samples: same losses, same
    parameters (fp32 bias / first-layer gradients go through float atomics: last-bit freedom), gradients left at zero."""
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    from cinema_amd import parallel as par
    if as_rank_of_two:
        # the code path of one rank of several in ONE process (two ranks sharing this box's single GPU stall for seconds
        # per collective): separate optimiser launch for the MLP after the all-reduce, everything on one stream.  The
        # collective itself is replaced by the identity (a replica whose peer computed the same gradient).
        monkeypatch.setattr(par, "world_info", lambda pg=None: (0, 2))
        monkeypatch.setattr(par, "broadcast_", lambda t, src=0, pg=None: t)
        monkeypatch.setattr(par, "allreduce_mean_", lambda g, pg=None, async_op=False: None)
        monkeypatch.setattr(par, "share_batch_normaliser_", lambda norm, pg=None: par._UnionNormaliser(norm, None, 1))
    args = base_args([48, 3, 3, 3], 128, 3, [1, 3], [2, 10], 0, precision, tf_weight=0.0)
    store = SubjectStore.synthetic(5, (24, 24, 24), 155.0 / 24, [155, 155, 155], 2, 10, seed=7, device=DEV)
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=700)
    trainers = []
    for keep in (True, False):
        torch.manual_seed(21)
        tr = HotPathTrainer(args, store, DEV, keep_grads=keep)
        with torch.no_grad():
            tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(5)) - 0.5) * 0.05)
        trainers.append(tr)
    a, b = trainers
    assert torch.equal(a.eng.flat, b.eng.flat) and torch.equal(a.latents, b.latents)
    assert (b.side is None and b.opt_mlp is not None) if as_rank_of_two else (b.side is not None and b.opt_mlp is None)
    for i in range(4):
        for tr in (a, b):
            tr.step(700 * i, 700 * (i + 1))
        la, lb = a.loss_dict(), b.loss_dict()
        assert abs(la["total"] - lb["total"]) <= 1e-5 * abs(la["total"]), (i, la, lb)
        assert float(b.eng.flat_grad.abs().max()) == 0.0 and float(b.lat_grad.abs().max()) == 0.0 and \
            float(b.tf_grad.abs().max()) == 0.0
        assert float(a.eng.flat_grad.abs().max()) > 0.0
    for name, x, y in (("flat", a.eng.flat, b.eng.flat), ("lat", a.latents, b.latents), ("tfs", a.tfs, b.tfs)):
        assert rel_err(y.detach().cpu().numpy(), x.detach().cpu().numpy()) < 1e-4, name


def test_trainer_loss_decreases_over_an_epoch_slice():
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = base_args([32, 3, 3, 3], 128, 3, [1, 3], [2, 10], 0, "bf16")
    torch.manual_seed(2)
    store = SubjectStore.synthetic(4, (32, 32, 32), 155.0 / 32, [155, 155, 155], 2, 10, seed=3, device=DEV)
    tr = HotPathTrainer(args, store, DEV)
    store.shuffle(torch.Generator(device=DEV).manual_seed(4))
    n = 2000
    first = last = None
    for i in range(60):
        lo = (i * n) % (len(store) - n)
        tr.step(lo, lo + n)
        if i == 0:
            first = tr.loss_dict()["total"]
    last = tr.loss_dict()["total"]
    assert np.isfinite(last) and last < 0.8 * first, (first, last)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tf_dim,cond_dims,order", [(6, 0, "grouped"), (12, 2, "grouped"), (9, 0, "shuffled")])
def test_sorted_k1_forward_is_bit_identical(precision, tf_dim, cond_dims, order):
    """cinema_affine_trilerp_fwd_sorted (table in LDS, API-layout latents, affine built in the kernel) against
    cinema_affine_from_embedding + cinema_latent_to_node_major + cinema_affine_trilerp_fwd: same bits in coords_t and
    lat (pad rows and pad columns included), for grouped and for arbitrary sample order, ragged N."""
    from cinema_amd import _lib, engine as E
    S, Cc, n = 7, 48, 1000 + 37
    lat_size = Cc + cond_dims
    d = E.make_desc(3, lat_size, 12, 128, 4, [1, 3], 30, 30, _lib.PREC_BF16 if precision == "bf16" else _lib.PREC_F32)
    eng = E.SirenEngine(d, DEV)
    g = torch.Generator().manual_seed(3)
    latents = (torch.randn(S, Cc, 3, 3, 3, generator=g) * 0.1).to(DEV)
    tfs = ((torch.rand(S, tf_dim, generator=g) - 0.5) * 0.2).to(DEV)
    coords = (torch.rand(n, 3, generator=g) * 2.4 - 1.2).to(DEV)          # some outside [-1, 1]: border clamp
    idx = torch.randint(0, S - 1, (n,), generator=g, dtype=torch.int32)   # subject S-1 has no samples
    if order == "grouped":
        idx = idx.sort().values
    idx = idx.to(DEV)
    cond = torch.randn(n, cond_dims, generator=g).to(DEV) if cond_dims else None
    assert E.k1_sorted_eligible(tuple(latents.shape), d.LP)
    ws = eng.workspace(n)
    st = E.TrilerpState()
    ws.coords_t.fill_(float("nan")); ws.lat.fill_(float("nan"))
    E.k1_forward(eng, st, coords, idx, latents, cond, tfs, False, n)
    ref_c, ref_l = ws.coords_t.clone(), ws.lat.clone()
    ws.coords_t.fill_(float("nan")); ws.lat.fill_(float("nan"))
    E.k1_forward_sorted(eng, coords, idx, latents, cond, tfs, n)
    torch.cuda.synchronize()
    npad = E.round_up(n, 128)
    assert torch.equal(ws.coords_t[:npad].view(torch.int32), ref_c[:npad].view(torch.int32))
    bits = torch.int16 if precision == "bf16" else torch.int32
    assert torch.equal(ws.lat[:npad].view(bits), ref_l[:npad].view(bits))
    # no embedding table at all: coordinates pass through
    E.k1_forward_sorted(eng, coords, idx, latents, cond, None, n)
    assert torch.equal(ws.coords_t[:n], coords)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_validation_split_step_matches_oracle(precision):
    """split='val' (build_atlas.py:63,74,99-126,423-425): the MLP is frozen (dX-only backward), seg_weight = 0, the
    per-subject condition rows are learnable.  Gradients of latents / transformations / conditions == CPU oracle, the
    MLP does not move, the tables move as torch.optim.AdamW moves them."""
    from oracle import cinema_oracle as O
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    tol = 1e-4 if precision == "fp32" else 2e-2
    args = base_args([48, 3, 3, 3], 128, 3, [1, 3], [2, 10], 1, precision, tf_weight=0.0)
    torch.manual_seed(33)
    store = SubjectStore.synthetic(4, (24, 24, 24), 155.0 / 24, [155, 155, 155], 2, 10, seed=9, device=DEV, cond_dims=1)
    n = 600
    store.shuffle(torch.Generator(device=DEV).manual_seed(2), group_chunk=n)
    tr = HotPathTrainer(args, store, DEV, split="val", keep_grads=True)
    assert tr.cond_param is not None and not tr.train_mlp and tr.seg_weight == 0.0
    with torch.no_grad():
        tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(6)) - 0.5) * 0.05)
        tr.cond_param.copy_(torch.randn(tr.cond_param.shape, generator=torch.Generator().manual_seed(7)) * 0.05)
    lat0, tf0, c0 = (x.detach().cpu().clone() for x in (tr.latents, tr.tfs, tr.cond_param))
    flat0 = tr.eng.flat.detach().cpu().clone()
    coords, voxel, subj = store.coords[:n].cpu(), store.voxel[:n].cpu(), store.subj[:n].cpu()
    base = subj.long() * store.V + voxel
    target = torch.cat((store.intens.cpu()[base], store.labels.cpu()[base].float().unsqueeze(1)), 1)
    params = {k: v.clone().requires_grad_(True) for k, v in tr.eng.param_views(flat0).items()}
    lat_o, tf_o, c_o = (x.clone().requires_grad_(True) for x in (lat0, tf0, c0))
    cond_rows = c_o[subj.long()]
    cfg = dict(num_hidden_layers=3, omega=[30, 30], sr_dims=2, loss_metric="mse",
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=0.0, seg_weight=0.0)
    loss_o, _ = O.training_step(params, lat_o, tf_o, coords, target, subj, cond_rows, cfg)
    tr.step(0, n)
    got = tr.loss_dict()
    assert abs(got["total"] - float(loss_o["total"])) < tol * abs(float(loss_o["total"]))
    errs = {"glat": rel_err(tr.lat_grad.cpu().numpy(), lat_o.grad.numpy()),
            "gtf": rel_err(tr.tf_grad.cpu().numpy(), tf_o.grad.numpy()),
            "gcond": rel_err(tr.cond_param.grad.cpu().numpy(), c_o.grad.numpy())}
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, bad
    assert torch.equal(tr.eng.flat.detach().cpu(), flat0)           # frozen MLP
    ref_p = [torch.nn.Parameter(lat0.clone()), torch.nn.Parameter(tf0.clone()), torch.nn.Parameter(c0.clone())]
    ref_p[0].grad, ref_p[1].grad, ref_p[2].grad = tr.lat_grad.cpu(), tr.tf_grad.cpu(), tr.cond_param.grad.cpu()
    torch.optim.AdamW([{"params": [ref_p[0]], "lr": 5e-4, "weight_decay": 0.0},
                       {"params": [ref_p[1]], "lr": 7.5e-3, "weight_decay": 0.0},
                       {"params": [ref_p[2]], "lr": 5e-4, "weight_decay": 0.0}]).step()
    for x, r in ((tr.latents, ref_p[0]), (tr.tfs, ref_p[1]), (tr.cond_param, ref_p[2])):
        assert rel_err(x.detach().cpu().numpy(), r.detach().numpy()) < 1e-5


@pytest.mark.parametrize("split", ["train", "val"])
def test_trainer_step_cfgV_full_shape(split):
    """BASELINE config 5 (cfg-V) at its real shape: latent 256 x 3^3 + 1 condition (lat 257, LP 320), hidden 512, 5 hidden
    layers modulated 1/3/5, out 1 + 10, cond_scale 0.12, bf16, batch grouped by subject (the per-subject condition table is
    gathered inside K1; for split='val' its gradient is reduced per subject in cinema_cond_table_bwd).  Gradients == CPU
    oracle, max-norm and L2."""
    from oracle import cinema_oracle as O
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = base_args([256, 3, 3, 3], 512, 5, [1, 3, 5], [1, 10], 1, "bf16", tf_weight=0.0)
    torch.manual_seed(55)
    S, n = 12, 6000
    store = SubjectStore.synthetic(S, (24, 24, 24), 155.0 / 24, [155, 155, 155], 1, 10, seed=19, device=DEV, cond_dims=1,
                                   cond_scale=0.12)
    store.shuffle(torch.Generator(device=DEV).manual_seed(4), group_chunk=n)
    tr = HotPathTrainer(args, store, DEV, split=split, keep_grads=True)
    assert tr.eng.desc.lat == 257 and tr.eng.desc.LP == 320
    with torch.no_grad():
        tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(6)) - 0.5) * 0.05)
        if split == "val":
            tr.cond_param.copy_(torch.randn(tr.cond_param.shape, generator=torch.Generator().manual_seed(7)) * 0.05)
    cond_tab = (tr.cond_param if split == "val" else store.conditions).detach().cpu().clone()
    lat0, tf0, flat0 = tr.latents.detach().cpu().clone(), tr.tfs.detach().cpu().clone(), tr.eng.flat.detach().cpu().clone()
    coords, voxel, subj = store.coords[:n].cpu(), store.voxel[:n].cpu(), store.subj[:n].cpu()
    base = subj.long() * store.V + voxel
    target = torch.cat((store.intens.cpu()[base], store.labels.cpu()[base].float().unsqueeze(1)), 1)
    params = {k: v.clone().requires_grad_(True) for k, v in tr.eng.param_views(flat0).items()}
    lat_o, tf_o, c_o = (x.clone().requires_grad_(True) for x in (lat0, tf0, cond_tab))
    cfg = dict(num_hidden_layers=5, omega=[30, 30], sr_dims=1, loss_metric="mse",
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=0.0,
               seg_weight=0.0 if split == "val" else 1.0)
    loss_o, _ = O.training_step(params, lat_o, tf_o, coords, target, subj, c_o[subj.long()], cfg)
    tr.step(0, n)
    got = tr.loss_dict()
    assert abs(got["total"] - float(loss_o["total"])) < 2e-2 * abs(float(loss_o["total"]))
    errs = both_errs("glat", tr.lat_grad.cpu().numpy(), lat_o.grad.numpy())
    errs.update(both_errs("gtf", tr.tf_grad.cpu().numpy(), tf_o.grad.numpy()))
    if split == "val":
        errs.update(both_errs("gcond", tr.cond_param.grad.cpu().numpy(), c_o.grad.numpy()))
        assert torch.equal(tr.eng.flat.detach().cpu(), flat0)
    else:
        gviews = tr.eng.param_views(tr.eng.flat_grad.cpu())
        for k, p in params.items():
            errs.update(both_errs("g_" + k, gviews[k].numpy(), p.grad.numpy()))
    bad = failing(errs, 2e-2, L2_TOL["bf16"])
    assert not bad, bad


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_trainer_step_with_conv_modulator(precision):
    """cnn_kernel_size = 3 (models/inr_decoder.py:109-123): K1 interpolates conv(latents); gradients of the latents, the
    conv parameters and the MLP against the CPU oracle fed with torch's conv3d of the table, and the AdamW update of the
    conv parameters in the MLP's group."""
    import torch.nn.functional as F
    from oracle import cinema_oracle as O
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    tol = 1e-4 if precision == "fp32" else 2e-2
    args = base_args([48, 3, 3, 3], 128, 3, [1, 3], [2, 10], 0, precision, tf_weight=0.0)
    args["inr_decoder"]["cnn_kernel_size"] = 3
    torch.manual_seed(77)
    S, n = 5, 700
    store = SubjectStore.synthetic(S, (24, 24, 24), 155.0 / 24, [155, 155, 155], 2, 10, seed=7, device=DEV)
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=n)
    tr = HotPathTrainer(args, store, DEV, keep_grads=True)
    assert tr.conv is not None
    with torch.no_grad():
        tr.latents.mul_(10.0)                                   # N(0, 0.1): the conv output is not dominated by its bias
        tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(5)) - 0.5) * 0.05)
    lat0, tf0, flat0 = tr.latents.detach().cpu().clone(), tr.tfs.detach().cpu().clone(), tr.eng.flat.detach().cpu().clone()
    cw0, cb0 = tr.conv.weight.detach().cpu().clone(), tr.conv.bias.detach().cpu().clone()
    coords, voxel, subj = store.coords[:n].cpu(), store.voxel[:n].cpu(), store.subj[:n].cpu()
    base = subj.long() * store.V + voxel
    target = torch.cat((store.intens.cpu()[base], store.labels.cpu()[base].float().unsqueeze(1)), 1)
    params = {k: v.clone().requires_grad_(True) for k, v in tr.eng.param_views(flat0).items()}
    lat_o, tf_o, cw, cb = (x.clone().requires_grad_(True) for x in (lat0, tf0, cw0, cb0))
    cfg = dict(num_hidden_layers=3, omega=[30, 30], sr_dims=2, loss_metric="mse",
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=0.0)
    loss_o, _ = O.training_step(params, F.conv3d(lat_o, cw, cb, padding="same"), tf_o, coords, target, subj, None, cfg)
    tr.step(0, n)
    got = tr.loss_dict()
    assert abs(got["total"] - float(loss_o["total"])) < tol * abs(float(loss_o["total"]))
    errs = both_errs("glat", tr.lat_grad.cpu().numpy(), lat_o.grad.numpy())
    errs.update(both_errs("gtf", tr.tf_grad.cpu().numpy(), tf_o.grad.numpy()))
    errs.update(both_errs("g_conv_w", tr.conv.weight.grad.cpu().numpy(), cw.grad.numpy()))
    errs.update(both_errs("g_conv_b", tr.conv.bias.grad.cpu().numpy(), cb.grad.numpy()))
    gviews = tr.eng.param_views(tr.eng.flat_grad.cpu())
    for k, p in params.items():
        errs.update(both_errs("g_" + k, gviews[k].numpy(), p.grad.numpy()))
    bad = failing(errs, tol, L2_TOL[precision])
    assert not bad, bad
    ref = torch.nn.Parameter(cw0.clone())
    ref.grad = tr.conv.weight.grad.cpu()
    torch.optim.AdamW([{"params": [ref], "lr": 1e-4, "weight_decay": 0.0}]).step()
    assert rel_err(tr.conv.weight.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert not torch.equal(tr.conv.weight.detach().cpu(), cw0)


def test_epoch_walks_the_whole_list_and_steps_the_schedule():
    """HotPathTrainer.epoch (build_atlas.py:65-68, 455-463): ceil(m / n_samples) steps including the ragged last one,
    every chunk grouped by subject (the grouped K1 kernels take them), cosine schedule stepped once per epoch."""
    import math
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = base_args([32, 3, 3, 3], 128, 3, [1, 3], [2, 10], 0, "bf16")
    torch.manual_seed(8)
    store = SubjectStore.synthetic(3, (16, 16, 16), 155.0 / 16, [155, 155, 155], 2, 10, seed=12, device=DEV)
    tr = HotPathTrainer(args, store, DEV)
    m, n = len(store), 1000
    assert m % n != 0
    lat0, flat0 = tr.latents.detach().clone(), tr.eng.flat.detach().clone()
    steps = tr.epoch(n, generator=torch.Generator(device=DEV).manual_seed(3))
    assert steps == math.ceil(m / n)
    assert store.group_chunk == n
    for lo in range(0, m, n):
        chunk = store.subj[lo:lo + n]
        assert bool((chunk[1:] >= chunk[:-1]).all())
    assert tr.n_last == m - (steps - 1) * n                          # the ragged last batch
    loss = tr.loss_dict()
    assert np.isfinite(loss["total"]) and loss["total"] > 0
    assert not torch.equal(tr.latents.detach(), lat0) and not torch.equal(tr.eng.flat.detach(), flat0)
    assert bool(torch.isfinite(tr.eng.flat).all()) and bool(torch.isfinite(tr.latents).all())
    o = args["optimizer"]
    want = o["scheduler"]["eta_min"] + (o["lr_inr"] - o["scheduler"]["eta_min"]) * (1 + math.cos(math.pi / args["epochs"]["train"])) / 2
    lrs = [g["lr"] for g in tr.opt_local.param_groups]
    assert abs(lrs[-1] - want) < 1e-12 * max(1.0, want) + 1e-15, (lrs, want)
    assert tr.epoch(n, generator=torch.Generator(device=DEV).manual_seed(4), max_steps=2) == 2


def test_step_head_equals_the_separate_launches():
    """cinema_step_head (one launch) leaves the same bits as cinema_siren_pack + cinema_affine_trilerp_fwd_sorted +
    cinema_latent_to_node_major + cinema_step_prepare: packed weight copies, registered coordinates, interpolated latent
    rows, node-major shadow, cleared dcoords and the loss sums with the class-weight normaliser."""
    from cinema_amd import _lib, engine as E
    S, Cc, n, V = 9, 64, 3000 + 11, 4096
    d = E.make_desc(3, Cc, 12, 256, 4, [1, 3], 30, 30, _lib.PREC_BF16)
    eng = E.SirenEngine(d, DEV)
    assert eng.can_skip_saved()
    g = torch.Generator().manual_seed(17)
    eng.flat.copy_((torch.randn(eng.flat.shape, generator=g) * 0.05).to(DEV))
    latents = (torch.randn(S, Cc, 3, 3, 3, generator=g) * 0.1).to(DEV)
    tfs = ((torch.rand(S, 6, generator=g) - 0.5) * 0.2).to(DEV)
    coords = (torch.rand(n, 3, generator=g) * 2.2 - 1.1).to(DEV)
    idx = torch.randint(0, S, (n,), generator=g, dtype=torch.int32).sort().values.to(DEV)
    voxel = torch.randint(0, V, (n,), generator=g, dtype=torch.int64).to(DEV)
    labels = torch.randint(0, 10, (S * V,), generator=g, dtype=torch.uint8).to(DEV)
    class_w = (torch.rand(10, generator=g) + 0.5).to(DEV)
    ws = eng.workspace(n)
    ws.ensure_backward()
    npad = E.round_up(n, 128)
    st_a, st_b = E.TrilerpState(), E.TrilerpState()
    sums_a, sums_b = torch.full((4,), 7.0, device=DEV), torch.full((4,), 7.0, device=DEV)
    scratch = torch.zeros(_lib.PREPARE_SCRATCH, device=DEV)
    eng.packed.zero_()
    eng.pack(3 | 8)     # the head also leaves the row-major linear_lats copies of the node-space backward
    ws.coords_t.fill_(float("nan")); ws.lat.fill_(float("nan")); ws.dcoords.fill_(3.0)
    E.k1_forward_sorted(eng, coords, idx, latents, None, tfs, n)
    E.k1_shadow(eng, st_a, latents)
    E.step_prepare(ws.dcoords, sums_a, n, 10, class_w, voxel, idx, labels, V, scratch)
    ref = (eng.packed.clone(), ws.coords_t[:npad].clone(), ws.lat[:npad].clone(), ws.dcoords.clone())
    eng.packed.zero_()
    ws.coords_t.fill_(float("nan")); ws.lat.fill_(float("nan")); ws.dcoords.fill_(3.0)
    E.step_head(eng, st_b, coords, idx, latents, None, tfs, n,
                prepare=dict(dcoords=ws.dcoords, sums=sums_b, n_cls=10, class_w=class_w, voxel=voxel, labels=labels, V=V,
                             scratch=scratch))
    torch.cuda.synchronize()
    assert torch.equal(eng.packed.view(torch.int16), ref[0].view(torch.int16))
    assert torch.equal(ws.coords_t[:npad].view(torch.int32), ref[1].view(torch.int32))
    assert torch.equal(ws.lat[:npad].view(torch.int16), ref[2].view(torch.int16))
    assert torch.equal(ws.dcoords, ref[3]) and float(ws.dcoords.abs().max()) == 0.0
    assert torch.equal(st_a.shadow, st_b.shadow)
    assert torch.equal(sums_a, sums_b) and float(sums_a[2]) > 0 and float(sums_a[0]) == 0.0
    assert float(scratch[256:].abs().max()) == 0.0          # the arrival counter was reset


def test_checkpoint_round_trip(tmp_path):
    """save_state / load_checkpoint (build_atlas.py:344-370): the reference's keys without the pandas frames, readable
    with weights_only=True; a fresh trainer that loads it holds the same decoder and tables, starts with a fresh
    optimiser, and its next step equals the next step of a trainer built from the same state by hand."""
    import os
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = base_args([32, 3, 3, 3], 128, 3, [1, 3], [2, 10], 0, "bf16")
    args.update(output_dir=str(tmp_path), save_model=True)
    torch.manual_seed(2)
    store = SubjectStore.synthetic(4, (24, 24, 24), 155.0 / 24, [155, 155, 155], 2, 10, seed=3, device=DEV)
    tr = HotPathTrainer(args, store, DEV)
    store.shuffle(torch.Generator(device=DEV).manual_seed(4), group_chunk=1500)
    for i in range(3):
        tr.step(i * 1500, (i + 1) * 1500)
    path = tr.save_state(7)
    assert os.path.basename(path) == "checkpoint_epoch_7.pth"
    chk = torch.load(path, weights_only=True)
    assert set(chk) == {"epoch", "latents", "transformations", "inr_decoder", "args"} and chk["epoch"] == 7
    assert list(chk["inr_decoder"])[0] == "sr_net.final_linear.weight"          # reference's state_dict order
    torch.manual_seed(99)
    tr2 = HotPathTrainer(args, store, DEV)                                       # different initialisation
    assert not torch.equal(tr2.eng.flat, tr.eng.flat)
    tr2.step(0, 1500)                                                            # gives its optimiser a history to forget
    tr2.load_checkpoint(str(tmp_path), 7)
    assert torch.equal(tr2.eng.flat, tr.eng.flat) and torch.equal(tr2.latents, tr.latents) and torch.equal(tr2.tfs, tr.tfs)
    assert len(tr2.opt_local.state) == 0
    # reference semantics after a load: same parameters, NEW optimiser -> compare with a trainer set up that way by hand
    tr3 = HotPathTrainer(args, store, DEV)
    with torch.no_grad():
        tr3.eng.flat.copy_(tr.eng.flat); tr3.latents.copy_(tr.latents); tr3.tfs.copy_(tr.tfs)
    tr2.step(4500, 6000)
    tr3.step(4500, 6000)
    for name, x, y in (("flat", tr3.eng.flat, tr2.eng.flat), ("lat", tr3.latents, tr2.latents), ("tfs", tr3.tfs, tr2.tfs)):
        assert rel_err(y.detach().cpu().numpy(), x.detach().cpu().numpy()) < 1e-4, name
    with pytest.raises(FileNotFoundError):
        tr2.load_checkpoint(str(tmp_path), 8)
    args["save_model"] = False
    assert tr.save_state(9) is None and not os.path.exists(os.path.join(str(tmp_path), "checkpoint_epoch_9.pth"))
    tr.re_init_latents()
    assert float(tr.tfs.detach().abs().max()) == 0.0 and 0.005 < float(tr.latents.detach().std()) < 0.02


def test_reinit_and_checkpoint_load_start_afresh(tmp_path):
    """re_init_latents zeroes the gradient of every parameter group (optimizer.zero_grad(), build_atlas.py:427-430) and
    load_checkpoint restarts optimiser state AND the cosine schedule (the reference builds both anew after a load)."""
    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = base_args([16, 3, 3, 3], 64, 2, [1, 2], [1, 4], 1, "bf16")
    args["optimizer"]["scheduler"] = {"type": "cosine", "eta_min": 1e-6}
    args["epochs"] = {"train": 5, "val": 5}
    args["output_dir"] = str(tmp_path)
    store = SubjectStore.synthetic(3, (12, 12, 12), 155.0 / 12, [155, 155, 155], 1, 4, seed=2, device=DEV, cond_dims=1)
    tr = HotPathTrainer(args, store, DEV, keep_grads=True, split="val")
    lr0 = [g["lr"] for g in tr.opt_local.param_groups]
    tr.epoch(400, torch.Generator(device=DEV).manual_seed(1), max_steps=2)
    tr.epoch(400, torch.Generator(device=DEV).manual_seed(2), max_steps=2)
    assert [g["lr"] for g in tr.opt_local.param_groups] != lr0
    assert float(tr.cond_param.grad.abs().max()) > 0 and float(tr.lat_grad.abs().max()) > 0
    tr.re_init_latents()
    assert float(tr.cond_param.grad.abs().max()) == 0 and float(tr.lat_grad.abs().max()) == 0 and float(tr.tf_grad.abs().max()) == 0
    assert tr.eng.grads_clean and float(tr.eng.flat_grad.abs().max()) == 0
    path = tr.save_state(1)
    assert path is not None
    tr.load_checkpoint(str(tmp_path), 1)
    assert [g["lr"] for g in tr.opt_local.param_groups] == lr0
    assert all(s.last_epoch == 0 for s in tr.schedulers) and len(tr.opt_local.state) == 0
    tr.epoch(400, torch.Generator(device=DEV).manual_seed(3), max_steps=1)      # and it still trains
    assert [g["lr"] for g in tr.opt_local.param_groups] != lr0
