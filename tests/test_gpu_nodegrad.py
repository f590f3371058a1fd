"""GPU tests of the node-space backward (include/cinema_hip.h, "node gradients"): the same trainer step with the latent
side of the backward reduced over S*32 node rows (default) and over the N sample rows (CINEMA_NODE_GRAD=0: SIREN backward
+ K1 backward), on the same parameters and samples.  Both are the same function in different rounding; the oracle
comparisons of tests/test_gpu_trainer.py run on the default."""
import pytest
import torch

from conftest import both_errs, failing
from test_gpu_trainer import DEV, base_args

pytestmark = pytest.mark.gpu


def _one_step(monkeypatch, node_grad, args, store, n, split="train", split_backward=False):
    from cinema_amd.trainer import HotPathTrainer
    from cinema_amd import engine as E
    monkeypatch.setenv("CINEMA_NODE_GRAD", "1" if node_grad else "0")
    torch.manual_seed(21)
    tr = HotPathTrainer(args, store, DEV, keep_grads=True, split=split)
    with torch.no_grad():
        tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(5)) - 0.5) * 0.05)
        # make the modulation (and with it the latent-side gradients) matter
        for k, v in tr.eng.param_views().items():
            if "linear_lats.weight" in k:
                v.mul_(3.0)
    assert E.nodegrad_eligible(tr.eng, tuple(tr.latents.shape)) == node_grad
    tr.split_backward = split_backward
    tr.step(0, n)
    torch.cuda.synchronize()
    out = {"glat": tr.lat_grad.clone(), "gtf": tr.tf_grad.clone(), "gflat": tr.eng.flat_grad.clone()}
    if tr.cond_param is not None:
        out["gcond"] = tr.cond_param.grad.clone()
    views = tr.eng.param_views(tr.eng.flat_grad.clone())
    return out, views


@pytest.mark.parametrize("shape", ["small", "ragged", "bench"])
def test_node_gradients_equal_per_sample_backward(shape, monkeypatch):
    from cinema_amd.trainer import SubjectStore
    if shape == "bench":
        S, side, latent_dim, hidden, layers, mod, n = 64, 28, [256, 3, 3, 3], 512, 5, [1, 3, 5], 25000
    elif shape == "small":
        S, side, latent_dim, hidden, layers, mod, n = 5, 24, [48, 3, 3, 3], 128, 3, [1, 3], 700
    else:       # a step that is not a multiple of anything, subjects of very different sizes, layer 0 modulated
        S, side, latent_dim, hidden, layers, mod, n = 7, 20, [32, 3, 3, 3], 192, 2, [0, 2], 1003
    args = base_args(latent_dim, hidden, layers, mod, [2, 10], 0, "bf16")
    store = SubjectStore.synthetic(S, (side, side, side), 155.0 / side, [155, 155, 155], 2, 10, seed=7, device=DEV)
    if shape == "ragged":       # drop most samples of two subjects and every sample of another one
        keep = ~(((store.subj == 1) | (store.subj == 4)) & (torch.arange(len(store), device=DEV) % 37 != 0)) & (store.subj != 2)
        store.coords, store.voxel, store.subj = store.coords[keep], store.voxel[keep], store.subj[keep]
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=n)
    ref, ref_v = _one_step(monkeypatch, False, args, store, n)
    got, got_v = _one_step(monkeypatch, True, args, store, n)
    errs = {}
    for k in ("glat", "gtf"):
        errs.update(both_errs(k, got[k].cpu().numpy(), ref[k].cpu().numpy()))
    for k in ref_v:
        errs.update(both_errs("g_" + k, got_v[k].cpu().numpy(), ref_v[k].cpu().numpy()))
    bad = failing(errs, 1e-2, 6e-3)
    assert not bad, bad


def test_node_gradients_validation_split_with_conditions(monkeypatch):
    """split='val' (frozen MLP, learnable per-subject conditions, lat = C + 1): table, embedding and condition gradients."""
    from cinema_amd.trainer import SubjectStore
    S, side, n = 6, 20, 900
    args = base_args([63, 3, 3, 3], 128, 3, [1, 3], [1, 10], 1, "bf16")
    store = SubjectStore.synthetic(S, (side, side, side), 155.0 / side, [155, 155, 155], 1, 10, seed=9, device=DEV, cond_dims=1,
                                   cond_scale=0.12)
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=n)
    ref, _ = _one_step(monkeypatch, False, args, store, n, split="val")
    got, _ = _one_step(monkeypatch, True, args, store, n, split="val")
    errs = {}
    for k in ("glat", "gtf", "gcond"):
        errs.update(both_errs(k, got[k].cpu().numpy(), ref[k].cpu().numpy()))
    assert float(got["gflat"].abs().max()) == 0.0
    bad = failing(errs, 1e-2, 6e-3)
    assert not bad, bad


@pytest.mark.parametrize("node_grad", [True, False])
def test_two_call_backward_of_a_rank_of_several_equals_the_single_call(node_grad, monkeypatch):
    """world > 1 calls the backward twice (BWD_WEIGHTS_FIRST, all-reduce starts, BWD_DLAT_ONLY): same gradients as one call."""
    from cinema_amd.trainer import SubjectStore
    S, side, n = 9, 22, 2900
    args = base_args([64, 3, 3, 3], 256, 3, [1, 3], [2, 10], 0, "bf16")
    store = SubjectStore.synthetic(S, (side, side, side), 155.0 / side, [155, 155, 155], 2, 10, seed=11, device=DEV)
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=n)
    ref, ref_v = _one_step(monkeypatch, node_grad, args, store, n)
    got, got_v = _one_step(monkeypatch, node_grad, args, store, n, split_backward=True)
    errs = {}
    for k in ("glat", "gtf"):
        errs.update(both_errs(k, got[k].cpu().numpy(), ref[k].cpu().numpy()))
    for k in ref_v:
        errs.update(both_errs("g_" + k, got_v[k].cpu().numpy(), ref_v[k].cpu().numpy()))
    bad = failing(errs, 1e-5, 1e-5)        # the same kernels on the same operands; float atomics in the bias sums
    assert not bad, bad


@pytest.mark.parametrize("cond_dims", [0, 1])
def test_node_tables_riding_in_head_and_forward_equal_the_stand_alone_preparation(cond_dims):
    """cinema_step_head(node_work) + cinema_siren_forward_train(node_work) leave the same bits in the work buffer as
    cinema_nodegrad_prepare on its own (node rows, centred rows, interpolation weights, axes, subject offsets, projections),
    and the forward's outputs are those of cinema_siren_forward."""
    from cinema_amd import engine as E
    S, C, n, hidden = 7, 60, 1500, 256          # lat = 60 (+ 1 condition): LP = 64 with zero padding
    d = E.make_desc(3, C + cond_dims, 12, hidden, 4, [1, 3], 30, 30, E._lib.PREC_BF16)
    eng = E.SirenEngine(d, torch.device(DEV))
    g = torch.Generator().manual_seed(3)
    eng.flat.copy_((torch.randn(eng.flat.shape, generator=g) * 0.05).to(DEV))
    latents = (torch.randn(S, C, 3, 3, 3, generator=g) * 0.1).to(DEV)
    cond = (torch.randn(S, cond_dims, generator=g) * 0.1).to(DEV) if cond_dims else None
    tfs = ((torch.rand(S, 9, generator=g) - 0.5) * 0.1).to(DEV)
    coords = (torch.rand(n, 3, generator=g) * 2.2 - 1.1).to(DEV)          # some samples outside the grid (clamped axes)
    idx = torch.randint(0, S, (n,), generator=g, dtype=torch.int32).sort().values.to(DEV)
    assert E.nodegrad_eligible(eng, tuple(latents.shape))
    eng.workspace(n)
    st = E.TrilerpState()
    a, b = E.NodeGrad(), E.NodeGrad()
    E.step_head(eng, st, coords, idx, latents, cond, tfs, n, shadow=False, cond_by_subject=True, ng=a)
    out_a = eng.forward(n, ng=a, latents_shape=tuple(latents.shape)).clone()
    saved_a = eng.ws.saved.clone()
    E.nodegrad_prepare(eng, b, coords, idx, latents, cond, tfs, n)          # rows + weights + projections, two launches
    out_b = eng.forward(n).clone()
    torch.cuda.synchronize()
    assert a.work is not b.work and torch.equal(a.work, b.work)
    assert torch.equal(out_a, out_b) and torch.equal(saved_a, eng.ws.saved)
    assert int(a.work.count_nonzero()) > 0


@pytest.mark.parametrize("case", ["one_subject_few_rows", "grid_1x1x1", "grid_2x3x4", "no_tf", "single_rows"])
def test_node_gradients_odd_shapes(case, monkeypatch):
    """Corners of the node-space backward against the per-sample one: a single subject with fewer rows than one tile, a
    1-node and a non-cubic 24-node grid (the batch then goes through the generic K1 forward and the stand-alone
    preparation), no transformation table (tf_dim = 0), subjects with a single sample each."""
    from cinema_amd.trainer import SubjectStore
    S, side, latent_dim, hidden, layers, mod, n, tf_dim = 4, 16, [32, 3, 3, 3], 128, 2, [0, 2], 333, 6
    if case == "one_subject_few_rows":
        S, n = 1, 5
    elif case == "grid_1x1x1":
        latent_dim = [32, 1, 1, 1]
    elif case == "grid_2x3x4":
        latent_dim = [32, 2, 3, 4]
    elif case == "no_tf":
        tf_dim = 0
    args = base_args(latent_dim, hidden, layers, mod, [2, 10], 0, "bf16", tf_dim=tf_dim)
    store = SubjectStore.synthetic(S, (side, side, side), 155.0 / side, [155, 155, 155], 2, 10, seed=13, device=DEV)
    if case == "single_rows":       # one sample per subject
        first = torch.zeros(len(store), dtype=torch.bool, device=DEV)
        for s in range(S):
            first[int((store.subj == s).nonzero()[0])] = True
        store.coords, store.voxel, store.subj = store.coords[first], store.voxel[first], store.subj[first]
        n = S
    store.shuffle(torch.Generator(device=DEV).manual_seed(1), group_chunk=n)

    def one(node_grad):
        from cinema_amd.trainer import HotPathTrainer
        monkeypatch.setenv("CINEMA_NODE_GRAD", "1" if node_grad else "0")
        torch.manual_seed(21)
        tr = HotPathTrainer(args, store, DEV, keep_grads=True)
        with torch.no_grad():
            if tf_dim:
                tr.tfs.copy_((torch.rand(tr.tfs.shape, generator=torch.Generator().manual_seed(5)) - 0.5) * 0.05)
            for k, v in tr.eng.param_views().items():
                if "linear_lats.weight" in k:
                    v.mul_(3.0)
        tr.step(0, n)
        torch.cuda.synchronize()
        return tr.lat_grad.clone(), tr.tf_grad.clone(), tr.eng.param_views(tr.eng.flat_grad.clone())

    ref, got = one(False), one(True)
    errs = both_errs("glat", got[0].cpu().numpy(), ref[0].cpu().numpy())
    if tf_dim:
        errs.update(both_errs("gtf", got[1].cpu().numpy(), ref[1].cpu().numpy()))
    for k in ref[2]:
        errs.update(both_errs("g_" + k, got[2][k].cpu().numpy(), ref[2][k].cpu().numpy()))
    bad = failing(errs, 1e-2, 6e-3)
    assert not bad, bad
