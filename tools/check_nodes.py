"""Developer check: nodes path (node tables + interpolating chain kernels) against the per-sample path on the same inputs,
at the bench shape, with timings."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cinema_amd import _lib, engine as E  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "25000"))
S = int(os.environ.get("S", "64"))
HID = int(os.environ.get("HID", "512"))
C_ = int(os.environ.get("C", "256"))
MOD = [int(x) for x in os.environ.get("MOD", "1,3,5").split(",")]
LAYERS = int(os.environ.get("LAYERS", "5"))
d = E.make_desc(3, C_, 12, HID, LAYERS + 1, MOD, 30, 30, _lib.PREC_BF16)
eng = E.SirenEngine(d, dev)
g = torch.Generator().manual_seed(5)
from cinema_amd.models.siren import Siren  # noqa: E402
net = Siren(3, C_, 12, HID, LAYERS, 30, 30, True, MOD, precision=_lib.PREC_BF16)
sd = net.state_dict()
views = eng.param_views()
for k, v in views.items():
    v.copy_(sd[k].to(dev))
for k in views:
    if "linear_lats" in k:
        views[k].mul_(3.0)      # make the modulation matter
latents = (torch.randn(S, C_, 3, 3, 3, generator=g) * 0.1).to(dev)
tfs = ((torch.rand(S, 6, generator=g) - 0.5) * 0.1).to(dev)
coords = (torch.rand(N, 3, generator=g) * 2.0 - 1.0).to(dev)
idx = torch.randint(0, S, (N,), generator=g, dtype=torch.int32).sort().values.to(dev)
if os.environ.get("SHUFFLE") == "1":
    idx = idx[torch.randperm(N, generator=g).to(dev)].contiguous()
ws = eng.workspace(N)
ws.ensure_backward()
npad = E.round_up(N, 128)
ws.dout.zero_()
ws.dout[:N].copy_((torch.randn(N, 12, generator=g) * 1e-3).to(dev))


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm().clamp_min(1e-30))


# ---- per-sample path
eng.pack(3)
E.k1_forward_sorted(eng, coords, idx, latents, None, tfs, N)
ct_ref = ws.coords_t[:npad].clone()
out_ref = eng.forward(N)[:N].clone()
saved_ref = ws.saved.clone()
st = E.TrilerpState()
E.k1_shadow(eng, st, latents)
eng.backward(N, ws.dout, weight_grads=True)
g_ref, dlat_ref, dco_ref = eng.flat_grad.clone(), ws.dlat[:N].clone(), ws.dcoords[:N].clone()
gb_ref = ws.gbuf.clone()
torch.cuda.synchronize()

# ---- nodes path
nt = E.NodeTables()
eng.pack(4 | 2)
E.node_weights(eng, nt, coords, idx, tuple(latents.shape), tfs, N)
E.node_tables(eng, nt, latents, None)
torch.cuda.synchronize()
print("coords_t bit-identical:", torch.equal(ws.coords_t[:npad].view(torch.int32), ct_ref.view(torch.int32)))
for round_s in (True, False):
    ws.saved.fill_(float("nan"))
    out = eng.forward_nodes(N, nt, idx, keep=True, round_s=round_s)[:N].clone()
    torch.cuda.synchronize()
    print(f"round_s={round_s}: out max-rel %.3e l2-rel %.3e" % rel(out, out_ref))
    for l in range(d.n_layers):
        for nm, slot in (("y", d.sv_y[l]), ("c", d.sv_c[l]), ("z", d.sv_z[l]), ("s", d.sv_s[l])):
            if slot >= 0 and (nm != "s" or round_s):
                print(f"   layer {l} {nm}: max-rel %.3e l2-rel %.3e" % rel(ws.saved[slot, :N], saved_ref[slot, :N]))
    eng.flat_grad.zero_()
    eng.grads_clean = False
    eng.backward(N, ws.dout, weight_grads=True, nodes=(nt, idx), round_s=round_s)
    torch.cuda.synchronize()
    print("   flat_grad max-rel %.3e l2-rel %.3e" % rel(eng.flat_grad, g_ref))
    for key, v in eng.param_views(eng.flat_grad).items():
        print(f"     {key:28s} max-rel %.3e l2 %.3e" % rel(v, eng.param_views(g_ref)[key]))
    print("   dlat  max-rel %.3e l2-rel %.3e" % rel(ws.dlat[:N], dlat_ref))
    print("   dcoords max-rel %.3e l2-rel %.3e" % rel(ws.dcoords[:N], dco_ref))
out_inf = eng.forward_nodes(N, nt, idx, keep=False)[:N].clone()
print("inference (nothing kept) equals keep run:", torch.equal(out_inf, out))


def timeit(tag, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:52s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)


timeit("per-sample: pack(3)", lambda: eng.pack(3))
timeit("per-sample: K1 forward (grouped)", lambda: E.k1_forward_sorted(eng, coords, idx, latents, None, tfs, N))
timeit("per-sample: forward", lambda: eng.forward(N))
timeit("per-sample: forward, nothing kept", lambda: eng.forward(N, keep=False))
timeit("per-sample: backward chain", lambda: eng.backward(N, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN))
timeit("nodes: pack(4|2)", lambda: eng.pack(6))
timeit("nodes: weights", lambda: E.node_weights(eng, nt, coords, idx, tuple(latents.shape), tfs, N))
timeit("nodes: tables (rows + projection)", lambda: E.node_tables(eng, nt, latents, None))
timeit("nodes: forward", lambda: eng.forward_nodes(N, nt, idx))
timeit("nodes: forward, nothing kept", lambda: eng.forward_nodes(N, nt, idx, keep=False))
timeit("nodes: backward chain", lambda: eng.backward(N, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN, nodes=(nt, idx)))
timeit("per-sample: backward chain, frozen weights", lambda: eng.backward(N, ws.dout, weight_grads=False, phases=_lib.BWD_PHASE_CHAIN))
timeit("nodes: backward chain, frozen weights", lambda: eng.backward(N, ws.dout, weight_grads=False, phases=_lib.BWD_PHASE_CHAIN, nodes=(nt, idx)))
