"""Developer timing of k_node_grad alone at the bench shape (random ds / dt rows), with ablation bits:
1 = no exchange / d coords, 2 = no dU products, 4 = no dwn products."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cinema_amd import _lib, engine as E  # noqa: E402

dev = torch.device("cuda:0")
N, S = int(os.environ.get("N", "25000")), int(os.environ.get("S", "64"))
d = E.make_desc(3, 256, 12, 512, 6, [1, 3, 5], 30, 30, _lib.PREC_BF16)
eng = E.SirenEngine(d, dev)
g = torch.Generator().manual_seed(5)
eng.flat.copy_((torch.randn(eng.flat.shape, generator=g) * 0.05).to(dev))
latents = (torch.randn(S, 256, 3, 3, 3, generator=g) * 0.1).to(dev)
tfs = ((torch.rand(S, 6, generator=g) - 0.5) * 0.1).to(dev)
coords = (torch.rand(N, 3, generator=g) * 2.0 - 1.0).to(dev)
idx = torch.randint(0, S, (N,), generator=g, dtype=torch.int32).sort().values.to(dev)
ws = eng.workspace(N)
ws.ensure_backward()
ws.gbuf.view(torch.int16)[: ws.gbuf.numel() // 2].copy_(
    (torch.randn(ws.gbuf.numel() // 2, generator=g) * 1e-3).to(torch.bfloat16).view(torch.int16).to(dev))
eng.pack(3 | 8)
E.k1_forward_sorted(eng, coords, idx, latents, None, tfs, N)
ng = E.NodeGrad()
E.nodegrad_prepare(eng, ng, coords, idx, latents, None, tfs, N)
npad = E.round_up(N, 128)
lib = eng.lib
lib.cinema_debug_node_grad.argtypes = [C.POINTER(_lib.SirenDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]


def run(dbg):
    _lib.check(lib.cinema_debug_node_grad(C.byref(d), _lib.ptr(ng.work), _lib.ptr(ws.gbuf), _lib.ptr(ws.coords_t), N, npad, S, 3, 3, 3,
                                          dbg, _lib.ptr(eng.packed), _lib.ptr(eng.flat), eng.stream()), "debug_node_grad")


if os.environ.get("PREP"):
    for _ in range(3):
        E.nodegrad_prepare(eng, ng, coords, idx, latents, None, tfs, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        E.nodegrad_prepare(eng, ng, coords, idx, latents, None, tfs, N)
    e1.record()
    torch.cuda.synchronize()
    print(f"k_ng_prepare skip={os.environ.get('CINEMA_NG_PREP_SKIP', '0')}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
    sys.exit(0)

for dbg in [int(x) for x in os.environ.get("DBG", "0,1,2,4,6,7").split(",")]:
    for _ in range(3):
        run(dbg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run(dbg)
    e1.record()
    torch.cuda.synchronize()
    print(f"k_node_grad dbg={dbg}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
