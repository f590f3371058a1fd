"""Timing probe: K1 backward (coordinate gradient, latent-table reduction, tf-table gradient) at the bench shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cinema_amd import _lib, engine as E  # noqa: E402

dev = torch.device("cuda:0")
N, S = 25000, 64
d = E.make_desc(3, 256, 12, 512, 6, [1, 3, 5], 30, 30, _lib.PREC_BF16)
eng = E.SirenEngine(d, dev)
g = torch.Generator().manual_seed(5)
latents = (torch.randn(S, 256, 3, 3, 3, generator=g) * 0.1).to(dev)
tfs = ((torch.rand(S, 6, generator=g) - 0.5) * 0.1).to(dev)
coords = (torch.rand(N, 3, generator=g) * 2.0 - 1.0).to(dev)
idx = torch.randint(0, S, (N,), generator=g, dtype=torch.int32).sort().values.to(dev)
ws = eng.workspace(N)
ws.ensure_backward()
ws.dlat.normal_(0, 1e-3)
k1 = E.TrilerpState()
eng.pack()
E.k1_forward_sorted(eng, coords, idx, latents, None, tfs, N)
E.k1_shadow(eng, k1, latents)
lat_grad, tf_grad = torch.zeros_like(latents), torch.zeros_like(tfs)


def timeit(tag, fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:60s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)


shape = tuple(latents.shape)
timeit("coord only (no table grads)", lambda: E.k1_backward(eng, k1, coords, idx, shape, 0, None, False, N, want_latents=False, grouped=True))
timeit("coord + latent reduce", lambda: E.k1_backward(eng, k1, coords, idx, shape, 0, None, False, N, want_latents=True, dlat_out=lat_grad, grouped=True))
timeit("coord + latent reduce + tf table", lambda: E.k1_backward(eng, k1, coords, idx, shape, 0, tfs, False, N, want_latents=True, dtf_out=tf_grad, dlat_out=lat_grad, grouped=True))
