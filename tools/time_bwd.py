"""Developer timing: the backward chain alone at the bench shape (N = 25 000, hidden 512, cfg-D)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cinema_amd import _lib  # noqa: E402
from cinema_amd.models.siren import Siren  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "25000"))
net = Siren(3, 256, 12, 512, 5, 30, 30, True, [1, 3, 5], precision=_lib.PREC_BF16).to(dev)
eng = net.hot_state(dev).engine
ws = eng.workspace(N)
ws.ensure_backward()
ws.coords_t.uniform_(-1, 1)
ws.lat.normal_(0, 0.05)
ws.dout.normal_(0, 1e-3)
eng.pack()
eng.forward(N)


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


timeit("forward", lambda: eng.forward(N))
timeit("backward chain", lambda: eng.backward(N, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN))
timeit("backward chain, frozen weights", lambda: eng.backward(N, ws.dout, weight_grads=False, phases=_lib.BWD_PHASE_CHAIN))
