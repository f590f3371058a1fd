"""Profiling target: a few SIREN forward stacks at the bench shape (use under rocprofv3)."""
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
ws.coords_t.uniform_(-1, 1)
ws.lat.normal_(0, 0.05)
eng.pack()
for _ in range(int(os.environ.get("REPS", "5"))):
    eng.forward(N)
torch.cuda.synchronize()
