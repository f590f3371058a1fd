"""Timing probe (not a test): forward stack with debug switches."""
import os, sys, torch
sys.path.insert(0, ".")
from cinema_amd import _lib, engine as E
from cinema_amd.models.siren import Siren
dev = torch.device("cuda:0")
N = 25000
net = Siren(3, 256, 12, 512, 5, 30, 30, True, [1, 3, 5], precision=_lib.PREC_BF16).to(dev)
st = net.hot_state(dev); eng = st.engine
ws = eng.workspace(N)
ws.coords_t.uniform_(-1, 1); ws.lat.normal_(0, 0.05)
eng.pack()
def run(tag):
    for _ in range(3): eng.forward(N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.forward(N)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:40s} {e0.elapsed_time(e1)/20*1e3:8.1f} us per forward stack")
for dbg, tag in ((0, "full"), (1, "no mainloop (first GEMM of each layer)"), (2, "no sincos"), (4, "no y/c stores"), (6, "no sincos, no stores"), (7, "nothing but mod GEMMs")):
    os.environ["CINEMA_DBG"] = str(dbg)
    run(tag)
