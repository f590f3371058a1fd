"""Phase timeline of one wave of the fused SIREN forward (developer build: python -c "from cinema_amd import build;
build.build(trace=True)"; run with CINEMA_HIP_LIB=cinema_amd/libcinema_hip_trace.so)."""
import ctypes as C
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
lib = _lib.load()
fn = lib.cinema_debug_fused_trace
fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong)]
names = {2: "s-seg", 3: "z-seg", 4: "t-seg", 5: "barrier A", 6: "epilogue", 7: "barrier B"}
for blk, wave in ((0, 0), (0, 7), (int(os.environ.get("BLK", "300")), 3)):
    for _ in range(3):
        eng.forward(N)
    fn(blk, wave, None)
    eng.forward(N)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 128)()
    fn(-1, 0, buf)
    t = list(buf)
    t0 = t[0]
    print(f"--- block {blk} wave {wave}: total {(t[127] - t0) / 100:.1f} us (100 MHz clock)")
    print(f"   prologue (zero, lat, bias, ring init): {(t[1] - t0) / 100:.2f} us")
    prev = t[1]
    for l in range(6):
        parts = []
        for k in range(2, 8):
            v = t[k + 8 * l]
            if v < prev:      # phase not executed in this layer (stale value)
                continue
            parts.append(f"{names[k]} {(v - prev) / 100:.2f}")
            prev = v
        print(f"   layer {l}: " + ", ".join(parts))
    print(f"   final linear: {(t[127] - prev) / 100:.2f} us")
