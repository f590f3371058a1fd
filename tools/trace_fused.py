"""Phase timeline of one wave of the fused SIREN kernels (developer build: python -c "from cinema_amd import build;
build.build(trace=True)"; run with CINEMA_HIP_LIB=cinema_amd/libcinema_hip_trace.so; MODE=fwd|bwd)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cinema_amd import _lib  # noqa: E402
from cinema_amd.models.siren import Siren  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "25000"))
mode = os.environ.get("MODE", "fwd")
net = Siren(3, 256, 12, 512, 5, 30, 30, True, [1, 3, 5], precision=_lib.PREC_BF16).to(dev)
eng = net.hot_state(dev).engine
ws = eng.workspace(N)
ws.ensure_backward()
ws.coords_t.uniform_(-1, 1)
ws.lat.normal_(0, 0.05)
ws.dout.normal_(0, 1e-3)
eng.pack()
eng.forward(N)
lib = _lib.load()
fn = lib.cinema_debug_fused_trace
fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong)]
names = {2: "s-seg", 3: "z-seg" if mode == "fwd" else "dy GEMM", 4: "t-seg", 5: "barrier A", 6: "epilogue", 7: "barrier B"}


def run():
    if mode == "fwd":
        eng.forward(N)
    else:
        eng.backward(N, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN)


if mode == "wide":
    fw = lib.cinema_debug_wide_trace
    fw.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    for blk, wave in ((0, 0), (0, 7), (int(os.environ.get("BLK", "200")), 3)):
        for _ in range(3):
            eng.backward(N, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN)
        torch.cuda.synchronize()
        fw(blk, wave, None)
        eng.backward(N, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN)
        torch.cuda.synchronize()
        buf = (C.c_longlong * 128)()
        fw(-1, 0, buf)
        t = list(buf)
        print(f"--- wide backward block {blk} wave {wave}: total {(t[127] - t[125]) / 100:.1f} us; prologue {(t[126] - t[125]) / 100:.2f}")
        prev = t[126]
        tot = [0.0] * 6
        for l in range(5, -1, -1):
            v = [t[8 * l + k] for k in range(6)]
            names_w = ["GEMM 0", "epilogue 0", "GEMM 1", "barrier A", "epilogue 1", "barrier B"]
            parts = []
            for k in range(6):
                parts.append(f"{names_w[k]} {(v[k] - prev) / 100:.2f}")
                tot[k] += (v[k] - prev) / 100
                prev = v[k]
            print(f"   layer {l}: " + ", ".join(parts))
        print("   sums: " + ", ".join(f"{n} {x:.1f}" for n, x in zip(names_w, tot)) + f"; tail {(t[127] - prev) / 100:.2f}")
    sys.exit(0)

for blk, wave in ((0, 0), (0, 7), (int(os.environ.get("BLK", "300")), 3)):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    fn(blk, wave, None)
    run()
    torch.cuda.synchronize()
    buf = (C.c_longlong * 128)()
    fn(-1, 0, buf)
    t = list(buf)
    t0 = t[0]
    print(f"--- {mode} block {blk} wave {wave}: total {(t[127] - t0) / 100:.1f} us (100 MHz clock)")
    print(f"   prologue: {(t[1] - t0) / 100:.2f} us")
    prev = t[1]
    layers = range(6) if mode == "fwd" else range(5, -1, -1)
    for l in layers:
        parts = []
        for k in range(2, 8):
            v = t[k + 8 * l]
            if v < prev:      # phase not executed in this layer (stale value)
                continue
            parts.append(f"{names[k]} {(v - prev) / 100:.2f}")
            prev = v
        print(f"   layer {l}: " + ", ".join(parts))
    print(f"   tail: {(t[127] - prev) / 100:.2f} us")
