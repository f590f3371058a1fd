"""Developer check: the wide backward chain (k_fused_bwdw) against k_fused_bwd on the same inputs, over a sweep of shapes.
Run without arguments: starts itself twice per shape (CINEMA_BWD_WIDE is read once per process) and compares the dumps."""
import itertools
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [  # N, hidden, hidden layers, modulated, out_dim, latent
    (64, 512, 5, [1, 3, 5], 12, 256), (100, 256, 2, [1], 3, 64), (1000, 256, 4, [1, 3], 12, 48), (4096, 512, 1, [1], 16, 32),
    (5000, 512, 3, [2], 1, 64), (17000, 512, 2, [1, 2], 12, 32), (20500, 256, 3, [], 7, 32), (25000, 512, 5, [1, 3, 5], 12, 256),
    (33000, 512, 2, [1], 12, 32), (70000, 256, 2, [2], 5, 32),
]


def worker(idx: int, out: str):
    import torch
    from cinema_amd import _lib
    from cinema_amd.models.siren import Siren
    n, hid, layers, mod, od, lat = SHAPES[idx]
    dev = torch.device("cuda:0")
    torch.manual_seed(100 + idx)
    net = Siren(3, lat, od, hid, layers, 30, 30, True, mod, precision=_lib.PREC_BF16).to(dev)
    eng = net.hot_state(dev).engine
    ws = eng.workspace(n)
    ws.ensure_backward()
    g = torch.Generator().manual_seed(7 + idx)
    ws.coords_t.zero_()
    ws.coords_t[:n].copy_((torch.rand(n, 3, generator=g) * 2 - 1).to(dev))
    ws.lat.zero_()
    ws.lat[:n].copy_((torch.randn(n, ws.lat.shape[1], generator=g) * 0.05).to(ws.lat.dtype).to(dev))
    ws.dout.zero_()
    ws.dout[:n].copy_((torch.randn(n, od, generator=g) * 1e-3).to(dev))
    eng.pack()
    eng.forward(n)
    for wg in (True, False):
        eng.flat_grad.zero_()
        eng.grads_clean = False
        eng.backward(n, ws.dout, weight_grads=wg)
        torch.cuda.synchronize()
        np.savez(out + ("_w" if wg else "_f") + ".npz", grad=eng.flat_grad.cpu().numpy(), dcoords=ws.dcoords[:n].cpu().numpy(),
                 dlat=ws.dlat[:n].float().cpu().numpy())


def rel(a, b):
    d = float(np.abs(a - b).max())
    s = float(np.abs(b).max())
    l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    return d / max(s, 1e-30), l2


if __name__ == "__main__":
    if len(sys.argv) == 3:
        worker(int(sys.argv[1]), sys.argv[2])
        sys.exit(0)
    bad = 0
    for idx in range(len(SHAPES)):
        for w in ("1", "0"):
            env = dict(os.environ, CINEMA_BWD_WIDE=w)
            subprocess.run([sys.executable, __file__, str(idx), f"/tmp/abw_{idx}_{w}"], check=True, env=env, timeout=300)
        for tag in ("_w", "_f"):
            a, b = np.load(f"/tmp/abw_{idx}_1{tag}.npz"), np.load(f"/tmp/abw_{idx}_0{tag}.npz")
            line = f"{SHAPES[idx]} {'weights' if tag == '_w' else 'frozen '}:"
            for k in ("grad", "dcoords", "dlat"):
                mx, l2 = rel(a[k], b[k])
                ok = (mx < 2e-2 and l2 < 6e-3) or float(np.abs(b[k]).max()) == 0.0
                bad += 0 if ok else 1
                line += f" {k} max {mx:.2e} l2 {l2:.2e}{'' if ok else ' <-- BAD'};"
            print(line, flush=True)
    print("FAILED" if bad else "all shapes agree", flush=True)
    sys.exit(1 if bad else 0)
