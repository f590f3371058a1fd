"""cinema_affine_trilerp_bwd_noalloc (the entry point that allocates nothing: its grouping scratch is a device array of
the library) against cinema_affine_trilerp_bwd with a caller-owned work buffer, through the C ABI: same outputs to the
fp32 bar (1e-4 max-norm relative; the reductions are atomics-free, so in practice the bits agree), on two calls in a row,
the second with a smaller input, and an over-size call is refused with an error.  Collected last on purpose (file name)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _case(N, S, Cc, tf_dim, seed):
    g = torch.Generator().manual_seed(seed)
    G = 27
    LP = (Cc + 63) // 64 * 64
    npad = (N + 127) // 128 * 128
    t = {
        "coords": torch.rand(N, 3, generator=g) * 1.8 - 0.9,
        "idx": torch.randint(0, S, (N,), generator=g, dtype=torch.int32),
        "embed": (torch.rand(S, tf_dim, generator=g) - 0.5) * 0.1,
        "shadow": torch.randn(S, G, Cc, generator=g) * 0.01,
        "coords_t": torch.zeros(npad, 3),
        "dlat": torch.zeros(npad, LP),
        "dcoords0": torch.zeros(npad, 3),
    }
    t["coords_t"][:N] = torch.rand(N, 3, generator=g) * 2.2 - 1.1        # also outside [-1, 1]: border clamp
    t["dlat"][:N, :Cc] = torch.randn(N, Cc, generator=g)
    t["dcoords0"][:N] = torch.randn(N, 3, generator=g)
    return {k: v.to(DEV).contiguous() for k, v in t.items()}, npad, LP


def _run(lib, t, N, npad, S, Cc, tf_dim, LP, noalloc):
    from cinema_amd._lib import check, ptr
    dlatents = torch.full((S, Cc, 3, 3, 3), float("nan"), device=DEV)
    dembed = torch.full((S, tf_dim), float("nan"), device=DEV)
    dcoords = t["dcoords0"].clone()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [ptr(t["coords"]), ptr(t["idx"]), ptr(t["embed"]), tf_dim, 0, None, ptr(t["shadow"]), ptr(t["coords_t"]),
            ptr(t["dlat"]), ptr(dcoords), N, npad, S, Cc, 3, 3, 3, 0, LP, ptr(dlatents), ptr(dembed), None]
    if noalloc:
        check(lib.cinema_affine_trilerp_bwd_noalloc(*args, stream), "cinema_affine_trilerp_bwd_noalloc")
    else:
        work = torch.empty(2 * S + 2 + 13 * N, device=DEV, dtype=torch.int32)
        check(lib.cinema_affine_trilerp_bwd(*args, ptr(work), stream), "cinema_affine_trilerp_bwd")
    torch.cuda.synchronize()
    return dlatents.cpu().numpy(), dembed.cpu().numpy(), dcoords[:N].cpu().numpy()


def _rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def test_noalloc_k1_backward_matches_the_work_buffer_entry_point():
    from cinema_amd import _lib
    lib = _lib.load()
    # two calls in a row on the same device array: a step-sized batch (one-launch grouping), then a smaller one;
    # then a batch above 32768 samples (three-launch grouping, which clears its counters itself) and a small one again
    for N, S, Cc, tf_dim, seed in ((20000, 48, 64, 6, 1), (777, 5, 32, 9, 2), (40000, 700, 32, 12, 3), (130, 3, 64, 6, 4)):
        t, npad, LP = _case(N, S, Cc, tf_dim, seed)
        ref = _run(lib, t, N, npad, S, Cc, tf_dim, LP, noalloc=False)
        got = _run(lib, t, N, npad, S, Cc, tf_dim, LP, noalloc=True)
        for name, a, b in zip(("dlatents", "dembed", "dcoords"), got, ref):
            assert np.isfinite(a).all() and np.isfinite(b).all(), (name, N)
            assert _rel(a, b) <= TOL, (name, N, _rel(a, b))


def test_noalloc_k1_backward_rejects_sizes_beyond_its_scratch():
    from cinema_amd import _lib
    lib = _lib.load()
    t, npad, LP = _case(256, 4, 32, 6, 9)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = _lib.ptr
    for N, S in ((_lib.NOALLOC_MAX_N + 1, 4), (256, _lib.NOALLOC_MAX_S + 1)):
        rc = lib.cinema_affine_trilerp_bwd_noalloc(p(t["coords"]), p(t["idx"]), p(t["embed"]), 6, 0, None, p(t["shadow"]),
                                                   p(t["coords_t"]), p(t["dlat"]), p(t["dcoords0"]), N, npad, S, 32, 3, 3, 3,
                                                   0, LP, None, None, None, stream)
        assert rc != 0
        assert b"exceed the static scratch" in lib.cinema_last_error()
