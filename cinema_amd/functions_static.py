"""Autograd wrappers for the reference's two static helpers, INR_Decoder.transform and
INR_Decoder.spatial_interpolation (models/inr_decoder.py:64-85), on the K1 HIP kernels.
They exist for API completeness and parity tests; the training path uses the fused
DecoderFunction instead."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr
from .engine import ROW_ALIGN, round_up


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _require_gpu(t):
    if t.device.type != "cuda":
        raise _lib.CinemaHipError("cinema_amd runs on an MI355X only; there is no CPU fallback")


class TransformFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, tfs):
        _require_gpu(coords)
        lib, dev = _lib.load(), coords.device
        n = coords.shape[0]
        c = coords.detach().float().contiguous()
        e = tfs.detach().float().contiguous()
        aff = torch.empty(n, 12, device=dev)
        check(lib.cinema_affine_from_embedding(ptr(e), n, e.shape[1], ptr(aff), _stream(dev)), "affine_from_embedding")
        npad = round_up(n, ROW_ALIGN)
        shadow = torch.zeros(1, 1, 1, device=dev)
        idx = torch.zeros(n, dtype=torch.int32, device=dev)
        ct = torch.empty(npad, 3, device=dev)
        lat = torch.empty(npad, 64, device=dev)
        check(lib.cinema_affine_trilerp_fwd(ptr(c), ptr(idx), ptr(aff), 1, ptr(shadow), None, n, npad, 1, 1, 1, 1, 1, 0,
                                            _lib.PREC_F32, ptr(ct), ptr(lat), 64, _stream(dev)), "affine_trilerp_fwd")
        ctx.save_for_backward(c, e, idx, ct)
        return ct[:n].clone()

    @staticmethod
    def backward(ctx, g):
        c, e, idx, ct = ctx.saved_tensors
        lib, dev = _lib.load(), c.device
        n = c.shape[0]
        dct = g.detach().float().contiguous()
        dtf = torch.empty(n, e.shape[1], device=dev)
        check(lib.cinema_affine_trilerp_bwd(ptr(c), ptr(idx), ptr(e), e.shape[1], 1, None, None, ptr(ct), None, ptr(dct), n,
                                            round_up(n, ROW_ALIGN), 1, 1, 1, 1, 1, 0, 64, None, ptr(dtf), None, None,
                                            _stream(dev)), "affine_trilerp_bwd")
        return None, dtf


class InterpFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, latents, cond):
        _require_gpu(coords)
        lib, dev = _lib.load(), coords.device
        n, Cc, lx, ly, lz = latents.shape
        G = lx * ly * lz
        c = coords.detach().float().contiguous()
        lat = latents.detach().float().contiguous()
        cd = 0 if cond is None else cond.shape[1]
        cnd = cond.detach().float().contiguous() if cd else None
        LP = round_up(Cc + cd, 64)
        npad = round_up(n, ROW_ALIGN)
        shadow = torch.empty(n, G, Cc, device=dev)
        check(lib.cinema_latent_to_node_major(ptr(lat), n, Cc, G, ptr(shadow), _stream(dev)), "latent_to_node_major")
        idx = torch.arange(n, dtype=torch.int32, device=dev)
        ct = torch.empty(npad, 3, device=dev)
        out = torch.empty(npad, LP, device=dev)
        check(lib.cinema_affine_trilerp_fwd(ptr(c), ptr(idx), None, 0, ptr(shadow), ptr(cnd), n, npad, n, Cc, lx, ly, lz, cd,
                                            _lib.PREC_F32, ptr(ct), ptr(out), LP, _stream(dev)), "affine_trilerp_fwd")
        ctx.save_for_backward(c, idx, shadow, ct)
        ctx.dims = (n, Cc, lx, ly, lz, cd, LP, npad)
        ctx.has_cond = cond is not None
        return out[:n, :Cc + cd].clone()

    @staticmethod
    def backward(ctx, g):
        c, idx, shadow, ct = ctx.saved_tensors
        n, Cc, lx, ly, lz, cd, LP, npad = ctx.dims
        lib, dev = _lib.load(), c.device
        dlat = torch.zeros(npad, LP, device=dev)
        dlat[:n, :Cc + cd] = g
        dct = torch.zeros(npad, 3, device=dev)
        dl = torch.empty(n, Cc, lx, ly, lz, device=dev)
        dcond = torch.empty(n, cd, device=dev) if cd else None
        work = torch.empty(2 * n + 2 + n, dtype=torch.int32, device=dev)
        check(lib.cinema_affine_trilerp_bwd(ptr(c), ptr(idx), None, 0, 0, None, ptr(shadow), ptr(ct), ptr(dlat), ptr(dct), n,
                                            npad, n, Cc, lx, ly, lz, cd, LP, ptr(dl), None, ptr(dcond), ptr(work),
                                            _stream(dev)), "affine_trilerp_bwd")
        gc = dct[:n].clone() if ctx.needs_input_grad[0] else None
        return gc, dl, (dcond if ctx.has_cond else None)
