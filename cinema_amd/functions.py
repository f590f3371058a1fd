"""torch.autograd.Functions that route the reference's module API into libcinema_hip.so."""
from __future__ import annotations

from typing import Optional

import ctypes as C

import torch

from . import _lib
from . import engine as E


class DecoderState:
    """Per-module mutable state shared between forward and backward."""

    def __init__(self, eng: E.SirenEngine):
        self.engine = eng
        self.k1 = E.TrilerpState()
        self.pack_sig = None
        self.keep_saved = True      # False for a forward under no_grad: the engine then keeps nothing for a backward


def _as_f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _maybe_pack(state: DecoderState, params):
    """Refresh the packed bf16/f32 weight copies.  Always: optimisers that write through raw pointers
    (HipAdamW) do not bump tensor version counters, and one pack launch costs ~25 us."""
    state.engine.pack()


class DecoderFunction(torch.autograd.Function):
    """INR_Decoder.forward of the reference (models/inr_decoder.py:24-36) as ONE autograd node:
    transform -> [idcs_df] gather -> trilinear interpolation -> Siren."""

    @staticmethod
    def forward(ctx, state: DecoderState, coords, latents, cond, tfs, idx, tf_per_sample: bool, *params):
        eng = state.engine
        dev = eng.device
        n = coords.shape[0]
        coords_c = _as_f32(coords, dev)
        lat_c = _as_f32(latents, dev)
        cond_c = _as_f32(cond, dev) if (cond is not None and cond.shape[-1] > 0) else None
        tfs_c = _as_f32(tfs, dev) if tfs is not None else None
        idx_c = idx.detach().to(device=dev, dtype=torch.int32).contiguous()
        eng.workspace(n)
        _maybe_pack(state, params)
        E.k1_forward(eng, state.k1, coords_c, idx_c, lat_c, cond_c, tfs_c, tf_per_sample, n)
        out = eng.forward(n, keep=state.keep_saved)[:n].clone()     # no_grad / inference: nothing is kept
        ctx.state, ctx.n, ctx.version = state, n, eng.version
        ctx.tf_per_sample = tf_per_sample
        ctx.lat_shape = tuple(lat_c.shape)
        ctx.cond_dims = 0 if cond_c is None else cond_c.shape[1]
        ctx.has_cond = cond is not None
        ctx.cond_shape = None if cond is None else tuple(cond.shape)
        ctx.save_for_backward(coords_c, idx_c, tfs_c if tfs_c is not None else coords_c.new_empty(0))
        ctx.has_tfs = tfs_c is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        state, n = ctx.state, ctx.n
        eng = state.engine
        if eng.version != ctx.version:
            raise RuntimeError("cinema_amd: the saved activations of this forward were overwritten by a later forward of "
                               "the same module; call backward() before the next forward (or use a separate module copy)")
        coords_c, idx_c, tfs_c = ctx.saved_tensors
        tfs_c = tfs_c if ctx.has_tfs else None
        ws = eng.ws
        ws.ensure_backward()
        npad = E.round_up(n, E.ROW_ALIGN)
        ws.dout[:n].copy_(gout)
        if npad > n:
            ws.dout[n:npad].zero_()
        need = ctx.needs_input_grad
        want_w = any(need[7:])
        eng.backward(n, ws.dout, weight_grads=want_w)
        want_lat, want_cond, want_tf = need[2], ctx.has_cond and need[3], ctx.has_tfs and need[4]
        dlat, dtfs, dcond = E.k1_backward(eng, state.k1, coords_c, idx_c, ctx.lat_shape, ctx.cond_dims,
                                          tfs_c if want_tf else None, ctx.tf_per_sample, n,
                                          want_latents=want_lat, want_cond=want_cond)
        if ctx.has_cond and need[3] and dcond is None:   # zero-width condition tensor
            dcond = torch.zeros(ctx.cond_shape, device=eng.device)
        pgrads = [None] * (len(need) - 7)
        if want_w:
            g = eng.flat_grad.clone()
            for k, (key, off, shape) in enumerate(eng.layout):
                if need[7 + k]:
                    cnt = 1
                    for s in shape:
                        cnt *= s
                    pgrads[k] = g[off:off + cnt].view(shape)
        return (None, None, dlat, dcond, dtfs, None, None, *pgrads)


class SirenFunction(torch.autograd.Function):
    """Siren.forward of the reference (models/siren.py:59-61) on explicit (coords, latent) rows."""

    @staticmethod
    def forward(ctx, state: DecoderState, coords, latent, *params):
        eng = state.engine
        n = coords.shape[0]
        ws = eng.workspace(n)
        _maybe_pack(state, params)
        npad = E.round_up(n, E.ROW_ALIGN)
        ws.coords_t[:npad].zero_()
        ws.coords_t[:n].copy_(coords.detach())
        ws.lat[:npad].zero_()
        ws.lat[:n, :eng.desc.lat].copy_(latent.detach())
        out = eng.forward(n, keep=state.keep_saved)[:n].clone()
        ctx.state, ctx.n, ctx.version = state, n, eng.version
        return out

    @staticmethod
    def backward(ctx, gout):
        state, n = ctx.state, ctx.n
        eng = state.engine
        if eng.version != ctx.version:
            raise RuntimeError("cinema_amd: saved activations were overwritten by a later forward of the same Siren")
        ws = eng.ws
        ws.ensure_backward()
        npad = E.round_up(n, E.ROW_ALIGN)
        ws.dout[:n].copy_(gout)
        if npad > n:
            ws.dout[n:npad].zero_()
        need = ctx.needs_input_grad
        want_w = any(need[3:])
        eng.backward(n, ws.dout, weight_grads=want_w)
        dcoords = ws.dcoords[:n].clone() if need[1] else None
        dlat = ws.dlat[:n, :eng.desc.lat].clone() if need[2] else None
        pgrads = [None] * (len(need) - 3)
        if want_w:
            g = eng.flat_grad.clone()
            for k, (key, off, shape) in enumerate(eng.layout):
                if need[3 + k]:
                    cnt = 1
                    for s in shape:
                        cnt *= s
                    pgrads[k] = g[off:off + cnt].view(shape)
        return (None, dcoords, dlat, *pgrads)


class LossFunction(torch.autograd.Function):
    """Criterion's data terms (utils.py:84-95) with the gradient w.r.t. the network output
    produced in the same launch.  Returns (total_data, sr_mean, ce_mean); only `total_data`
    = sr_scale*sr_mean + seg_scale*ce_mean carries gradient."""

    @staticmethod
    def forward(ctx, output, target, sr_dims: int, loss_metric: str, class_w: Optional[torch.Tensor], sr_scale: float,
                seg_scale: float):
        n, od = output.shape
        dev = output.device
        out_c = output.detach().contiguous().float()
        tgt_c = target.detach().to(device=dev, dtype=torch.float32).contiguous()
        npad = E.round_up(n, E.ROW_ALIGN)
        dout = torch.empty(npad, od, device=dev, dtype=torch.float32)
        sums = torch.empty(4, device=dev, dtype=torch.float32)
        cw = None if class_w is None else class_w.detach().to(device=dev, dtype=torch.float32).contiguous()
        E.sample_loss(None, out_c, n, sr_dims, od - sr_dims, loss_metric, cw, sr_scale, seg_scale, dout, sums,
                      target=tgt_c)
        sr_mean = sums[0] / float(n * sr_dims)
        if seg_scale > 0:
            den = sums[2] if cw is not None else float(n)
            ce_mean = sums[1] / den
        else:
            ce_mean = torch.zeros((), device=dev)
        total = sr_scale * sr_mean + seg_scale * ce_mean
        ctx.save_for_backward(dout)
        ctx.n = n
        ctx.mark_non_differentiable(sr_mean, ce_mean)
        return total, sr_mean, ce_mean

    @staticmethod
    def backward(ctx, g_total, g_sr, g_ce):
        (dout,) = ctx.saved_tensors
        return dout[:ctx.n] * g_total, None, None, None, None, None, None


class SineLayerFunction(torch.autograd.Function):
    """SineLayer.forward of the reference (models/siren.py:26-36) as one C-ABI call each way
    (cinema_sine_layer_forward / _backward, include/cinema_hip.h).  latent / wl / bl are None for an unmodulated layer."""

    @staticmethod
    def forward(ctx, desc, x, latent, w, b, wl, bl):
        lib = _lib.load()
        n = x.shape[0]
        mod = desc.lat_feat > 0
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        x32, lat32, w32, b32, wl32, bl32 = f32(x), f32(latent) if mod else None, f32(w), f32(b), f32(wl) if mod else None, \
            f32(bl) if mod else None
        nbytes = lib.cinema_sine_layer_work_bytes(C.byref(desc), n)
        if nbytes < 0:
            _lib.check(-1, "cinema_sine_layer_work_bytes")
        work = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty(n, desc.out_feat, dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.cinema_sine_layer_forward(C.byref(desc), _lib.ptr(x32), _lib.ptr(lat32), _lib.ptr(w32), _lib.ptr(b32),
                                                 _lib.ptr(wl32), _lib.ptr(bl32), n, _lib.ptr(work), _lib.ptr(out), stream),
                   "cinema_sine_layer_forward")
        ctx.desc, ctx.n, ctx.work, ctx.w32, ctx.mod = desc, n, work, w32, mod
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        d, n, dev = ctx.desc, ctx.n, gout.device
        z = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        dx, dw, db = z(n, d.in_feat), z(d.out_feat, d.in_feat), z(d.out_feat)
        dlat = z(n, d.lat_feat) if ctx.mod else None
        dwl = z(2 * d.out_feat, d.lat_feat) if ctx.mod else None
        dbl = z(2 * d.out_feat) if ctx.mod else None
        g32 = gout.detach().to(torch.float32).contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.cinema_sine_layer_backward(C.byref(d), _lib.ptr(ctx.w32), n, _lib.ptr(ctx.work), _lib.ptr(g32), _lib.ptr(dx),
                                                  _lib.ptr(dlat), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(dwl), _lib.ptr(dbl), stream),
                   "cinema_sine_layer_backward")
        return None, dx, dlat, dw, db, dwl, dbl


class ConvModFunction(torch.autograd.Function):
    """Modulator's Conv3d(C, C, k, padding='same') over the latent table (models/inr_decoder.py:109-123) through
    cinema_conv_mod_forward / _backward (include/cinema_hip.h)."""

    @staticmethod
    def forward(ctx, table, weight, bias, precision: int):
        lib = _lib.load()
        S, Cc, lx, ly, lz = table.shape
        k = weight.shape[2]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        x, w, b = f32(table), f32(weight), f32(bias)
        nbytes = lib.cinema_conv_mod_work_bytes(S, Cc, lx, ly, lz, k, precision)
        if nbytes < 0:
            _lib.check(-1, "cinema_conv_mod_work_bytes")
        work = torch.empty(nbytes, dtype=torch.uint8, device=table.device)
        out = torch.empty_like(x)
        stream = C.c_void_p(torch.cuda.current_stream(table.device).cuda_stream)
        _lib.check(lib.cinema_conv_mod_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), S, Cc, lx, ly, lz, k, precision, _lib.ptr(work),
                                               _lib.ptr(out), stream), "cinema_conv_mod_forward")
        ctx.save_for_backward(x, w)
        ctx.work, ctx.geom = work, (S, Cc, lx, ly, lz, k, precision)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        S, Cc, lx, ly, lz, k, precision = ctx.geom
        g = gout.detach().to(torch.float32).contiguous()
        din, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty(Cc, dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.cinema_conv_mod_backward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(g), S, Cc, lx, ly, lz, k, precision,
                                                _lib.ptr(ctx.work), _lib.ptr(din), _lib.ptr(dw), _lib.ptr(db), stream),
                   "cinema_conv_mod_backward")
        return din, dw, db, None
