"""Drop-in for the reference's `models/siren.py` (SineLayer, Siren) on MI355X.

Same class names, constructor signatures, parameter names/shapes (reference checkpoints load
with `load_state_dict`) and call convention (`Siren.forward((coords, latent))`), but the whole
stack runs as hand-written HIP kernels (cinema_amd/csrc/siren.hip): every sine layer is an
MFMA GEMM with the FiLM modulation, sin() and its derivative fused into the epilogue.

The parameters live in ONE flat fp32 device buffer (state_dict order of the reference); each
nn.Parameter is a view into it, so the kernels, a fused optimiser and a single RCCL all-reduce
of the gradients all see one contiguous tensor.
"""
from __future__ import annotations

import copy
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .. import engine as E
from ..functions import DecoderState, SirenFunction


class SineLayer(nn.Module):
    """Parameter container of one sine layer (reference: models/siren.py:6-36).

    linear: Linear(in_feat, out_feat); linear_lats: Linear(lat_feat, 2*out_feat) when modulated.
    Initialisation follows models/siren.py:17-24.  The arithmetic of the layer,
        sin(omega * z)            or   sin(omega * z * s + t),  [s | t] = linear_lats(latent),
    is executed fused inside `Siren.forward`; a stand-alone call `layer((x, latent)) -> (out, latent)` runs the same
    layer kernels through cinema_sine_layer_forward / _backward (fp32 rows in and out; GEMM arithmetic per
    `precision`: 'bf16' / 'fp32' / None = engine default).
    """

    def __init__(self, in_feat, lat_feat, out_feat, bias=True, is_first=False, omega=30):
        super().__init__()
        if not bias:
            raise NotImplementedError("cinema_amd SineLayer requires bias=True (the reference never disables it)")
        self.omega = omega
        self.is_first = is_first
        self.in_features = in_feat
        self.out_features = out_feat
        self.linear = nn.Linear(in_feat, out_feat, bias=True)
        self.linear_lats = nn.Linear(lat_feat, out_feat * 2, bias=True) if lat_feat > 0 else None
        self.init_weights()

    def init_weights(self):
        bound = 1.0 / self.in_features if self.is_first else float(np.sqrt(6.0 / self.in_features)) / self.omega
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)

    precision = None

    def forward(self, input):
        """models/siren.py:26-36: (x (..., in_feat), latent (..., lat_feat)) -> (out (..., out_feat), latent)."""
        from ..functions import SineLayerFunction
        x, latent = input
        if not x.is_cuda:
            raise RuntimeError("cinema_amd SineLayer.forward runs HIP kernels: the input must live on the GPU")
        prec = self.precision
        if isinstance(prec, str):
            prec = _lib.PREC_BF16 if prec.lower() in ("bf16", "bfloat16") else _lib.PREC_F32
        if prec is None:
            prec = E.default_precision()
        lat_feat = self.linear_lats.in_features if self.linear_lats is not None else 0
        desc = _lib.SineLayerDesc(self.in_features, lat_feat, self.out_features, prec, float(self.omega))
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.in_features)
        if x2.shape[0] == 0:
            return x.new_zeros(lead + (self.out_features,)), latent
        if lat_feat:
            out = SineLayerFunction.apply(desc, x2, latent.reshape(-1, lat_feat), self.linear.weight, self.linear.bias,
                                          self.linear_lats.weight, self.linear_lats.bias)
        else:
            out = SineLayerFunction.apply(desc, x2, None, self.linear.weight, self.linear.bias, None, None)
        return out.reshape(lead + (self.out_features,)), latent


class Siren(nn.Module):
    """Reference: models/siren.py:39-61.  `forward((coords (N,in), latent (N,lat))) -> (N,out)`."""

    def __init__(self, in_size, lat_size, out_size, hidden_size, num_layers, f_om, h_om, outermost_linear,
                 modulated_layers, precision: Optional[int] = None):
        super().__init__()
        if not outermost_linear:
            raise NotImplementedError("outermost_linear=False is unused by the reference (inr_decoder.py:22) and not built")
        self._ctor = dict(in_size=in_size, lat_size=lat_size, out_size=out_size, hidden_size=hidden_size,
                          num_layers=num_layers, f_om=f_om, h_om=h_om, outermost_linear=outermost_linear,
                          modulated_layers=list(modulated_layers))
        self.hidden_size = hidden_size
        self.in_size, self.lat_size, self.out_size = in_size, lat_size, out_size
        self.num_layers = num_layers
        self.modulated_layers = [int(i) for i in modulated_layers if 0 <= int(i) <= num_layers]
        self.omega = (float(f_om), float(h_om))
        layers = [SineLayer(in_size, lat_size * (0 in modulated_layers), hidden_size, is_first=True, omega=f_om)]
        for i in range(num_layers):
            layers.append(SineLayer(hidden_size, lat_size * ((i + 1) in modulated_layers), hidden_size, is_first=False,
                                    omega=h_om))
        # final_linear is registered before `net`, as in the reference, so state_dict() lists it first
        self.final_linear = nn.Linear(hidden_size + lat_size, out_size, bias=True)
        with torch.no_grad():
            b = float(np.sqrt(6.0 / hidden_size)) / h_om
            self.final_linear.weight.uniform_(-b, b)
        self.net = nn.Sequential(*layers)
        self.precision = precision
        self._state: Optional[DecoderState] = None

    # ---- flat parameter buffer -------------------------------------------------------
    def _ordered_params(self):
        sd = dict(self.named_parameters())
        return sd

    def _make_desc(self, precision: int):
        return E.make_desc(self.in_size, self.lat_size, self.out_size, self.hidden_size, self.num_layers + 1,
                           self.modulated_layers, self.omega[0], self.omega[1], precision)

    def set_precision(self, precision):
        """'bf16' (MFMA bf16, fp32 accumulate) or 'fp32' (exact-f32 MFMA)."""
        if isinstance(precision, str):
            precision = _lib.PREC_BF16 if precision.lower() in ("bf16", "bfloat16") else _lib.PREC_F32
        if precision != self.precision:
            self.precision = precision
            self._state = None

    def hot_state(self, device) -> DecoderState:
        """Engine for `device`, with every nn.Parameter re-homed as a view of its flat buffer."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        prec = self.precision if self.precision is not None else E.default_precision()
        named = dict(self.named_parameters())
        st = self._state
        if st is not None and st.engine.device == device and st.engine.desc.precision == prec:
            views = st.views
            if all(named[k].data_ptr() == views[k].data_ptr() for k in views):
                return st
        eng = E.SirenEngine(self._make_desc(prec), device)
        views = eng.param_views()
        if set(views) != set(named):
            raise RuntimeError(f"parameter set mismatch: {sorted(set(views) ^ set(named))}")
        with torch.no_grad():
            for k, v in views.items():
                v.copy_(named[k].data)
                named[k].data = v
                if named[k].grad is not None:
                    named[k].grad = None
        st = DecoderState(eng)
        st.views = views
        st.param_list = [named[k] for k, _, _ in eng.layout]
        self._state = st
        return st

    def flat_parameters(self):
        """(flat fp32 parameter tensor, flat gradient tensor) of the current engine."""
        return self._state.engine.flat, self._state.engine.flat_grad

    def __deepcopy__(self, memo):
        new = Siren(**self._ctor, precision=self.precision)
        new.load_state_dict({k: v.detach().clone() for k, v in self.state_dict().items()})
        p = next(self.parameters())
        new.to(p.device)
        new.train(self.training)
        for a, b in zip(new.parameters(), self.parameters()):
            a.requires_grad_(b.requires_grad)
        memo[id(self)] = new
        return new

    def __getstate__(self):
        s = self.__dict__.copy()
        s["_state"] = None
        return s

    # ---- forward ---------------------------------------------------------------------
    def forward(self, x):
        coords, latent = x
        st = self.hot_state(coords.device)
        if coords.shape[0] == 0:
            return coords.new_zeros((0, self.out_size))
        st.keep_saved = torch.is_grad_enabled()
        return SirenFunction.apply(st, coords, latent, *st.param_list)
