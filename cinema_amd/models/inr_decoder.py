"""Drop-in for the reference's `models/inr_decoder.py` (INR_Decoder, Modulator, LatentRegressor).

Same constructor (`INR_Decoder(args, device)`), `forward(coords, latent_vecs, condition_vecs,
tfs=None, idcs_df=None)`, `inference(...)`, static `transform` / `spatial_interpolation`, and the
same state_dict keys (`sr_net.net.{i}.linear.*`, `sr_net.net.{i}.linear_lats.*`,
`sr_net.final_linear.*`, `modulator.conv.*`).  `forward` is one autograd node that calls the
fused HIP path: per-sample affine registration -> latent gather by subject id (no (N,C,lx,ly,lz)
materialisation) -> trilinear interpolation -> SIREN stack.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from .. import engine as E
from ..functions import DecoderFunction
from .siren import Siren


class Modulator(nn.Module):
    """Reference: models/inr_decoder.py:109-123.  Identity for kernel_size == 0 (the default,
    config_atlas.yaml:74), else Conv3d(C, C, k, padding='same') over the latent table (S, C, lx, ly, lz).  `conv` is the
    reference's nn.Conv3d as the parameter container (state_dict keys conv.weight / conv.bias, torch's initialisation);
    its arithmetic runs in the HIP library (cinema_conv_mod_forward / _backward: im2col + split-K MFMA GEMMs, `precision`
    'bf16' | 'fp32' | None = engine default)."""

    def __init__(self, latent_dims, kernel_size=3, precision=None):
        super().__init__()
        self.kernel_size = kernel_size
        self.precision = precision
        if kernel_size > 0:
            self.conv = nn.Conv3d(latent_dims[0], latent_dims[0], kernel_size, padding="same")
        else:
            self.conv = nn.Identity()

    def forward(self, latent_vecs):
        if self.kernel_size <= 0:
            return latent_vecs
        if not latent_vecs.is_cuda:
            raise RuntimeError("cinema_amd Modulator runs HIP kernels: the latent table must live on the GPU")
        from ..functions import ConvModFunction
        prec = self.precision
        if isinstance(prec, str):
            prec = _lib.PREC_BF16 if prec.lower() in ("bf16", "bfloat16") else _lib.PREC_F32
        if prec is None:
            prec = E.default_precision()
        return ConvModFunction.apply(latent_vecs, self.conv.weight, self.conv.bias, prec)


class LatentRegressor(nn.Module):
    """Reference: models/inr_decoder.py:126-141 (tiny 3-layer MLP on pooled latents; off the hot path)."""

    def __init__(self, latent_dims):
        super().__init__()
        self.sequence = nn.Sequential(nn.Linear(latent_dims[0], 64), nn.ReLU(), nn.Linear(64, 32), nn.ReLU(),
                                      nn.Linear(32, 1))

    def forward(self, latent_vecs):
        return self.sequence(latent_vecs)


class INR_Decoder(nn.Module):
    def __init__(self, args, device):
        super().__init__()
        self.args = args
        a = args["inr_decoder"]
        self.device = device
        self.sr_dims = sum(a["out_dim"][:-1])
        self.out_dim = sum(a["out_dim"])
        self.mod_dim = a["hidden_size"] * len(a["modulated_layers"]) * 2
        precision = a.get("precision", None)
        if isinstance(precision, str):
            precision = _lib.PREC_BF16 if precision.lower() in ("bf16", "bfloat16") else _lib.PREC_F32
        if precision is None:
            precision = E.default_precision(args.get("amp", None))
        self.modulator = Modulator(a["latent_dim"], kernel_size=a["cnn_kernel_size"], precision=precision)
        self.sr_net = Siren(a["in_dim"], a["latent_dim"][0] + a["cond_dims"], self.out_dim, a["hidden_size"],
                            a["num_hidden_layers"], f_om=a["omega"][0], h_om=a["omega"][1], outermost_linear=True,
                            modulated_layers=a["modulated_layers"], precision=precision)
        self._nodes = None           # node tables of the dense-inference path (engine.NodeTables), made on first use
        self.min_inference_chunk = 1 << 19
        self.last_minmax = None

    # ------------------------------------------------------------------------------------
    def forward(self, coords, latent_vecs, condition_vecs, tfs=None, idcs_df=None, tf_table: bool = False):
        """
        coords (N,3); latent_vecs (S,C,lx,ly,lz) -- the whole subject table; condition_vecs (N,c) or
        None; tfs (N,6|9|12) per sample as in the reference, or the (S,d) table when tf_table=True
        (extension: rows are then picked by idcs_df inside the kernel); idcs_df (N,) subject ids.
        Returns (N, sum(out_dim)).
        """
        if idcs_df is None:
            if latent_vecs.shape[0] != coords.shape[0] and latent_vecs.shape[0] != 1:
                raise ValueError("idcs_df is required unless latent_vecs has one grid per sample (or a single grid)")
            idcs_df = (torch.arange(coords.shape[0], device=coords.device) if latent_vecs.shape[0] != 1
                       else torch.zeros(coords.shape[0], dtype=torch.int32, device=coords.device))
        st = self.sr_net.hot_state(coords.device)
        n = coords.shape[0]
        if n == 0:
            return coords.new_zeros((0, self.out_dim))
        if idcs_df.dim() != 1:
            idcs_df = idcs_df.reshape(-1)
        if idcs_df.shape[0] != n:
            raise ValueError(f"idcs_df has {idcs_df.shape[0]} entries for {n} coordinates")
        if tfs is not None and not tf_table and tfs.shape[0] != n:
            raise ValueError(f"tfs has {tfs.shape[0]} rows for {n} coordinates (pass tf_table=True for a per-subject table)")
        table = self.modulator(latent_vecs)
        if table.dim() != 5:
            raise ValueError("latent_vecs must be (S, C, lx, ly, lz)")
        st.keep_saved = torch.is_grad_enabled()      # grad mode is off inside Function.forward: read it here
        return DecoderFunction.apply(st, coords, table, condition_vecs, tfs, idcs_df, not tf_table, *st.param_list)

    # ------------------------------------------------------------------------------------
    def inference(self, coords, latent_vec, condition_vec, img_shape, tfs=None, step_size=100000):
        """Dense volume evaluation (reference: models/inr_decoder.py:38-62): chunks of `step_size` coordinates through
        K1 + the SIREN stack (nothing kept), each chunk's clamp / argmax / softmax written straight into its rows of the
        result and folded into the per-modality whole-volume min / max, one rescale pass at the end.  `coords` is the
        (G, 3) tensor of the reference or a `utils.WorldGrid` (axis vectors; each chunk's coordinates are then made on the
        GPU).  No (G, out_dim) intermediate.  Returns (X, Y, Z, sr_dims + 1 + n_cls) fp32."""
        from ..utils import WorldGrid
        lazy = isinstance(coords, WorldGrid)
        dev = torch.device(coords.device)
        n = len(coords) if lazy else coords.shape[0]
        sr = self.sr_dims
        vc = self.out_dim + 1
        if n == 0:
            return torch.zeros(list(img_shape) + [vc], device=dev)
        lib = _lib.load()
        st = self.sr_net.hot_state(dev)
        eng = st.engine
        # step_size bounds the reference's per-forward memory.  On the activation-stationary bf16 kernels a forward keeps
        # nothing per coordinate beyond its workspace rows (coords 12 B + bf16 latent row 2 LP B + 4 out_dim B, 0.57 KB at
        # cfg-D; the saved-activation tensor is never allocated), and the result does not depend on the chunking, so small
        # values are raised to where the launches of a chunk stop being the cost (0.3 GB at 2^19 rows; measured 221 -> 250
        # M coords/s at 310^3 from 100 000 to 400 000).  Every other path (fp32, hidden > 512, CINEMA_SIREN_LAYERED=1)
        # writes its activations (n_saved x HP elements per row: 18.4 KB bf16 / 36.9 KB fp32 at cfg-D) and honours step_size.
        floor = self.min_inference_chunk if eng.can_skip_saved() else 1
        chunk = int(min(max(step_size, floor), n))
        ws_before = eng.ws
        with torch.no_grad():
            table = self.modulator(latent_vec).detach().to(device=dev, dtype=torch.float32).contiguous()
            if table.dim() != 5 or table.shape[0] != 1:
                raise ValueError("inference: latent_vec must be one grid (1, C, lx, ly, lz)")
            cond = condition_vec.reshape(1, -1).to(device=dev, dtype=torch.float32)
            cond = cond.expand(chunk, -1).contiguous() if cond.shape[1] > 0 else None
            tf_row = None if tfs is None else tfs.detach().reshape(1, -1).to(device=dev, dtype=torch.float32).contiguous()
            idx0 = torch.zeros(chunk, dtype=torch.int32, device=dev)
            eng.workspace(chunk)
            stream = eng.stream()
            sorted_k1 = E.k1_sorted_eligible(tuple(table.shape), eng.desc.LP)
            # one subject, so every chunk is "grouped": FiLM modulation from this volume's node projections (computed once)
            nodes = E.nodes_inference_eligible(eng, tuple(table.shape))
            if nodes:
                eng.pack(4)
                if self._nodes is None:
                    self._nodes = E.NodeTables()
                E.node_tables(eng, self._nodes, table, None if cond is None else cond[:1])
            else:
                eng.pack()
            minmax = torch.cat((torch.ones(sr, device=dev), torch.zeros(sr, device=dev)))
            vol = torch.empty((n, vc), device=dev, dtype=torch.float32)
            cbuf = torch.empty((chunk, 3), device=dev, dtype=torch.float32) if lazy else None
            for i in range(0, n, chunk):
                m = min(chunk, n - i)
                if lazy:
                    ax, ay, az = coords.axes
                    _lib.check(lib.cinema_grid_coords(_lib.ptr(ax), _lib.ptr(ay), _lib.ptr(az), coords.dims[0], coords.dims[1],
                                                      coords.dims[2], i, m, _lib.ptr(cbuf), stream), "cinema_grid_coords")
                    c = cbuf
                else:
                    c = coords[i:i + m].detach().to(torch.float32).contiguous()
                if nodes:
                    E.node_weights(eng, self._nodes, c, idx0, tuple(table.shape), tf_row, m)
                    out = eng.forward_nodes(m, self._nodes, idx0, keep=False)
                elif sorted_k1:
                    E.k1_forward_sorted(eng, c, idx0, table, cond, tf_row, m)
                    out = eng.forward(m, keep=False)
                else:
                    E.k1_forward(eng, st.k1, c, idx0, table, cond, tf_row, False, m)
                    out = eng.forward(m, keep=False)
                _lib.check(lib.cinema_infer_chunk(_lib.ptr(out), m, self.out_dim, sr, _lib.ptr(minmax), _lib.ptr(vol[i:i + m]),
                                                  stream), "cinema_infer_chunk")
            _lib.check(lib.cinema_infer_rescale(_lib.ptr(vol), n, self.out_dim, sr, _lib.ptr(minmax), stream),
                       "cinema_infer_rescale")
        if ws_before is not None and eng.ws is not ws_before:
            eng.ws = ws_before      # a training workspace shared with this engine keeps its size (the chunk's one is dropped)
        self.last_minmax = minmax       # [min per modality | max per modality] of the clamped intensities (device tensor)
        modalities_rec = vol.reshape(list(img_shape) + [-1])
        if self.args.get("mask_reconstruction", False):
            from ..postmask import mask_reconstruction       # models/inr_decoder.py:59-60, 87-107, on the GPU
            return mask_reconstruction(modalities_rec, sr, self.args["dataset"]["label_names"])
        return modalities_rec

    # ------------------------------------------------------------------------------------
    @staticmethod
    def transform(coords, tfs, inverse=False):
        """c' = A(tfs) c + t per sample (reference: models/inr_decoder.py:64-71) through the K1 kernels."""
        if inverse:
            # c' = R^-1 c - R^-1 t (models/inr_decoder.py:67-69; no caller in the reference): R, t from the K1 kernel, the
            # 3 x 3 inverses by the adjugate in elementwise device ops.  Forward only -- no gradient reaches tfs / coords.
            from ..utils import embed2affine
            R, t = embed2affine(tfs)
            c = coords.detach().float()
            a, b, cc, d, e, f, g, h, i = (R[:, r, k] for r in range(3) for k in range(3))
            A, B, Cf = e * i - f * h, -(d * i - f * g), d * h - e * g
            inv = torch.stack((torch.stack((A, -(b * i - cc * h), b * f - cc * e), -1),
                               torch.stack((B, a * i - cc * g, -(a * f - cc * d)), -1),
                               torch.stack((Cf, -(a * h - b * g), a * e - b * d), -1)), -2) / (a * A + b * B + cc * Cf)[:, None, None]
            return torch.einsum("nxy,ny->nx", inv, c - t)
        from ..functions_static import TransformFunction
        return TransformFunction.apply(coords, tfs)

    @staticmethod
    def spatial_interpolation(coords, latents, condition_vecs=None):
        """Trilinear interpolation of PER-SAMPLE grids latents (N,C,lx,ly,lz) at coords (N,3)
        (reference: models/inr_decoder.py:73-85), through the K1 kernels with idx = arange(N)."""
        from ..functions_static import InterpFunction
        return InterpFunction.apply(coords, latents, condition_vecs)
