"""Host-side driver of the HIP hot path: owns the flat parameter buffer, the packed bf16/f32
weight copies and the per-batch workspaces, and sequences the C-ABI calls.

PyTorch is plumbing here (device memory, streams); every arithmetic step of the path runs in
libcinema_hip.so.  There is no eager fallback: constructing an engine without the library or
without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SirenDesc, check, ptr

ROW_ALIGN = 128


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def default_precision(amp: Optional[bool] = None) -> int:
    """bf16 MFMA unless CINEMA_PRECISION=fp32 or the reference's `amp: False` is configured."""
    env = os.environ.get("CINEMA_PRECISION", "").lower()
    if env in ("fp32", "f32", "float32"):
        return _lib.PREC_F32
    if env in ("bf16", "bfloat16"):
        return _lib.PREC_BF16
    if amp is not None and not amp:
        return _lib.PREC_F32
    return _lib.PREC_BF16


def make_desc(in_dim: int, lat: int, out_dim: int, hidden: int, n_sine_layers: int, modulated_layers: Sequence[int],
              omega_first: float, omega_hidden: float, precision: int) -> SirenDesc:
    d = SirenDesc()
    d.n_layers, d.in_dim, d.hidden, d.lat, d.out_dim, d.precision = n_sine_layers, in_dim, hidden, lat, out_dim, precision
    mask = 0
    for i in modulated_layers:
        if 0 <= i < n_sine_layers:
            mask |= 1 << i
    d.modulated_mask = mask
    d.omega_first, d.omega_hidden = float(omega_first), float(omega_hidden)
    check(_lib.load().cinema_siren_plan(C.byref(d)), "cinema_siren_plan")
    return d


def param_layout(d: SirenDesc) -> List[Tuple[str, int, Tuple[int, ...]]]:
    """(state_dict key, offset in floats, shape) in the flat-buffer order (= the reference's
    state_dict order: final_linear first, models/siren.py:50-57)."""
    H, L = d.hidden, d.lat
    out = [("final_linear.weight", d.off_wf, (d.out_dim, H + L)), ("final_linear.bias", d.off_bf, (d.out_dim,))]
    for i in range(d.n_layers):
        fan_in = d.in_dim if i == 0 else H
        out.append((f"net.{i}.linear.weight", d.off_w[i], (H, fan_in)))
        out.append((f"net.{i}.linear.bias", d.off_b[i], (H,)))
        if (d.modulated_mask >> i) & 1:
            out.append((f"net.{i}.linear_lats.weight", d.off_wl[i], (2 * H, L)))
            out.append((f"net.{i}.linear_lats.bias", d.off_bl[i], (2 * H,)))
    return out


class Workspace:
    """Device buffers for one batch size; reused across steps (the library allocates nothing)."""

    def __init__(self, d: SirenDesc, n_rows: int, device, in_dim: int):
        self.npad_cap = round_up(max(n_rows, 1), ROW_ALIGN)
        dt = torch.bfloat16 if d.precision == _lib.PREC_BF16 else torch.float32
        n = self.npad_cap
        self.coords_t = torch.empty(n, in_dim, device=device, dtype=torch.float32)
        self.lat = torch.empty(n, d.LP, device=device, dtype=dt)
        self._saved = None     # (n_saved, n, HP): allocated by the first access -- a forward that keeps nothing never pays for it
        self.out = torch.empty(n, d.out_dim, device=device, dtype=torch.float32)
        self.gbuf = None       # backward-only buffers are allocated on first use
        self.dlat = None
        self.dcoords = None
        self.dout = None
        self._d = d
        self._dt = dt
        self._device = device
        self._in_dim = in_dim

    @property
    def saved(self) -> torch.Tensor:
        """Activations kept for the backward (18 x Npad x HP elements at cfg-D: 18.4 KB per row in bf16)."""
        if self._saved is None:
            d = self._d
            self._saved = torch.empty(d.n_saved, self.npad_cap, d.HP, device=self._device, dtype=self._dt)
        return self._saved

    def ensure_backward(self):
        if self.gbuf is None:
            d, n = self._d, self.npad_cap
            esize = 2 if self._dt == torch.bfloat16 else 4
            nbytes = round_up(d.n_gbuf * n * d.HP * esize, 256) + int(d.gb_extra_bytes)
            self.gbuf = torch.empty(nbytes, device=self._device, dtype=torch.uint8)
            self.dlat = torch.empty(n, d.LP, device=self._device, dtype=torch.float32)
            self.dcoords = torch.empty(n, self._in_dim, device=self._device, dtype=torch.float32)
            self.dout = torch.zeros(n, d.out_dim, device=self._device, dtype=torch.float32)


class SirenEngine:
    """Flat parameters + packed copies + workspaces for one Siren stack on one device."""

    def __init__(self, desc: SirenDesc, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.CinemaHipError(
                f"the CINeMA hot path runs on an MI355X only (got device '{device}'); there is no CPU fallback")
        self.lib = _lib.load()
        self.desc = desc
        self.device = device
        self.layout = param_layout(desc)
        self.flat = torch.zeros(desc.n_params, device=device, dtype=torch.float32)
        self.flat_grad = torch.zeros(desc.n_params, device=device, dtype=torch.float32)
        pk_dt = torch.bfloat16 if desc.precision == _lib.PREC_BF16 else torch.float32
        self.packed = torch.zeros(desc.pk_total, device=device, dtype=pk_dt)
        self.ws: Optional[Workspace] = None
        self.version = 0          # bumped by every forward: guards stale backward calls
        self.grads_clean = True   # flat_grad holds zeros (a clearing optimiser step ran after the last backward)
        self._packed_version = None

    # -- views ------------------------------------------------------------------------
    def param_views(self, flat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        flat = self.flat if flat is None else flat
        out = {}
        for key, off, shape in self.layout:
            n = 1
            for s in shape:
                n *= s
            out[key] = flat[off:off + n].view(shape)
        return out

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, n_rows: int) -> Workspace:
        if self.ws is None or self.ws.npad_cap < round_up(n_rows, ROW_ALIGN):
            self.ws = None
            self.ws = Workspace(self.desc, n_rows, self.device, self.desc.in_dim)
        return self.ws

    # -- steps ------------------------------------------------------------------------
    def pack(self, part: int = 3):
        """Refresh the packed compute copies from the flat fp32 parameters.
        part 1: what the forward reads, 2: what only the backward reads, 3: both."""
        check(self.lib.cinema_siren_pack_part(C.byref(self.desc), ptr(self.flat), ptr(self.packed), part, self.stream()),
              "cinema_siren_pack")

    def can_skip_saved(self) -> bool:
        """True when a forward may run without keeping activations (the activation-stationary bf16 kernels)."""
        return self.desc.pk_fs >= 0 and os.environ.get("CINEMA_SIREN_LAYERED") != "1"

    def forward(self, n: int, keep: bool = True, ng: Optional["NodeGrad"] = None, latents_shape=None) -> torch.Tensor:
        """Run the stack on ws.coords_t / ws.lat (already filled for n rows). Returns ws.out.
        keep=False (inference): nothing is saved for a backward where the kernels allow it.
        ng (with latents_shape): the node projections of the node-space backward ride in the same launch
        (cinema_siren_forward_node_prep; step_head(..., ng=ng) of this step wrote the node rows and weights)."""
        ws = self.ws
        npad = round_up(n, ROW_ALIGN)
        self.version += 1
        saved = ws.saved if (keep or not self.can_skip_saved()) else None
        if ng is not None:
            S, _, lx, ly, lz = latents_shape
            ng.ensure(self, S, npad)
            check(self.lib.cinema_siren_forward_node_prep(C.byref(self.desc), ptr(self.flat), ptr(self.packed), ptr(ws.coords_t),
                                                          ptr(ws.lat), n, npad, ptr(saved), ptr(ws.out), ptr(ng.work), S, lx, ly,
                                                          lz, self.stream()), "cinema_siren_forward_node_prep")
            return ws.out
        check(self.lib.cinema_siren_forward(C.byref(self.desc), ptr(self.flat), ptr(self.packed), ptr(ws.coords_t),
                                            ptr(ws.lat), n, npad, ptr(saved), ptr(ws.out), self.stream()),
              "cinema_siren_forward")
        return ws.out

    def forward_nodes(self, n: int, nt: "NodeTables", idx: torch.Tensor, keep: bool = True, round_s: bool = False) -> torch.Tensor:
        """forward() on the nodes path: ws.coords_t / nt.wrows filled by node_weights, nt.U / nt.V by node_tables."""
        ws = self.ws
        npad = round_up(n, ROW_ALIGN)
        self.version += 1
        check(self.lib.cinema_siren_forward_nodes(C.byref(self.desc), ptr(self.flat), ptr(self.packed), ptr(ws.coords_t),
                                                  ptr(nt.wrows), ptr(nt.tiles), ptr(nt.U), ptr(nt.V), nt.S, n, npad,
                                                  ptr(ws.saved) if keep else None, ptr(ws.out),
                                                  _lib.NODES_ROUND_S if round_s else 0, self.stream()),
              "cinema_siren_forward_nodes")
        return ws.out

    def backward(self, n: int, dout_padded: torch.Tensor, weight_grads: bool = True, phases: int = 0,
                 dcoords_cleared: bool = False, pending=None, nodes=None, round_s: bool = False, part: int = 0):
        """dout_padded: [>= Npad x out_dim] fp32 with zero rows >= n.  Fills flat_grad, ws.dlat, ws.dcoords.
        dcoords_cleared: ws.dcoords already holds zeros (step_prepare); flat_grad is not cleared again when a
        clearing optimiser step left it zeroed (grads_clean).
        part: 0 = the whole backward; _lib.BWD_WEIGHTS_FIRST = everything but the d-latent GEMM (flat_grad complete on
        return: a data-parallel rank starts its all-reduce here); _lib.BWD_DLAT_ONLY = the d-latent GEMM afterwards."""
        ws = self.ws
        ws.ensure_backward()
        npad = round_up(n, ROW_ALIGN)
        flags = (0 if weight_grads else _lib.BWD_SKIP_WEIGHT_GRADS) | phases | part
        if dcoords_cleared:
            flags |= _lib.BWD_DCOORDS_CLEARED
        if weight_grads and part != _lib.BWD_DLAT_ONLY:
            if self.grads_clean and phases == 0:
                flags |= _lib.BWD_GRADS_CLEARED
            self.grads_clean = False
        if nodes is not None:
            nt, idx = nodes
            check(self.lib.cinema_siren_backward_nodes(C.byref(self.desc), ptr(self.flat), ptr(self.packed), ptr(ws.coords_t),
                                                       ptr(ws.lat), ptr(nt.wrows), ptr(nt.tiles), ptr(nt.U), nt.S, ptr(ws.saved),
                                                       ptr(dout_padded), n, npad, ptr(ws.gbuf), ptr(self.flat_grad), ptr(ws.dlat),
                                                       ptr(ws.dcoords), flags | (_lib.NODES_ROUND_S if round_s else 0),
                                                       C.byref(pending) if pending is not None else None, self.stream()),
                  "cinema_siren_backward_nodes")
            return
        if pending is not None:
            # the split reduction of the weight-gradient tiles is left to the caller (k1_backward(pending=...) runs it
            # beside the K1 coordinate gradient): flat_grad is complete only after that call
            check(self.lib.cinema_siren_backward_deferred(C.byref(self.desc), ptr(self.flat), ptr(self.packed),
                                                          ptr(ws.coords_t), ptr(ws.lat), ptr(ws.saved), ptr(dout_padded), n,
                                                          npad, ptr(ws.gbuf), ptr(self.flat_grad), ptr(ws.dlat),
                                                          ptr(ws.dcoords), flags, C.byref(pending), self.stream()),
                  "cinema_siren_backward_deferred")
            return
        check(self.lib.cinema_siren_backward(C.byref(self.desc), ptr(self.flat), ptr(self.packed), ptr(ws.coords_t),
                                             ptr(ws.lat), ptr(ws.saved), ptr(dout_padded), n, npad, ptr(ws.gbuf),
                                             ptr(self.flat_grad), ptr(ws.dlat), ptr(ws.dcoords), flags, self.stream()),
              "cinema_siren_backward")


    def backward_nodegrad(self, n: int, dout_padded: torch.Tensor, ng: "NodeGrad", coords: torch.Tensor, idx: torch.Tensor,
                          tfs: Optional[torch.Tensor], latents_shape, cond_dims: int, dlatents: torch.Tensor,
                          dtfs: Optional[torch.Tensor], dcond: Optional[torch.Tensor], weight_grads: bool = True,
                          dcoords_cleared: bool = False, part: int = 0, phases: int = 0):
        """SIREN backward + K1 backward in one call with the latent side reduced in node space (include/cinema_hip.h,
        "node gradients"): fills flat_grad, dlatents (S, C, lx, ly, lz), dtfs (S, tf_dim), dcond (S, cond_dims).
        The batch must be grouped by subject and nodegrad_prepare must have run on the current parameters."""
        ws = self.ws
        ws.ensure_backward()
        npad = round_up(n, ROW_ALIGN)
        S, Cc, lx, ly, lz = latents_shape
        flags = (0 if weight_grads else _lib.BWD_SKIP_WEIGHT_GRADS) | part | phases     # phases: single kernels (bench.py)
        if dcoords_cleared:
            flags |= _lib.BWD_DCOORDS_CLEARED
        if weight_grads and part != _lib.BWD_DLAT_ONLY:
            if self.grads_clean and phases == 0:
                flags |= _lib.BWD_GRADS_CLEARED
            self.grads_clean = False
        check(self.lib.cinema_siren_backward_nodegrad(
            C.byref(self.desc), ptr(self.flat), ptr(self.packed), ptr(ws.coords_t), ptr(ws.saved), ptr(dout_padded), n, npad,
            ptr(ws.gbuf), ptr(self.flat_grad), ptr(ws.dcoords), flags, ptr(coords), ptr(idx), ptr(tfs),
            0 if tfs is None else tfs.shape[1], S, Cc, lx, ly, lz, cond_dims, ptr(ng.work), ptr(dlatents),
            ptr(dtfs) if tfs is not None else None, ptr(dcond), self.stream()), "cinema_siren_backward_nodegrad")


class NodeGrad:
    """Work buffer of the node-space backward for one (subject count, batch size)."""

    def __init__(self):
        self.work = None
        self._bufs = {}     # (S, Npad) -> buffer: the layout depends on both, and its pad rows must stay zero

    def ensure(self, eng: "SirenEngine", S: int, npad: int):
        buf = self._bufs.get((S, npad))
        if buf is None:
            nbytes = eng.lib.cinema_nodegrad_work_bytes(C.byref(eng.desc), S, npad)
            if nbytes < 0:
                check(-1, "cinema_nodegrad_work_bytes")
            if len(self._bufs) >= 2:      # a step-sized batch and the epoch's short last chunk
                self._bufs.clear()
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=eng.device)
            self._bufs[(S, npad)] = buf
        self.work = buf


def nodegrad_eligible(eng: "SirenEngine", latents_shape) -> bool:
    """Shapes the node-space backward takes (bf16 activation-stationary stack with a modulated layer, at most 30 grid
    nodes); CINEMA_NODE_GRAD=0 keeps the per-sample backward (A/B switch)."""
    _, _, lx, ly, lz = latents_shape
    return (os.environ.get("CINEMA_NODE_GRAD", "1") != "0" and eng.can_skip_saved()
            and eng.lib.cinema_nodegrad_eligible(C.byref(eng.desc), lx, ly, lz) == 1)


def nodegrad_prepare(eng: "SirenEngine", ng: NodeGrad, coords: torch.Tensor, idx: torch.Tensor, table: torch.Tensor,
                     cond_subject: Optional[torch.Tensor], tfs: Optional[torch.Tensor], n: int, rows_ready: bool = False):
    """Interpolation weights of the batch + node rows / centred node projections of the current table (after the packed
    weight copies were refreshed: step_head, or eng.pack(2 | 8)).  rows_ready: step_head(..., ng=ng) of this step wrote
    the node rows and the interpolation weights already (only the projections are left)."""
    S, Cc, lx, ly, lz = table.shape
    cond_dims = 0 if cond_subject is None else cond_subject.shape[1]
    npad = round_up(n, ROW_ALIGN)
    ng.ensure(eng, S, npad)
    check(eng.lib.cinema_nodegrad_prepare(C.byref(eng.desc), ptr(eng.packed), ptr(coords), ptr(idx), ptr(tfs),
                                          0 if tfs is None else tfs.shape[1], ptr(table), ptr(cond_subject) if cond_dims else None,
                                          n, npad, S, Cc, lx, ly, lz, cond_dims, ptr(ng.work), 3 if rows_ready else 0,
                                          eng.stream()),
          "cinema_nodegrad_prepare")


class NodeTables:
    """Buffers of the nodes path for one latent table (include/cinema_hip.h, "Nodes" section): node rows T, their
    projections U / V, and the per-sample weight rows of the current batch."""

    def __init__(self):
        self.T = self.U = self.V = self.wrows = self.tiles = None
        self.S = 0

    def ensure(self, eng: "SirenEngine", S: int, rows: int):
        if self.S != S or self.T is None:
            t, u, v = C.c_int64(), C.c_int64(), C.c_int64()
            check(eng.lib.cinema_nodes_sizes(C.byref(eng.desc), S, C.byref(t), C.byref(u), C.byref(v)), "cinema_nodes_sizes")
            self.T = torch.empty(t.value, device=eng.device, dtype=torch.bfloat16)
            self.U = torch.zeros(u.value, device=eng.device, dtype=torch.uint8)
            self.V = torch.zeros(v.value, device=eng.device, dtype=torch.uint8)
            self.S = S
        if self.wrows is None or self.wrows.shape[0] < rows:
            self.wrows = torch.empty(rows, _lib.NODE_SLOTS, device=eng.device, dtype=torch.float16)
        need = 4 * (rows // 16 + S + 2)
        if self.tiles is None or self.tiles.numel() < need:
            self.tiles = torch.zeros(need, device=eng.device, dtype=torch.int32)


def nodes_eligible(eng: "SirenEngine", latents_shape) -> bool:
    """Shapes the nodes path takes: bf16 activation-stationary stack, at most 30 grid nodes.  Opt-in (CINEMA_NODES=1): its
    chain kernels are not yet faster than the per-sample ones end to end (nodes path)."""
    _, _, lx, ly, lz = latents_shape
    return eng.can_skip_saved() and eng.desc.pk_ns >= 0 and lx * ly * lz <= 30 and os.environ.get("CINEMA_NODES", "0") == "1"


def nodes_inference_eligible(eng: "SirenEngine", latents_shape) -> bool:
    """Dense inference takes the nodes path by default where the shape allows (forward only, nothing kept: measured 24 %
    faster than the per-sample forward at cfg-D); CINEMA_NODES=0 turns it off."""
    _, _, lx, ly, lz = latents_shape
    return eng.can_skip_saved() and eng.desc.pk_ns >= 0 and lx * ly * lz <= 30 and os.environ.get("CINEMA_NODES", "1") != "0"


def node_weights(eng: "SirenEngine", nt: NodeTables, coords: torch.Tensor, idx: torch.Tensor, latents_shape,
                 tfs: Optional[torch.Tensor], n: int):
    """transform + interpolation weights into eng.ws.coords_t / nt.wrows (rows up to Npad)."""
    S, _, lx, ly, lz = latents_shape
    npad = round_up(n, ROW_ALIGN)
    nt.ensure(eng, S, eng.ws.npad_cap)
    check(eng.lib.cinema_node_weights(ptr(coords), ptr(idx), ptr(tfs), 0 if tfs is None else tfs.shape[1], n, npad, S, lx, ly, lz,
                                      ptr(eng.ws.coords_t), ptr(nt.wrows), eng.stream()), "cinema_node_weights")
    check(eng.lib.cinema_node_tiles(ptr(idx), n, npad, S, ptr(nt.tiles), eng.stream()), "cinema_node_tiles")


def node_tables(eng: "SirenEngine", nt: NodeTables, latents: torch.Tensor, cond_subject: Optional[torch.Tensor]):
    """Node rows + projections of the current parameters (after eng.pack(4))."""
    S, Cc, lx, ly, lz = latents.shape
    cond_dims = 0 if cond_subject is None else cond_subject.shape[1]
    nt.ensure(eng, S, eng.ws.npad_cap if eng.ws is not None else ROW_ALIGN)
    check(eng.lib.cinema_node_tables(C.byref(eng.desc), ptr(eng.flat), ptr(eng.packed), ptr(latents),
                                     ptr(cond_subject) if cond_dims else None, S, Cc, lx, ly, lz, cond_dims, ptr(nt.T), ptr(nt.U),
                                     ptr(nt.V), eng.stream()), "cinema_node_tables")


class TrilerpState:
    """Buffers of the K1 stage for one latent table."""

    def __init__(self):
        self.shadow = None
        self.affine = None
        self.work = None


def k1_forward(eng: SirenEngine, st: TrilerpState, coords: torch.Tensor, idx: torch.Tensor, latents: torch.Tensor,
               cond: Optional[torch.Tensor], tfs: Optional[torch.Tensor], tf_per_sample: bool, n: int):
    """transform + latent gather + trilinear interpolation into eng.ws.coords_t / eng.ws.lat."""
    lib, d, ws = eng.lib, eng.desc, eng.ws
    S, Cc = latents.shape[0], latents.shape[1]
    lx, ly, lz = latents.shape[2], latents.shape[3], latents.shape[4]
    G = lx * ly * lz
    cond_dims = 0 if cond is None else cond.shape[1]
    if Cc + cond_dims != d.lat:
        raise ValueError(f"latent channels {Cc} + condition dims {cond_dims} != lat_size {d.lat} of the SIREN")
    if st.shadow is None or st.shadow.numel() != S * G * Cc:
        st.shadow = torch.empty(S, G, Cc, device=eng.device, dtype=torch.float32)
    s = eng.stream()
    check(lib.cinema_latent_to_node_major(ptr(latents), S, Cc, G, ptr(st.shadow), s), "cinema_latent_to_node_major")
    aff = None
    if tfs is not None:
        rows = tfs.shape[0]
        if st.affine is None or st.affine.shape[0] < rows:
            st.affine = torch.empty(rows, 12, device=eng.device, dtype=torch.float32)
        check(lib.cinema_affine_from_embedding(ptr(tfs), rows, tfs.shape[1], ptr(st.affine), s),
              "cinema_affine_from_embedding")
        aff = st.affine
    npad = round_up(n, ROW_ALIGN)
    check(lib.cinema_affine_trilerp_fwd(ptr(coords), ptr(idx), ptr(aff), 1 if tf_per_sample else 0, ptr(st.shadow),
                                        ptr(cond) if cond_dims else None, n, npad, S, Cc, lx, ly, lz, cond_dims,
                                        d.precision, ptr(ws.coords_t), ptr(ws.lat), d.LP, s),
          "cinema_affine_trilerp_fwd")


def k1_sorted_eligible(latents_shape, LP: int) -> bool:
    """Shapes the grouped-batch forward kernel takes (cinema_affine_trilerp_fwd_sorted)."""
    _, Cc, lx, ly, lz = latents_shape
    return (lx, ly, lz) == (3, 3, 3) and 4 <= Cc <= 256 and Cc % 4 == 0 and LP % 4 == 0


def k1_forward_sorted(eng: SirenEngine, coords: torch.Tensor, idx: torch.Tensor, latents: torch.Tensor,
                      cond: Optional[torch.Tensor], tfs: Optional[torch.Tensor], n: int, cond_by_subject: bool = False):
    """k1_forward for a batch grouped by subject with a per-subject embedding table: one launch, reads the latent
    table in its API layout (no shadow, no affine table).  Bit-identical outputs.
    cond_by_subject: cond is the (S, cond_dims) table, rows picked by idx in the kernel (no host-side gather)."""
    lib, d, ws = eng.lib, eng.desc, eng.ws
    S, Cc, lx, ly, lz = latents.shape
    cond_dims = 0 if cond is None else cond.shape[1]
    if Cc + cond_dims != d.lat:
        raise ValueError(f"latent channels {Cc} + condition dims {cond_dims} != lat_size {d.lat} of the SIREN")
    npad = round_up(n, ROW_ALIGN)
    check(lib.cinema_affine_trilerp_fwd_sorted(ptr(coords), ptr(idx), ptr(tfs), 0 if tfs is None else tfs.shape[1],
                                               ptr(latents), ptr(cond) if cond_dims else None, n, npad, S, Cc, lx, ly, lz,
                                               cond_dims, 1 if cond_by_subject else 0, d.precision, ptr(ws.coords_t), ptr(ws.lat),
                                               d.LP, eng.stream()),
          "cinema_affine_trilerp_fwd_sorted")


def cond_table_grad(idx_sorted: torch.Tensor, dcond: torch.Tensor, n: int, out: torch.Tensor):
    """out[s] = sum of dcond rows of subject s for a batch sorted by subject (the index_add_ of the reference's autograd
    for conditions[idcs_df], build_atlas.py:74-75): one launch, ranges by binary search, no atomics."""
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
    check(lib.cinema_cond_table_bwd(ptr(idx_sorted), ptr(dcond), n, out.shape[0], out.shape[1], ptr(out), stream),
          "cinema_cond_table_bwd")


def step_head(eng: SirenEngine, st: "TrilerpState", coords: torch.Tensor, idx: torch.Tensor, latents: torch.Tensor,
              cond: Optional[torch.Tensor], tfs: Optional[torch.Tensor], n: int, prepare: Optional[dict] = None,
              shadow: bool = True, cond_by_subject: bool = False, ng: Optional["NodeGrad"] = None):
    """eng.pack() + k1_forward_sorted (+ k1_shadow, + step_prepare) in ONE launch (cinema_step_head); the caller
    checked k1_sorted_eligible and eng.can_skip_saved() (= the activation-stationary bf16 kernels are in use).
    prepare: dict(dcoords, sums, n_cls, class_w, voxel, labels, V, scratch) or None.
    ng: work buffer of the node-space backward -- its node rows and interpolation weights are written in the same launch
    (then eng.forward(n, ng=ng, ...) adds the projections, or nodegrad_prepare(..., rows_ready=True) does)."""
    lib, d, ws = eng.lib, eng.desc, eng.ws
    S, Cc, lx, ly, lz = latents.shape
    cond_dims = 0 if cond is None else cond.shape[1]
    npad = round_up(n, ROW_ALIGN)
    sh = None
    if shadow:
        G = lx * ly * lz
        if st.shadow is None or st.shadow.numel() != S * G * Cc:
            st.shadow = torch.empty(S, G, Cc, device=eng.device, dtype=torch.float32)
        sh = st.shadow
    if ng is not None:
        ng.ensure(eng, S, npad)
    p = prepare or {}
    dco = p.get("dcoords")
    check(lib.cinema_step_head(C.byref(d), ptr(eng.flat), ptr(eng.packed), ptr(coords), ptr(idx), ptr(tfs),
                               0 if tfs is None else tfs.shape[1], ptr(latents), ptr(cond) if cond_dims else None, n, npad,
                               S, Cc, lx, ly, lz, cond_dims, 1 if cond_by_subject else 0, ptr(ws.coords_t), ptr(ws.lat), ptr(sh),
                               None if ng is None else ptr(ng.work), ptr(dco),
                               0 if dco is None else dco.numel(), ptr(p.get("sums")), ptr(p.get("voxel")), ptr(p.get("labels")),
                               int(p.get("V", 0)), int(p.get("n_cls", 0)), ptr(p.get("class_w")), ptr(p.get("scratch")),
                               eng.stream()), "cinema_step_head")


def k1_shadow(eng: SirenEngine, st: TrilerpState, latents: torch.Tensor):
    """Node-major shadow of the latent table for the K1 backward (on the current stream)."""
    S, Cc = latents.shape[0], latents.shape[1]
    G = latents.shape[2] * latents.shape[3] * latents.shape[4]
    if st.shadow is None or st.shadow.numel() != S * G * Cc:
        st.shadow = torch.empty(S, G, Cc, device=eng.device, dtype=torch.float32)
    check(eng.lib.cinema_latent_to_node_major(ptr(latents), S, Cc, G, ptr(st.shadow), eng.stream()),
          "cinema_latent_to_node_major")


def k1_backward(eng: SirenEngine, st: TrilerpState, coords: torch.Tensor, idx: torch.Tensor, latents_shape,
                cond_dims: int, tfs: Optional[torch.Tensor], tf_per_sample: bool, n: int,
                want_latents: bool = True, want_cond: bool = False, dtf_out: Optional[torch.Tensor] = None,
                dlat_out: Optional[torch.Tensor] = None, grouped: bool = False, pending=None):
    """Returns (dlatents | None, dtfs | None, dcond | None) from eng.ws.dlat / eng.ws.dcoords.
    grouped=True: the caller guarantees idx is non-decreasing (batch sorted by subject)."""
    lib, d, ws = eng.lib, eng.desc, eng.ws
    S, Cc, lx, ly, lz = latents_shape
    need = 2 * S + 2 + 13 * n   # counts, offsets, perm, per-sample embedding gradients (fp32 x 12)
    if st.work is None or st.work.numel() < need:
        st.work = torch.empty(need, device=eng.device, dtype=torch.int32)
    dlatents = None
    if want_latents:
        dlatents = dlat_out if dlat_out is not None else torch.empty(S, Cc, lx, ly, lz, device=eng.device,
                                                                      dtype=torch.float32)
    dtfs = None
    if tfs is not None:
        if dtf_out is not None:
            dtfs = dtf_out
        elif tf_per_sample:
            dtfs = torch.empty(n, tfs.shape[1], device=eng.device, dtype=torch.float32)
        else:
            dtfs = torch.zeros(tfs.shape[0], tfs.shape[1], device=eng.device, dtype=torch.float32)
    dcond = torch.empty(n, cond_dims, device=eng.device, dtype=torch.float32) if (want_cond and cond_dims) else None
    npad = round_up(n, ROW_ALIGN)
    args = (ptr(coords), ptr(idx), ptr(tfs), 0 if tfs is None else tfs.shape[1], 1 if tf_per_sample else 0,
            ptr(st.affine) if (tfs is not None and st.affine is not None) else None, ptr(st.shadow), ptr(ws.coords_t),
            ptr(ws.dlat), ptr(ws.dcoords), n, npad, S, Cc, lx, ly, lz, cond_dims, d.LP, ptr(dlatents), ptr(dtfs), ptr(dcond),
            ptr(st.work))
    if pending is not None:
        if not grouped:
            raise ValueError("k1_backward: a pending weight-gradient reduction rides with the grouped entry point only")
        check(lib.cinema_affine_trilerp_bwd_sorted_pending(*args, C.byref(pending), eng.stream()),
              "cinema_affine_trilerp_bwd_sorted_pending")
    else:
        fn = lib.cinema_affine_trilerp_bwd_sorted if grouped else lib.cinema_affine_trilerp_bwd
        check(fn(*args, eng.stream()), "cinema_affine_trilerp_bwd")
    return dlatents, dtfs, dcond


def sample_loss(eng_or_device, out: torch.Tensor, n: int, sr_dims: int, n_cls: int, loss_metric: str,
                class_w: Optional[torch.Tensor], sr_scale: float, seg_scale: float, dout: torch.Tensor,
                sums: torch.Tensor, target: Optional[torch.Tensor] = None, voxel: Optional[torch.Tensor] = None,
                subj: Optional[torch.Tensor] = None, intens: Optional[torch.Tensor] = None,
                labels: Optional[torch.Tensor] = None, voxels_per_subject: int = 0,
                target_out: Optional[torch.Tensor] = None, prepared: bool = False, norm: Optional[torch.Tensor] = None):
    """K3: GT sampling + loss sums + dLoss/dOutput in one launch (see include/cinema_hip.h).
    prepared: `sums` was set up by step_prepare (no memset, no class-weight pre-pass).
    norm: device fp32[2] {samples, class-weight sum} per rank of the union batch (data-parallel ranks; needs prepared)."""
    lib = _lib.load()
    device = out.device
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    npad = min(round_up(n, ROW_ALIGN), dout.shape[0])
    if norm is not None:
        if not prepared or norm.dtype != torch.float32 or norm.numel() != 2 or norm.device != device:
            raise ValueError("sample_loss: norm is a device fp32[2] tensor and needs prepared sums")
        check(lib.cinema_sample_loss_fwd_bwd_normed(ptr(out), ptr(target), ptr(voxel), ptr(subj), ptr(intens), ptr(labels),
                                                    voxels_per_subject, n, npad, sr_dims, n_cls, 0 if loss_metric == "mse" else 1,
                                                    ptr(class_w), float(sr_scale), float(seg_scale), ptr(sums), ptr(dout),
                                                    ptr(target_out), ptr(norm), stream), "cinema_sample_loss_fwd_bwd_normed")
        return
    fn = lib.cinema_sample_loss_fwd_bwd_prepared if prepared else lib.cinema_sample_loss_fwd_bwd
    check(fn(ptr(out), ptr(target), ptr(voxel), ptr(subj), ptr(intens), ptr(labels),
             voxels_per_subject, n, npad, sr_dims, n_cls, 0 if loss_metric == "mse" else 1,
             ptr(class_w), float(sr_scale), float(seg_scale), ptr(sums), ptr(dout),
             ptr(target_out), stream), "cinema_sample_loss_fwd_bwd")


def step_prepare(dcoords: Optional[torch.Tensor], sums: torch.Tensor, n: int, n_cls: int,
                 class_w: Optional[torch.Tensor], voxel: Optional[torch.Tensor], subj: Optional[torch.Tensor],
                 labels: Optional[torch.Tensor], voxels_per_subject: int, scratch: torch.Tensor):
    """Per-step clearing (dcoords, loss sums) + the class-weight normaliser of the batch in one launch on the
    CURRENT stream (the trainer runs it on its side stream, beside the forward).
    scratch: fp32 [_lib.PREPARE_SCRATCH], zeroed once at allocation."""
    if scratch.numel() < _lib.PREPARE_SCRATCH or scratch.dtype != torch.float32:
        raise ValueError("step_prepare: scratch must be fp32 with at least PREPARE_SCRATCH entries")
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(sums.device).cuda_stream)
    check(lib.cinema_step_prepare(ptr(dcoords), 0 if dcoords is None else dcoords.numel(), ptr(sums), ptr(voxel), ptr(subj),
                                  ptr(labels), voxels_per_subject, n, n_cls, ptr(class_w), ptr(scratch), stream), "cinema_step_prepare")
