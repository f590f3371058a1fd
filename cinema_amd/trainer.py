"""Data-parallel hot loop of CINeMA's AtlasBuilder (reference: build_atlas.py:48-97, 414-463) on MI355X.

One process per GPU.  Subjects -- their latent grids, rigid/affine embeddings, resident volumes,
foreground lists and AdamW moments -- are SHARDED across ranks (each subject lives on exactly one
rank); the SIREN weights are replicated and only their 8.46 MB flat gradient is all-reduced
(RCCL over xGMI through torch.distributed), overlapped with the latent-side backward + optimiser.

A step is the reference's inner loop body (build_atlas.py:69-90): zero_grad -> forward -> loss ->
backward -> AdamW, on `n_samples` coordinates sliced from the epoch's shuffled foreground list.
It calls the C ABI directly (no autograd graph): pack -> K1 -> SIREN fwd -> K3 -> SIREN bwd ->
K1 bwd -> optimiser.
"""
from __future__ import annotations

import os

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib
from . import engine as E
from . import parallel as par
from .models.inr_decoder import INR_Decoder
from .optim import HipAdamW


class SubjectStore:
    """Resident ground truth of this rank's subjects + the epoch's shuffled sample list.

    intens (S*V, sr_dims) fp32 and labels (S*V,) uint8 are the (normalised) modality volumes in
    C-order; `coords`, `voxel`, `subj` list every foreground voxel of every local subject
    (data_loading/dataset.py:83-98) and are permuted once per epoch like collate_fn
    (dataset.py:36-47) so that a step is a contiguous slice."""

    def __init__(self, intens: torch.Tensor, labels: torch.Tensor, voxels_per_subject: int, coords: torch.Tensor,
                 voxel: torch.Tensor, subj: torch.Tensor, conditions: Optional[torch.Tensor] = None,
                 n_subjects: Optional[int] = None):
        # voxels_per_subject = 0: compact tables (cinema_amd/dataset.py) -- `voxel` is then a global row of intens/labels
        self.intens, self.labels, self.V = intens, labels, int(voxels_per_subject)
        self.coords, self.voxel, self.subj = coords, voxel, subj
        self.conditions = conditions          # (S, cond_dims) normalised, or None
        self.group_chunk = 0                  # > 0: every chunk of this many samples is sorted by subject
        self.n_subjects = int(n_subjects) if n_subjects is not None else labels.numel() // self.V

    def __len__(self):
        return self.coords.shape[0]

    def shuffle(self, generator: Optional[torch.Generator] = None, group_chunk: Optional[int] = None):
        """Random permutation of the sample list (the reference's collate_fn shuffle).  With group_chunk = n_samples
        every chunk [k*n, (k+1)*n) -- one step's batch -- is additionally ordered by subject: a batch is a set (the
        step's loss is a mean over it), and a grouped batch lets the backward skip its counting sort and read each
        subject's rows in place (cinema_affine_trilerp_bwd_sorted)."""
        m = len(self)
        perm = torch.randperm(m, device=self.coords.device, generator=generator)
        if group_chunk:
            key = (torch.arange(m, device=perm.device) // int(group_chunk)) * self.n_subjects + self.subj[perm].long()
            perm = perm[torch.argsort(key, stable=True)]
            del key
        self.coords = self.coords[perm]
        self.voxel = self.voxel[perm]
        self.subj = self.subj[perm]
        self.group_chunk = int(group_chunk) if group_chunk else 0

    @staticmethod
    def synthetic(n_subjects: int, shape, spacing: float, world_bbox, sr_dims: int, n_labels: int, seed: int, device,
                  cond_dims: int = 0, cond_scale: float = 0.05):
        from . import synthetic as syn
        V = shape[0] * shape[1] * shape[2]
        intens = torch.empty(n_subjects * V, sr_dims, device=device, dtype=torch.float32)
        labels = torch.empty(n_subjects * V, device=device, dtype=torch.uint8)
        cs, vs, ss = [], [], []
        for s in range(n_subjects):
            it, lb, aff = syn.make_subject(shape, spacing, sr_dims, n_labels, seed * 100003 + s, device)
            intens[s * V:(s + 1) * V] = it
            labels[s * V:(s + 1) * V] = lb
            c, v = syn.foreground_lists(lb, shape, aff, world_bbox)
            cs.append(c)
            vs.append(v)
            ss.append(torch.full((v.shape[0],), s, dtype=torch.int32, device=device))
            del it, lb
        cond = None
        if cond_dims:
            g = torch.Generator().manual_seed(seed + 17)
            cond = ((torch.rand(n_subjects, cond_dims, generator=g) * 2 - 1) * cond_scale).to(device)
        return SubjectStore(intens, labels, V, torch.cat(cs), torch.cat(vs), torch.cat(ss), cond)


class HotPathTrainer:
    def __init__(self, args: dict, store: SubjectStore, device, process_group=None, split: str = "train",
                 decoder: Optional[INR_Decoder] = None, keep_grads: bool = False):
        """keep_grads=True leaves the step's gradients in lat_grad / tf_grad / eng.flat_grad for inspection; by default the
        optimiser launch zeroes them as it consumes them (the next step then needs no zero_grad fill)."""
        import torch.distributed as dist
        self.args, self.store, self.device, self.split = args, store, torch.device(device), split
        self.pg = process_group
        self.keep_grads = bool(keep_grads)
        self.rank, self.world = par.world_info(process_group)
        a, o = args["inr_decoder"], args["optimizer"]
        self.sr_dims = sum(a["out_dim"][:-1])
        self.n_cls = a["out_dim"][-1]
        self.cond_dims = a["cond_dims"]
        self.decoder = decoder if decoder is not None else INR_Decoder(args, str(self.device)).to(self.device)
        self.state = self.decoder.sr_net.hot_state(self.device)
        self.eng = self.state.engine
        par.broadcast_(self.eng.flat, 0, process_group)        # replicas start from rank 0's initialisation
        self.k1 = self.state.k1
        S = store.n_subjects
        tf_w = max(a["tf_dim"], 6)
        # parameter tables (reference: build_atlas.py:414-425)
        self.latents = nn.Parameter(torch.normal(0, 0.01, size=(S, *a["latent_dim"])).to(self.device))
        self.tfs = nn.Parameter(torch.zeros(S, tf_w, device=self.device), requires_grad=a["tf_dim"] > 0)
        self.lat_grad = torch.zeros_like(self.latents)
        self.tf_grad = torch.zeros_like(self.tfs)
        self.flat_param = nn.Parameter(self.eng.flat)
        self.flat_param.grad = self.eng.flat_grad
        # Conv3d modulator (cnn_kernel_size > 0, models/inr_decoder.py:109-123): K1 reads conv(latents) instead of the
        # latents; its parameters train in the MLP's group (they are part of inr_decoder.parameters(), build_atlas.py:
        # 432-453) and their gradients are all-reduced like the flat one
        self.conv = self.decoder.modulator.conv if a.get("cnn_kernel_size", 0) > 0 else None
        self.conv_params = []
        if self.conv is not None:
            lib = _lib.load()
            k = a["cnn_kernel_size"]
            nbytes = lib.cinema_conv_mod_work_bytes(S, a["latent_dim"][0], *a["latent_dim"][1:], k, self.eng.desc.precision)
            if nbytes < 0:
                _lib.check(-1, "cinema_conv_mod_work_bytes")
            self.conv_work = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.table = torch.empty_like(self.latents.data)
            self.table_grad = torch.zeros_like(self.latents.data)
            for prm in (self.conv.weight, self.conv.bias):
                par.broadcast_(prm.data, 0, process_group)
                prm.grad = torch.zeros_like(prm.data)
                self.conv_params.append(prm)
        self.train_mlp = split == "train"
        self.seg_weight = o["seg_weight"] if split == "train" else 0.0      # build_atlas.py:63
        cw = args["dataset"]["class_weights"]
        self.class_w = torch.tensor(cw, dtype=torch.float32, device=self.device) if cw is not None else None
        self.cond_param = None
        if split == "val" and self.cond_dims:
            self.cond_param = nn.Parameter(torch.normal(0, 0.01, size=(S, self.cond_dims)).to(self.device))
        groups = [{"params": [self.latents], "lr": o["lr_latent"], "weight_decay": o["latent_weight_decay"]}]
        if a["tf_dim"] > 0:
            groups.append({"params": [self.tfs], "lr": o["lr_tf"], "weight_decay": o["tf_weight_decay"]})
        if self.cond_param is not None:
            groups.append({"params": [self.cond_param], "lr": o["lr_latent"], "weight_decay": o["latent_weight_decay"]})
        mlp_group = {"params": [self.flat_param] + self.conv_params, "lr": o["lr_inr"], "weight_decay": o["inr_weight_decay"]}
        if self.train_mlp and self.world == 1:
            # one process: nothing to wait for between the table-side and the MLP update -> one launch for all groups
            self.opt_local = HipAdamW(groups + [mlp_group], clear_grads=not self.keep_grads)
            self.opt_mlp = None
        else:
            self.opt_local = HipAdamW(groups, clear_grads=not self.keep_grads)
            self.opt_mlp = HipAdamW([mlp_group], clear_grads=not self.keep_grads) if self.train_mlp else None
        self._base_lrs = [[g["lr"] for g in opt.param_groups] for opt in (self.opt_local, self.opt_mlp) if opt is not None]
        self._build_schedulers()
        self.sums = torch.zeros(4, device=self.device)
        self.norm = torch.zeros(2, device=self.device)               # union-batch normaliser of a data-parallel rank
        self.cond_grad = None
        # Grouped batches on the bf16 chain kernels need no second stream: everything in front of the forward is ONE
        # launch (cinema_step_head).  The fallback shapes keep the backward-only preparation (transposed linear_lats copies,
        # node-major shadow, per-step clearing + class-weight normaliser) on a side stream that joins before the loss
        # kernel; a rank of several keeps one stream there too (the collective library brings its own).
        self.side = torch.cuda.Stream(device=self.device) if self.world == 1 else None
        self.prep_scratch = torch.zeros(_lib.PREPARE_SCRATCH, device=self.device)
        self.ev_prep = torch.cuda.Event()
        self.ev_head = torch.cuda.Event()
        self.split_backward = False      # tests: a single rank walks the call order of a rank of several (weights first, table side second)
        self.pending = C.create_string_buffer(_lib.PENDING_BYTES)    # host-side hand-over of the deferred reduction
        self.nodes = E.NodeTables()                                    # node-space tables of the nodes path
        self.ng = E.NodeGrad()                                         # work buffer of the node-space backward
        self.n_last, self.lo_last = 0, 0
        self.tf_weight = o["tf_weight"]
        self.loss_metric = "mse" if o["loss_metric"] == "mse" else "l1"

    # ------------------------------------------------------------------------------------
    def step(self, lo: int, hi: int):
        """One optimisation step on samples [lo, hi) of the epoch list (build_atlas.py:69-88)."""
        import torch.distributed as dist
        st, eng = self.store, self.eng
        coords, voxel, subj = st.coords[lo:hi], st.voxel[lo:hi], st.subj[lo:hi]
        n = coords.shape[0]
        ws = eng.workspace(n)
        ws.ensure_backward()
        cond, cond_table = None, None
        if self.cond_dims:
            cond_table = (self.cond_param.data if self.cond_param is not None else st.conditions).contiguous()
        tfs = self.tfs.data
        table = self._latent_table()
        main = torch.cuda.current_stream(self.device)
        use_cw = self.class_w is not None and self.seg_weight > 0
        grouped = st.group_chunk > 0 and lo % st.group_chunk == 0 and hi - lo <= st.group_chunk
        fast_k1 = grouped and E.k1_sorted_eligible(tuple(self.latents.shape), eng.desc.LP)

        if cond_table is not None and not fast_k1:
            cond = cond_table[subj.long()].contiguous()  # generic K1 entry: per-sample rows (the grouped one gathers in the kernel)
        one_head = fast_k1 and eng.can_skip_saved()      # grouped batch on the activation-stationary bf16 path
        # nodes path: FiLM modulation from per-subject node tables (conditions are per subject here)
        use_nodes = grouped and self.conv is None and E.nodes_eligible(eng, tuple(self.latents.shape))
        if use_nodes:
            return self._step_nodes(lo, hi)
        # node gradients: the latent side of the backward (d table, d linear_lats, d coords through the interpolation)
        # reduced in node space -- one call instead of SIREN backward + K1 backward
        use_ng = grouped and E.nodegrad_eligible(eng, tuple(table.shape))

        def k1_fwd():
            if fast_k1:
                E.k1_forward_sorted(eng, coords, subj, table, cond_table, tfs, n, cond_by_subject=True)
            else:
                E.k1_forward(eng, self.k1, coords, subj, table, cond, tfs, False, n)

        def backward_only_prep():
            eng.pack(2 | (8 if use_ng else 0))                    # transposed (and row-major) linear_lats copies,
            if fast_k1 and not use_ng:
                E.k1_shadow(eng, self.k1, table)                  # node-major shadow (only the K1 backward reads it),
            E.step_prepare(ws.dcoords, self.sums, n, self.n_cls, self.class_w if use_cw else None, voxel, subj, st.labels, st.V,
                           self.prep_scratch)                     # cleared sums / dcoords + class-weight normaliser

        joined = True
        if one_head:
            # everything in front of the forward as ONE launch (cinema_step_head): no second stream, no join
            E.step_head(eng, self.k1, coords, subj, table, cond_table, tfs, n, cond_by_subject=True,
                        prepare=dict(dcoords=ws.dcoords, sums=self.sums, n_cls=self.n_cls, class_w=self.class_w if use_cw else None,
                                     voxel=voxel, labels=st.labels, V=st.V, scratch=self.prep_scratch), shadow=not use_ng,
                        ng=self.ng if use_ng else None)
        elif self.side is not None:
            side = self.side
            side.wait_stream(main)                  # the previous step's optimiser wrote the parameters
            with torch.cuda.stream(side):
                backward_only_prep()                # nothing the forward reads: beside it, joined before the loss kernel
                self.ev_prep.record(side)
            eng.pack(1)                             # what the forward reads stays on this stream (a cross-stream wait
            k1_fwd()                                # costs 6-10 us of idle queue)
            joined = False
        else:
            eng.pack(1)
            k1_fwd()
            backward_only_prep()
        norm = self._union_normaliser(n, use_cw, joined)      # rank of several: started here, hidden under the forward
        ng_on_side = use_ng and one_head and self.side is not None and os.environ.get("CINEMA_NG_SIDE", "0") == "1"
        if ng_on_side:
            # node tables of the backward (interpolation weights, node rows, centred projections): they need the packed
            # weights of the head launch and nothing of the forward -> beside it, joined before the backward
            self.ev_head.record(main)
            self.side.wait_event(self.ev_head)
            with torch.cuda.stream(self.side):
                E.nodegrad_prepare(eng, self.ng, coords, subj, table, cond_table, tfs, n, rows_ready=True)
                self.ev_prep.record(self.side)
        elif use_ng and not one_head:
            E.nodegrad_prepare(eng, self.ng, coords, subj, table, cond_table, tfs, n)     # one launch (+ the node rows), in line
        if use_ng and one_head and not ng_on_side:
            # the head wrote node rows and interpolation weights; the projections ride in the forward launch
            eng.forward(n, ng=self.ng, latents_shape=tuple(table.shape))
        else:
            eng.forward(n)
        if not joined:
            main.wait_event(self.ev_prep)
        E.sample_loss(None, ws.out, n, self.sr_dims, self.n_cls, self.loss_metric, self.class_w, 1.0,
                      float(self.seg_weight) ** 2, ws.dout, self.sums, voxel=voxel, subj=subj, intens=st.intens,
                      labels=st.labels, voxels_per_subject=st.V, prepared=True, norm=None if norm is None else norm.wait())
        # one rank: nobody needs the flat gradient before the optimiser, so the split reduction of the weight-gradient
        # tiles rides with the K1 backward's first launch (beside the coordinate gradient) instead of a launch of its own
        pending = self.pending if (grouped and self.train_mlp and self.world == 1 and not self.split_backward) else None
        work = None
        want_tf = self.tfs.requires_grad
        if ng_on_side:
            main.wait_event(self.ev_prep)
        if use_ng:
            if self.cond_param is not None and self.cond_grad is None:
                self.cond_grad = torch.zeros_like(self.cond_param)
            ng_args = dict(ng=self.ng, coords=coords, idx=subj, tfs=tfs, latents_shape=tuple(table.shape), cond_dims=self.cond_dims,
                           dlatents=self.lat_grad if self.conv is None else self.table_grad, dtfs=self.tf_grad if want_tf else None,
                           dcond=self.cond_grad if self.cond_param is not None else None)
            if self.train_mlp and (self.world > 1 or self.split_backward):
                eng.backward_nodegrad(n, ws.dout, weight_grads=True, dcoords_cleared=True, part=_lib.BWD_WEIGHTS_FIRST, **ng_args)
                if self.world > 1:
                    work = par.allreduce_mean_(eng.flat_grad, self.pg, async_op=True)
                eng.backward_nodegrad(n, ws.dout, weight_grads=True, part=_lib.BWD_DLAT_ONLY, **ng_args)
            else:
                eng.backward_nodegrad(n, ws.dout, weight_grads=self.train_mlp, dcoords_cleared=True, **ng_args)
        elif self.train_mlp and (self.world > 1 or self.split_backward):
            # rank of several: weight gradients first, their all-reduce starts at once and runs beside the d-latent GEMM,
            # the K1 backward and the table-side AdamW
            eng.backward(n, ws.dout, weight_grads=True, dcoords_cleared=True, part=_lib.BWD_WEIGHTS_FIRST)
            if self.world > 1:
                work = par.allreduce_mean_(eng.flat_grad, self.pg, async_op=True)
            eng.backward(n, ws.dout, weight_grads=True, part=_lib.BWD_DLAT_ONLY)
        else:
            eng.backward(n, ws.dout, weight_grads=self.train_mlp, dcoords_cleared=True, pending=pending)
        dcond = None
        if not use_ng:
            _, _, dcond = E.k1_backward(eng, self.k1, coords, subj, tuple(self.latents.shape), self.cond_dims,
                                        tfs if want_tf else None, False, n, want_latents=True,
                                        want_cond=self.cond_param is not None, dtf_out=self.tf_grad if want_tf else None,
                                        dlat_out=self.lat_grad if self.conv is None else self.table_grad, grouped=grouped,
                                        pending=pending)
        work_conv = []
        if self.conv is not None:
            self._latent_table_backward()
            if self.train_mlp and self.world > 1:
                work_conv = [par.allreduce_mean_(prm.grad, self.pg, async_op=True) for prm in self.conv_params]
        if want_tf and self.tf_weight != 0.0:
            self._add_tf_regulariser_grad(subj, n)
        self.latents.grad = self.lat_grad
        if want_tf:
            self.tfs.grad = self.tf_grad
        if self.cond_param is not None:
            self.cond_param.grad = self.cond_grad if use_ng else self._cond_table_grad(subj, dcond, n, grouped)
        self.opt_local.step()
        for w in [work] + work_conv:
            if w is not None:
                w.wait()
        if self.opt_mlp is not None:
            self.opt_mlp.step()
        if self.train_mlp and not self.keep_grads:
            eng.grads_clean = True                   # the clearing AdamW consumed and zeroed flat_grad
        self.n_last, self.lo_last = n, lo

    def _latent_table(self):
        """What K1 interpolates: the latents, or their Conv3d modulation (one C-ABI call per step)."""
        if self.conv is None:
            return self.latents.data
        a = self.args["inr_decoder"]
        S, Cc, lx, ly, lz = self.latents.shape
        _lib.check(self.eng.lib.cinema_conv_mod_forward(_lib.ptr(self.latents.data), _lib.ptr(self.conv.weight.data),
                                                        _lib.ptr(self.conv.bias.data), S, Cc, lx, ly, lz, a["cnn_kernel_size"],
                                                        self.eng.desc.precision, _lib.ptr(self.conv_work), _lib.ptr(self.table),
                                                        self.eng.stream()), "cinema_conv_mod_forward")
        return self.table

    def _latent_table_backward(self):
        """table gradient (left by the K1 backward in self.table_grad) -> latents, conv.weight, conv.bias gradients."""
        a = self.args["inr_decoder"]
        S, Cc, lx, ly, lz = self.latents.shape
        _lib.check(self.eng.lib.cinema_conv_mod_backward(_lib.ptr(self.latents.data), _lib.ptr(self.conv.weight.data),
                                                         _lib.ptr(self.table_grad), S, Cc, lx, ly, lz, a["cnn_kernel_size"],
                                                         self.eng.desc.precision, _lib.ptr(self.conv_work), _lib.ptr(self.lat_grad),
                                                         _lib.ptr(self.conv.weight.grad), _lib.ptr(self.conv.bias.grad),
                                                         self.eng.stream()), "cinema_conv_mod_backward")

    def _cond_table_grad(self, subj, dcond, n: int, grouped: bool):
        """d loss / d conditions_val: the per-sample rows summed per subject (in the kernel for grouped batches)."""
        if self.cond_grad is None:
            self.cond_grad = torch.zeros_like(self.cond_param)
        if grouped:
            E.cond_table_grad(subj, dcond, n, self.cond_grad)
        else:
            self.cond_grad.zero_()
            self.cond_grad.index_add_(0, subj.long(), dcond[:n])
        return self.cond_grad

    def _union_normaliser(self, n: int, use_cw: bool, prepared_on_main: bool):
        """Rank of several: {samples, class-weight sum} of the union batch / world (parallel.share_batch_normaliser_), as
        the loss kernel's normaliser.  None for one process."""
        if self.world == 1:
            return None
        if not prepared_on_main:
            torch.cuda.current_stream(self.device).wait_event(self.ev_prep)      # sums[2] comes from the side stream
        self.norm[0] = float(n)
        if use_cw:
            self.norm[1:2].copy_(self.sums[2:3])
        else:
            self.norm[1] = float(n)
        return par.share_batch_normaliser_(self.norm, self.pg)

    def _step_nodes(self, lo: int, hi: int):
        """step() on the nodes path (grouped batch, bf16 chain kernels)."""
        st, eng = self.store, self.eng
        coords, voxel, subj = st.coords[lo:hi], st.voxel[lo:hi], st.subj[lo:hi]
        n = coords.shape[0]
        ws = eng.workspace(n)
        ws.ensure_backward()
        cond_tab = None
        cond = None
        if self.cond_dims:
            cond_tab = (self.cond_param.data if self.cond_param is not None else st.conditions).contiguous()
        tfs = self.tfs.data
        use_cw = self.class_w is not None and self.seg_weight > 0
        nt = self.nodes
        eng.pack(4 | 2)
        E.k1_forward_sorted(eng, coords, subj, self.latents.data, cond_tab, tfs, n, cond_by_subject=True)   # lat rows: d-latent / dW GEMMs
        E.node_weights(eng, nt, coords, subj, tuple(self.latents.shape), tfs, n)
        E.node_tables(eng, nt, self.latents.data, cond_tab)
        E.k1_shadow(eng, self.k1, self.latents.data)
        E.step_prepare(ws.dcoords, self.sums, n, self.n_cls, self.class_w if use_cw else None, voxel, subj, st.labels, st.V,
                       self.prep_scratch)
        norm = self._union_normaliser(n, use_cw, True)
        eng.forward_nodes(n, nt, subj)
        E.sample_loss(None, ws.out, n, self.sr_dims, self.n_cls, self.loss_metric, self.class_w, 1.0,
                      float(self.seg_weight) ** 2, ws.dout, self.sums, voxel=voxel, subj=subj, intens=st.intens,
                      labels=st.labels, voxels_per_subject=st.V, prepared=True, norm=None if norm is None else norm.wait())
        work = None
        if self.train_mlp and self.world > 1:
            eng.backward(n, ws.dout, weight_grads=True, dcoords_cleared=True, nodes=(nt, subj), part=_lib.BWD_WEIGHTS_FIRST)
            work = par.allreduce_mean_(eng.flat_grad, self.pg, async_op=True)
            eng.backward(n, ws.dout, weight_grads=True, nodes=(nt, subj), part=_lib.BWD_DLAT_ONLY)
        else:
            eng.backward(n, ws.dout, weight_grads=self.train_mlp, dcoords_cleared=True, nodes=(nt, subj))
        want_tf = self.tfs.requires_grad
        _, _, dcond = E.k1_backward(eng, self.k1, coords, subj, tuple(self.latents.shape), self.cond_dims,
                                    tfs if want_tf else None, False, n, want_latents=True,
                                    want_cond=self.cond_param is not None, dtf_out=self.tf_grad if want_tf else None,
                                    dlat_out=self.lat_grad, grouped=True)
        if want_tf and self.tf_weight != 0.0:
            self._add_tf_regulariser_grad(subj, n)
        self.latents.grad = self.lat_grad
        if want_tf:
            self.tfs.grad = self.tf_grad
        if self.cond_param is not None:
            self.cond_param.grad = self._cond_table_grad(subj, dcond, n, True)
        self.opt_local.step()
        if work is not None:
            work.wait()
        if self.opt_mlp is not None:
            self.opt_mlp.step()
        if self.train_mlp and not self.keep_grads:
            eng.grads_clean = True
        self.n_last, self.lo_last = n, lo

    def _build_schedulers(self):
        """build_atlas.py:455-457: fresh CosineAnnealingLR objects on the configured learning rates (also after a checkpoint
        load: the reference builds optimiser and scheduler anew there)."""
        o = self.args["optimizer"]
        self.schedulers = []
        opts = [opt for opt in (self.opt_local, self.opt_mlp) if opt is not None]
        for opt, lrs in zip(opts, self._base_lrs):
            for g, lr in zip(opt.param_groups, lrs):
                g["lr"] = lr
                g.pop("initial_lr", None)
        if o.get("scheduler", {}).get("type") == "cosine":
            T = self.args["epochs"][self.split]
            for opt in opts:
                self.schedulers.append(torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T, eta_min=o["scheduler"]["eta_min"]))

    def _add_tf_regulariser_grad(self, subj, n):
        """d/d table of tf_weight * (mean(tfs[idx,0:3]^2) + mean(tfs[idx,3:6]^2) [+ scale term])  (utils.py:98-103)."""
        counts = torch.bincount(subj.long(), minlength=self.tfs.shape[0]).to(torch.float32).unsqueeze(1)
        w = self.tfs.shape[1]
        scale = torch.zeros(w, device=self.device)
        scale[:6] = 1.0
        if w == 9:
            scale[6:9] = 1.0
        self.tf_grad.add_(self.tfs.data * counts * scale * (2.0 * self.tf_weight / (3.0 * n)))

    def epoch(self, n_samples: int, generator: Optional[torch.Generator] = None, max_steps: Optional[int] = None):
        """One pass over the shuffled foreground list in chunks of n_samples (build_atlas.py:65-68)."""
        self.store.shuffle(generator, group_chunk=n_samples)
        m = len(self.store)
        total = (m + n_samples - 1) // n_samples
        if max_steps is not None:
            total = min(total, max_steps)
        total = par.agree_steps(total, self.pg, self.device)   # every rank must join every gradient all-reduce
        steps = 0
        for lo in range(0, total * n_samples, n_samples):
            self.step(lo, min(lo + n_samples, m))
            steps += 1
        for s in self.schedulers:
            s.step()
        return steps

    # ------------------------------------------------------------------------------------
    # state (build_atlas.py:344-370, 427-430)
    def checkpoint_dict(self, epoch: int) -> dict:
        """The reference's checkpoint dictionary (build_atlas.py:347-355) without its two pandas frames (`tsv_file`,
        `dataset_df`: subject selection is outside this package): tensors and plain containers only, so the file loads with
        `torch.load(weights_only=True)`.  Under a process group the tables of all ranks, in rank order (collective)."""
        return {"epoch": int(epoch),
                "latents": par.gather_rows(self.latents.data, self.pg).cpu(),
                "transformations": par.gather_rows(self.tfs.data, self.pg).cpu(),
                "inr_decoder": {k: v.detach().cpu().clone() for k, v in self.decoder.state_dict().items()},
                "args": self.args}

    def save_state(self, epoch: int, log_dir: Optional[str] = None) -> Optional[str]:
        """checkpoint_epoch_{epoch}.pth in `log_dir` (default args['output_dir']), written by rank 0; honours
        args['save_model'] like the reference."""
        if not self.args.get("save_model", True):
            return None
        st = self.checkpoint_dict(epoch)
        path = os.path.join(log_dir if log_dir is not None else self.args["output_dir"], f"checkpoint_epoch_{epoch}.pth")
        if self.rank == 0:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(st, path)
        return path

    def load_checkpoint(self, chkp_path: str, epoch: int) -> dict:
        """build_atlas.py:360-370: decoder weights and this rank's rows of the tables from checkpoint_epoch_{epoch}.pth;
        the optimiser starts fresh (build_atlas.py:385).  The file is read with `weights_only=True`: a checkpoint written
        by the reference itself carries pickled pandas frames and is refused by that loader -- re-save it without them."""
        path = os.path.join(chkp_path, f"checkpoint_epoch_{epoch}.pth")
        if not os.path.exists(path):
            raise FileNotFoundError(f"State file {path} not found!")
        chkp = torch.load(path, map_location="cpu", weights_only=True)
        S = self.latents.shape[0]
        sizes = par.gather_rows(torch.tensor([[S]], dtype=torch.int64, device=self.device), self.pg).flatten().tolist()
        lo = sum(sizes[:self.rank])
        lat, tfs = chkp["latents"], chkp["transformations"]
        if lat.shape[0] != sum(sizes) or tuple(lat.shape[1:]) != tuple(self.latents.shape[1:]) or \
                tuple(tfs.shape) != (sum(sizes), self.tfs.shape[1]):
            raise ValueError(f"checkpoint tables {tuple(lat.shape)}, {tuple(tfs.shape)} do not fit {sum(sizes)} subjects of "
                             f"{tuple(self.latents.shape[1:])} / {self.tfs.shape[1]}")
        self.decoder.load_state_dict(chkp["inr_decoder"])       # copies into the views of the flat parameter buffer
        with torch.no_grad():
            self.latents.data.copy_(lat[lo:lo + S])
            self.tfs.data.copy_(tfs[lo:lo + S])
        for opt in (self.opt_local, self.opt_mlp):
            if opt is not None:
                opt.state.clear()
        self._build_schedulers()        # a load starts optimiser AND schedule afresh (build_atlas.py:385, 455-457)
        return chkp

    def re_init_latents(self):
        """build_atlas.py:427-430."""
        with torch.no_grad():
            self.latents.data.normal_(0, 0.01)
            self.tfs.data.zero_()
        # optimizer.zero_grad() of the reference covers every parameter group
        grads = [self.lat_grad, self.tf_grad, self.eng.flat_grad, self.cond_grad] + [prm.grad for prm in self.conv_params]
        if self.conv is not None:
            grads.append(self.table_grad)
        if self.cond_param is not None:
            grads.append(self.cond_param.grad)
        for g in grads:
            if g is not None:
                g.zero_()
        self.eng.grads_clean = True

    def loss_dict(self) -> Dict[str, float]:
        """Losses of the LAST step with the reference's eight keys (Criterion.forward, utils.py:88-104): `seg` carries
        seg_weight and `total` applies it again, the tf terms are means over the batch's gathered rows tfs[idcs_df]
        (here: the table's rows weighted by the subjects' sample counts), `total` includes tf_weight * tf.
        sr / seg are the sums the last step's loss kernel left (parameters BEFORE that step's update, as the reference
        logs them); the tf terms are evaluated on the table as it is now, i.e. after that update -- one AdamW step away
        from the reference's logged value (the gradient of the term, _add_tf_regulariser_grad, uses the pre-update
        table).  One device->host sync; call every k steps."""
        s = self.sums.tolist()
        n = max(self.n_last, 1)
        sr = s[0] / (n * self.sr_dims)
        den = s[2] if self.class_w is not None else float(n)
        ce = s[1] / den if (self.seg_weight > 0 and den > 0) else 0.0
        out = {"seg": self.seg_weight * ce, "sr": sr, "trafo": 0.0, "total": 0.0}
        tf = 0.0
        if self.tfs.requires_grad and self.n_last > 0:        # tf_dim > 0: the reference hands tfs[idcs_df] to the criterion
            lo = self.lo_last
            subj = self.store.subj[lo:lo + self.n_last]
            counts = torch.bincount(subj.long(), minlength=self.tfs.shape[0]).to(torch.float64)
            sq = (self.tfs.data.double() ** 2 * counts.unsqueeze(1)).sum(0) / (3.0 * self.n_last)
            sq = sq.tolist()
            out["tf_rot"], out["tf_trans"] = sum(sq[0:3]), sum(sq[3:6])
            out["tf_scale"] = sum(sq[6:9]) if self.tfs.shape[1] == 9 else 0.0
            out["tf"] = tf = out["tf_rot"] + out["tf_trans"] + out["tf_scale"]
        out["total"] = out["sr"] + self.seg_weight * out["seg"] + self.tf_weight * tf
        return out
