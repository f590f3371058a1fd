#!/usr/bin/env python3
"""Headline benchmark: coordinate samples/s through fwd + loss + bwd + AdamW of the conditional-INR
training step (BASELINE.json metric) at the dhcp_neo_mm shape: latent 256x3x3x3, hidden 512,
5 hidden layers (modulated 1,3,5), out 2+10, 64 synthetic subjects of 310^3 voxels at 0.5 mm per GPU,
n_samples = 25 000 coordinates per step per GPU (config_atlas.yaml:126).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (see the contract in the task description): whole-job samples/s with the
inputs resident in HBM, the roofline object of the dominant kernel (timed live with HIP events on the
launch stream) and the CPU baseline (the CPU oracle timed on the host cores of the same box).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0               # HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is what streaming kernels reach


def make_args(hidden, precision, loss_metric, n_samples):
    return {
        "inr_decoder": {"in_dim": 3, "out_dim": [2, 10], "latent_dim": [256, 3, 3, 3], "cond_dims": 0,
                        "hidden_size": hidden, "num_hidden_layers": 5, "modulated_layers": [1, 3, 5], "omega": [30, 30],
                        "cnn_kernel_size": 0, "tf_dim": 6, "precision": precision},
        "optimizer": {"tf_weight": 0.0, "loss_metric": loss_metric, "seg_weight": 1.0, "lr_inr": 1e-4, "inr_weight_decay": 0.0,
                      "lr_latent": 5e-4, "latent_weight_decay": 0.0, "lr_tf": 7.5e-3, "tf_weight_decay": 0.0,
                      "scheduler": {"type": "cosine", "eta_min": 2.5e-5}},
        "dataset": {"class_weights": [0.0] + [1.0] * 9, "world_bbox": [155, 155, 155], "label_names": []},
        "epochs": {"train": 1, "val": 5}, "n_samples": n_samples, "mask_reconstruction": False, "amp": precision == "bf16",
    }


def flops_per_sample(hidden=512, lat=256, layers=5, n_mod=3, out=12, in_dim=3):
    """Algorithmic MACs of SURVEY.md section 8d: fwd = in*H + L*H^2 + |mod|*lat*2H + (H+lat)*out; fwd+bwd = 3x."""
    mac = in_dim * hidden + layers * hidden * hidden + n_mod * lat * 2 * hidden + (hidden + lat) * out
    return 2 * mac, 6 * mac


def cpu_baseline(args, trainer, n_samples, lat0, tf0, flat0, budget_s=25.0):
    """The CPU oracle (a port of the reference path, pinned to the reference by tests/golden) on the host cores, on the
    bench's own first batch: the store's coordinates, subject ids and sampled targets, the trainer's initial parameters.
    All host threads first, then ONE thread on a quarter of the batch (SURVEY.md section 8d asks for both)."""
    from oracle import cinema_oracle as O
    a = args["inr_decoder"]
    st = trainer.store
    n = n_samples
    coords, subj = st.coords[:n].cpu(), st.subj[:n].cpu().long()
    base = st.subj[:n].long() * st.V + st.voxel[:n]
    target = torch.cat((st.intens[base], st.labels[base].float().unsqueeze(1)), 1).cpu()
    params = {k: v.clone().requires_grad_(True) for k, v in trainer.eng.param_views(flat0).items()}
    lat, tft = lat0.clone().requires_grad_(True), tf0.clone().requires_grad_(True)
    cfg = dict(num_hidden_layers=a["num_hidden_layers"], omega=[30, 30], sr_dims=2, loss_metric=args["optimizer"]["loss_metric"],
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=0.0)

    def timed(m, max_iters, budget):
        times, t_start = [], time.time()
        for it in range(max_iters):
            for p in list(params.values()) + [lat, tft]:
                p.grad = None
            t0 = time.time()
            O.training_step(params, lat, tft, coords[:m], target[:m], subj[:m], None, cfg)
            if it > 0:
                times.append(time.time() - t0)
            if time.time() - t_start > budget and len(times) >= 1:
                break
        times.sort()
        return times[len(times) // 2], len(times)

    cores = torch.get_num_threads()
    med, k = timed(n, 6, budget_s * 0.6)
    torch.set_num_threads(1)
    m1 = max(n // 4, 1)
    med1, k1 = timed(m1, 3, budget_s * 0.4)
    torch.set_num_threads(cores)
    return {"value": n / med, "unit": "coord samples/s", "cores": cores, "kind": "port",
            "sample": f"{k} timed fwd+loss+bwd steps (after 1 warm-up) of the CPU oracle on the bench's first batch "
                      f"(N={n}, S={lat0.shape[0]}, fp32, torch CPU ops, median)",
            "one_thread": {"value": m1 / med1, "cores": 1, "sample": f"{k1} timed step(s) on the first {m1} samples of that batch"}}


PROFILE_TAG = "r03"                 # profiles/<tag>_*: rocprofv3 passes of `bench.py --profile-step` on this tree (tools/profile_round.sh)
PMC_FILES = {"FETCH_SIZE": PROFILE_TAG + "_pmc_fetch.txt", "WRITE_SIZE": PROFILE_TAG + "_pmc_write.txt"}


def _pmc_values(kernel=None):
    """{kernel line -> bytes} from the committed PMC passes: counters are in KiB; FETCH_SIZE is doubled as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (128-B requests tallied at 64 B),
    WRITE_SIZE is exact.  None when the files are absent."""
    per = {}
    for key, name in PMC_FILES.items():
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None
        for line in open(path):
            if key + "=" in line and (kernel is None or line.startswith(kernel)):
                per.setdefault(line.split("(")[0].split("<")[0].strip(), {})[key] = float(line.split(key + "=")[1].split()[0])
    out = {k: (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 for k, v in per.items() if len(v) == 2}
    return out or None


def pmc_traffic(kernel: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this workload (separate --pmc runs
    of `bench.py --profile-step`).  A live run cannot collect counters on itself, so this is the profile of the same
    command, not of this process; None when the files are absent."""
    v = _pmc_values(kernel)
    return None if not v else sum(v.values())


def instep_ms(kernel: str):
    """Average duration of `kernel` INSIDE the training step (cold L2, neighbours' tails) from the committed
    rocprofv3 --kernel-trace --stats summary of `bench.py --profile-step` (profiles/<tag>_kernel_stats.csv); None if absent."""
    import csv
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_kernel_stats.csv")
    if not os.path.exists(path):
        return None
    tot = None
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if name.startswith(kernel):
            tot = (tot or 0.0) + float(r["AverageNs"]) * 1e-6
    return tot


def event_ms(fn, reps):
    """Average duration of fn() over `reps` back-to-back calls, HIP events on the stream the kernels are launched on
    (the library launches on torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def box_peaks(dev):
    """What this box reaches with library code, beside the vendor peaks: a large bf16 GEMM through hipBLASLt (torch.matmul)
    and a 1 GiB device-to-device copy (read + write bytes)."""
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    ms = event_ms(lambda: torch.matmul(a, b), 10)
    src = torch.empty(1 << 28, device=dev, dtype=torch.float32)
    dst = torch.empty_like(src)
    ms_c = event_ms(lambda: dst.copy_(src), 10)
    return {"mfma_bf16_tflops_measured": 2.0 * 8192 ** 3 / (ms * 1e-3) / 1e12, "mfma_probe": "torch.matmul bf16 8192^3 (hipBLASLt)",
            "stream_copy_gbs_measured": 2.0 * src.numel() * 4 / (ms_c * 1e-3) / 1e9, "stream_probe": "1 GiB device-to-device copy, read + write"}


def rooflines(trainer, n, opt, peak):
    """Per-kernel rooflines of one training step.  SIREN kernels against the MFMA peak with the
    ALGORITHMIC flops of what they compute (not the flops executed: padding counts as the reference's), the HBM-bound ones
    (k_node_grad, K1, K3) against HBM with their compulsory bytes.  `avg_launch_ms`: the kernel timed alone here, `reps`
    launches back to back, HIP events on the launch stream; `in_step_ms`: the same kernel inside the step, from the
    committed rocprofv3 summary of this command (profiles/<tag>_kernel_stats.csv); `frac` uses the in-step time when
    the profile is there (it is the one that adds up to the step), else the back-to-back one."""
    import ctypes as C
    from cinema_amd import _lib
    from cinema_amd import engine as E
    eng, st = trainer.eng, trainer.store
    ws, d, reps = eng.ws, eng.desc, opt.kernel_reps
    H, L, nl, od, ind = d.hidden, d.lat, d.n_layers - 1, d.out_dim, d.in_dim
    n_mod = bin(d.modulated_mask).count("1")
    mac_main, mac_mod, mac_fin, mac_first = nl * H * H, n_mod * L * 2 * H, (H + L) * od, ind * H
    mac_fwd = mac_first + mac_main + mac_mod + mac_fin
    bf16 = opt.precision == "bf16"
    coords, voxel, subj = st.coords[:n], st.voxel[:n], st.subj[:n]
    tfs, lat = trainer.tfs.data, trainer.latents.data
    S = lat.shape[0]
    grouped = st.group_chunk > 0
    use_ng = grouped and E.nodegrad_eligible(eng, tuple(lat.shape))
    profiled = bf16 and n == 25000          # the committed profiles are of the default command

    def obj(kernel, what, bound, work, unit_work, ms, names, executed=None):
        ins = sum(filter(None, (instep_ms(k) for k in names))) if profiled else None
        ins = ins or None
        t = ins if ins else ms
        ach = n * work / (t * 1e-3) / (1e12 if bound == "mfma" else 1e9)
        pk = peak if bound == "mfma" else HBM_PEAK_GBS
        tr = [pmc_traffic(k) for k in names] if profiled else [None]
        o = {"kernel": f"{kernel} ({what})", "bound": bound, "achieved": ach, "peak": pk, "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
             "frac": ach / pk, "traffic": sum(tr) if all(x is not None for x in tr) else None, "avg_launch_ms": ms,
             "in_step_ms": ins, "frac_from": "in_step_ms" if ins else "avg_launch_ms", unit_work: work}
        if executed is not None:
            o["flop_per_sample_executed"] = executed
        return o

    objs = []
    if use_ng:
        # the default path: chain kernels + node-space reductions (include/cinema_hip.h, "node gradients")
        ng = trainer.ng
        eng.pack(3 | 8)
        E.k1_forward_sorted(eng, coords, subj, lat, None, tfs, n)
        E.nodegrad_prepare(eng, ng, coords, subj, lat, None, tfs, n)
        npad = E.round_up(n, 128)
        lib = eng.lib
        lib.cinema_debug_node_grad.argtypes = [C.POINTER(_lib.SirenDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                               C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        lx, ly, lz = lat.shape[2:]
        ng_args = dict(ng=ng, coords=coords, idx=subj, tfs=tfs, latents_shape=tuple(lat.shape), cond_dims=0,
                       dlatents=trainer.lat_grad, dtfs=trainer.tf_grad, dcond=None)
        F = n_mod * 2 * d.HP
        node_rows = S * 32 / n          # node rows per sample row
        ms = event_ms(lambda: eng.forward(n), reps)
        objs.append(obj("k_fused_fwd", "forward stack incl. final_linear", "mfma", 2 * mac_fwd, "flop_per_sample", ms, ["k_fused_fwd"]))
        ms = event_ms(lambda: eng.backward(n, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN), reps)
        # hidden 256 / 512: k_fused_bwdw (fused_wide.hip: workgroups of up to 112 samples); the name filter matches both
        objs.append(obj("k_fused_bwdw" if d.HP in (256, 512) else "k_fused_bwd", "backward chain: dX of every layer", "mfma",
                        2 * (mac_main + mac_first + H * od), "flop_per_sample", ms, ["k_fused_bwd"]))
        ms = event_ms(lambda: eng.backward_nodegrad(n, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_DW,
                                                    **ng_args), reps)
        objs.append(obj("k_dw256 + k_ng_tail", "all weight-gradient GEMMs (linear_lats on S*32 node rows) + split reduction, table / "
                        "embedding gradients", "mfma", 2 * mac_fwd, "flop_per_sample", ms, ["k_dw256", "k_ng_tail"],
                        executed=2 * (mac_first + mac_main + H * od) + 2 * node_rows * (mac_mod + L * od)))
        ms = event_ms(lambda: _lib.check(lib.cinema_debug_node_grad(C.byref(d), _lib.ptr(ng.work), _lib.ptr(ws.gbuf), _lib.ptr(ws.coords_t),
                                                                     n, npad, S, lx, ly, lz, 0, _lib.ptr(eng.packed), _lib.ptr(eng.flat),
                                                                     eng.stream()), "k_node_grad"), reps)
        # what the d-latent GEMM + the K1 backward did per sample, now one pass over the ds / dt / dOut rows
        node = obj("k_node_grad", "node-space reductions of ds / dt / dOut: dU, d coords through the interpolation, d node rows",
                   "hbm", (F + 16) * 2, "bytes_per_sample", ms, ["k_node_grad"])
        node["flop_per_sample"] = 2 * (mac_mod + L * od)                       # the algorithmic share it replaces (k_dlat's)
        node["flop_per_sample_executed"] = 2 * 2 * 32 * F + 2 * node_rows * (F * L + L * od)
        objs.append(node)
    else:
        names = {"fwd": "k_fused_fwd", "chain": "k_fused_bwd", "dlat": "k_dlat256", "dw": "k_dw256"} if bf16 else \
            {"fwd": "k_layer_fwd x layers + k_final_fwd", "chain": "k_layer_bwd x layers", "dlat": "k_dlat", "dw": "k_dw_grouped"}
        siren = [
            ("fwd", "forward stack incl. final_linear", 2 * mac_fwd, lambda: eng.forward(n)),
            ("chain", "backward chain: dX of every layer", 2 * (mac_main + mac_first + H * od),
             lambda: eng.backward(n, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_CHAIN)),
            ("dlat", "d-latent GEMM", 2 * (mac_mod + L * od), lambda: eng.backward(n, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_DLAT)),
            ("dw", "all weight-gradient GEMMs" + (" (+ split reduction)" if bf16 else ""), 2 * mac_fwd,
             lambda: eng.backward(n, ws.dout, weight_grads=True, phases=_lib.BWD_PHASE_DW)),
        ]
        for key, what, fl, fn in siren:
            objs.append(obj(names[key], what, "mfma", fl, "flop_per_sample", event_ms(fn, reps), [names[key]] if bf16 else []))
    tot_ms = sum(o["in_step_ms"] or o["avg_launch_ms"] for o in objs)
    b2b_ms = sum(o["avg_launch_ms"] for o in objs)
    agg = n * 6 * mac_fwd / (tot_ms * 1e-3) / 1e12
    siren_total = {"kernels": "SIREN fwd + bwd (" + ", ".join(o["kernel"].split(" (")[0] for o in objs) + ")",
                   "flop_per_sample": 6 * mac_fwd, "ms": tot_ms, "ms_back_to_back": b2b_ms, "achieved": agg, "peak": peak,
                   "unit": "TFLOP/s", "frac": agg / peak}
    chain = next(o for o in objs if o["kernel"].startswith("k_fused_bwd" if bf16 else "k_layer_bwd"))
    mfma_objs = sorted((o for o in objs if o["bound"] == "mfma"), key=lambda o: -(o["in_step_ms"] or o["avg_launch_ms"]))
    objs = mfma_objs + [o for o in objs if o["bound"] != "mfma"]
    # HBM view of the dominant kernel: the bytes are the saved-activation traffic this design chose, not compulsory ones
    esize = 2 if bf16 else 4
    tensors = 2 * d.n_layers + 4 * n_mod                            # c + dz per layer; z, s, ds, dt per modulated layer
    bps = tensors * d.HP * esize + 4 * (od + 2 * ind)
    t_chain = chain["in_step_ms"] or chain["avg_launch_ms"]
    hbm_view = {"kernel": chain["kernel"], "bound": "hbm", "bytes": "saved-activation traffic of this design",
                "achieved": n * bps / (t_chain * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "bytes_per_sample": bps, "traffic": chain["traffic"]}
    hbm_view["frac"] = hbm_view["achieved"] / HBM_PEAK_GBS
    # K1 / K3: compulsory bytes per sample (the interpolated latent leaves K1 in the activation dtype)
    cl = d.lat * esize
    k1f = (lambda: E.k1_forward_sorted(eng, coords, subj, lat, None, tfs, n)) if grouped and E.k1_sorted_eligible(tuple(lat.shape), d.LP) \
        else (lambda: E.k1_forward(eng, trainer.k1, coords, subj, lat, None, tfs, False, n))
    k3 = lambda: E.sample_loss(None, ws.out, n, trainer.sr_dims, trainer.n_cls, trainer.loss_metric, trainer.class_w, 1.0, 1.0,
                               ws.dout, trainer.sums, voxel=voxel, subj=subj, intens=st.intens, labels=st.labels,
                               voxels_per_subject=st.V, prepared=True)
    legs = [("K1 forward (affine + trilinear gather; in the step it is a block range of k_step_head)", 12 + 4 + cl, k1f, ["k_step_head"]),
            ("K3 (ground-truth sampling + loss + dOut)", 4 * od + 4 * trainer.sr_dims + 1 + 4 + 4 * od + 1, k3, ["k_sample_loss"])]
    if not use_ng:
        E.k1_shadow(eng, trainer.k1, lat)
        k1b = lambda: E.k1_backward(eng, trainer.k1, coords, subj, tuple(lat.shape), 0, tfs, False, n, want_latents=True,
                                    want_cond=False, dtf_out=trainer.tf_grad, dlat_out=trainer.lat_grad, grouped=grouped)
        legs.insert(1, ("K1 backward (coordinate gradient + latent / tf reductions)", 4 * d.lat + 12 + 4, k1b, []))
    hbm = []
    for name, bytes_ps, fn, prof_names in legs:
        ms = event_ms(fn, reps)
        ach = n * bytes_ps / (ms * 1e-3) / 1e9
        ins = sum(filter(None, (instep_ms(k) for k in prof_names))) if profiled else None
        hbm.append({"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "avg_launch_ms": ms, "in_step_ms": ins or None, "bytes_per_sample": bytes_ps})
    trainer.lat_grad.zero_()
    trainer.tf_grad.zero_()
    eng.flat_grad.zero_()
    eng.grads_clean = True

    def optimiser():
        trainer.opt_local.step()
        if trainer.opt_mlp is not None:
            trainer.opt_mlp.step()
    opt_ms = event_ms(optimiser, reps)          # the rate is reported with and without the optimiser step
    return objs, siren_total, hbm_view, hbm, opt_ms


def step_profile_summary(n):
    """Whole-step figures of the committed profile of this command: kernels of one step with their in-step durations and
    HBM-side bytes (PMC), their sums, bytes per sample."""
    import csv
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_kernel_stats.csv")
    traffic = _pmc_values()
    if not os.path.exists(path) or traffic is None:
        return None
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].strip()
        if name.startswith("k_") and int(r["Calls"]) >= 20:         # launched every step (the profile run has 25 steps)
            rows.append({"kernel": name, "in_step_ms": float(r["AverageNs"]) * 1e-6, "traffic": traffic.get(name)})
    tot_b = sum(x["traffic"] or 0.0 for x in rows)
    return {"source": f"profiles/{PROFILE_TAG}_kernel_stats.csv, {PROFILE_TAG}_pmc_fetch.txt, {PROFILE_TAG}_pmc_write.txt",
            "kernels": rows, "kernel_ms_sum": sum(x["in_step_ms"] for x in rows), "hbm_bytes_per_step": tot_b,
            "hbm_bytes_per_sample": tot_b / n}


def atlas_measure(opt, dev, steps, warmup, chunk):
    """BASELINE config 4 (cfg-A): one dense 310^3 volume (29.8 M coordinates at 0.5 mm) through INR_Decoder.inference."""
    from cinema_amd import engine as E
    from cinema_amd.models.inr_decoder import INR_Decoder
    from cinema_amd.utils import generate_world_grid
    args = make_args(512, opt.precision, opt.loss, opt.n_samples)
    sp = 155.0 / opt.grid
    args["atlas_gen"] = {"spacing": [sp, sp, sp]}
    args["device"] = str(dev)
    torch.manual_seed(42)
    dec = INR_Decoder(args, str(dev)).to(dev)
    lat = torch.randn(1, 256, 3, 3, 3, device=dev) * 0.01
    cond = torch.zeros(0, device=dev)
    grid, shape, _ = generate_world_grid(args, device=dev, lazy=True)
    G = len(grid)
    for _ in range(warmup):
        vol = dec.inference(grid, lat, cond, shape, None, step_size=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vol = dec.inference(grid, lat, cond, shape, None, step_size=chunk)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert tuple(vol.shape) == (shape[0], shape[1], shape[2], 13) and bool(torch.isfinite(vol[..., 2:]).all())
    del vol
    eng = dec.sr_net.hot_state(dev).engine
    d = eng.desc
    m = min(max(chunk, dec.min_inference_chunk) if eng.can_skip_saved() else chunk, G)
    nodes = E.nodes_inference_eligible(eng, (1, 256, 3, 3, 3))
    idx0 = torch.zeros(m, dtype=torch.int32, device=dev)
    eng.workspace(m)
    ms = event_ms((lambda: eng.forward_nodes(m, dec._nodes, idx0, keep=False)) if nodes else (lambda: eng.forward(m, keep=False)),
                  opt.kernel_reps)
    fwd_f, _ = flops_per_sample()
    # flops the nodes kernel executes: the per-sample linear_lats GEMM (lat x 2H per modulated layer) and the latent columns
    # of final_linear are one k = 32 step over the node projections each
    H, L, od = d.hidden, d.lat, d.out_dim
    n_mod = bin(d.modulated_mask).count("1")
    exec_f = 2 * (d.in_dim * H + (d.n_layers - 1) * H * H + n_mod * 32 * 2 * H + (H + 32) * od) if nodes else fwd_f
    peak = MFMA_BF16_PEAK_TFLOPS if opt.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
    ach = m * fwd_f / (ms * 1e-3) / 1e12
    value = steps * G / elapsed
    kern = (("k_nodes_fwd" if nodes else "k_fused_fwd") + ", nothing kept (one chunk)") if opt.precision == "bf16" else "k_layer_fwd x layers"
    return {"value": value, "unit": "coords/s", "ms_per_volume": elapsed / steps * 1e3, "volumes_timed": steps, "warmup": warmup,
            "shape": shape, "coords": G, "spacing_mm": sp, "step_size": chunk, "chunk": m, "dtype": opt.precision,
            "end_to_end_mfma_tflops": value * fwd_f / 1e12, "end_to_end_mfma_frac": value * fwd_f / 1e12 / peak,
            "roofline": {"kernel": kern, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                         "frac_executed": m * exec_f / (ms * 1e-3) / 1e12 / peak, "traffic": None, "avg_launch_ms": ms,
                         "flop_per_sample": fwd_f, "flop_per_sample_executed": exec_f}}


def run_atlas(opt, dev):
    a = atlas_measure(opt, dev, opt.steps, opt.warmup, opt.chunk)
    shape, G = a["shape"], a["coords"]
    res = {"metric": "atlas coords/sec, dense inference (hidden 512)", "value": a["value"], "unit": "coords/s", "n_gpus": 1,
           "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": a["ms_per_volume"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": opt.precision, "data": "synthetic",
           "config": {"workload": f"cfg-A: one {shape[0]}x{shape[1]}x{shape[2]} volume ({G:,} coordinates, spacing {a['spacing_mm']:.3f} mm) "
                                  f"through INR_Decoder.inference(step_size={opt.chunk}), evaluated in chunks of {a['chunk']}: grid "
                                  "coordinates made per chunk, K1, SIREN forward (nothing kept), fused clamp / min-max / argmax / "
                                  "softmax epilogue, rescale"},
           "end_to_end_mfma_tflops": a["end_to_end_mfma_tflops"], "end_to_end_mfma_frac": a["end_to_end_mfma_frac"],
           "roofline": a["roofline"]}
    print(json.dumps(res))


def data_parallel_report(trainer, opt, dev, world, rank, n, total_steps, barrier, ms_per_step):
    """What makes a multi-GPU line auditable (collective): the collective library actually in use, the gradient
    all-reduce timed alone, and how much of it the step does NOT hide (step time minus step time with the collective
    switched off, CINEMA_DP_SKIP_ALLREDUCE=1 -- measurement only, the replicas drift apart afterwards, so it runs last)."""
    import torch.distributed as dist
    from cinema_amd import parallel as par
    backend = dist.get_backend()
    names = [None] * world
    dist.all_gather_object(names, torch.cuda.get_device_name(dev) + f" (cuda:{dev.index})")
    info = {"backend": backend, "world_size": dist.get_world_size(), "devices": names,
            "library_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None}
    g = trainer.eng.flat_grad
    reps = 20
    par.allreduce_mean_(g, trainer.pg)
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        par.allreduce_mean_(g, trainer.pg)
    barrier()
    ar_ms = (time.perf_counter() - t0) / reps * 1e3
    g.zero_()
    trainer.eng.grads_clean = True
    os.environ["CINEMA_DP_SKIP_ALLREDUCE"] = "1"
    lo = total_steps - opt.steps                      # the same samples as the timed steps
    for i in range(lo, lo + min(3, opt.steps)):
        trainer.step(i * n, (i + 1) * n)
    barrier()
    t0 = time.perf_counter()
    for i in range(lo, total_steps):
        trainer.step(i * n, (i + 1) * n)
    barrier()
    skip = time.perf_counter() - t0
    del os.environ["CINEMA_DP_SKIP_ALLREDUCE"]
    t = torch.tensor([skip, ar_ms], device="cpu" if backend != "nccl" else dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    skip_ms = float(t[0]) / opt.steps * 1e3
    return {"rccl": info, "allreduce_bytes": g.numel() * 4, "allreduce_ms": float(t[1]),
            "ms_per_step_without_allreduce": skip_ms, "allreduce_exposed_ms": ms_per_step - skip_ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-samples", type=int, default=25000)
    ap.add_argument("--subjects", type=int, default=64, help="subjects per GPU")
    ap.add_argument("--grid", type=int, default=310, help="voxels per axis of each synthetic subject")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--hidden", type=int, default=512, help="hidden width (the headline metric is quoted at 512; 1024 = the "
                    "reference's config default, layer-by-layer kernels: use with --profile-step, the roofline objects assume 512)")
    ap.add_argument("--loss", default="mse", choices=["mse", "l1"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-atlas", action="store_true", help="train mode: leave out the \"atlas\" object (one 310^3 volume, ~1 s)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--mode", default="train", choices=["train", "atlas"],
                    help="train: the headline training step (default); atlas: one dense 310^3 volume through inference (cfg-A)")
    ap.add_argument("--chunk", type=int, default=100000, help="atlas mode: coordinates per chunk (reference default step_size)")
    ap.add_argument("--profile-step", action="store_true",
                    help="only the timed steps: no per-kernel timing, no box probes, no CPU baseline (for rocprofv3 passes)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank dry run on a 1-GPU box: every rank uses cuda:0 and the gradient travels over gloo "
                         "(checks the rank logic; the number it prints is not a result)")
    opt = ap.parse_args()

    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched bare (`python bench.py --gpus N`): start the N ranks ourselves, one process per GPU, through the same
        # launcher the driver uses.  This parent has made no GPU call (importing torch makes none); it only waits for the
        # launcher and hands its exit status on -- a failed rank fails the run.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={opt.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if opt.gpus != world:
        raise SystemExit(f"bench.py: --gpus {opt.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {opt.gpus})")
    if not opt.rehearse_on_one_gpu and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s); one process per GPU "
                         "(--rehearse-on-one-gpu runs the rank logic over gloo on one card)")
    if opt.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if opt.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if opt.mode == "atlas":
        if world > 1:
            raise SystemExit("bench.py --mode atlas runs on one GPU (volumes shard over ranks as replicas, cinema_amd/atlas.py)")
        if opt.steps == 50 and opt.warmup == 10:      # the training defaults: a volume is 1 200 training batches
            opt.steps, opt.warmup = 3, 1
        return run_atlas(opt, dev)

    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = make_args(opt.hidden, opt.precision, opt.loss, opt.n_samples)
    torch.manual_seed(42)           # identical MLP init on every rank (replicated weights)
    spacing = 155.0 / opt.grid
    t0 = time.time()
    store = SubjectStore.synthetic(opt.subjects, (opt.grid,) * 3, spacing, [155, 155, 155], 2, 10, seed=42 + rank, device=dev)
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    store.shuffle(gen, group_chunk=opt.n_samples)     # as HotPathTrainer.epoch does: batches grouped by subject
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] synthetic store: {opt.subjects} subjects x {opt.grid}^3, {len(store):,} foreground samples/rank, "
              f"{time.time() - t0:.1f}s", file=sys.stderr)
    trainer = HotPathTrainer(args, store, dev)
    lat0, tf0, flat0 = trainer.latents.detach().cpu().clone(), trainer.tfs.detach().cpu().clone(), trainer.eng.flat.detach().cpu().clone()
    n = opt.n_samples
    total_steps = opt.warmup + opt.steps
    if total_steps * n > len(store):
        raise SystemExit(f"not enough foreground samples ({len(store)}) for {total_steps} steps of {n}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(opt.warmup):
        trainer.step(i * n, (i + 1) * n)
    barrier()
    t0 = time.perf_counter()
    for i in range(opt.warmup, total_steps):
        trainer.step(i * n, (i + 1) * n)
    enqueued = time.perf_counter() - t0       # host side done: how far ahead of the GPU the launch loop runs
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if opt.rehearse_on_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = trainer.loss_dict()
    dp = None
    if world > 1:
        dp = data_parallel_report(trainer, opt, dev, world, rank, n, total_steps, barrier, elapsed / opt.steps * 1e3)

    peak = MFMA_BF16_PEAK_TFLOPS if opt.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
    siren = siren_total = hbm_view = hbm = peaks = opt_ms = None
    if rank == 0 and not opt.profile_step:
        siren, siren_total, hbm_view, hbm, opt_ms = rooflines(trainer, n, opt, peak)
        peaks = box_peaks(dev)

    cpu_baseline_obj = None
    if rank == 0 and world == 1 and not opt.no_cpu_baseline and not opt.profile_step:
        cpu_baseline_obj = cpu_baseline(args, trainer, n, lat0, tf0, flat0)
    fwd_f, tot_f = flops_per_sample()
    value = world * opt.steps * n / elapsed
    step_tflops = value * tot_f / 1e12
    res = {
        "metric": f"coord samples/sec fwd+bwd (hidden {opt.hidden})", "value": value, "unit": "coord samples/s", "n_gpus": world,
        "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": elapsed / opt.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": opt.precision, "data": "synthetic" if not opt.rehearse_on_one_gpu else "synthetic (REHEARSAL on one GPU over gloo: not a result)",
        "config": {"workload": f"dhcp_neo_mm shape on MI355X: latent 256x3x3x3, hidden {opt.hidden}, 5 hidden layers (mod 1,3,5), "
                               f"out 2+10, {opt.subjects} synthetic subjects of {opt.grid}^3 voxels per GPU, "
                               f"n_samples {n} per GPU per step, {opt.loss} + weighted CE, AdamW step included",
                   "global_batch": world * n, "parallelism": f"dp{world} (subjects sharded, MLP grads all-reduced)"},
        "step_mfma_tflops_per_gpu": step_tflops / world, "step_mfma_frac": step_tflops / world / peak,
        "loss_last_step": losses, "host_enqueue_ms_per_step": enqueued / opt.steps * 1e3,
    }
    if rank == 0:
        if siren is not None:
            res["roofline"] = siren[0]                       # the SIREN kernel with the longest launch, against MFMA
            res["roofline_siren_kernels"] = siren[1:]
            res["roofline_siren_fwd_bwd"] = siren_total
            res["roofline_hbm_view"] = hbm_view
            res["roofline_hbm_kernels"] = hbm
            res["box_peaks"] = peaks
            ms = res["ms_per_step"]
            res["optimiser_ms"] = opt_ms                     # the fused AdamW launch(es), timed alone
            res["value_without_optimiser"] = world * n / ((ms - opt_ms) * 1e-3)
            res["step_profile"] = step_profile_summary(n) if (opt.precision == "bf16" and n == 25000) else None
        if dp is not None:
            res.update(dp)
        if world == 1 and not opt.profile_step and not opt.no_atlas:
            # BASELINE config 4 beside the headline: one 310^3 volume, 1 warm-up + 2 timed (0.3 s)
            del trainer
            torch.cuda.empty_cache()
            res["atlas"] = atlas_measure(opt, dev, 2, 1, opt.chunk)
        if world == 1 and not opt.no_cpu_baseline and not opt.profile_step:
            res["cpu_baseline"] = cpu_baseline_obj
        print(json.dumps(res))
    if world > 1:
        dist.barrier()              # rank 0 times single kernels after the steps: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
