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


def cpu_baseline(args, n_samples, n_subjects, budget_s=25.0):
    """The CPU oracle (a port of the reference path, pinned to the reference by tests/golden) on the host cores."""
    from oracle import cinema_oracle as O
    a = args["inr_decoder"]
    g = torch.Generator().manual_seed(42)
    params = O.init_siren_params(3, a["latent_dim"][0], sum(a["out_dim"]), a["hidden_size"], a["num_hidden_layers"], 30, 30,
                                 a["modulated_layers"], g)
    for p in params.values():
        p.requires_grad_(True)
    lat = (torch.randn(n_subjects, *a["latent_dim"], generator=g) * 0.01).requires_grad_(True)
    tft = torch.zeros(n_subjects, 6, requires_grad=True)
    coords = torch.rand(n_samples, 3, generator=g) * 1.6 - 0.8
    idx = torch.randint(0, n_subjects, (n_samples,), generator=g)
    target = torch.cat((torch.rand(n_samples, 2, generator=g), torch.randint(1, 10, (n_samples, 1), generator=g).float()), 1)
    cfg = dict(num_hidden_layers=a["num_hidden_layers"], omega=[30, 30], sr_dims=2, loss_metric=args["optimizer"]["loss_metric"],
               class_weights=torch.tensor(args["dataset"]["class_weights"]), tf_weight=0.0)
    cores = torch.get_num_threads()
    times = []
    t_start = time.time()
    for it in range(6):
        for p in list(params.values()) + [lat, tft]:
            p.grad = None
        t0 = time.time()
        O.training_step(params, lat, tft, coords, target, idx, None, cfg)
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > budget_s and len(times) >= 2:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": n_samples / med, "unit": "coord samples/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed fwd+loss+bwd steps (after 1 warm-up) of the CPU oracle at the same shape: "
                      f"N={n_samples}, S={n_subjects}, fp32, torch CPU ops, median"}


def pmc_traffic(kernel: str):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this workload
    (profiles/r01f_pmc_fetch.txt, _write.txt: separate --pmc runs of this script, tools/profile_round.sh).  Counters
    are in KiB; FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on
    gfx950 (128-B requests tallied at 64 B), WRITE_SIZE is exact.  A live run cannot collect counters on itself, so
    this is the profile of the same command, not of this process; (None, None) when the files are absent."""
    here = os.path.dirname(os.path.abspath(__file__))
    vals = {}
    for key, name in (("FETCH_SIZE", "r01f_pmc_fetch.txt"), ("WRITE_SIZE", "r01f_pmc_write.txt")):
        path = os.path.join(here, "profiles", name)
        if not os.path.exists(path):
            return None, None
        for line in open(path):
            if line.startswith(kernel) and key + "=" in line:
                vals[key] = float(line.split(key + "=")[1].split()[0])
    if len(vals) != 2:
        return None, None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "profiles/r01f_pmc_fetch.txt (x2) + profiles/r01f_pmc_write.txt"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-samples", type=int, default=25000)
    ap.add_argument("--subjects", type=int, default=64, help="subjects per GPU")
    ap.add_argument("--grid", type=int, default=310, help="voxels per axis of each synthetic subject")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--loss", default="mse", choices=["mse", "l1"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=20)
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

    from cinema_amd.trainer import HotPathTrainer, SubjectStore
    args = make_args(512, opt.precision, opt.loss, opt.n_samples)
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
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if opt.rehearse_on_one_gpu else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = trainer.loss_dict()

    # ---- roofline of the dominant kernel, timed alone with events on its launch stream.  By time (profiles/) that is the
    # fused backward chain k_fused_bwd: per sample it reads the saved c of every sine layer and z, s of the modulated
    # ones and writes dz of every layer and ds, dt of the modulated ones, HP bf16 values each; its MFMA work
    # (2*H*H*(L) FLOP per sample) would take a third of that at peak, so HBM is the roofline that binds it.
    roofline = None
    others = []
    if rank == 0:
        from cinema_amd import _lib
        eng = trainer.eng
        ws = eng.ws
        reps = opt.kernel_reps
        d = eng.desc
        esize = 2 if opt.precision == "bf16" else 4

        def time_phase(phase):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eng.backward(n, ws.dout, weight_grads=True, phases=phase)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(reps):
                eng.backward(n, ws.dout, weight_grads=True, phases=phase)
            ev1.record()
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) / reps

        n_mod = bin(d.modulated_mask).count("1")
        ms = time_phase(_lib.BWD_PHASE_CHAIN)
        tensors = 2 * d.n_layers + 4 * n_mod                       # c + dz per layer, z, s, ds, dt per modulated layer
        bytes_per_sample = tensors * d.HP * esize + 4 * (d.out_dim + 2 * d.in_dim)
        ach = n * bytes_per_sample / (ms * 1e-3) / 1e9
        kname = ("k_fused_bwd (backward chain of the SIREN stack, one launch)" if opt.precision == "bf16"
                 else "k_layer_bwd x layers (backward chain, fp32 layer-by-layer kernels)")
        traffic, traffic_src = pmc_traffic("k_fused_bwd") if opt.precision == "bf16" and n == 25000 else (None, None)
        roofline = {"kernel": kname, "bound": "hbm",
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_src, "avg_launch_ms": ms, "bytes_per_launch": n * bytes_per_sample}
        peak = MFMA_BF16_PEAK_TFLOPS if opt.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
        ms_dw = time_phase(_lib.BWD_PHASE_DW)
        macs = n * (5 * d.hidden * d.hidden + 3 * 2 * d.hidden * d.lat)           # dW of 5 linear + 3 linear_lats
        others.append({"kernel": "k_dw256 + k_dw256_reduce (all weight-gradient GEMMs: 256x256 tiles, split partials, reduce)", "bound": "mfma",
                       "achieved": 2.0 * macs / (ms_dw * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                       "frac": 2.0 * macs / (ms_dw * 1e-3) / 1e12 / peak, "avg_launch_ms": ms_dw})

    fwd_f, tot_f = flops_per_sample()
    value = world * opt.steps * n / elapsed
    step_tflops = value * tot_f / 1e12
    peak = MFMA_BF16_PEAK_TFLOPS if opt.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
    res = {
        "metric": "coord samples/sec fwd+bwd (hidden 512)", "value": value, "unit": "coord samples/s", "n_gpus": world,
        "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": elapsed / opt.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": opt.precision, "data": "synthetic" if not opt.rehearse_on_one_gpu else "synthetic (REHEARSAL on one GPU over gloo: not a result)",
        "config": {"workload": "dhcp_neo_mm shape on MI355X: latent 256x3x3x3, hidden 512, 5 hidden layers (mod 1,3,5), "
                               f"out 2+10, {opt.subjects} synthetic subjects of {opt.grid}^3 voxels per GPU, "
                               f"n_samples {n} per GPU per step, {opt.loss} + weighted CE, AdamW step included",
                   "global_batch": world * n, "parallelism": f"dp{world} (subjects sharded, MLP grads all-reduced)"},
        "step_mfma_tflops_per_gpu": step_tflops / world, "step_mfma_frac": step_tflops / world / peak,
        "loss_last_step": losses,
    }
    if rank == 0:
        res["roofline"] = roofline
        res["roofline_other_kernels"] = others
        if world == 1 and not opt.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, n, opt.subjects)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
