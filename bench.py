#!/usr/bin/env python3
"""bench.py — EEG clips/s (60 s, 19 ch, K=2, 2 layers x 64 units), fwd + bwd + optimiser step, on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the DCRNN hot path over one batch of synthetic clips that is already
resident in HBM: zero_grad -> forward -> loss -> full BPTT backward (all parameter gradients) ->
[RCCL all-reduce of one flat gradient bucket] -> clip_grad_norm_(5) -> Adam.  Weak scaling: the
per-GPU batch is fixed (256 clips), ranks hold different clips, no data-path collective except
the gradient all-reduce.  Rank 0 prints ONE JSON line (contract in the task statement), extended
with `roofline` (live HIP-event timing of every kernel, algorithmic FLOPs/bytes from DESIGN.md)
and `cpu_baseline` (the torch-eager oracle timed on the host cores on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_NODES, H_UNITS, D_IN, K_DIFF, LAYERS = 19, 64, 100, 2, 2
PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec (6290 GB/s measured copy)

WORKLOADS = {
    # name: (task, filter_type, T, per-GPU batch, classes)
    "cfg2": ("detection", "laplacian", 60, 256, 1),
    "cfg3": ("detection", "dual_random_walk", 60, 256, 1),
    "cfg4": ("classification", "laplacian", 60, 256, 4),
    "cfg1": ("detection", "laplacian", 12, 4, 1),
    "cfg5": ("ssl", "dual_random_walk", 60, 512, 0),
}
T_OUT = 12        # SSL prediction horizon (args.py:52-56)
DESCR = {
    "cfg2": "BASELINE cfg2: DCRNN detection, distance graph, clip_len=60, batch=256/GPU, K=2, 2x64, synthetic FFT inputs",
    "cfg3": "BASELINE cfg3: DCRNN detection, correlation graph (per-clip adj), clip_len=60, batch=256/GPU",
    "cfg4": "BASELINE cfg4: DCRNN 4-class classification, distance graph, clip_len=60, batch=256/GPU (2048 over 8)",
    "cfg1": "BASELINE cfg1: DCRNN detection, distance graph, clip_len=12, batch=4 (plumbing)",
    "cfg5": "BASELINE cfg5: SSL seq2seq pretrain (encoder 60 s + decoder 12 s), correlation graph, batch=512/GPU (4096 over 8)",
}


def make_args(filter_type):
    import types
    return types.SimpleNamespace(num_nodes=N_NODES, num_rnn_layers=LAYERS, rnn_units=H_UNITS, input_dim=D_IN,
                                 output_dim=D_IN, max_diffusion_step=K_DIFF, dcgru_activation="tanh",
                                 filter_type=filter_type, dropout=0.0, cl_decay_steps=3000,
                                 use_curriculum_learning=False)


def synthetic_batch(task, filter_type, t_len, batch, classes, seed):
    """x ~ N(0,1) (z-scored log-FFT amplitudes), seq_lengths = T (detection) or
    U[T/2, T] with zero padding (classification), distance-graph scaled Laplacian or per-clip
    top-3 dual random-walk supports, labels from a fixed statistic of the clip."""
    from eeg_gnn_ssl_amd import utils
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, t_len, N_NODES, D_IN, generator=g)
    if task == "classification":
        lengths = torch.randint(t_len // 2, t_len + 1, (batch,), generator=g)
        for i in range(batch):
            x[i, int(lengths[i]):] = 0
    else:
        lengths = torch.full((batch,), t_len, dtype=torch.int64)
    stat = x[:, :, :, :10].mean(dim=(1, 2, 3))
    if task == "ssl":
        y = torch.randn(batch, T_OUT, N_NODES, D_IN, generator=g)     # independent next clip (loss parity only)
    elif classes == 1:
        y = (stat > 0).float()
    else:
        q = torch.quantile(stat, torch.tensor([0.25, 0.5, 0.75]))
        y = torch.bucketize(stat, q).to(torch.int64)
    if filter_type == "laplacian":
        adj = np.load(os.path.join(ROOT, "eeg_gnn_ssl_amd", "data", "electrode_adj_3d.npy"))
        s = utils.compute_supports(adj, "laplacian")[0]
        supports = [s.unsqueeze(0).repeat(batch, 1, 1)]     # the trainers always pass batched supports (Q5)
    else:
        s1, s2 = [], []
        xn = x.numpy()
        for i in range(batch):
            a = utils.correlation_graph(xn[i], top_k=3)
            s = utils.compute_supports(a, "dual_random_walk")
            s1.append(s[0])
            s2.append(s[1])
        supports = [torch.stack(s1), torch.stack(s2)]
    return x, y, lengths, supports


def algorithmic_work(filter_type, t_len, batch, task="detection"):
    """Per-step algorithmic FLOPs / bytes of every kernel class."""
    m = (2 if filter_type == "dual_random_walk" else 1) * K_DIFF + 1
    n, h = N_NODES, H_UNITS
    s = t_len * batch
    r = s * n
    fins = [D_IN] + [h] * (LAYERS - 1)
    w = {"seq_fwd": 0.0, "seq_bwd": 0.0, "gemm_nn": 0.0, "gemm_tn": 0.0, "diffuse_fwd": 0.0, "diffuse_adj": 0.0}
    if filter_type == "dual_random_walk":
        w["corr_gram"] = 4.0 * s * n * D_IN          # per-clip correlation graph: every clip read once
    for l, fin in enumerate(fins):
        w["seq_fwd"] += s * (2 * (m - 1) * 2 * n * n * h + 2 * n * (h * m) * 3 * h)
        w["seq_bwd"] += s * ((m - 1) * 2 * n * n * 3 * h + 2 * n * (h * m) * 3 * h)
        w["gemm_nn"] += 2.0 * r * (m * fin) * 3 * h
        w["gemm_tn"] += 2.0 * r * (m * fin) * 3 * h + 2.0 * r * (m * h) * 2 * h + 2.0 * r * (m * h) * h
        # layer 0 only: read X once, write M-1 planes + the time-major copy of the batch-major input (the hop
        # planes of h and r*h -- and with them the input planes of the layers above -- are by-products of seq_fwd)
        if l == 0:
            w["diffuse_fwd"] += 4.0 * s * n * fin * (m + 1)
        if l > 0:
            w["gemm_nn"] += 2.0 * r * 3 * h * (m * fin)
            w["diffuse_adj"] += 4.0 * s * n * fin * (m + 1)
    if task == "ssl":       # decoder: T_OUT autoregressive steps; every layer's dx is needed (feedback / layer below)
        sd = T_OUT * batch
        rd = sd * n
        for k in list(w):
            w["dec_" + k] = 0.0
        for l, fin in enumerate(fins):
            w["dec_seq_fwd"] += sd * (2 * (m - 1) * 2 * n * n * h + 2 * n * (h * m) * 3 * h)
            w["dec_seq_bwd"] += sd * ((m - 1) * 2 * n * n * 3 * h + 2 * n * (h * m) * 3 * h)
            w["dec_gemm_nn"] += 2.0 * rd * (m * fin) * 3 * h * 2
            w["dec_gemm_tn"] += 2.0 * rd * (m * fin) * 3 * h + 2.0 * rd * (m * h) * 3 * h
            if l == 0:
                w["dec_diffuse_fwd"] += 4.0 * sd * n * fin * m     # first decoder layer only (as above)
            w["dec_diffuse_adj"] += 4.0 * sd * n * fin * (m + 1)
        w["dec_gemm_nn"] += 2 * 2.0 * rd * h * D_IN          # projection forward + d h_top
        w["dec_gemm_tn"] += 2.0 * rd * h * D_IN              # projection weight gradient
    return w


def cpu_baseline(workload, sample_clips=32, budget_s=60.0):
    """The oracle (torch-eager restatement of the reference's op sequence, autograd backward)
    timed on this host's cores on a bounded sample of the same workload.  The op stream is ~10^4
    tiny ATen calls per step, so more threads is not faster: a few thread counts are probed on a
    small sample and the best one is used (and reported as `cores`)."""
    from oracle import dcrnn_oracle as orc
    task, filt, t_len, _, classes = WORKLOADS[workload]
    cfg = orc.DCRNNConfig(filter_type=filt, num_classes=max(classes, 1))
    kind = "nextTimePred" if task == "ssl" else "classification"
    params = {k: v.requires_grad_(True) for k, v in orc.init_params(cfg, kind, seed=0).items()}
    x, y, lengths, sup = synthetic_batch(task, filt, t_len, sample_clips, classes, seed=123)

    def one(nclips):
        for p in params.values():
            p.grad = None
        t0 = time.perf_counter()
        if task == "ssl":
            pred = orc.next_time_pred_forward(params, cfg, x[:nclips], y[:nclips], [s[:nclips] for s in sup])
            loss = orc.regression_loss(y[:nclips], pred, loss_fn="MAE")
        else:
            logits = orc.classification_forward(params, cfg, x[:nclips], lengths[:nclips], [s[:nclips] for s in sup])
            loss = orc.bce_with_logits(logits, y[:nclips]) if classes == 1 else orc.cross_entropy(logits, y[:nclips])
        loss.backward()
        return time.perf_counter() - t0

    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    probe = {}
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        one(4)                                   # warm-up
        probe[nt] = one(4)
        print(f"[bench] cpu baseline probe: {nt} threads -> {4 / probe[nt]:.1f} clips/s", file=sys.stderr, flush=True)
        if time.perf_counter() - t_start > budget_s / 2:
            break
    best_nt = min(probe, key=probe.get)
    torch.set_num_threads(best_nt)
    one(sample_clips)
    best = float("inf")
    reps = 0
    while reps < 3 and time.perf_counter() - t_start < budget_s:
        best = min(best, one(sample_clips))
        reps += 1
    return {"value": round(sample_clips / best, 2), "unit": "clips/s", "cores": best_nt, "host_logical_cpus": ncpu,
            "kind": "port",
            "sample": f"{sample_clips} clips x T={t_len} of {workload} (fwd+loss+bwd, best of {reps} after 1 warm-up; "
                      f"torch-eager oracle = op-for-op restatement of the reference; thread count chosen by probe "
                      f"{ {k: round(4 / v, 1) for k, v in probe.items()} } clips/s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (default: workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="disable the live per-kernel HIP-event timing")
    ap.add_argument("--host-supports", action="store_true", help="correlation-graph workloads: use supports prepared "
                    "on the host (the reference's DataLoader path) instead of building them on the GPU every step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the "
                    "captured HIP graph of forward+loss+backward")
    ap.add_argument("--graph", action="store_true", help="replay the HIP graph also when launched on several GPUs "
                    "(default there: eager launches; the replay measured no faster at 1 GPU, the host runs ahead anyway)")
    ap.add_argument("--tune", action="append", default=[], help="development knob key=value (eeg_dcrnn_set_tuning)")
    args = ap.parse_args()

    t_boot = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device: eeg_gnn_ssl_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")                     # "nccl" IS RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from eeg_gnn_ssl_amd import DCRNNModel_classification, _lib, ops
    from eeg_gnn_ssl_amd.train_step import TrainStep

    for kv in args.tune:
        k, v = kv.split("=")
        _lib.get_lib().call("eeg_dcrnn_set_tuning", int(k), int(v))
    task, filt, t_len, batch, classes = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    torch.manual_seed(123)                                   # identical replicas on every rank
    if task == "ssl":
        from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
        model = DCRNNModel_nextTimePred(make_args(filt), device=dev).to(dev)
    else:
        model = DCRNNModel_classification(make_args(filt), classes, device=dev).to(dev)
    model.train()
    stepper = TrainStep(model, task=task, lr=3e-4, weight_decay=5e-4, max_grad_norm=5.0)
    x, y, lengths, supports = synthetic_batch(task, filt, t_len, batch, classes, seed=123 + rank)
    x, y, lengths = x.to(dev), y.to(dev), lengths.to(dev)
    supports = [s.to(dev) for s in supports]
    device_graph = filt == "dual_random_walk" and not args.host_supports
    if device_graph:
        # per-clip correlation graph + supports are rebuilt from the clips on the GPU inside every step
        # (eeg_dcrnn_corr_graph); they must match what the host pipeline prepared for the same clips
        chk = ops.correlation_supports(x, top_k=3)
        bad = sum((a - b_).abs().amax(dim=(1, 2)) > 1e-5 for a, b_ in zip(chk, supports)).clamp(max=1).sum().item()
        if bad > max(1, batch // 100):       # a rare top-3 near-tie (fp32 vs the host's fp64 Gram) may flip one edge
            raise SystemExit(f"device correlation-graph supports differ from the host pipeline on {bad} clips")
        supports = None

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_boot:6.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"inputs on device, {args.warmup} warm-up steps")
    lib = _lib.get_lib()
    graphed = False
    if not args.no_graph and (world == 1 or args.graph):
        # forward + loss + backward replayed as ONE HIP graph; all-reduce + fused clip/Adam stay eager
        try:
            stepper.capture(x, y, lengths, supports)
            graphed = True
        except Exception as e:                                   # noqa: BLE001 -- fall back to eager launches
            log(f"HIP graph capture failed ({type(e).__name__}: {e}); launching eagerly")
            torch.cuda.synchronize()
    one_step = stepper.replay_step if graphed else (lambda: stepper.step(x, y, lengths, supports))
    for _ in range(args.warmup):
        one_step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    log(f"timed {args.steps} steps ({'graph replay' if graphed else 'eager'}): {elapsed / args.steps * 1e3:.3f} ms/step")
    prof = {}
    if not args.no_prof:
        # per-kernel durations: the SAME K steps once more, launched eagerly with a HIP-event pair
        # around every launch on the launch stream (events cannot be read back from a graph replay;
        # the pairs themselves cost ~0.2 ms/step, which is why they are kept out of the timed region)
        lib.query("eeg_dcrnn_prof_enable", 1)
        for _ in range(args.steps):
            stepper.step(x, y, lengths, supports)
        torch.cuda.synchronize()
        lib.query("eeg_dcrnn_prof_enable", 0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.call("eeg_dcrnn_prof_report", buf, len(buf))
        for line in buf.value.decode().strip().splitlines():
            name, cnt, ms = line.split()
            prof[name] = (int(cnt), float(ms))
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(loss.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    clips_per_s = batch * world / (elapsed / args.steps)
    work = algorithmic_work(filt, t_len, batch, task)
    kernels = {}
    for name, (cnt, ms) in prof.items():
        per_step_ms = ms / args.steps
        ent = {"launches_per_step": cnt / args.steps, "ms_per_step": round(per_step_ms, 4)}
        if name in work and work[name] > 0 and per_step_ms > 0:
            if "diffuse" in name or name == "corr_gram":
                gbs = work[name] / (per_step_ms * 1e-3) / 1e9
                ent.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
            else:
                tf = work[name] / (per_step_ms * 1e-3) / 1e12
                ent.update(bound="mfma", achieved=round(tf, 2), peak=PEAK_MFMA_F32_TFLOPS, unit="TFLOP/s",
                           frac=round(tf / PEAK_MFMA_F32_TFLOPS, 4))
        kernels[name] = ent
    roofline = None
    timed = {k: v for k, v in kernels.items() if "bound" in v}
    if timed:
        dom = max(timed, key=lambda k: timed[k]["ms_per_step"])
        d = timed[dom]
        traffic = None            # HBM bytes per launch from the committed PMC passes (same command, cfg2)
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
        for name, tb in (traffic or {}).items():
            if name in kernels:
                kernels[name]["traffic_bytes_per_launch_pmc"] = tb
        roofline = {"kernel": dom, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"], "unit": d["unit"],
                    "frac": d["frac"], "traffic": (traffic or {}).get(dom), "avg_launch_ms": round(d["ms_per_step"] / d["launches_per_step"], 4),
                    "kernels": kernels,
                    "kernel_ms_per_step_total": round(sum(v["ms_per_step"] for v in kernels.values()), 3),
                    "whole_step_flops": round(sum(v for k, v in work.items() if "diffuse" not in k and k != "corr_gram") / 1e9, 1),
                    "whole_step_mfma_frac": round(sum(v for k, v in work.items() if "diffuse" not in k and k != "corr_gram")
                                                  / (ms_per_step * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, 4)}
    out = {
        "metric": "EEG clips/sec (60s, 19ch, K=2, 2-layer x64) fwd+bwd",
        "value": round(clips_per_s, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": DESCR[args.workload], "per_gpu_batch": batch, "global_batch": batch * world,
                   "clip_len": t_len, "parallelism": f"dp{world}", "optimizer_step_included": True,
                   "supports": ("per-clip correlation graph + dual random-walk supports built on the GPU inside the step"
                                if device_graph else "prepared on the host (distance graph is fixed)"
                                if filt == "laplacian" else "prepared on the host"),
                   "launch": "hip-graph replay (fwd+loss+bwd) + eager all-reduce/clip+Adam" if graphed else "eager",
                   "final_loss": round(loss_val, 5)},
        # SURVEY.md §8d asks for both figures: `value` is the training-loop rate (optimiser step included); the
        # kernel figure takes the optimiser tail (norm + fused clip/Adam, live HIP-event times) out of the step
        "fwd_bwd_only": (None if not prof or world > 1 else {
            "clips_per_s": round(batch / ((ms_per_step - sum(prof.get(k, (0, 0.0))[1] for k in ("grad_sqnorm", "clip_adam")) / args.steps) * 1e-3), 1),
            "excluded_ms_per_step": round(sum(prof.get(k, (0, 0.0))[1] for k in ("grad_sqnorm", "clip_adam")) / args.steps, 4)}),
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(args.workload)
        out["speedup_vs_cpu_baseline"] = round(clips_per_s / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
