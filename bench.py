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
import hashlib
import json
import os
import sys
import time

if __name__ == "__main__" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # the CPU baseline leg (oracle on the host cores) is reported with pinned threads: its rate moved 131.9 <-> 162.7 clips/s
    # between boxes of the pool with floating threads.  Must be set before the OpenMP runtime starts (= before torch).
    # Only when run as the benchmark: `import bench` (tests, smoke) must not touch the environment -- a process that imports this
    # module after torch and later loads a second OpenMP runtime (scikit-learn's) ends up with every thread bound to the same
    # cores, and its torch-eager code crawls (the CPU test suite lost half an hour in one oracle test to exactly that).
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_NODES, H_UNITS, D_IN, K_DIFF, LAYERS = 19, 64, 100, 2, 2
PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_CLOCK_MHZ = 2400.0           # the clock that peak is quoted at
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
ACHIEVABLE_HBM_GBS = 6290.0       # MI355X_MICROARCH.md: measured float4 copy (79 % of spec) = what any stream reaches

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


def make_args(filter_type, dropout=0.0, layers=LAYERS):
    import types
    return types.SimpleNamespace(num_nodes=N_NODES, num_rnn_layers=layers, rnn_units=H_UNITS, input_dim=D_IN,
                                 output_dim=D_IN, max_diffusion_step=K_DIFF, dcgru_activation="tanh",
                                 filter_type=filter_type, dropout=dropout, cl_decay_steps=3000,
                                 use_curriculum_learning=False)


def synthetic_batch(task, filter_type, t_len, batch, classes, seed, host_supports=True):
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
    elif not host_supports:
        supports = None          # per-clip graphs are built on the GPU (eeg_dcrnn_corr_graph): skip the per-clip numpy loop
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


def algorithmic_work(filter_type, t_len, batch, task="detection", layers=LAYERS):
    """Per-step algorithmic FLOPs / bytes of every profiled kernel role.  Roles = the names the
    library's event recorder uses; at the benchmark shapes each role is ONE kernel symbol per layer
    (`ROLE_SYMBOLS`), so `roofline.kernels` is a by-symbol table."""
    m = (2 if filter_type == "dual_random_walk" else 1) * K_DIFF + 1
    n, h = N_NODES, H_UNITS
    s = t_len * batch
    r = s * n
    fins = [D_IN] + [h] * (layers - 1)
    w = {k: 0.0 for k in ("seq_fwd", "seq_bwd", "gemm_nn_xw", "gemm_nn_dx", "gemm_tn_x", "gemm_tn_hg", "gemm_tn_hc",
                          "diffuse_fwd", "diffuse_adj")}
    if filter_type == "dual_random_walk":
        w["corr_gram"] = 4.0 * s * n * D_IN          # per-clip correlation graph: every clip read once
    for l, fin in enumerate(fins):
        w["seq_fwd"] += s * (2 * (m - 1) * 2 * n * n * h + 2 * n * (h * m) * 3 * h)
        w["seq_bwd"] += s * ((m - 1) * 2 * n * n * 3 * h + 2 * n * (h * m) * 3 * h)
        w["gemm_nn_xw"] += 2.0 * r * (m * fin) * 3 * h
        w["gemm_tn_x"] += 2.0 * r * (m * fin) * 3 * h
        w["gemm_tn_hg"] += 2.0 * r * (m * h) * 2 * h
        w["gemm_tn_hc"] += 2.0 * r * (m * h) * h
        # layer 0 only (the layers above take their input planes from the recurrent kernel below): SURVEY.md §8(d)'s
        # bytes = read X once, write M-1 planes
        if l == 0:
            w["diffuse_fwd"] += 4.0 * s * n * fin * m
        if l > 0:
            w["gemm_nn_dx"] += 2.0 * r * 3 * h * (m * fin)
            w["diffuse_adj"] += 4.0 * s * n * fin * (m + 1)
    if task == "ssl":       # decoder: T_OUT autoregressive steps; every layer's dx is needed (feedback / layer below)
        sd = T_OUT * batch
        rd = sd * n
        for k in ("seq_fwd", "seq_bwd", "gemm_nn", "gemm_tn_x", "gemm_tn_hg", "gemm_tn_hc", "gemm_tn", "diffuse_fwd", "diffuse_adj"):
            w["dec_" + k] = 0.0
        for l, fin in enumerate(fins):
            w["dec_seq_fwd"] += sd * (2 * (m - 1) * 2 * n * n * h + 2 * n * (h * m) * 3 * h)
            w["dec_seq_bwd"] += sd * ((m - 1) * 2 * n * n * 3 * h + 2 * n * (h * m) * 3 * h)
            w["dec_gemm_nn"] += 2.0 * rd * (m * fin) * 3 * h * 2
            w["dec_gemm_tn_x"] += 2.0 * rd * (m * fin) * 3 * h
            w["dec_gemm_tn_hg"] += 2.0 * rd * (m * h) * 2 * h
            w["dec_gemm_tn_hc"] += 2.0 * rd * (m * h) * h
            if l == 0:
                w["dec_diffuse_fwd"] += 4.0 * sd * n * fin * m     # first decoder layer only (as above)
            w["dec_diffuse_adj"] += 4.0 * sd * n * fin * (m + 1)
        w["dec_gemm_nn"] += 2 * 2.0 * rd * h * D_IN          # projection forward + d h_top
        w["dec_gemm_tn"] += 2.0 * rd * h * D_IN              # projection weight gradient
        # the persistent decoder kernels (kernels_decoder.h) do the work of the per-step roles in ONE launch each:
        # forward = recurrence + x-part GEMMs + input hop mixes + projection; backward = BPTT + input gradients + d h_top.
        # Priced for their own rows only (`*_persist` is left out of the whole-step sum: the roles above already hold it).
        w["dec_fwd_persist"] = w["dec_seq_fwd"] + sum(2.0 * rd * (m * fin) * 3 * h for fin in fins) \
            + sd * (m - 1) * 2.0 * n * n * D_IN + 2.0 * rd * h * D_IN
        w["dec_bwd_persist"] = w["dec_seq_bwd"] + sum(2.0 * rd * (m * fin) * 3 * h for fin in fins) + 2.0 * rd * h * D_IN
    return w


# kernel symbol behind every role at the cfg2 shapes (64 units, M = 3, 19 nodes)
ROLE_SYMBOLS = {
    "seq_fwd": "seq_fwd2_kernel<64,3,5>", "seq_bwd": "seq_bwd2_kernel<64,3,5>",
    "gemm_nn_xw": "gemm_nnr_kernel<4,2> (layer 0: K=300 in 19 chunks; layer 1: K=192)",
    "gemm_nn_dx": "gemm_nnr_kernel<4,2>", "gemm_tn_x": "gemm_tnq_kernel<5,6,16,bt> (layer 0) + gemm_tnq_kernel<6,6,16,planar> (layer 1)",
    "gemm_tn_hg": "gemm_tnq_kernel<6,4,16,planar>", "gemm_tn_hc": "gemm_tnq_kernel<6,2,16,planar>",
    "diffuse_fwd": "diffuse_fwd_stream_kernel<19>", "diffuse_adj": "diffuse_adj_stream_kernel<19>",
}
# per-clip algorithmic FLOPs (fwd+bwd) and compulsory HBM bytes -> the roofs the whole step is priced against
CLIP_GFLOP = {"cfg1": 1.302 * 12 / 60, "cfg2": 1.302, "cfg3": 2.221, "cfg4": 1.302, "cfg5": 2.674}
CLIP_BYTES = {"cfg1": 8.208e6 * 12 / 60, "cfg2": 8.208e6, "cfg3": 8.213776e6, "cfg4": 8.208e6, "cfg5": 10.1e6}
# the GENUINE reference on the survey container's 8 Xeon vCPUs at the same per-GPU batch (clips/s)
REFERENCE_CPU_CLIPS_PER_S = {"cfg1": 136.0, "cfg2": 173.0, "cfg3": 108.0, "cfg4": 237.0, "cfg5": 110.0}


def kernel_sources_sha256():
    """Hash of everything the HIP library is built from (csrc + the C ABI headers): the build id of the PMC stamps."""
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "eeg_gnn_ssl_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".h", ".cpp", ".hip")) or f == "Makefile":
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def split_bf16_experiment():
    """Gated experiment (never the headline): run the lab binary tools/micro/bf16x3_lab and parse its report -- the layer-1 NN
    GEMM (R x 192 x 192 at the cfg2 row count) with fp32 operands / results computed (a) by the product's true-fp32 MFMA kernel
    and (b) as a three-term bf16 split (6 or 3 of the 9 partial products) on v_mfma_f32_16x16x32_bf16, each against an fp64
    host sum over 2000 sampled rows."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "micro", "bf16x3_lab")
    if not os.path.exists(exe):
        return {"error": "tools/micro/bf16x3_lab not built (make -C tools/micro bf16x3_lab)"}
    txt = subprocess.run([exe, "11"], capture_output=True, text=True, timeout=300).stdout
    return parse_split_bf16_report(txt)


def parse_split_bf16_report(txt):
    """The report of tools/micro/bf16x3_lab -> the `experimental_split_bf16` object of the bench line."""
    import re
    res = {"scope": "lab kernel tools/micro/bf16x3_lab.hip, hoisted NN GEMM of layer 1 only (R=291840, K=192, O=192); NOT in the "
                    "product path, `value` and `dtype` above are true fp32 MFMA", "raw": txt.strip().splitlines()}
    err = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^\s+(.*?)\s+max \|err\| vs fp64 ([0-9.e+-]+)", txt, flags=re.M)}
    tms = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^\s+(fp32 MFMA|bf16 split, \d products)\s+([0-9.]+) ms", txt, flags=re.M)}
    if "fp32 MFMA" in tms:
        res["fp32_mfma_ms"] = tms["fp32 MFMA"]
        for k in ("bf16 split, 6 products", "bf16 split, 3 products"):
            if k in tms:
                key = "six_products" if ", 6 products" in k else "three_products"
                res[key] = {"ms": tms[k], "speedup_vs_fp32_mfma": round(tms["fp32 MFMA"] / tms[k], 3), "max_abs_err_vs_fp64": err.get(k)}
        res["fp32_mfma_max_abs_err_vs_fp64"] = err.get("fp32 MFMA (gemm_nnr_kernel)")
    return res


def cpu_baseline(workload, budget_s=45.0):
    """The oracle (torch-eager restatement of the reference's op sequence, autograd backward) timed on this host's
    cores at the workload's PER-GPU batch (SURVEY.md §8(d) / BASELINE.md §4: B=256 for cfg2; ~2 s per step), fwd + loss +
    bwd, one warm-up + best of up to 3 inside the time budget.  The op stream is ~10^4 small ATen calls per step, so
    more threads is not faster: a few thread counts are probed on a 4-clip sample and the best one is used (`cores`)."""
    from oracle import dcrnn_oracle as orc
    task, filt, t_len, batch, classes = WORKLOADS[workload]
    cfg = orc.DCRNNConfig(filter_type=filt, num_classes=max(classes, 1))
    kind = "nextTimePred" if task == "ssl" else "classification"
    params = {k: v.requires_grad_(True) for k, v in orc.init_params(cfg, kind, seed=0).items()}
    x, y, lengths, sup = synthetic_batch(task, filt, t_len, batch, classes, seed=123)

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
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32)}):
        torch.set_num_threads(nt)
        one(4)                                   # warm-up
        probe[nt] = one(4)
        print(f"[bench] cpu baseline probe: {nt} threads -> {4 / probe[nt]:.1f} clips/s", file=sys.stderr, flush=True)
    best_nt = min(probe, key=probe.get)
    torch.set_num_threads(best_nt)
    one(min(batch, 32))                          # warm-up of the allocator at a larger size
    times = []
    while len(times) < 3 and (not times or time.perf_counter() - t_start < budget_s):
        times.append(one(batch))
    best, reps = min(times), len(times)
    value = batch / best
    ref = REFERENCE_CPU_CLIPS_PER_S.get(workload)
    return {"value": round(value, 2), "unit": "clips/s", "cores": best_nt, "host_logical_cpus": ncpu, "kind": "port",
            "sample": f"{batch} clips x T={t_len} of {workload} = the per-GPU batch (fwd+loss+bwd, {best:.2f} s/step, best of {reps} "
                      f"after a warm-up; torch-eager oracle = op-for-op restatement of the reference; thread count chosen by probe "
                      f"{ {k: round(4 / v, 1) for k, v in probe.items()} } clips/s on 4 clips)",
            "runs_clips_per_s": [round(batch / t, 1) for t in times],
            "spread": round((max(times) - min(times)) / min(times), 3),
            "threads_pinned": os.environ.get("OMP_PROC_BIND", "") + "/" + os.environ.get("OMP_PLACES", ""),
            "note": "host-dependent: the same code measured 131.9-162.7 clips/s on different boxes of the pool (+-20 %); a "
                    "reported baseline, not the target",
            "reference_8vcpu_clips_per_s": ref,
            "ratio_to_reference_8vcpu": None if not ref else round(value / ref, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (default: workload's)")
    ap.add_argument("--layers", type=int, default=LAYERS, help="num_rnn_layers (BASELINE's configs: 2; the reference's SSL recipe "
                    "README.md:91 and its shipped checkpoints use 3 -- decoder layers >= 1 then share one cell)")
    ap.add_argument("--dropout", type=float, default=0.0, help="nn.Dropout probability of the model in train() mode (the "
                    "reference trains the 4-class model of cfg4 with --dropout 0.5, README.md:83; the masks are generated inside "
                    "the head / decoder kernels from a device-resident Philox state, so the captured graph draws fresh ones "
                    "on every replay)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="disable the live per-kernel HIP-event timing")
    ap.add_argument("--host-supports", action="store_true", help="correlation-graph workloads: use supports prepared "
                    "on the host (the reference's DataLoader path) instead of building them on the GPU every step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the "
                    "captured HIP graph of forward+loss+backward (default: replay, at any number of GPUs)")
    ap.add_argument("--no-stream-inputs", action="store_true", help="skip the second timed pass that feeds a fresh pinned "
                    "host batch into the step's input tensors on a side stream every step")
    ap.add_argument("--force-dist", action="store_true", help="single process: create a world-size-1 process group over "
                    "the nccl (= RCCL) backend and issue the gradient all-reduce every step (exercises the RCCL path on one GPU)")
    ap.add_argument("--split-bf16-experiment", action="store_true", help="also run tools/micro/bf16x3_lab (a LAB kernel, not "
                    "the product path: the hoisted NN GEMM as a three-term bf16 split on the bf16 matrix pipe) and report its "
                    "time and error beside the true-fp32 kernel under `experimental_split_bf16`; `value` / `dtype` are untouched")
    ap.add_argument("--tune", action="append", default=[], help="development knob key=value (eeg_dcrnn_set_tuning); loads "
                    "the DEV build libeeg_dcrnn_hip_dev.so instead of the product library")
    ap.add_argument("--lib", default=None, help="development A/B runs only: load this build of the C ABI (e.g. a library built "
                    "from an older commit, kept under build/ab/) instead of the product library; named in config.library")
    args = ap.parse_args()

    t_boot = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device: eeg_gnn_ssl_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the stream the input batches travel on is created FIRST: HIP maps streams round-robin onto a few hardware queues, and a
    # copy stream created after the capture / RCCL streams can share the compute stream's queue (copy and step then
    # serialise: measured 5.5 instead of 3.0 ms/step under --force-dist)
    copy_stream = torch.cuda.Stream()
    with torch.cuda.stream(copy_stream):          # (the queue is bound at the first submission, not at creation)
        torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)     # "nccl" IS RCCL on ROCm
    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))            # host threads per rank (8 ranks share the host)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from eeg_gnn_ssl_amd import DCRNNModel_classification, _lib, ops
    from eeg_gnn_ssl_amd.train_step import TrainStep

    if args.lib:                                                     # development A/B runs only
        _lib._LIB = _lib.EegDcrnnLib(os.path.abspath(args.lib), strict=False)
    if args.tune:                                                    # development A/B runs only
        if not args.lib:
            _lib._LIB = _lib.EegDcrnnLib(_lib.DEV_LIB_PATH)
        for kv in args.tune:
            k, v = kv.split("=")
            _lib._LIB.call("eeg_dcrnn_set_tuning", int(k), int(v))
    task, filt, t_len, batch, classes = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    torch.manual_seed(123)                                   # identical replicas on every rank
    if task == "ssl":
        from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
        model = DCRNNModel_nextTimePred(make_args(filt, args.dropout, args.layers), device=dev).to(dev)
    else:
        model = DCRNNModel_classification(make_args(filt, args.dropout, args.layers), classes, device=dev).to(dev)
    model.train()
    stepper = TrainStep(model, task=task, lr=3e-4, weight_decay=5e-4, max_grad_norm=5.0, always_reduce=args.force_dist)
    device_graph = filt == "dual_random_walk" and not args.host_supports
    # the host-side per-clip graph loop (numpy, 256-512 clips) is only needed to CHECK the device graphs: rank 0 of a
    # single-GPU run does it; data-parallel ranks build their supports on the GPU only
    check_graphs = device_graph and world == 1
    hx, hy, hlen, hsup = synthetic_batch(task, filt, t_len, batch, classes, seed=123 + rank, host_supports=not device_graph or check_graphs)
    x, y, lengths = hx.to(dev), hy.to(dev), hlen.to(dev)
    supports = [s.to(dev) for s in hsup] if hsup is not None else None
    if device_graph and not check_graphs:
        supports = None
    elif device_graph:
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
    if not args.no_graph:
        # forward + loss + backward replayed as ONE HIP graph at any number of GPUs (one launch per step and rank
        # instead of ~60: what the multi-GPU scaling hinges on); the all-reduce + fused clip/Adam stay eager
        try:
            stepper.capture(x, y, lengths, supports)
            graphed = True
        except Exception as e:                                   # noqa: BLE001 -- fall back to eager launches
            log(f"HIP graph capture failed ({type(e).__name__}: {e}); launching eagerly")
            torch.cuda.synchronize()
    one_step = stepper.replay_step if graphed else (lambda: stepper.step(x, y, lengths, supports))

    clock_buf = torch.zeros(3, dtype=torch.int64, device=dev)

    def timed(step_fn, marks=None):
        """the contract's timed region: W untimed steps, barrier + synchronize, K steps on the wall clock, barrier + synchronize.
        marks: a HIP event is recorded on the launch stream in front of every timed step and behind the last one (K + 1 records of
        ~1 us each) so that the line can show the per-step durations the wall-clock mean is made of."""
        for _ in range(args.warmup):
            step_fn()
        sync_all()
        cur = torch.cuda.current_stream()
        t0 = time.perf_counter()
        for k in range(args.steps):
            if marks is not None:
                marks[k].record(cur)
            loss = step_fn()
        if marks is not None:
            marks[args.steps].record(cur)
        sync_all()
        dt = time.perf_counter() - t0
        # shader clock under sustained fp32-MFMA load right behind the timed steps (200 us on every SIMD, outside the timed region)
        if marks is not None and hasattr(lib._dll, "eeg_dcrnn_prof_clock_probe"):
            clock_buf.zero_()
            lib.call("eeg_dcrnn_prof_clock_probe", ctypes.c_void_p(clock_buf.data_ptr()), ctypes.c_void_p(cur.cuda_stream))
            torch.cuda.synchronize()
        return dt, loss

    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    elapsed, loss = timed(one_step, marks)
    step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps)]
    cyc, ticks = (int(v) for v in clock_buf.tolist()[:2])
    sclk_mhz = round(cyc / ticks * 100.0, 1) if ticks > 0 else None
    log(f"timed {args.steps} steps ({'graph replay' if graphed else 'eager'}): {elapsed / args.steps * 1e3:.3f} ms/step "
        f"(first {step_ms[0]:.3f}, median {sorted(step_ms)[len(step_ms) // 2]:.3f}, last {step_ms[-1]:.3f}; shader clock under MFMA load behind the last step {sclk_mhz} MHz)")

    # second timed pass: every step first receives a FRESH batch from pinned host memory (the trainer's situation: at
    # 85 k clips/s the input stream is ~40 GB/s per GPU).  The step is captured on TWO input sets; the host-to-device copy
    # of batch k+1 runs on a side stream straight into the set the NEXT replay reads while batch k computes: no staging
    # buffer and no device-to-device refresh (round 2 paid 0.39 ms/step for that).  Eager launches: one input set, the copy
    # waits for the step that reads it.
    streamed = None
    if not args.no_stream_inputs:
        pin = [t.pin_memory() for t in (hx, hy)]
        side = copy_stream
        sets = [(x, y)]
        if graphed:
            x2, y2 = torch.empty_like(x), torch.empty_like(y)
            x2.copy_(x); y2.copy_(y)
            stepper.capture(x2, y2, lengths, supports, slot=1)
            sets.append((x2, y2))
        landed = [torch.cuda.Event() for _ in sets]      # batch has arrived in set i
        done = [torch.cuda.Event() for _ in sets]        # the step that read set i has finished
        for e in done:
            e.record()
        state = {"k": 0}

        def fetch(i):
            with torch.cuda.stream(side):
                side.wait_event(done[i])                         # the previous contents of set i were consumed
                sets[i][0].copy_(pin[0], non_blocking=True)
                sets[i][1].copy_(pin[1], non_blocking=True)
                landed[i].record(side)

        fetch(0)

        def streamed_step():
            i = state["k"] % len(sets)
            state["k"] += 1
            cur = torch.cuda.current_stream()
            if len(sets) > 1:
                fetch((i + 1) % len(sets))                       # next batch travels while this step computes
            cur.wait_event(landed[i])
            out = stepper.replay_step(i) if graphed else stepper.step(sets[i][0], sets[i][1], lengths, supports)
            done[i].record(cur)
            if len(sets) == 1:
                fetch(0)
            return out

        el2, _ = timed(streamed_step)
        t2 = torch.tensor([el2], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        el2 = float(t2.item())
        streamed = {"value": round(batch * world / (el2 / args.steps), 1), "unit": "clips/s",
                    "ms_per_step": round(el2 / args.steps * 1e3, 3),
                    "host_bytes_per_step_per_gpu": int(hx.numel() * 4 + hy.numel() * hy.element_size()),
                    "note": ("a fresh batch per step from pinned host memory: the step is captured on two input sets and the H2D "
                             "copy of batch k+1 lands in the idle set on a side stream while batch k computes (no staging buffer, "
                             "no device-side copy)") if graphed else
                            "a fresh batch per step from pinned host memory, copied on a side stream between eager steps"}
        log(f"streamed inputs: {streamed['ms_per_step']} ms/step")

    prof = {}
    kernel_clock_mhz = {}
    if not args.no_prof:
        # per-kernel durations: the SAME K steps once more, launched eagerly with a HIP-event pair
        # around every launch on the launch stream (events cannot be read back from a graph replay;
        # the pairs themselves cost ~0.2 ms/step, which is why they are kept out of the timed region)
        clk = torch.zeros(4, dtype=torch.int64, device=dev)     # in-kernel clock samples of the two-wave recurrent kernels
        has_clk = hasattr(lib._dll, "eeg_dcrnn_prof_clock_samples") and not lib.is_dev_build
        if has_clk:
            lib.call("eeg_dcrnn_prof_clock_samples", ctypes.c_void_p(clk.data_ptr()))
        lib.query("eeg_dcrnn_prof_enable", 1)
        for _ in range(args.steps):
            stepper.step(x, y, lengths, supports)
        torch.cuda.synchronize()
        lib.query("eeg_dcrnn_prof_enable", 0)
        if has_clk:
            lib.call("eeg_dcrnn_prof_clock_samples", None)
            c = clk.tolist()
            kernel_clock_mhz = {"seq_fwd": round(c[0] / c[1] * 100.0, 1) if c[1] > 0 else None,
                                "seq_bwd": round(c[2] / c[3] * 100.0, 1) if c[3] > 0 else None}
        buf = ctypes.create_string_buffer(1 << 16)
        lib.call("eeg_dcrnn_prof_report", buf, len(buf))
        for line in buf.value.decode().strip().splitlines():
            name, cnt, ms = line.split()
            prof[name] = (int(cnt), float(ms))
    # exchange + optimiser tail (all-reduce of the flat bucket when a process group exists, norm + fused clip/Adam):
    # HIP events around reduce_and_update() on the launch stream, gradients left as they are
    # (measured on a scratch copy of the optimiser state: with a process group every call all-reduces -- SUMS -- the bucket in
    # place, so repeating it on the live buffers would grow the gradient by world^steps and apply `steps` extra updates)
    keep = [t.clone() for t in (stepper.fp.flat, stepper.fp.flat_grad, stepper.exp_avg, stepper.exp_avg_sq)]
    keep_count = stepper.step_count
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tail_ms = 0.0
    for _ in range(args.steps):
        stepper.fp.flat_grad.copy_(keep[1])                      # the same (finite) gradient every time
        e0.record()
        stepper.reduce_and_update()
        e1.record()
        torch.cuda.synchronize()
        tail_ms += e0.elapsed_time(e1) / args.steps
    with torch.no_grad():
        for dst, src in zip((stepper.fp.flat, stepper.fp.flat_grad, stepper.exp_avg, stepper.exp_avg_sq), keep):
            dst.copy_(src)
    stepper.step_count = keep_count
    per_rank = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        allr = [torch.empty_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in allr]
        elapsed = max(float(t.item()) for t in allr)             # MAX over ranks
        backend = dist.get_backend()
        world_seen = dist.get_world_size()
    else:
        per_rank_ms = [round(elapsed / args.steps * 1e3, 3)]
        backend = dist.get_backend() if dist.is_initialized() else None
        world_seen = dist.get_world_size() if dist.is_initialized() else 1
    loss_val = float(loss.item())
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    clips_per_s = batch * world / (elapsed / args.steps)
    work = algorithmic_work(filt, t_len, batch, task, args.layers)
    kernels = {}
    for name, (cnt, ms) in prof.items():
        per_step_ms = ms / args.steps
        ent = {"launches_per_step": cnt / args.steps, "ms_per_step": round(per_step_ms, 4),
               "avg_launch_ms": round(per_step_ms / (cnt / args.steps), 4)}
        if name in ROLE_SYMBOLS and args.workload in ("cfg2", "cfg4"):
            ent["symbol"] = ROLE_SYMBOLS[name]
        if name in work and work[name] > 0 and per_step_ms > 0:
            if "diffuse" in name or name == "corr_gram":
                gbs = work[name] / (per_step_ms * 1e-3) / 1e9
                ent.update(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                           frac_of_achievable=round(gbs / ACHIEVABLE_HBM_GBS, 4))
            else:
                tf = work[name] / (per_step_ms * 1e-3) / 1e12
                ent.update(bound="mfma", achieved=round(tf, 2), peak=PEAK_MFMA_F32_TFLOPS, unit="TFLOP/s",
                           frac=round(tf / PEAK_MFMA_F32_TFLOPS, 4))
        kernels[name] = ent
    roofline = None
    timed_k = {k: v for k, v in kernels.items() if "bound" in v}
    if timed_k:
        dom = max(timed_k, key=lambda k: timed_k[k]["ms_per_step"])     # the kernel SYMBOL with the most time per step
        d = timed_k[dom]
        # HBM bytes per launch from the committed PMC passes of the same command (tools/pmc_traffic.sh).  The file is stamped
        # with the hash of the kernel sources it was collected on: a stale file yields `traffic: null` + a warning
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel_sources_sha256") == kernel_sources_sha256():
                traffic = tj["traffic_bytes_per_launch"]
            else:
                traffic_note = (f"profiles/pmc_traffic_{args.workload}.json was collected on other kernel sources "
                                f"(stamp {str(tj.get('kernel_sources_sha256'))[:12]} != {kernel_sources_sha256()[:12]}): not used")
                print("[bench] WARNING: " + traffic_note, file=sys.stderr, flush=True)
        for name, tb in (traffic or {}).items():
            for k in kernels:
                if k == name or (k.startswith(name + "_") and name in ("gemm_nn", "gemm_tn")):
                    kernels[k]["traffic_bytes_per_launch_pmc" + ("" if k == name else "_class_avg")] = tb
        classes_ms = {}
        for k, v in timed_k.items():
            cls = "gemm_tn" if k.startswith("gemm_tn") else "gemm_nn" if k.startswith("gemm_nn") else k
            c = classes_ms.setdefault(cls, {"ms_per_step": 0.0, "work": 0.0, "bound": v["bound"]})
            c["ms_per_step"] += v["ms_per_step"]
            c["work"] += work[k]
        by_class = {k: {"ms_per_step": round(v["ms_per_step"], 4),
                        "frac": round(v["work"] / (v["ms_per_step"] * 1e-3) / (PEAK_HBM_GBS * 1e9 if v["bound"] == "hbm" else PEAK_MFMA_F32_TFLOPS * 1e12), 4)}
                    for k, v in classes_ms.items()}
        flops = sum(v for k, v in work.items() if "diffuse" not in k and k != "corr_gram" and not k.endswith("_persist"))
        # frac_at_held_clock: the fraction of the cycles the chip actually ran.  For the two-wave recurrent kernels the clock is
        # sampled INSIDE the kernel (eeg_dcrnn_prof_clock_samples), for the others by the MFMA-burn probe behind the timed steps.
        # In steady state both read 2.37-2.42 GHz (the peak is a 2.4 GHz figure); a process that has just started runs its first
        # tens of milliseconds at 1.9-2.2 GHz, which is what the few-step PMC passes see (profiles/README.md) and what first5_ms shows.
        for k, v in kernels.items():
            if v.get("bound") != "mfma":
                continue
            mhz = kernel_clock_mhz.get(k) or sclk_mhz
            if mhz:
                v["clock_mhz"] = mhz
                v["clock_source"] = "in-kernel sample" if kernel_clock_mhz.get(k) else "MFMA-burn probe behind the step"
                v["frac_at_held_clock"] = round(v["frac"] * PEAK_CLOCK_MHZ / mhz, 4)
        roofline = {"kernel": dom, "symbol": d.get("symbol"), "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                    "unit": d["unit"], "frac": d["frac"],
                    # the MFMA peak is a 2.4 GHz figure; the part holds less under sustained fp32 matrix load.  frac_at_held_clock =
                    # frac x 2400 / (shader clock measured right behind the last timed step): the share of the cycles the chip
                    # actually ran.  `frac` (against the spec-sheet peak) stays the reported figure.
                    "shader_clock_mhz_under_load": sclk_mhz, "kernel_clock_mhz": kernel_clock_mhz or None,
                    "frac_at_held_clock": d.get("frac_at_held_clock"), "clock_source": d.get("clock_source"),
                    "traffic": (traffic or {}).get(dom), "traffic_note": traffic_note, "avg_launch_ms": d["avg_launch_ms"],
                    "top_class": max(by_class, key=lambda k: by_class[k]["ms_per_step"]), "by_class": by_class,
                    "kernels": kernels,
                    "kernel_ms_per_step_total": round(sum(v["ms_per_step"] for v in kernels.values()), 3),
                    "whole_step_flops": round(flops / 1e9, 1),
                    "whole_step_mfma_frac": round(flops / (ms_per_step * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, 4)}
    per_gpu = clips_per_s / world
    mfma_roof = PEAK_MFMA_F32_TFLOPS * 1e12 / (CLIP_GFLOP[args.workload] * 1e9)          # clips/s/GPU, SURVEY.md §8(d)
    out = {
        "metric": "EEG clips/sec (60s, 19ch, K=2, 2-layer x64) fwd+bwd",
        "value": round(clips_per_s, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        # what the wall-clock mean is made of (HIP events in front of every timed step, rank 0): a fresh process shows whether
        # its first replays are slower than the rest (clock ramp / first-use costs) instead of hiding it in the mean
        "ms_per_step_p50": round(sorted(step_ms)[len(step_ms) // 2], 3),
        "first5_ms": [round(v, 3) for v in step_ms[:5]], "last5_ms": [round(v, 3) for v in step_ms[-5:]],
        "ms_per_step_events_mean": round(sum(step_ms) / len(step_ms), 3),
        "shader_clock_mhz_under_load": sclk_mhz,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": DESCR[args.workload], "per_gpu_batch": batch, "global_batch": batch * world,
                   "clip_len": t_len, "parallelism": f"dp{world}", "optimizer_step_included": True, "dropout": args.dropout, "num_rnn_layers": args.layers,
                   "supports": ("per-clip correlation graph + dual random-walk supports built on the GPU inside the step"
                                if device_graph else "prepared on the host (distance graph is fixed)"
                                if filt == "laplacian" else "prepared on the host"),
                   "launch": "hip-graph replay (fwd+loss+bwd) + eager all-reduce/clip+Adam" if graphed else "eager",
                   "timed_region": f"{args.steps} steps on one batch resident in HBM = {elapsed * 1e3:.1f} ms wall",
                   "library": (("A/B build " + args.lib + " ") if args.lib else "") + ("DEV build with tuning knobs " + ",".join(args.tune) if args.tune else ("" if args.lib else "product")),
                   "final_loss": round(loss_val, 5)},
        "distributed": {"world_size": world_seen, "backend": backend, "per_rank_ms_per_step": per_rank_ms,
                        "all_reduce_issued": bool(stepper.reduce),
                        "reduce_and_update_ms_per_step": round(tail_ms, 4),
                        "exchange": "one all-reduce of the flat fp32 gradient bucket per step "
                                    f"({stepper.fp.flat_grad.numel() * 4} bytes), outside the HIP graph"
                                    + ("" if stepper.reduce else " (no process group: not issued in this run)")},
        # the whole step against BOTH roofs: the binding fp32-MFMA roof and the HBM roof north_star names
        # executed_mfma_frac = the FLOPs the kernels actually execute / time / peak (leads); *_survey_flops prices the step
        # with SURVEY's per-clip figure, which includes the layer-0 dX that neither the reference's autograd nor this
        # library computes (21 % more FLOPs at cfg2)
        "whole_step": {"executed_mfma_frac": None if roofline is None else roofline["whole_step_mfma_frac"],
                       "executed_mfma_frac_at_held_clock": None if (roofline is None or not sclk_mhz) else
                       round(roofline["whole_step_mfma_frac"] * PEAK_CLOCK_MHZ / sclk_mhz, 4),
                       "executed_gflop": None if roofline is None else roofline["whole_step_flops"],
                       "mfma_roof_clips_per_s_per_gpu_survey_flops": round(mfma_roof, 0),
                       "mfma_roof_frac_survey_flops": round(per_gpu / mfma_roof, 4),
                       "hbm_frac": round(per_gpu * CLIP_BYTES[args.workload] / (PEAK_HBM_GBS * 1e9), 4),
                       "hbm_roof_clips_per_s_per_gpu": round(PEAK_HBM_GBS * 1e9 / CLIP_BYTES[args.workload], 0)},
        # `value` is the training-loop rate (optimiser step included); the kernel figure takes the optimiser tail
        # (norm + fused clip/Adam, live HIP-event times) out of the step
        "fwd_bwd_only": (None if not prof or world > 1 else {
            "clips_per_s": round(batch / ((ms_per_step - sum(prof.get(k, (0, 0.0))[1] for k in ("grad_sqnorm", "clip_adam")) / args.steps) * 1e-3), 1),
            "excluded_ms_per_step": round(sum(prof.get(k, (0, 0.0))[1] for k in ("grad_sqnorm", "clip_adam")) / args.steps, 4)}),
        "streamed_inputs": streamed,
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle on host cores, per-GPU batch)")
        out["cpu_baseline"] = cpu_baseline(args.workload)
        out["speedup_vs_cpu_baseline"] = round(clips_per_s / out["cpu_baseline"]["value"], 1)
    if world == 1 and args.split_bf16_experiment:
        out["experimental_split_bf16"] = split_bf16_experiment()
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
