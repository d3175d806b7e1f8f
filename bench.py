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

if __name__ == "__main__":
    # multi-process GPU work on this image needs dmabuf IPC (RCCL / tensor sharing across processes fail with `hipIpcGetMemHandle:
    # invalid argument` in legacy mode); the driver's environment exports it already -- a bare shell may not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    # cfg3 fed with RAW signals: raw (B,19,60*200) -> log|FFT| features + z-score (eeg_dcrnn_fft_features) -> per-clip correlation
    # graph from the un-standardised features -> step: the reference's DataLoader-side chain on the device
    "raw": ("detection", "dual_random_walk", 60, 256, 1),
}
RAW_WINDOW = 200  # samples per 1-s step (dataloader_detection.py:25-26: FREQUENCY = 200)
T_OUT = 12        # SSL prediction horizon (args.py:52-56)
DESCR = {
    "cfg2": "BASELINE cfg2: DCRNN detection, distance graph, clip_len=60, batch=256/GPU, K=2, 2x64, synthetic FFT inputs",
    "cfg3": "BASELINE cfg3: DCRNN detection, correlation graph (per-clip adj), clip_len=60, batch=256/GPU",
    "cfg4": "BASELINE cfg4: DCRNN 4-class classification, distance graph, clip_len=60, batch=256/GPU (2048 over 8)",
    "cfg1": "BASELINE cfg1: DCRNN detection, distance graph, clip_len=12, batch=4 (plumbing)",
    "cfg5": "BASELINE cfg5: SSL seq2seq pretrain (encoder 60 s + decoder 12 s), correlation graph, batch=512/GPU (4096 over 8)",
    "raw": "cfg3 from RAW signals: (256,19,12000) resampled EEG -> log|FFT| + z-score -> per-clip correlation graph -> detection step, all on the GPU",
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


HBM_ROLES = ("corr_gram", "fft_features")


def is_hbm_role(name):
    """roles priced in bytes against the HBM roof (everything else: FLOPs against the fp32-MFMA roof)"""
    return "diffuse" in name or "spec_mix" in name or name in HBM_ROLES


def algorithmic_work(filter_type, t_len, batch, task="detection", layers=LAYERS, raw=False, spectral=False, fused=True):
    """Per-step algorithmic FLOPs / bytes of every profiled kernel role.  Roles = the names the
    library's event recorder uses; at the benchmark shapes each role is ONE kernel symbol per layer
    (reported by the library's recorder with every launch), so `roofline.kernels` carries the symbols and `roofline.by_symbol`
    ranks them like a kernel trace would."""
    m = (2 if filter_type == "dual_random_walk" else 1) * K_DIFF + 1
    n, h = N_NODES, H_UNITS
    s = t_len * batch
    r = s * n
    fins = [D_IN] + [h] * (layers - 1)
    w = {k: 0.0 for k in ("seq_fwd", "seq_bwd", "gemm_nn_xw", "gemm_nn_dx", "gemm_tn_x", "gemm_tn_hg", "gemm_tn_hc",
                          "diffuse_fwd", "diffuse_adj")}
    if filter_type == "dual_random_walk":
        w["corr_gram"] = 4.0 * s * n * D_IN          # per-clip correlation graph: every clip read once
    if raw:                                          # featurisation: raw windows read once, feat_raw + feat_std written once
        w["fft_features"] = 4.0 * s * n * (RAW_WINDOW + 2 * D_IN)
    for l, fin in enumerate(fins):
        w["seq_fwd"] += s * (2 * (m - 1) * 2 * n * n * h + 2 * n * (h * m) * 3 * h)
        w["seq_bwd"] += s * ((m - 1) * 2 * n * n * 3 * h + 2 * n * (h * m) * 3 * h)
        w["gemm_nn_xw"] += 2.0 * r * (m * fin) * 3 * h
        w["gemm_tn_x"] += 2.0 * r * (m * fin) * 3 * h
        w["gemm_tn_hg"] += 2.0 * r * (m * h) * 2 * h
        w["gemm_tn_hc"] += 2.0 * r * (m * h) * h
        if spectral:
            # fused: the two-wave recurrent kernels do the spectral node mixes themselves (csrc/kernels_seq.h SPEC): forward U Yh (3H
            # wide) + U^T h, U^T (r*h) (H wide each), backward U^T [dR|dU|dC] -- executed MFMA work on top of the cell's; the layers
            # above the first take their transformed input from the layer below.  Not fused: the same mixes as HBM-bound passes
            if fused:
                w["seq_fwd"] += s * (2.0 * n * n * 3 * h + 2 * 2.0 * n * n * h)
                w["seq_bwd"] += s * (2.0 * n * n * 3 * h)
            # spectral form of the hoisted x-part (shared symmetric support; csrc/spec_common.h): the GEMMs contract over Fin, not
            # M * Fin (EXECUTED FLOPs), framed by HBM-bound node mixes (bytes = every operand read once + every result written once)
            w["gemm_nn_xw"] += 2.0 * r * fin * 3 * h - 2.0 * r * (m * fin) * 3 * h
            w["gemm_tn_x"] += 2.0 * r * fin * 3 * h - 2.0 * r * (m * fin) * 3 * h
            w["gemm_tn_hg"] += 2.0 * r * h * 2 * h - 2.0 * r * (m * h) * 2 * h
            w["gemm_tn_hc"] += 2.0 * r * h * h - 2.0 * r * (m * h) * h
            for k_, wd in (("spec_mix_x", fin if (l == 0 or not fused) else 0), ("spec_mix_y", 0 if fused else 3 * h),
                           ("spec_mix_dy", 0 if fused else 3 * h), ("spec_mix_h", 0 if fused else 2 * h)):
                w[k_] = w.get(k_, 0.0) + 2 * 4.0 * r * wd
            if l > 0:
                w["gemm_nn_dx"] += 2.0 * r * 3 * h * fin
                w["spec_mix_dx"] = w.get("spec_mix_dx", 0.0) + 2 * 4.0 * r * fin
            continue
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


def synthetic_raw_signals(batch, t_len, seed):
    """Synthetic resampled EEG (B, N, T*200), in microvolts: four AR(1) sources per clip (different spectra) mixed into the 19
    electrodes with random weights + white sensor noise -- channels then differ in spectral SHAPE, so the correlation graph of
    their log|FFT| features has decided top-3 neighbours (white noise alone gives |correlation| in 0.984..0.987 for every pair and
    top-3 margins at fp32 rounding level: a graph of coin flips)."""
    from scipy.signal import lfilter
    g = torch.Generator().manual_seed(seed)
    n_src, total = 4, t_len * RAW_WINDOW
    e = torch.randn(batch, n_src, total, generator=g).numpy().astype(np.float64)
    src = np.stack([lfilter([1.0], [1.0, -r], e[:, k], axis=-1) * np.sqrt(1.0 - r * r) for k, r in enumerate((0.5, 0.8, 0.95, -0.5))], axis=1)
    mix = torch.randn(batch, N_NODES, n_src, generator=g).numpy().astype(np.float64)
    noise = torch.randn(batch, N_NODES, total, generator=g).numpy()
    return torch.from_numpy((20.0 * np.einsum("bnk,bkt->bnt", mix, src) + 5.0 * noise).astype(np.float32))


def per_launch_work(filter_type, t_len, batch, layers=LAYERS, spectral=False, fused=True):
    """Encoder roles: the algorithmic work of every launch of a step IN LAUNCH ORDER (forward roles bottom-up, backward roles
    top-down), so that a role whose layers run different kernel instantiations (e.g. `gemm_tn_x`: batch-major layer 0, planar
    above) can be priced per SYMBOL.  Sums equal algorithmic_work()."""
    m = (2 if filter_type == "dual_random_walk" else 1) * K_DIFF + 1
    n, h = N_NODES, H_UNITS
    s = t_len * batch
    r = s * n
    fins = [D_IN] + [h] * (layers - 1)
    fwd = {"seq_fwd": [s * (2 * (m - 1) * 2 * n * n * h + 2 * n * (h * m) * 3 * h) for _ in fins],
           "gemm_nn_xw": [2.0 * r * (m * fin) * 3 * h for fin in fins],
           "diffuse_fwd": [4.0 * s * n * fins[0] * m]}
    bwd = {"seq_bwd": [s * ((m - 1) * 2 * n * n * 3 * h + 2 * n * (h * m) * 3 * h) for _ in fins],
           "gemm_tn_x": [2.0 * r * (m * fin) * 3 * h for fin in fins],
           "gemm_tn_hg": [2.0 * r * (m * h) * 2 * h for _ in fins],
           "gemm_tn_hc": [2.0 * r * (m * h) * h for _ in fins],
           "gemm_nn_dx": [2.0 * r * 3 * h * (m * fin) for fin in fins[1:]],
           "diffuse_adj": [4.0 * s * n * fin * (m + 1) for fin in fins[1:]]}
    if spectral:
        fwd = {"seq_fwd": [v + (s * (2.0 * n * n * 3 * h + 2 * 2.0 * n * n * h) if fused else 0.0) for v in fwd["seq_fwd"]],
               "gemm_nn_xw": [2.0 * r * fin * 3 * h for fin in fins],
               "spec_mix_x": [8.0 * r * fin for fin in (fins[:1] if fused else fins)],
               "spec_mix_y": [] if fused else [8.0 * r * 3 * h for _ in fins]}
        bwd["seq_bwd"] = [v + (s * (2.0 * n * n * 3 * h) if fused else 0.0) for v in bwd["seq_bwd"]]
        bwd.update({"gemm_tn_x": [2.0 * r * fin * 3 * h for fin in fins], "gemm_nn_dx": [2.0 * r * 3 * h * fin for fin in fins[1:]],
                    "gemm_tn_hg": [2.0 * r * h * 2 * h for _ in fins], "gemm_tn_hc": [2.0 * r * h * h for _ in fins],
                    "spec_mix_h": [] if fused else [8.0 * r * h for _ in fins for _ in (0, 1)],
                    "spec_mix_dy": [] if fused else [8.0 * r * 3 * h for _ in fins], "spec_mix_dx": [8.0 * r * fin for fin in fins[1:]]})
        bwd.pop("diffuse_adj")
    out = dict(fwd)
    out.update({k: v[::-1] for k, v in bwd.items()})
    return out


def merge_paired_roles(work, launch_work, roles):
    """Round 5: the library launches the two h-part weight-gradient GEMMs of a cell (roles `gemm_tn_hg`, `gemm_tn_hc`) as ONE paired
    kernel where their plans agree (role `gemm_tn_h`, `dec_gemm_tn_h` inside the decoder operator): the work tables follow the
    roles the recorder actually reports -- the pair is priced with the sum of the two problems, nothing is counted twice."""
    for pre in ("", "dec_"):
        pair, a, b = pre + "gemm_tn_h", pre + "gemm_tn_hg", pre + "gemm_tn_hc"
        if pair in roles and a not in roles and b not in roles and a in work and b in work:
            work[pair] = work.pop(a) + work.pop(b)
            if a in launch_work and b in launch_work:
                launch_work[pair] = [x + y for x, y in zip(launch_work.pop(a), launch_work.pop(b))]
    # Round 6: under the spectral form the x-part and both h-part weight-gradient problems of a cell are ONE launch (role
    # `gemm_tn_f`, csrc/kernels_gemm_f.h): priced with the sum of the three
    fused, parts = "gemm_tn_f", ("gemm_tn_x", "gemm_tn_hg", "gemm_tn_hc")
    if fused in roles and not any(r in roles for r in parts + ("gemm_tn_h",)) and all(r in work for r in parts):
        work[fused] = sum(work.pop(r) for r in parts)
        if all(r in launch_work for r in parts):
            launch_work[fused] = [sum(v) for v in zip(*(launch_work.pop(r) for r in parts))]
    # ... and the input gradient of a spectral layer is one kernel (role `gemm_dx_f`: the K = 3H GEMM + the node mix back): priced with
    # the GEMM's FLOPs; the mix pass it replaces (`spec_mix_dx`) is no longer launched
    if "gemm_dx_f" in roles and "gemm_nn_dx" not in roles and "gemm_nn_dx" in work:
        work["gemm_dx_f"] = work.pop("gemm_nn_dx")
        if "gemm_nn_dx" in launch_work:
            launch_work["gemm_dx_f"] = launch_work.pop("gemm_nn_dx")
        work.pop("spec_mix_dx", None)
        launch_work.pop("spec_mix_dx", None)
    return work, launch_work


def short_symbol(sym):
    """the recorder's kernel spelling without blanks and without the trailing default `false` template flags"""
    sym = sym.replace(" ", "")
    while sym.endswith(",false>"):
        sym = sym[:-len(",false>")] + ">"
    return sym


# per-clip algorithmic FLOPs (fwd+bwd) and compulsory HBM bytes -> the roofs the whole step is priced against
CLIP_GFLOP = {"cfg1": 1.302 * 12 / 60, "cfg2": 1.302, "cfg3": 2.221, "cfg4": 1.302, "cfg5": 2.674, "raw": 2.221}
CLIP_BYTES = {"cfg1": 8.208e6 * 12 / 60, "cfg2": 8.208e6, "cfg3": 8.213776e6, "cfg4": 8.208e6, "cfg5": 10.1e6,
              "raw": 8.213776e6 + 4.0 * N_NODES * 60 * (RAW_WINDOW + 2 * D_IN)}
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


def _oracle_step_fn(workload, device):
    """fwd + loss + bwd of the oracle (torch-eager restatement of the reference's op sequence, autograd backward) on the
    workload's per-GPU batch, as a closure one(nclips) -> seconds (synchronised when on the GPU)"""
    from oracle import dcrnn_oracle as orc
    task, filt, t_len, batch, classes = WORKLOADS[workload]
    cfg = orc.DCRNNConfig(filter_type=filt, num_classes=max(classes, 1))
    kind = "nextTimePred" if task == "ssl" else "classification"
    params = {k: v.to(device).requires_grad_(True) for k, v in orc.init_params(cfg, kind, seed=0).items()}
    x, y, lengths, sup = synthetic_batch(task, filt, t_len, batch, classes, seed=123)
    x, y, sup = x.to(device), y.to(device), [s.to(device) for s in sup]
    on_gpu = torch.device(device).type == "cuda"

    def one(nclips):
        for p in params.values():
            p.grad = None
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if task == "ssl":
            pred = orc.next_time_pred_forward(params, cfg, x[:nclips], y[:nclips], [s[:nclips] for s in sup])
            loss = orc.regression_loss(y[:nclips], pred, loss_fn="MAE")
        else:
            logits = orc.classification_forward(params, cfg, x[:nclips], lengths[:nclips], [s[:nclips] for s in sup])
            loss = orc.bce_with_logits(logits, y[:nclips]) if classes == 1 else orc.cross_entropy(logits, y[:nclips])
        loss.backward()
        if on_gpu:
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    return one, batch, t_len


def aten_gpu_baseline(workload, dev, value):
    """Context, never credit: the reference's op sequence (the oracle: ~10^4 small ATen calls per step, autograd backward) on
    device tensors of the SAME MI355X, i.e. what PyTorch-ROCm's stock kernels do with the reference model -- fwd + loss + bwd
    at the per-GPU batch, eager launches, one warm-up + best of 3, outside any timed region of the product path."""
    one, batch, t_len = _oracle_step_fn(workload, dev)
    one(batch)
    times = [one(batch) for _ in range(3)]
    best = min(times)
    torch.cuda.empty_cache()
    return {"value": round(batch / best, 1), "unit": "clips/s", "kind": "oracle op sequence on stock ATen (rocBLAS / elementwise) kernels, "
            "eager, fp32, same GPU", "sample": f"{batch} clips x T={t_len} of {workload}, fwd+loss+bwd (no optimiser step), {best:.3f} s/step, "
            f"best of 3 after a warm-up", "runs_clips_per_s": [round(batch / t, 1) for t in times],
            "product_over_aten": round(value / (batch / best), 1),
            "note": "context only: launch-bound eager execution of the reference's Python loop (60 steps x 2 layers x ~40 ops)"}


def cpu_baseline(workload, budget_s=45.0):
    """The oracle (torch-eager restatement of the reference's op sequence, autograd backward) timed on this host's
    cores at the workload's PER-GPU batch (SURVEY.md §8(d) / BASELINE.md §4: B=256 for cfg2; ~2 s per step), fwd + loss +
    bwd, one warm-up + best of up to 3 inside the time budget.  The op stream is ~10^4 small ATen calls per step, so
    more threads is not automatically faster: thread counts {8, 16, 32, 64, 128, all} are probed on a 32-clip sample (after a
    warm-up at each count) and the best one is used and stated (`cores`)."""
    one, batch, t_len = _oracle_step_fn(workload, "cpu")
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()
    probe = {}
    n_probe = min(batch, 32)
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)}):
        torch.set_num_threads(nt)
        one(min(batch, 8))                       # warm-up
        probe[nt] = one(n_probe)
        print(f"[bench] cpu baseline probe: {nt} threads -> {n_probe / probe[nt]:.1f} clips/s on {n_probe} clips", file=sys.stderr, flush=True)
        if probe[nt] > 2.5 * min(probe.values()):
            break                                # far past the optimum: more threads only get slower
    best_nt = min(probe, key=probe.get)
    torch.set_num_threads(best_nt)
    one(n_probe)                                 # warm-up of the allocator at the chosen thread count
    times = []
    while len(times) < 3 and (not times or time.perf_counter() - t_start < budget_s):
        times.append(one(batch))
    best, reps = min(times), len(times)
    value = batch / best
    ref = REFERENCE_CPU_CLIPS_PER_S.get(workload)
    return {"value": round(value, 2), "unit": "clips/s", "cores": best_nt, "host_logical_cpus": ncpu, "kind": "port",
            "sample": f"{batch} clips x T={t_len} of {workload} = the per-GPU batch (fwd+loss+bwd, {best:.2f} s/step, best of {reps} "
                      f"after a warm-up; torch-eager oracle = op-for-op restatement of the reference; thread count chosen by probe "
                      f"{ {k: round(n_probe / v, 1) for k, v in probe.items()} } clips/s on {n_probe} clips)",
            "runs_clips_per_s": [round(batch / t, 1) for t in times],
            "spread": round((max(times) - min(times)) / min(times), 3),
            "threads_pinned": os.environ.get("OMP_PROC_BIND", "") + "/" + os.environ.get("OMP_PLACES", ""),
            "note": "host-dependent: the same code measured 131.9-162.7 clips/s on different boxes of the pool (+-20 %); a "
                    "reported baseline, not the target",
            "reference_8vcpu_clips_per_s": ref,
            "ratio_to_reference_8vcpu": None if not ref else round(value / ref, 3)}


class Ctx:
    """what every workload of one bench process shares"""

    def __init__(self, args):
        self.args = args
        self.t_boot = time.perf_counter()
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def log(self, msg):
        if self.rank == 0:
            print(f"[bench +{time.perf_counter() - self.t_boot:6.1f}s] {msg}", file=sys.stderr, flush=True)

    def sync_all(self):
        if self.world > 1:
            dist.barrier()
        self.sync()

    # --emulator (TEST INFRASTRUCTURE, tests/test_ddp_gloo.py): the same measure() code on host tensors through the SIMT-emulator
    # build of the kernel sources, gloo instead of RCCL -- what exercises the world > 1 branch on a machine without a GPU
    @property
    def on_gpu(self):
        return self.dev.type == "cuda"

    def sync(self):
        if self.on_gpu:
            torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True) if self.on_gpu else _HostEvent()

    def stream(self):
        return torch.cuda.current_stream() if self.on_gpu else None


class _HostEvent:
    """stands in for torch.cuda.Event in --emulator runs (host clock; nothing is asynchronous there)"""

    def __init__(self):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def parse_prof_report(text):
    """`eeg_dcrnn_prof_report` lines "role launches total_ms symbol" -> {role: {"count", "ms", "symbols": {sym: (count, ms)}}}
    (symbols in order of their first launch)"""
    prof = {}
    for line in text.strip().splitlines():
        parts = line.split(None, 3)
        name, cnt, ms = parts[0], int(parts[1]), float(parts[2])
        sym = short_symbol(parts[3]) if len(parts) > 3 else "?"
        ent = prof.setdefault(name, {"count": 0, "ms": 0.0, "symbols": {}})
        ent["count"] += cnt
        ent["ms"] += ms
        c0, m0 = ent["symbols"].get(sym, (0, 0.0))
        ent["symbols"][sym] = (c0 + cnt, m0 + ms)
    return prof


def measure(ctx, workload, steps, warmup, primary):
    """One workload: build, capture, the contract's timed region, [streamed-input pass], live per-kernel event pass.
    Returns the bench line of that workload (rank 0) or None (other ranks).  primary=False: the short secondary passes of
    the other BASELINE configs (no streamed-input pass, no tail timing)."""
    args, world, rank, dev, log = ctx.args, ctx.world, ctx.rank, ctx.dev, ctx.log
    from eeg_gnn_ssl_amd import DCRNNModel_classification, _lib, ops
    from eeg_gnn_ssl_amd.train_step import TrainStep

    task, filt, t_len, batch, classes = WORKLOADS[workload]
    if args.batch and primary:
        batch = args.batch
    dropout = args.dropout if primary else 0.0
    layers = args.layers if primary else LAYERS
    raw_in = workload == "raw"
    torch.manual_seed(123)                                   # identical replicas on every rank
    if task == "ssl":
        from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
        margs = make_args(filt, dropout, layers)
        margs.use_curriculum_learning = bool(args.curriculum) and primary
        model = DCRNNModel_nextTimePred(margs, device=dev).to(dev)
    else:
        model = DCRNNModel_classification(make_args(filt, dropout, layers), classes, device=dev).to(dev)
    model.train()
    raw_kw = {}
    if raw_in:
        # the scaler of the synthetic data set (synthetic_raw_signals): log|FFT| ~ 5.68 +- 0.87
        raw_kw = dict(raw_window=RAW_WINDOW, raw_mean=5.68, raw_std=0.87)
    stepper = TrainStep(model, task=task, lr=3e-4, weight_decay=5e-4, max_grad_norm=5.0, always_reduce=args.force_dist, **raw_kw)
    device_graph = filt == "dual_random_walk" and not args.host_supports
    # the host-side per-clip graph loop (numpy, 256-512 clips) is only needed to CHECK the device graphs: rank 0 of a
    # single-GPU run does it; data-parallel ranks build their supports on the GPU only
    check_graphs = device_graph and world == 1
    if raw_in:
        if not device_graph:
            raise SystemExit("--workload raw builds its graphs on the device (no --host-supports)")
        hx = synthetic_raw_signals(batch, t_len, seed=123 + rank)                          # (B, N, T*200) resampled signals
        hy = (hx[:, :, :10].mean(dim=(1, 2)) > 0).float()
        hlen = torch.full((batch,), t_len, dtype=torch.int64)
        hsup = None
        if check_graphs:                             # host pipeline on the same signals: numpy FFT features -> correlation graphs
            from eeg_gnn_ssl_amd import utils
            amp = np.abs(np.fft.fft(hx.numpy().astype(np.float64).reshape(batch, N_NODES, t_len, RAW_WINDOW), axis=-1))[..., :D_IN]
            amp[amp == 0.0] = 1e-8
            feats = np.log(amp).transpose(0, 2, 1, 3)                                      # (B, T, N, 100)
            s1, s2 = [], []
            for i in range(batch):
                sp = utils.compute_supports(utils.correlation_graph(feats[i], top_k=3), "dual_random_walk")
                s1.append(sp[0]); s2.append(sp[1])
            hsup = [torch.stack(s1), torch.stack(s2)]
            host_feats = torch.from_numpy(feats.astype(np.float32))
    else:
        hx, hy, hlen, hsup = synthetic_batch(task, filt, t_len, batch, classes, seed=123 + rank, host_supports=not device_graph or check_graphs)
    x, y, lengths = hx.to(dev), hy.to(dev), hlen.to(dev)
    supports = [s.to(dev) for s in hsup] if hsup is not None else None
    graph_check = None
    if device_graph and not check_graphs:
        supports = None
    elif device_graph:
        # per-clip correlation graph + supports are rebuilt from the clips on the GPU inside every step
        # (eeg_dcrnn_corr_graph); they must match what the host pipeline prepared for the same clips
        if raw_in:
            fr, _ = ops.fft_features(x, window=RAW_WINDOW)
            fdiff = float((fr.cpu() - host_feats).abs().max().item())
            if fdiff > 1e-4:
                raise SystemExit(f"device log|FFT| features differ from numpy's on the same signals by {fdiff:.2e}")
            chk = ops.correlation_supports(fr, top_k=3)
        else:
            chk = ops.correlation_supports(x, top_k=3)
        odd = torch.nonzero(sum((a - b_).abs().amax(dim=(1, 2)) > 1e-5 for a, b_ in zip(chk, supports)) > 0).view(-1).tolist()
        # a clip may differ from the host pipeline (fp64 Gram) only where the host's own margin between the 3rd and the 4th
        # strongest neighbour of some electrode is a rounding error of an fp32 Gram (|correlation| values ~1, 114 000 terms)
        clips_np = host_feats.numpy() if raw_in else hx.numpy()
        worst = 0.0
        for i in odd:
            flat = np.transpose(clips_np[i], (1, 0, 2)).reshape(N_NODES, -1).astype(np.float64)
            nrm = np.sqrt((flat * flat).sum(axis=1))
            a_ = np.abs(flat @ flat.T / np.outer(nrm, nrm))
            np.fill_diagonal(a_, 0.0)
            srt = -np.sort(-a_, axis=1)
            worst = max(worst, float((srt[:, 2] - srt[:, 3]).min()))
        graph_check = {"clips": batch, "clips_with_a_different_edge_set": len(odd), "largest_top3_margin_among_them": worst,
                       "note": "device supports vs the host pipeline (fp64 Gram): edge sets may differ only where the host's margin "
                               "between the 3rd and 4th strongest neighbour is below 2e-6 (an fp32 Gram cannot resolve it)"}
        if worst > 2e-6:
            raise SystemExit(f"device correlation-graph supports differ from the host pipeline on {len(odd)} clips (top-3 margin up to {worst:.2e})")
        supports = None

    log(f"{workload}: inputs on device, {warmup} warm-up steps")
    lib = _lib.get_lib()
    graphed, whole_graph = False, False
    if not args.no_graph:
        # forward + loss + backward replayed as ONE HIP graph at any number of GPUs (one launch per step and rank
        # instead of ~60: what the multi-GPU scaling hinges on).  --graph-update: the exchange (RCCL all-reduce, when there is
        # a process group) and the fused clip/Adam are captured too -- ONE launch per rank and step.
        try:
            snap = stepper.snapshot() if args.graph_update else None
            stepper.capture(x, y, lengths, supports, include_update=args.graph_update)
            if snap is not None:
                stepper.restore(snap)
            graphed, whole_graph = True, bool(args.graph_update)
        except Exception as e:                                   # noqa: BLE001 -- fall back to eager launches
            log(f"HIP graph capture failed ({type(e).__name__}: {e}); launching eagerly")
            ctx.sync()
    one_step = stepper.replay_step if graphed else (lambda: stepper.step(x, y, lengths, supports))

    clock_buf = torch.zeros(3, dtype=torch.int64, device=dev)

    def timed(step_fn, marks=None):
        """the contract's timed region: W untimed steps, barrier + synchronize, K steps on the wall clock, barrier + synchronize.
        marks: a HIP event is recorded on the launch stream in front of every timed step and behind the last one (K + 1 records of
        ~1 us each) so that the line can show the per-step durations the wall-clock mean is made of."""
        for _ in range(warmup):
            step_fn()
        ctx.sync_all()
        cur = ctx.stream()
        t0 = time.perf_counter()
        for k in range(steps):
            if marks is not None:
                marks[k].record(cur)
            loss = step_fn()
        if marks is not None:
            marks[steps].record(cur)
        ctx.sync_all()
        dt = time.perf_counter() - t0
        # shader clock under sustained fp32-MFMA load right behind the timed steps (200 us on every SIMD, outside the timed region)
        if marks is not None and ctx.on_gpu and hasattr(lib._dll, "eeg_dcrnn_prof_clock_probe"):
            clock_buf.zero_()
            lib.call("eeg_dcrnn_prof_clock_probe", ctypes.c_void_p(clock_buf.data_ptr()), ctypes.c_void_p(cur.cuda_stream))
            torch.cuda.synchronize()
        return dt, loss

    marks = [ctx.event() for _ in range(steps + 1)]
    elapsed, loss = timed(one_step, marks)
    step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(steps)]
    cyc, ticks = (int(v) for v in clock_buf.tolist()[:2])
    sclk_mhz = round(cyc / ticks * 100.0, 1) if ticks > 0 else None
    log(f"{workload}: timed {steps} steps ({'graph replay' if graphed else 'eager'}): {elapsed / steps * 1e3:.3f} ms/step "
        f"(first {step_ms[0]:.3f}, median {sorted(step_ms)[len(step_ms) // 2]:.3f}, last {step_ms[-1]:.3f}; shader clock under MFMA load behind the last step {sclk_mhz} MHz)")

    # second timed pass: every step first receives a FRESH batch from pinned host memory (the trainer's situation: at
    # 85 k clips/s the input stream is ~40 GB/s per GPU).  The step is captured on TWO input sets; the host-to-device copy
    # of batch k+1 runs on a side stream straight into the set the NEXT replay reads while batch k computes: no staging
    # buffer and no device-to-device refresh (round 2 paid 0.39 ms/step for that).  Eager launches: one input set, the copy
    # waits for the step that reads it.
    streamed = None
    if primary and not args.no_stream_inputs:
        pin = [t.pin_memory() for t in (hx, hy)]
        side = ctx.copy_stream
        sets = [(x, y)]
        if graphed:
            x2, y2 = torch.empty_like(x), torch.empty_like(y)
            x2.copy_(x); y2.copy_(y)
            snap = stepper.snapshot() if whole_graph else None
            stepper.capture(x2, y2, lengths, supports, slot=1, include_update=whole_graph)
            if snap is not None:
                stepper.restore(snap)
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
        streamed = {"value": round(batch * world / (el2 / steps), 1), "unit": "clips/s",
                    "ms_per_step": round(el2 / steps * 1e3, 3),
                    "host_bytes_per_step_per_gpu": int(hx.numel() * 4 + hy.numel() * hy.element_size()),
                    "note": ("a fresh batch per step from pinned host memory: the step is captured on two input sets and the H2D "
                             "copy of batch k+1 lands in the idle set on a side stream while batch k computes (no staging buffer, "
                             "no device-side copy)") if graphed else
                            "a fresh batch per step from pinned host memory, copied on a side stream between eager steps"}
        log(f"{workload}: streamed inputs: {streamed['ms_per_step']} ms/step")

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
        for _ in range(steps):
            stepper.step(x, y, lengths, supports)
        ctx.sync()
        lib.query("eeg_dcrnn_prof_enable", 0)
        if has_clk:
            lib.call("eeg_dcrnn_prof_clock_samples", None)
            c = clk.tolist()
            kernel_clock_mhz = {"seq_fwd": round(c[0] / c[1] * 100.0, 1) if c[1] > 0 else None,
                                "seq_bwd": round(c[2] / c[3] * 100.0, 1) if c[3] > 0 else None}
        buf = ctypes.create_string_buffer(1 << 17)
        lib.call("eeg_dcrnn_prof_report", buf, len(buf))
        prof = parse_prof_report(buf.value.decode())
    # exchange + optimiser tail (all-reduce of the flat bucket when a process group exists, norm + fused clip/Adam):
    # HIP events around reduce_and_update() on the launch stream, on a snapshot of the optimiser state that is restored afterwards
    # (with a process group every call all-reduces -- SUMS -- the bucket in place, so the same finite gradient is re-installed
    # before every call)
    tail_ms = None
    if primary:
        snap = stepper.snapshot()
        keep_grad = stepper.fp.flat_grad.clone()
        e0, e1 = ctx.event(), ctx.event()
        tail_ms = 0.0
        for _ in range(steps):
            stepper.fp.flat_grad.copy_(keep_grad)
            e0.record()
            stepper.reduce_and_update()
            e1.record()
            ctx.sync()
            tail_ms += e0.elapsed_time(e1) / steps
        stepper.restore(snap)
        stepper.fp.flat_grad.copy_(keep_grad)
    per_rank = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        allr = [torch.empty_like(per_rank) for _ in range(world)]
        dist.all_gather(allr, per_rank)
        per_rank_ms = [round(float(t.item()) / steps * 1e3, 3) for t in allr]
        elapsed = max(float(t.item()) for t in allr)             # MAX over ranks
        backend = dist.get_backend()
        world_seen = dist.get_world_size()
        # the line's n_gpus / global batch are WORLD_SIZE's: the process group must have exactly that many ranks, all of them here
        assert world_seen == world and len(per_rank_ms) == world, (world_seen, world, per_rank_ms)
    else:
        per_rank_ms = [round(elapsed / steps * 1e3, 3)]
        backend = dist.get_backend() if dist.is_initialized() else None
        world_seen = dist.get_world_size() if dist.is_initialized() else 1
    loss_val = float(loss.item())
    n_grad = stepper.fp.flat_grad.numel()
    reduce_issued = bool(stepper.reduce)
    del stepper, model, x, y, supports
    if ctx.on_gpu:
        torch.cuda.empty_cache()
    if rank != 0:
        return None

    ms_per_step = elapsed / steps * 1e3
    clips_per_s = batch * world / (elapsed / steps)
    spectral = any(k.startswith("spec_mix") for k in prof)      # the encoder ran the spectral form of its hoisted x-part
    fused = spectral and "spec_mix_y" not in prof                 # ... and its recurrent kernels did the 3H-wide node mixes themselves
    work = algorithmic_work(filt, t_len, batch, task, layers, raw=raw_in, spectral=spectral, fused=fused)
    launch_work = per_launch_work(filt, t_len, batch, layers, spectral=spectral, fused=fused)
    merge_paired_roles(work, launch_work, prof)

    def rate(w, ms, hbm):
        """(bound, achieved, peak, unit, frac[, frac_of_achievable]) of `w` algorithmic bytes / FLOPs in `ms`"""
        if hbm:
            gbs = w / (ms * 1e-3) / 1e9
            return dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                        frac_of_achievable=round(gbs / ACHIEVABLE_HBM_GBS, 4))
        tf = w / (ms * 1e-3) / 1e12
        return dict(bound="mfma", achieved=round(tf, 2), peak=PEAK_MFMA_F32_TFLOPS, unit="TFLOP/s", frac=round(tf / PEAK_MFMA_F32_TFLOPS, 4))

    kernels, by_symbol = {}, {}
    for name, pr in prof.items():
        cnt, ms = pr["count"], pr["ms"]
        per_step_ms = ms / steps
        ent = {"launches_per_step": cnt / steps, "ms_per_step": round(per_step_ms, 4),
               "avg_launch_ms": round(per_step_ms / (cnt / steps), 4)}
        syms = pr["symbols"]
        ent["symbol"] = " + ".join(syms)
        hbm = is_hbm_role(name)
        priced = name in work and work[name] > 0 and per_step_ms > 0
        if priced:
            ent.update(rate(work[name], per_step_ms, hbm))
        # by SYMBOL: a role whose launches run different instantiations (layers) is split with the per-launch work table
        lw = launch_work.get(name)
        pos = 0
        for sym, (c, m_) in syms.items():
            k = int(round(c / steps))
            if len(syms) == 1:
                w_sym = work.get(name, 0.0) if priced else 0.0
            elif lw is not None and sum(int(round(cc / steps)) for cc, _ in syms.values()) == len(lw):
                w_sym = sum(lw[pos:pos + k])
            else:
                w_sym = None                        # (decoder roles: no per-launch table) -> priced with the role only
            pos += k
            if len(syms) > 1:
                sub = {"launches_per_step": c / steps, "ms_per_step": round(m_ / steps, 4)}
                if w_sym:
                    sub.update(rate(w_sym, m_ / steps, hbm))
                ent.setdefault("by_symbol", {})[sym] = sub
            if priced and w_sym is not None:
                b = by_symbol.setdefault(sym, {"ms_per_step": 0.0, "work": 0.0, "hbm": hbm, "launches_per_step": 0.0, "roles": []})
                b["ms_per_step"] += m_ / steps
                b["work"] += w_sym
                b["launches_per_step"] += c / steps
                b["roles"].append(name)
        kernels[name] = ent
    roofline = None
    timed_k = {k: v for k, v in kernels.items() if "bound" in v}
    if timed_k:
        # HBM bytes per launch from the committed PMC passes of the same command (tools/pmc_traffic.sh).  The file is stamped
        # with the hash of the kernel sources it was collected on: a stale file yields `traffic: null` + a warning
        traffic, traffic_note, traffic_sym = None, None, {}
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel_sources_sha256") == kernel_sources_sha256():
                traffic = tj["traffic_bytes_per_launch"]
                traffic_sym = tj.get("traffic_bytes_per_launch_by_symbol") or {}
            else:
                traffic_note = (f"profiles/pmc_traffic_{workload}.json was collected on other kernel sources "
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
        flops = sum(v for k, v in work.items() if not is_hbm_role(k) and not k.endswith("_persist"))
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
        # the dominant kernel = the SYMBOL with the most time per step, over all the roles it serves (e.g. `gemm_nnr_kernel<4,2>`
        # = the pre-activation GEMMs of both layers + the input-gradient GEMM), as a rocprofv3 kernel trace would rank it
        sym_table = {}
        for sym, b in by_symbol.items():
            e = {"ms_per_step": round(b["ms_per_step"], 4), "launches_per_step": b["launches_per_step"], "roles": sorted(set(b["roles"])),
                 "avg_launch_ms": round(b["ms_per_step"] / b["launches_per_step"], 4)}
            e.update(rate(b["work"], b["ms_per_step"], b["hbm"]))
            e["algorithmic_work_per_launch"] = round(b["work"] / b["launches_per_step"], 1)
            # PMC traffic of the role(s) behind the symbol, weighted by launches (a class average where the PMC pass only has the class)
            tr = [(kernels[r_].get("traffic_bytes_per_launch_pmc") or kernels[r_].get("traffic_bytes_per_launch_pmc_class_avg"),
                   kernels[r_]["launches_per_step"]) for r_ in e["roles"]]
            if sym in traffic_sym:                      # the PMC pass's own per-symbol average
                e["traffic"] = int(traffic_sym[sym])
            elif tr and all(t_ for t_, _ in tr):
                e["traffic"] = int(sum(t_ * n_ for t_, n_ in tr) / sum(n_ for _, n_ in tr))
            sym_table[sym] = e
        dom = max(sym_table, key=lambda k: sym_table[k]["ms_per_step"])
        d = sym_table[dom]
        dom_role = d["roles"][0]
        roofline = {"kernel": dom, "symbol": dom, "roles": d["roles"], "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                    "unit": d["unit"], "frac": d["frac"], "ms_per_step": d["ms_per_step"], "launches_per_step": d["launches_per_step"],
                    # the MFMA peak is a 2.4 GHz figure; the part holds less under sustained fp32 matrix load.  frac_at_held_clock =
                    # frac x 2400 / (shader clock measured right behind the last timed step): the share of the cycles the chip
                    # actually ran.  `frac` (against the spec-sheet peak) stays the reported figure.
                    "shader_clock_mhz_under_load": sclk_mhz, "kernel_clock_mhz": kernel_clock_mhz or None,
                    "frac_at_held_clock": (round(d["frac"] * PEAK_CLOCK_MHZ / (kernel_clock_mhz.get(dom_role) or sclk_mhz), 4)
                                           if d["bound"] == "mfma" and (kernel_clock_mhz.get(dom_role) or sclk_mhz) else None),
                    "traffic": d.get("traffic"), "traffic_note": traffic_note, "avg_launch_ms": d["avg_launch_ms"],
                    "algorithmic_work_per_launch": d["algorithmic_work_per_launch"],
                    "by_symbol": dict(sorted(sym_table.items(), key=lambda kv: -kv[1]["ms_per_step"])),
                    "top_class": max(by_class, key=lambda k: by_class[k]["ms_per_step"]), "by_class": by_class,
                    "kernels": kernels,
                    "kernel_ms_per_step_total": round(sum(v["ms_per_step"] for v in kernels.values()), 3),
                    "whole_step_flops": round(flops / 1e9, 1),
                    "whole_step_mfma_frac": round(flops / (ms_per_step * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, 4)}
    per_gpu = clips_per_s / world
    mfma_roof = PEAK_MFMA_F32_TFLOPS * 1e12 / (CLIP_GFLOP[workload] * 1e9)          # clips/s/GPU, SURVEY.md §8(d)
    out = {
        "metric": "EEG clips/sec (60s, 19ch, K=2, 2-layer x64) fwd+bwd",
        "value": round(clips_per_s, 1), "unit": "clips/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
        # what the wall-clock mean is made of (HIP events in front of every timed step, rank 0): a fresh process shows whether
        # its first replays are slower than the rest (clock ramp / first-use costs) instead of hiding it in the mean
        "ms_per_step_p50": round(sorted(step_ms)[len(step_ms) // 2], 3),
        "first5_ms": [round(v, 3) for v in step_ms[:5]], "last5_ms": [round(v, 3) for v in step_ms[-5:]],
        "ms_per_step_events_mean": round(sum(step_ms) / len(step_ms), 3),
        "shader_clock_mhz_under_load": sclk_mhz,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": DESCR[workload], "per_gpu_batch": batch, "global_batch": batch * world,
                   "clip_len": t_len, "parallelism": f"dp{world}", "optimizer_step_included": True, "dropout": dropout, "num_rnn_layers": layers,
                   "supports": ("per-clip correlation graph + dual random-walk supports built on the GPU inside the step"
                                if device_graph else "prepared on the host (distance graph is fixed)"
                                if filt == "laplacian" else "prepared on the host"),
                   "device_graph_check": graph_check,
                   "launch": ("ONE hip-graph replay per step (fwd+loss+bwd+" + ("all-reduce+" if reduce_issued else "") + "clip+Adam)" if whole_graph else
                              "hip-graph replay (fwd+loss+bwd) + eager all-reduce/clip+Adam") if graphed else "eager",
                   "timed_region": f"{steps} steps on one batch resident in HBM = {elapsed * 1e3:.1f} ms wall",
                   "library": (("A/B build " + args.lib + " ") if args.lib else "") + ("DEV build with tuning knobs " + ",".join(args.tune) if args.tune else ("" if args.lib else "product")),
                   "final_loss": round(loss_val, 5)},
        "distributed": {"world_size": world_seen, "backend": backend, "per_rank_ms_per_step": per_rank_ms,
                        "all_reduce_issued": reduce_issued,
                        "reduce_and_update_ms_per_step": None if tail_ms is None else round(tail_ms, 4),
                        "exchange": "one all-reduce of the flat fp32 gradient bucket per step "
                                    f"({n_grad * 4} bytes), " + ("inside" if whole_graph else "outside") + " the HIP graph"
                                    + ("" if reduce_issued else " (no process group: not issued in this run)")},
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
                       "hbm_frac": round(per_gpu * CLIP_BYTES[workload] / (PEAK_HBM_GBS * 1e9), 4),
                       "hbm_roof_clips_per_s_per_gpu": round(PEAK_HBM_GBS * 1e9 / CLIP_BYTES[workload], 0)},
        # `value` is the training-loop rate (optimiser step included); the kernel figure takes the optimiser tail
        # (norm + fused clip/Adam, live HIP-event times) out of the step
        "fwd_bwd_only": (None if not prof or world > 1 else {
            "clips_per_s": round(batch / ((ms_per_step - sum(prof.get(k, {"ms": 0.0})["ms"] for k in ("grad_sqnorm", "clip_adam")) / steps) * 1e-3), 1),
            "excluded_ms_per_step": round(sum(prof.get(k, {"ms": 0.0})["ms"] for k in ("grad_sqnorm", "clip_adam")) / steps, 4)}),
        "streamed_inputs": streamed,
        "roofline": roofline,
    }
    return out


def secondary_summary(line):
    """the compact entry of a secondary workload in the primary line's `secondary_workloads`"""
    r = line["roofline"] or {}
    top = list((r.get("by_symbol") or {}).items())[:6]
    return {"workload": line["config"]["workload"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
            "ms_per_step_p50": line["ms_per_step_p50"], "steps": line["steps"], "warmup": line["warmup"],
            "per_gpu_batch": line["config"]["per_gpu_batch"], "launch": line["config"]["launch"], "supports": line["config"]["supports"],
            "device_graph_check": line["config"]["device_graph_check"], "final_loss": line["config"]["final_loss"],
            "dominant_symbol": r.get("symbol"), "dominant_roles": r.get("roles"), "bound": r.get("bound"), "frac": r.get("frac"),
            "achieved": r.get("achieved"), "peak": r.get("peak"), "rate_unit": r.get("unit"), "dominant_ms_per_step": r.get("ms_per_step"),
            "traffic": r.get("traffic"), "traffic_note": r.get("traffic_note"),
            "executed_mfma_frac": line["whole_step"]["executed_mfma_frac"],
            "mfma_roof_frac_survey_flops": line["whole_step"]["mfma_roof_frac_survey_flops"],
            "top_symbols": {k: {"ms_per_step": v["ms_per_step"], "frac": v["frac"], "bound": v["bound"]} for k, v in top}}


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
    ap.add_argument("--curriculum", action="store_true", help="SSL workloads: use_curriculum_learning (model.py:194-200), teacher-"
                    "forcing flags drawn on the device by eeg_dcrnn_teacher_flags inside the captured step")
    ap.add_argument("--secondary", default=None, help="comma-separated workloads measured by short captured passes BEHIND the "
                    "primary workload's timed region and reported under `secondary_workloads` of the same line (default at one "
                    "GPU with the default workload: cfg3,cfg4,cfg5 = the other BASELINE configs + raw = cfg3 from raw signals; 'none' switches them off)")
    ap.add_argument("--secondary-steps", type=int, default=None, help="timed steps of every secondary pass (default: max(10, --steps))")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-side legs: `cpu_baseline` (the oracle on the host "
                    "cores) and `aten_gpu_baseline` (the oracle's op sequence on stock ATen kernels on this GPU)")
    ap.add_argument("--no-prof", action="store_true", help="disable the live per-kernel HIP-event timing")
    ap.add_argument("--host-supports", action="store_true", help="correlation-graph workloads: use supports prepared "
                    "on the host (the reference's DataLoader path) instead of building them on the GPU every step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the "
                    "captured HIP graph of forward+loss+backward (default: replay, at any number of GPUs)")
    ap.add_argument("--graph-update", action="store_true", help="capture the exchange + optimiser tail into the step's HIP graph too "
                    "(RCCL all-reduce when a process group exists, fused clip/Adam with device-resident step count / learning "
                    "rate): ONE launch per rank and step.  Off by default: the driver's line keeps the launch sequence of "
                    "rounds 1-4 (graph + eager tail)")
    ap.add_argument("--no-stream-inputs", action="store_true", help="skip the second timed pass that feeds a fresh pinned "
                    "host batch into the step's input tensors on a side stream every step")
    ap.add_argument("--force-dist", action="store_true", help="single process: create a world-size-1 process group over "
                    "the nccl (= RCCL) backend and issue the gradient all-reduce every step (exercises the RCCL path on one GPU)")
    ap.add_argument("--split-bf16-experiment", action="store_true", help="also run tools/micro/bf16x3_lab (a LAB kernel, not "
                    "the product path: the hoisted NN GEMM as a three-term bf16 split on the bf16 matrix pipe) and report its "
                    "time and error beside the true-fp32 kernel under `experimental_split_bf16`; `value` / `dtype` are untouched")
    ap.add_argument("--split-bf16", action="store_true", help="also measure the primary workload once more with the OPT-IN three-term "
                    "bf16 split of the hoisted NN GEMMs (ops.set_gemm_mode(1): x-part pre-activations and input gradient on "
                    "v_mfma_f32_16x16x32_bf16, fp32 operands / results / accumulation, fp32-level error) and report it under "
                    "`experimental_split_bf16_step` BESIDE the fp32 line: `value` / `dtype` stay the fp32-MFMA measurement (the default "
                    "single-GPU cfg2 line carries it anyway, behind the secondary workloads)")
    ap.add_argument("--no-split-bf16", action="store_true", help="leave the experimental split-bf16 pass out of the default single-GPU line")
    ap.add_argument("--sustained-steps", type=int, default=1500, help="replays of the long pass behind everything else in the default single-GPU "
                    "line (`sustained` in the line: the rate once the part's clocks have settled; never `value`); 0: leave it out")
    ap.add_argument("--tune", action="append", default=[], help="development knob key=value (eeg_dcrnn_set_tuning); loads "
                    "the DEV build libeeg_dcrnn_hip_dev.so instead of the product library")
    ap.add_argument("--lib", default=None, help="development A/B runs only: load this build of the C ABI (e.g. a library built "
                    "from an older commit, kept under build/ab/) instead of the product library; named in config.library")
    ap.add_argument("--emulator", action="store_true", help="TEST INFRASTRUCTURE, never a measurement: run measure() on host "
                    "tensors through the SIMT-emulator build of the kernel sources (tests/emu) with the gloo backend, eager launches, "
                    "no profiler / baselines -- tests/test_ddp_gloo.py drives the world > 1 branch of this file that way")
    args = ap.parse_args()

    ctx = Ctx(args)
    world, rank = ctx.world, ctx.rank
    if args.emulator:
        return main_emulator(ctx)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP) device: eeg_gnn_ssl_amd has no CPU path")
    torch.cuda.set_device(ctx.local_rank)
    ctx.dev = torch.device("cuda", ctx.local_rank)
    # the stream the input batches travel on is created FIRST: HIP maps streams round-robin onto a few hardware queues, and a
    # copy stream created after the capture / RCCL streams can share the compute stream's queue (copy and step then
    # serialise: measured 5.5 instead of 3.0 ms/step under --force-dist)
    ctx.copy_stream = torch.cuda.Stream()
    with torch.cuda.stream(ctx.copy_stream):          # (the queue is bound at the first submission, not at creation)
        torch.zeros(8, device=ctx.dev).add_(1)
    torch.cuda.synchronize()
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)     # "nccl" IS RCCL on ROCm
    if world > 1:
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))            # host threads per rank (8 ranks share the host)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from eeg_gnn_ssl_amd import _lib

    if args.lib:                                                     # development A/B runs only
        _lib._LIB = _lib.EegDcrnnLib(os.path.abspath(args.lib), strict=False)
    if args.tune:                                                    # development A/B runs only
        if not args.lib:
            _lib._LIB = _lib.EegDcrnnLib(_lib.DEV_LIB_PATH)
        for kv in args.tune:
            k, v = kv.split("=")
            _lib._LIB.call("eeg_dcrnn_set_tuning", int(k), int(v))

    out = measure(ctx, args.workload, args.steps, args.warmup, primary=True)
    # the other BASELINE configs, by short captured passes behind the primary workload's timed region (same process, same
    # library, same launch mode); `value` above is untouched by them
    if args.secondary is None:
        sec = ["cfg3", "cfg4", "cfg5", "raw"] if (world == 1 and args.workload == "cfg2" and not args.batch and not args.tune and not args.lib) else []
    else:
        sec = [w for w in args.secondary.split(",") if w and w != "none"]
    sec_lines = {}
    for w in sec:
        if w == args.workload:
            continue
        try:
            line = measure(ctx, w, args.secondary_steps or max(10, args.steps), args.warmup, primary=False)
            if line is not None:
                sec_lines[w] = secondary_summary(line)
        except BaseException as e:                               # noqa: BLE001 -- a secondary pass never takes the headline line down
            if isinstance(e, KeyboardInterrupt):
                raise
            sec_lines[w] = {"error": f"{type(e).__name__}: {e}"}
            ctx.log(f"{w}: secondary pass failed: {type(e).__name__}: {e}")
            torch.cuda.synchronize()
    # opt-in second line (never the headline): the same workload with the hoisted NN GEMMs as a three-term bf16 split
    # (part of the default single-GPU line, behind everything else: it cannot touch `value` or the secondary workloads)
    bf_line, bf_error = None, None
    default_line = world == 1 and args.workload == "cfg2" and not args.batch and not args.tune and not args.lib and args.secondary is None
    if args.split_bf16 or (default_line and not args.no_split_bf16):
        from eeg_gnn_ssl_amd import ops as _ops
        prev_mode = _ops.set_gemm_mode(1)
        try:
            bf_line = measure(ctx, args.workload, args.steps, args.warmup, primary=False)
        except BaseException as e:                               # noqa: BLE001 -- the experimental pass never takes the headline line down
            if isinstance(e, KeyboardInterrupt):
                raise
            bf_error = f"{type(e).__name__}: {e}"
            ctx.log(f"split-bf16 pass failed: {bf_error}")
            torch.cuda.synchronize()
        finally:
            _ops.set_gemm_mode(prev_mode)
    # a long pass of the SAME captured step behind everything else (default single-GPU line only; never `value`): the contract's K steps
    # start W steps after an idle phase (capture, graph instantiation), while the part is still raising its clocks -- the first timed
    # steps are 5-8 % slower than the median (`first5_ms` / `last5_ms`); this is the rate a training run sees after the first 50 ms
    sustained = None
    if default_line and args.sustained_steps > 0:
        keep_prof, args.no_prof = args.no_prof, True
        try:
            sl = measure(ctx, args.workload, args.sustained_steps, args.warmup, primary=False)
            sustained = {"steps": args.sustained_steps, "clips_per_s": sl["value"], "ms_per_step": sl["ms_per_step"], "ms_per_step_p50": sl.get("ms_per_step_p50"),
                         "first5_ms": sl.get("first5_ms"), "last5_ms": sl.get("last5_ms"), "ratio_to_value": round(sl["value"] / out["value"], 4),
                         "note": f"the same captured step replayed {args.sustained_steps} times in one timed region (same protocol: W warm-up "
                                 "steps, barrier + synchronize on both sides); `value` above is the K-step figure of the contract"}
        except BaseException as e:                               # noqa: BLE001 -- never takes the headline line down
            if isinstance(e, KeyboardInterrupt):
                raise
            sustained = {"error": f"{type(e).__name__}: {e}"}
            ctx.log(f"sustained pass failed: {type(e).__name__}: {e}")
            torch.cuda.synchronize()
        finally:
            args.no_prof = keep_prof
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    out["secondary_workloads"] = sec_lines or None
    out["sustained"] = sustained
    if bf_error is not None:
        out["experimental_split_bf16_step"] = {"error": bf_error}
    if bf_line is not None:
        r3 = bf_line["roofline"] or {}
        out["experimental_split_bf16_step"] = {
            "clips_per_s": bf_line["value"], "ms_per_step": bf_line["ms_per_step"], "ratio_to_fp32_value": round(bf_line["value"] / out["value"], 4),
            "final_loss": bf_line["config"]["final_loss"], "final_loss_fp32": out["config"]["final_loss"],
            # the bf16 kernels draw more power: the clock the part holds right behind the split-mode steps is lower than behind the
            # fp32 steps, which slows every OTHER kernel of the step -- why the step gains less than the NN GEMMs do
            "shader_clock_mhz_under_load": bf_line["shader_clock_mhz_under_load"], "shader_clock_mhz_under_load_fp32": out["shader_clock_mhz_under_load"],
            "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in (r3.get("kernels") or {}).items() if v["ms_per_step"] >= 0.02},
            "kernel_ms_per_step_fp32": {k: v["ms_per_step"] for k, v in ((out["roofline"] or {}).get("kernels") or {}).items() if v["ms_per_step"] >= 0.02},
            "scope": "OPT-IN (ops.set_gemm_mode(1) / EEG_DCRNN_SPLIT_BF16=1), never the headline: the two hoisted NN GEMMs of every encoder "
                     "layer (x-part pre-activations, input gradient) as a three-term bf16 split, 6 of 9 partial products on "
                     "v_mfma_f32_16x16x32_bf16 with fp32 accumulation; fp32 operands and results; the whole `-m gpu` suite passes with "
                     "the mode on (1e-4 vs the oracle); `value` / `dtype` above are the fp32-MFMA measurement",
            "kernels": {k: {"ms_per_step": v["ms_per_step"], "launches_per_step": v["launches_per_step"]}
                        for k, v in (r3.get("by_symbol") or {}).items() if "bf3" in k or "nnr" in k or "nn_dma" in k},
            "nn_gemm_ms_per_step_fp32": round(sum(v["ms_per_step"] for k, v in ((out["roofline"] or {}).get("kernels") or {}).items() if k.startswith("gemm_nn")), 4),
            "nn_gemm_ms_per_step_split": round(sum(v["ms_per_step"] for k, v in (r3.get("kernels") or {}).items() if k.startswith("gemm_nn")), 4)}
    if world == 1 and not args.no_cpu_baseline:
        ctx.log("aten gpu baseline (the oracle's op sequence on stock ATen kernels, same GPU, outside the timed region)")
        base_wl = "cfg3" if args.workload == "raw" else args.workload      # (the baselines time the model step on features)
        out["aten_gpu_baseline"] = aten_gpu_baseline(base_wl, ctx.dev, out["value"])
        ctx.log("cpu baseline (oracle on host cores, per-GPU batch)")
        out["cpu_baseline"] = cpu_baseline(base_wl)
        out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if world == 1 and args.split_bf16_experiment:
        out["experimental_split_bf16"] = split_bf16_experiment()
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def main_emulator(ctx):
    """--emulator: the contract's protocol (per-rank shards, barrier + MAX over ranks, ONE line from rank 0) on the CPU.  The figures
    it prints are emulator times: meaningless as measurements, and the line says so."""
    args, world, rank = ctx.args, ctx.world, ctx.rank
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_support
    emu_support.install_emulator()
    ctx.dev = torch.device("cpu")
    ctx.copy_stream = None
    args.no_prof = args.no_stream_inputs = args.no_cpu_baseline = True
    # (no graph capture on the host.  With --graph-update the capture attempt is left in: it raises, and the fall-back of measure()
    #  -- eager launches, exchange and update behind them -- is what runs; tests/test_ddp_gloo.py checks exactly that path)
    args.no_graph = not args.graph_update
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.set_num_threads(1)
    out = measure(ctx, args.workload, args.steps, args.warmup, primary=True)
    if rank == 0:
        out["data"] = "synthetic; EMULATOR RUN (host, SIMT emulator of the kernel sources, gloo): protocol test, not a measurement"
        out["roofline"] = None
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
