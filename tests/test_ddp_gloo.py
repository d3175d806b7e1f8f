"""CPU, world_size 2 over gloo: the data-parallel step (clips sharded over ranks, ONE flat-bucket
gradient all-reduce) must give every rank the same gradient as a single process on the
concatenated batch, and identical parameters after the Adam step.  Each rank drives the kernel
sources through the emulator build (there is no GPU here); on the MI355X node the very same
TrainStep runs over RCCL (bench.py --gpus N)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(filt, b):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np
    import cases
    import types
    adj = np.load(os.path.join(ROOT, "tests", "golden", "adj_mx_3d.npy"))
    args = types.SimpleNamespace(num_nodes=19, num_rnn_layers=2, rnn_units=16, input_dim=8, output_dim=8,
                                 max_diffusion_step=2, dcgru_activation="tanh", filter_type=filt, dropout=0.0,
                                 cl_decay_steps=3000, use_curriculum_learning=False)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(b, 3, 19, 8, generator=g)
    y = (torch.rand(b, generator=g) > 0.5).float()
    lengths = torch.tensor([3, 2, 3, 1][:b])
    sup = cases.supports_for(filt, adj, b)
    return args, x, y, lengths, sup


def _worker(rank, world, port, filt, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import emu_support
    emu_support.install_emulator()
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    from eeg_gnn_ssl_amd.train_step import TrainStep
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args, x, y, lengths, sup = _make(filt, 4)
    torch.manual_seed(0)
    model = DCRNNModel_classification(args, 1).train()
    step = TrainStep(model, task="detection", lr=1e-2)
    sl = slice(rank * 2, rank * 2 + 2)
    step.forward_backward(x[sl], y[sl], lengths[sl], [s[sl] for s in sup])
    step.reduce_and_update()
    torch.save({"grad": step.fp.flat_grad.clone(), "param": step.fp.flat.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("filt", ["laplacian", "dual_random_walk"])
def test_two_rank_allreduce_matches_single_process(filt, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_support
    emu_support.install_emulator()          # builds the emulator library once, before forking
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, filt, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["grad"], r1["grad"]) and torch.equal(r0["param"], r1["param"])
    # single process on the concatenated batch; mean-reduced BCE over 4 clips == mean of the two
    # half-batch means, which is what the rank-mean all-reduce produces
    from eeg_gnn_ssl_amd import DCRNNModel_classification
    from eeg_gnn_ssl_amd.train_step import TrainStep
    args, x, y, lengths, sup = _make(filt, 4)
    torch.manual_seed(0)
    model = DCRNNModel_classification(args, 1).train()
    step = TrainStep(model, task="detection", lr=1e-2)
    step.forward_backward(x, y, lengths, sup)
    ref_grad_before_clip = step.fp.flat_grad.clone()
    step.reduce_and_update()
    scale = ref_grad_before_clip.abs().max()
    # r0["grad"] holds the clipped gradient; compare directions/magnitudes after the same clipping
    assert (r0["grad"] - step.fp.flat_grad).abs().max() / scale < 1e-5
    assert (r0["param"] - step.fp.flat).abs().max() < 1e-5
    emu_support.uninstall()


# ---- world_size 4, uneven last shard, classification AND the SSL model (shared decoder cell) -----------------
SHARDS4 = [(0, 2), (2, 4), (4, 6), (6, 7)]          # 7 clips over 4 ranks: 2, 2, 2, 1


def _make4(task):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np
    import cases
    import types
    adj = np.load(os.path.join(ROOT, "tests", "golden", "adj_mx_3d.npy"))
    filt = "dual_random_walk"
    args = types.SimpleNamespace(num_nodes=19, num_rnn_layers=3 if task == "ssl" else 2, rnn_units=16, input_dim=8,
                                 output_dim=8, max_diffusion_step=2, dcgru_activation="tanh", filter_type=filt,
                                 dropout=0.0, cl_decay_steps=3000, use_curriculum_learning=False)
    g = torch.Generator().manual_seed(7)
    b = SHARDS4[-1][1]
    x = torch.randn(b, 3, 19, 8, generator=g)
    y = torch.randn(b, 2, 19, 8, generator=g) if task == "ssl" else torch.randint(0, 4, (b,), generator=g)
    lengths = torch.tensor([3, 2, 3, 1, 2, 3, 3])
    return args, x, y, lengths, cases.supports_for(filt, adj, b)


def _build4(task, args):
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import TrainStep
    torch.manual_seed(0)
    model = (DCRNNModel_nextTimePred(args) if task == "ssl" else DCRNNModel_classification(args, 4)).train()
    return TrainStep(model, task=task, lr=1e-2)


def _worker4(rank, world, port, task, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import emu_support
    emu_support.install_emulator()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args, x, y, lengths, sup = _make4(task)
    step = _build4(task, args)
    assert step.world == world
    lo, hi = SHARDS4[rank]
    step.step(x[lo:hi], y[lo:hi], lengths[lo:hi], [s[lo:hi] for s in sup])
    torch.save({"grad": step.fp.flat_grad.clone(), "param": step.fp.flat.clone(), "norm": step.grad_norm.clone(),
                "seen": step.samples_seen}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("task", ["classification", "ssl"])
def test_four_ranks_uneven_shards(task, tmp_path):
    """7 clips over 4 ranks (2,2,2,1): every rank ends with identical gradients and parameters, equal to the mean
    over ranks of the per-shard gradients computed by one process — for the 4-class model and for the 3-layer SSL
    model, whose shared decoder cell receives two contributions inside ONE flat bucket entry."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_support
    emu_support.install_emulator()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker4, args=(4, port, task, str(tmp_path)), nprocs=4, join=True)
    rs = [torch.load(tmp_path / f"r{r}.pt") for r in range(4)]
    for r in rs[1:]:
        assert torch.equal(r["grad"], rs[0]["grad"]) and torch.equal(r["param"], rs[0]["param"])
    args, x, y, lengths, sup = _make4(task)
    step = _build4(task, args)
    acc = torch.zeros_like(step.fp.flat_grad)
    for lo, hi in SHARDS4:
        step.forward_backward(x[lo:hi], y[lo:hi], lengths[lo:hi], [s[lo:hi] for s in sup])
        acc += step.fp.flat_grad
    mean_grad = acc / 4
    norm = mean_grad.norm()
    assert abs(rs[0]["norm"].item() - norm.item()) <= 1e-5 * max(1.0, norm.item())
    clipped = mean_grad * min(1.0, 5.0 / (norm.item() + 1e-6))
    assert (rs[0]["grad"] - clipped).abs().max() / mean_grad.abs().max() < 1e-5
    # global sample counter (drives scheduled sampling of the SSL model): per-rank batch x world, as on one process
    assert rs[0]["seen"] == 2 * 4
    emu_support.uninstall()


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    from eeg_gnn_ssl_amd.train_step import _all_gather_uneven
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = [5, 0, 3][rank]                                     # uneven evaluation shards, one of them empty
    prob = torch.arange(n, dtype=torch.float32) + 100 * rank
    soft = (torch.arange(n * 4, dtype=torch.float32) + 1000 * rank).view(n, 4)      # class probabilities (n, C)
    lab = torch.arange(n, dtype=torch.int64) + 10 * rank
    torch.save({"prob": _all_gather_uneven(prob), "soft": _all_gather_uneven(soft), "lab": _all_gather_uneven(lab)},
               os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_uneven_evaluation_shards_gather(tmp_path):
    """predict / evaluate gather the per-rank probabilities and labels of shards of DIFFERENT sizes (a data set whose
    size is not a multiple of the world size; here 5 / 0 / 3 samples): every rank must end with the concatenation in
    rank order, for 1-D and 2-D tensors and for integer labels."""
    port = 29600 + os.getpid() % 300
    mp.spawn(_gather_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    want_prob = torch.cat([torch.arange(5.0), torch.arange(3.0) + 200])
    want_soft = torch.cat([torch.arange(20.0).view(5, 4), (torch.arange(12.0) + 2000).view(3, 4)])
    want_lab = torch.cat([torch.arange(5), torch.arange(3) + 20])
    for r in range(3):
        got = torch.load(os.path.join(str(tmp_path), f"g{r}.pt"))
        assert torch.equal(got["prob"], want_prob) and torch.equal(got["soft"], want_soft) and torch.equal(got["lab"], want_lab), r


# ---- SSL under curriculum learning with DEVICE-side teacher-forcing flags + the data-parallel SSL evaluation pass ------
def _make_ssl64():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np
    import cases
    import types
    adj = np.load(os.path.join(ROOT, "tests", "golden", "adj_mx_3d.npy"))
    args = types.SimpleNamespace(num_nodes=19, num_rnn_layers=2, rnn_units=64, input_dim=20, output_dim=20,
                                 max_diffusion_step=1, dcgru_activation="tanh", filter_type="dual_random_walk", dropout=0.0,
                                 cl_decay_steps=4, use_curriculum_learning=True)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(6, 3, 19, 20, generator=g)
    y = torch.randn(6, 4, 19, 20, generator=g)
    return args, x, y, cases.supports_for("dual_random_walk", adj, 6)


SHARDS_SSL = [(0, 3), (3, 6)]          # even shards: the global sample counter is per-rank batch x world (train_step.py)


def _worker_ssl64(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import emu_support
    emu_support.install_emulator()
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import TrainStep, evaluate_ssl
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args, x, y, sup = _make_ssl64()
    torch.manual_seed(0)
    model = DCRNNModel_nextTimePred(args).train()
    step = TrainStep(model, task="ssl", lr=1e-2)
    lo, hi = SHARDS_SSL[rank]
    flags = []
    for _ in range(3):
        seed, off = model.decoder.dropout_rng_state()
        flags.append((seed, off, int(step.samples_seen_dev.item())))
        step.step(x[lo:hi], y[lo:hi], None, [s[lo:hi] for s in sup])
    ev = evaluate_ssl(model, [(x[lo:hi], y[lo:hi], [s[lo:hi] for s in sup])], 0.3, 1.7)
    torch.save({"param": step.fp.flat.clone(), "flags": flags, "device_curriculum": step.device_curriculum,
                "seen_dev": int(step.samples_seen_dev.item()), "seen": step.samples_seen, "eval": ev},
               os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_ssl_device_curriculum_and_evaluation(tmp_path):
    """Curriculum learning with the flags drawn on the device (eeg_dcrnn_teacher_flags): each rank flips the coins of ITS shard (the
    generator seed mixes the rank in, like the dropout masks it also serves: ops.make_rng_state) against the SAME threshold -- the
    global sample counter advances by the GLOBAL batch on every rank --, the ranks end with identical parameters, and
    `evaluate_ssl` returns on every rank the batch-size-weighted masked MAE of the union of the shards."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_support
    emu_support.install_emulator()
    port = 30500 + (os.getpid() % 2000)
    mp.spawn(_worker_ssl64, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"s{r}.pt") for r in range(2))
    assert r0["device_curriculum"] is True and r1["device_curriculum"] is True
    # (seed, offset, samples seen) per step: per-rank seeds, equal offsets and counters
    assert [f[1:] for f in r0["flags"]] == [f[1:] for f in r1["flags"]] and [f[2] for f in r0["flags"]] == [0, 6, 12]
    assert r0["flags"][0][0] != r1["flags"][0][0] and len({f[0] for f in r0["flags"]}) == 1
    assert r0["seen_dev"] == r1["seen_dev"] == r0["seen"] == r1["seen"] == 18
    assert torch.equal(r0["param"], r1["param"])
    assert r0["eval"] == r1["eval"]
    # the union's loss from one process with the ranks' final parameters
    from eeg_gnn_ssl_amd import DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import FlatParameters, evaluate_ssl
    args, x, y, sup = _make_ssl64()
    model = DCRNNModel_nextTimePred(args)
    fp = FlatParameters(model)
    with torch.no_grad():
        fp.flat.copy_(r0["param"])
    want = evaluate_ssl(model, [(x[lo:hi], y[lo:hi], [s[lo:hi] for s in sup]) for lo, hi in SHARDS_SSL], 0.3, 1.7)
    assert abs(want - r0["eval"]) < 1e-6 * max(1.0, abs(want))
    emu_support.uninstall()


# ---- a rank WITHOUT evaluation batches still takes part in the collectives of predict / evaluate_ssl -------------------------
def _worker_empty_shard(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    import types
    import emu_support
    emu_support.install_emulator()
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred
    from eeg_gnn_ssl_amd.train_step import evaluate_ssl, predict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = types.SimpleNamespace(num_nodes=19, num_rnn_layers=1, rnn_units=16, input_dim=4, output_dim=4, max_diffusion_step=1,
                                 dcgru_activation="tanh", filter_type="laplacian", dropout=0.0, cl_decay_steps=3000,
                                 use_curriculum_learning=False)
    g = torch.Generator().manual_seed(5)
    x, sup = torch.randn(3, 2, 19, 4, generator=g), [torch.rand(3, 19, 19, generator=g)]
    lens = torch.full((3,), 2, dtype=torch.int64)
    torch.manual_seed(0)
    det, cls, ssl = DCRNNModel_classification(args, 1), DCRNNModel_classification(args, 4), DCRNNModel_nextTimePred(args)
    y_ssl = torch.randn(3, 2, 19, 4, generator=g)
    mine = rank == 0                                        # rank 1 has no batches at all
    p1, l1 = predict(det, [(x, torch.tensor([0.0, 1.0, 1.0]), lens, sup)] if mine else [], "detection")
    p4, l4 = predict(cls, [(x, torch.tensor([0, 3, 2]), lens, sup)] if mine else [], "classification")
    ev = evaluate_ssl(ssl, [(x, y_ssl, sup)] if mine else [])
    torch.save({"p1": p1, "l1": l1, "p4": p4, "l4": l4, "ev": ev}, os.path.join(out_dir, f"e{rank}.pt"))
    dist.destroy_process_group()


def test_a_rank_without_evaluation_batches_joins_the_collectives(tmp_path):
    """Uneven evaluation shards can leave a rank with NO batch (fewer batches than ranks): it must still enter the gathers of
    `predict` and the all-reduce of `evaluate_ssl` -- the other ranks are waiting there -- and every rank returns the union."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_support
    emu_support.install_emulator()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_empty_shard, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"e{r}.pt", weights_only=False) for r in range(2))
    for k in ("p1", "l1", "p4", "l4"):
        assert r0[k].shape == r1[k].shape and (r0[k] == r1[k]).all(), k
    assert r0["p1"].shape == (3,) and r0["p4"].shape == (3, 4) and r0["l4"].tolist() == [0, 3, 2]
    assert r0["ev"] == r1["ev"] and r0["ev"] > 0
    emu_support.uninstall()
    # one process, no process group: an empty iterator is an error with a message
    import pytest
    from eeg_gnn_ssl_amd.train_step import evaluate_ssl, predict
    with pytest.raises(ValueError, match="no batches"):
        predict(torch.nn.Linear(1, 1), [])
    with pytest.raises(ValueError, match="no batches"):
        evaluate_ssl(torch.nn.Linear(1, 1), [])


def _run_bench_emulator(world, extra, port):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per "GPU"), on the host: --emulator = the SIMT
    emulator of the kernel sources + gloo.  Returns (the JSON lines rank 0 printed on stdout, stderr)."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_support
    emu_support.install_emulator()          # builds the emulator library once, before the ranks start
    emu_support.uninstall()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--emulator", "--workload", "cfg1", "--batch", "2", "--secondary", "none"] + list(extra)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return lines, r.stderr


def test_bench_world_size_2_branch_end_to_end():
    """The `world > 1` code of bench.py -- per-rank shards with different seeds, barrier + MAX over ranks of the timed region, the
    all_gather of the per-rank times, ONE JSON line from rank 0 -- had never executed anywhere (no multi-GPU node).
    Here it runs end to end with two gloo ranks on the emulator: the protocol is checked, the figures are not measurements."""
    lines, err = _run_bench_emulator(2, [], 29631)
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["per_gpu_batch"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    dd = d["distributed"]
    assert dd["world_size"] == 2 and dd["backend"] == "gloo" and dd["all_reduce_issued"] is True
    assert len(dd["per_rank_ms_per_step"]) == 2
    # value = clips of ALL ranks over the MAX of the ranks' timed regions
    slowest = max(dd["per_rank_ms_per_step"])
    assert abs(d["ms_per_step"] - slowest) <= 1e-3 * slowest + 1e-3
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) <= 0.06 * d["value"] + 0.1
    assert "EMULATOR RUN" in d["data"] and d["config"]["launch"] == "eager"
    assert "[bench" in err and err.count("timed 2 steps") == 1, "only rank 0 logs"


def test_bench_world_size_2_graph_update_falls_back_to_eager_exchange():
    """--graph-update asks for the whole step (exchange + update included) as one captured graph.  Where capture is not possible (here:
    no HIP device at all) measure() must fall back to eager launches with the all-reduce and the fused update issued behind them,
    on every rank alike -- the run completes, the line says `eager`, the exchange was issued."""
    lines, err = _run_bench_emulator(2, ["--graph-update"], 29633)
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["distributed"]["world_size"] == 2 and d["distributed"]["all_reduce_issued"] is True
    assert d["config"]["launch"] == "eager"
    assert "graph capture failed" in err
    assert d["config"]["final_loss"] == d["config"]["final_loss"] and d["config"]["final_loss"] > 0      # finite
