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
