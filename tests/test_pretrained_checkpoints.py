"""The reference's shipped SSL checkpoints (pretrained/*.pth.tar) load into the product modules unchanged
and produce the oracle's outputs; the fine-tune transplant (utils.build_finetune_model, reference
utils.py:166-176) feeds the classification model.  The checkpoints are DATA files of the reference; one of them
(distance graph, 12 s, 1.6 MB) is committed as a fixture under tests/golden/ so that the same check also runs on the
GPU box through the HIP library (`-m gpu`); the correlation-graph one is only read where /root/reference exists
(state_dict names/shapes of all four are pinned by tests/golden/pretrained_manifest.json)."""
import os

import numpy as np
import pytest
import torch

import parity_suite as ps
from oracle import dcrnn_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT_DIRS = ("/root/reference/pretrained", os.path.join(HERE, "golden"))
CASES = [("pretrained_distance_graph_12s.pth.tar", "laplacian"), ("pretrained_correlation_graph_12s.pth.tar", "dual_random_walk")]


def _find(name):
    for d in CKPT_DIRS:
        if os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    pytest.skip(f"{name} is not present here")


@pytest.fixture()
def emulator():
    import emu_support
    emu_support.install_emulator()
    yield
    emu_support.uninstall()


@pytest.mark.parametrize("name,filt", CASES)
def test_shipped_checkpoint_runs_and_transplants(name, filt, adj3d, emulator):
    check_checkpoint(name, filt, adj3d, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name,filt", CASES)
def test_shipped_checkpoint_runs_and_transplants_on_the_gpu(name, filt, adj3d):
    from eeg_gnn_ssl_amd import _lib
    _lib._LIB = None
    assert _lib.get_lib().is_device_build
    check_checkpoint(name, filt, adj3d, "cuda")


def check_checkpoint(name, filt, adj3d, dev):
    import cases
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred, utils
    cfg = orc.DCRNNConfig(filter_type=filt, num_rnn_layers=3, num_classes=1)
    ssl = DCRNNModel_nextTimePred(ps.make_args(cfg), device=dev)
    utils.load_model_checkpoint(_find(name), ssl)               # strict load_state_dict
    ssl = ssl.to(dev)
    params = {k: v.detach().cpu().clone() for k, v in ssl.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    b, t_in, t_out = 2, 3, 2
    x = torch.randn(b, t_in, 19, 100, generator=g)
    y = torch.randn(b, t_out, 19, 100, generator=g)
    sup = cases.supports_for(filt, adj3d, b)
    supd = [s_.to(dev) for s_ in sup]
    ssl.eval()
    with torch.no_grad():
        pred = ssl(x.to(dev), y.to(dev), supd)
    ref = orc.next_time_pred_forward(params, cfg, x, y, sup)
    ps.assert_close(pred.cpu().numpy(), ref.numpy(), f"{name}: SSL prediction vs oracle")
    # fine-tuning: transplant the pretrained encoder into a fresh detection model
    torch.manual_seed(1)
    clf = DCRNNModel_classification(ps.make_args(cfg), 1, device=dev).to(dev)
    clf = utils.build_finetune_model(model_new=clf, model_pretrained=ssl, num_rnn_layers=3)
    cp = {k: v.detach().cpu().clone() for k, v in clf.state_dict().items()}
    for k in params:
        if k.startswith("encoder."):
            assert torch.equal(cp[k], params[k]), k
    lengths = torch.tensor([t_in, t_in - 1])
    clf.eval()
    with torch.no_grad():
        logits = clf(x.to(dev), lengths.to(dev), supd)
    ref_logits = orc.classification_forward(cp, cfg, x, lengths, sup)
    ps.assert_close(logits.cpu().numpy(), ref_logits.numpy(), f"{name}: fine-tune logits vs oracle")
