"""The reference's shipped SSL checkpoints (pretrained/*.pth.tar) load into the product modules unchanged
and produce the oracle's outputs; the fine-tune transplant (utils.build_finetune_model, reference
utils.py:166-176) feeds the classification model.  The checkpoints are data files of the reference and
only exist in the build container: the test is skipped wherever /root/reference is absent (it never
runs on the GPU box; state_dict names/shapes are pinned there by tests/golden/pretrained_manifest.json)."""
import os

import numpy as np
import pytest
import torch

import parity_suite as ps
from oracle import dcrnn_oracle as orc

CKPT_DIR = "/root/reference/pretrained"
pytestmark = pytest.mark.skipif(not os.path.isdir(CKPT_DIR), reason="reference checkpoints not present")


@pytest.fixture(scope="module", autouse=True)
def emulator():
    import emu_support
    emu_support.install_emulator()
    yield
    emu_support.uninstall()


@pytest.mark.parametrize("name,filt", [("pretrained_distance_graph_12s.pth.tar", "laplacian"),
                                       ("pretrained_correlation_graph_12s.pth.tar", "dual_random_walk")])
def test_shipped_checkpoint_runs_and_transplants(name, filt, adj3d):
    import cases
    from eeg_gnn_ssl_amd import DCRNNModel_classification, DCRNNModel_nextTimePred, utils
    cfg = orc.DCRNNConfig(filter_type=filt, num_rnn_layers=3, num_classes=1)
    ssl = DCRNNModel_nextTimePred(ps.make_args(cfg), device="cpu")
    utils.load_model_checkpoint(os.path.join(CKPT_DIR, name), ssl)               # strict load_state_dict
    params = {k: v.detach().clone() for k, v in ssl.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    b, t_in, t_out = 2, 3, 2
    x = torch.randn(b, t_in, 19, 100, generator=g)
    y = torch.randn(b, t_out, 19, 100, generator=g)
    sup = cases.supports_for(filt, adj3d, b)
    ssl.eval()
    with torch.no_grad():
        pred = ssl(x, y, sup)
    ref = orc.next_time_pred_forward(params, cfg, x, y, sup)
    ps.assert_close(pred.numpy(), ref.numpy(), f"{name}: SSL prediction vs oracle")
    # fine-tuning: transplant the pretrained encoder into a fresh detection model
    torch.manual_seed(1)
    clf = DCRNNModel_classification(ps.make_args(cfg), 1, device="cpu")
    clf = utils.build_finetune_model(model_new=clf, model_pretrained=ssl, num_rnn_layers=3)
    cp = {k: v.detach().clone() for k, v in clf.state_dict().items()}
    for k in params:
        if k.startswith("encoder."):
            assert torch.equal(cp[k], params[k]), k
    lengths = torch.tensor([t_in, t_in - 1])
    clf.eval()
    with torch.no_grad():
        logits = clf(x, lengths, sup)
    ref_logits = orc.classification_forward(cp, cfg, x, lengths, sup)
    ps.assert_close(logits.numpy(), ref_logits.numpy(), f"{name}: fine-tune logits vs oracle")
