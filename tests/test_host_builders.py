"""CPU: the PRODUCT's host-side graph builders (eeg_gnn_ssl_amd/utils.py — the reference's
utils.calculate_scaled_laplacian / calculate_random_walk_matrix, data_utils.keep_topk and the per-clip
correlation graph of dataloader_detection.py:258-307) against the golden vectors of the genuine reference.
(The oracle's own copies are pinned in test_oracle_vs_golden.py; in the full-size GPU tests product and oracle
receive the same supports, so an error in these builders would cancel there — this file is where it shows.)"""
import numpy as np

from closed_form import cf
from eeg_gnn_ssl_amd import utils


def close(a, b, atol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= atol, np.abs(a - b).max()


def test_scaled_laplacian_matches_reference(golden, adj3d):
    close(utils.calculate_scaled_laplacian(adj3d, lambda_max=None), golden["supports/scaled_laplacian_adj3d"], 2e-7)
    close(utils.calculate_scaled_laplacian(adj3d), golden["supports/scaled_laplacian_adj3d_lmax2"], 2e-7)
    s = utils.compute_supports(adj3d, "laplacian")
    assert len(s) == 1 and s[0].dtype.is_floating_point and tuple(s[0].shape) == (19, 19)
    close(s[0].numpy(), golden["supports/scaled_laplacian_adj3d"].astype(np.float32), 1e-6)


def test_correlation_graph_pipeline_matches_reference(golden):
    clip = cf((12, 19, 100), scale=1.0, freq=0.7391, phase=0.2) + cf((12, 19, 100), scale=0.5, freq=0.0137, phase=1.0)
    adj = utils.correlation_graph(clip, top_k=3)
    close(adj, golden["corr/adj"], 1e-6)
    assert ((adj != 0) == (golden["corr/adj"] != 0)).all()          # the same edges survive keep_topk
    s = utils.compute_supports(adj, "dual_random_walk")
    close(s[0].numpy(), golden["corr/s1"].astype(np.float32), 1e-6)
    close(s[1].numpy(), golden["corr/s2"].astype(np.float32), 1e-6)
    rw = utils.compute_supports(adj, "random_walk")
    assert len(rw) == 1
    close(rw[0].numpy(), golden["corr/s1"].astype(np.float32), 1e-6)


def test_keep_topk_semantics():
    """data_utils.py:174-200: top-k neighbours per row by weight (self loops kept), directed / undirected"""
    rng = np.random.RandomState(4)
    a = rng.rand(19, 19).astype(np.float32)
    np.fill_diagonal(a, 1.0)
    d = utils.keep_topk(a, top_k=3, directed=True)
    for i in range(19):
        off = np.delete(np.arange(19), i)
        top = off[np.argsort(-a[i, off])[:3]]
        assert set(np.nonzero(d[i])[0]) == set(top) | {i}
        assert np.array_equal(d[i, top], a[i, top])
    u = utils.keep_topk(a, top_k=3, directed=False)
    assert ((u != 0) == ((d != 0) | (d != 0).T)).all()
