#!/usr/bin/env python3
"""Generates tests/golden/golden_reflect_v1.npz from the genuine reference (run in the build container only;
/root/reference never travels): the reflection side of the data augmentation --
  data/data_utils.py:37-62 `get_swap_pairs(INCLUDED_CHANNELS)`,
  data/dataloader_detection.py:233-246 `_random_reflect` (both outcomes of its coin, forced),
  data/dataloader_detection.py:248-256 `_random_scale` (use_fft: += log(scale), scale forced),
  data/dataloader_detection.py:309-333 `_get_combined_graph(swap_nodes)` on the shipped distance graph (adj_mx_3d.pkl[-1]),
  data/dataloader_detection.py:335-354 `_compute_supports` of the plain and the reflected adjacency (laplacian, dual_random_walk),
  data/dataloader_detection.py:258-307 `_get_indiv_graphs(eeg_clip, swap_nodes)`: the correlation graph of a reflected sample is
  built from the UN-reflected clip (its swapped name table is never read).
Only inputs that are not closed-form and the OUTPUTS are stored."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
for _m in ("h5py", "pyedflib"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
from constants import INCLUDED_CHANNELS  # noqa: E402
from data.data_utils import get_swap_pairs  # noqa: E402
from data.dataloader_detection import SeizureDataset  # noqa: E402


class _Self:
    """the attributes the methods under test read"""
    adj_mat_dir = os.path.join(REF, "data/electrode_graph/adj_mx_3d.pkl")
    sensor_ids = [c.split(" ")[-1] for c in INCLUDED_CHANNELS]
    top_k = 3
    use_fft = True
    filter_type = "laplacian"


def forced_choice(value):
    """np.random.choice([True, False]) of `_random_reflect` with a forced outcome"""
    return lambda *_a, **_k: value


me = _Self()
pairs = get_swap_pairs(INCLUDED_CHANNELS)
rng = np.random.RandomState(20260)
clip = rng.standard_normal((12, 19, 8)).astype(np.float64) * 1.5 + 3.0        # (T, N, D) log-amplitude-like features
out = {"pairs": np.asarray(pairs, dtype=np.int64), "clip": clip}
keep = np.random.choice
try:
    np.random.choice = forced_choice(True)
    refl, sw = SeizureDataset._random_reflect(me, clip)
    assert sw is not None
    out["clip_reflected"] = refl
    np.random.choice = forced_choice(False)
    same, sw0 = SeizureDataset._random_reflect(me, clip)
    assert sw0 is None and np.array_equal(same, clip)
finally:
    np.random.choice = keep
keep_u = np.random.uniform
try:
    np.random.uniform = lambda lo, hi: 1.1375
    out["clip_reflected_scaled"] = SeizureDataset._random_scale(me, refl.copy())
    out["scale"] = np.array([1.1375])
finally:
    np.random.uniform = keep_u
adj = SeizureDataset._get_combined_graph(me, None)
adj_r = SeizureDataset._get_combined_graph(me, pairs)
out["adj"], out["adj_reflected"] = adj, adj_r
for ft in ("laplacian", "dual_random_walk"):
    me.filter_type = ft
    for tag, a in (("plain", adj), ("reflected", adj_r)):
        for i, s in enumerate(SeizureDataset._compute_supports(me, a)):
            out[f"supports/{ft}/{tag}/{i}"] = s.numpy()
out["indiv_adj_plain"] = SeizureDataset._get_indiv_graphs(me, clip, None)
out["indiv_adj_swapped"] = SeizureDataset._get_indiv_graphs(me, clip, pairs)
np.savez_compressed(os.path.join(HERE, "golden_reflect_v1.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
print("reflected graph is a permutation similarity of the plain one:",
      np.allclose(np.sort(np.linalg.eigvalsh(adj)), np.sort(np.linalg.eigvalsh(adj_r))), "| symmetric:", np.allclose(adj_r, adj_r.T),
      "| indiv graphs equal:", np.array_equal(out["indiv_adj_plain"], out["indiv_adj_swapped"]))
