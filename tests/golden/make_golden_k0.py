#!/usr/bin/env python3
"""Golden vectors for max_diffusion_step = 0 (no graph mixing at all: the reference's cell.py:80-81 `pass` branch,
num_matrices = 1): the GENUINE reference DCGRUCell (imported from /root/reference, build container only) run forward +
backward on closed-form inputs; only outputs are stored -> golden_k0_v1.npz.  Run once: python tests/golden/make_golden_k0.py"""
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from closed_form import cf, cf_adjacency, cf_params  # noqa: E402

for _m in ("h5py", "pyedflib"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
import torch  # noqa: E402

from model.cell import DCGRUCell  # noqa: E402
import utils as ref_utils  # noqa: E402
from data.data_utils import keep_topk  # noqa: E402

torch.set_num_threads(4)
N = 19
G = {}
with open(os.path.join(REF, "data/electrode_graph/adj_mx_3d.pkl"), "rb") as f:
    ADJ = pickle.load(f)[-1].astype(np.float32)
LAP = ref_utils.calculate_scaled_laplacian(ADJ, lambda_max=None).toarray()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def supports(filt, b):
    if filt == "laplacian":
        return [torch.FloatTensor(LAP).unsqueeze(0).repeat(b, 1, 1)]
    s1, s2 = [], []
    for i in range(b):
        a = keep_topk(cf_adjacency(N, phase=0.3 + 1.7 * i), top_k=3, directed=True)
        s1.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a).T.toarray()))
        s2.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a.T).T.toarray()))
    return [torch.stack(s1), torch.stack(s2)]


def cell_case(tag, filt, din, h, b, act):
    cell = DCGRUCell(input_dim=din, num_units=h, max_diffusion_step=0, num_nodes=N, filter_type=filt, nonlinearity=act)
    shapes = {n: tuple(v.shape) for n, v in cell.state_dict().items()}
    cell.load_state_dict({n: T(v) for n, v in cf_params(shapes, base_phase=1.1).items()})
    x = T(cf((b, N * din), scale=1.0, freq=0.371, phase=0.1)).requires_grad_(True)
    s = T(cf((b, N * h), scale=0.8, freq=0.533, phase=0.7)).requires_grad_(True)
    out, _ = cell(supports(filt, b), x, s)
    (out * T(cf((b, N * h), scale=1.0, freq=0.291, phase=0.4))).sum().backward()
    G[f"cell/{tag}/out"] = out.detach().numpy()
    G[f"cell/{tag}/dx"], G[f"cell/{tag}/dh"] = x.grad.numpy(), s.grad.numpy()
    for n, p in cell.named_parameters():
        G[f"cell/{tag}/d_{n}"] = p.grad.numpy()


# must mirror the K = 0 entries of tests/cases.py CELL_K_CASES
cell_case("lap_k0", "laplacian", 8, 16, 3, "tanh")
cell_case("dual_k0_h64", "dual_random_walk", 100, 64, 2, "tanh")
np.savez_compressed(os.path.join(HERE, "golden_k0_v1.npz"), **G)
print(len(G), "arrays")
