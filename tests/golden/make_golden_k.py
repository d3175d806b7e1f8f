#!/usr/bin/env python3
"""Golden vectors for the diffusion orders and filter types the main golden file does not cover:
max_diffusion_step K = 1 and 3 (hop expansion incl. the carried-x0 quirk of cell.py:83-93 over three
recurrences and two supports) and filter_type "random_walk".  The GENUINE reference DCGRUCell (imported
from /root/reference, build container only) is run forward + backward on closed-form inputs; only outputs
are stored -> golden_k_v1.npz.   Run once, here:  python tests/golden/make_golden_k.py"""
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
    return [torch.stack(s1)] if filt == "random_walk" else [torch.stack(s1), torch.stack(s2)]


def cell_case(tag, filt, din, h, b, act, k):
    cell = DCGRUCell(input_dim=din, num_units=h, max_diffusion_step=k, num_nodes=N, filter_type=filt, nonlinearity=act)
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


# must mirror tests/cases.py CELL_K_CASES
cell_case("lap_k1", "laplacian", 8, 16, 3, "tanh", 1)
cell_case("lap_k3", "laplacian", 8, 16, 3, "tanh", 3)
cell_case("dual_k1", "dual_random_walk", 8, 16, 3, "tanh", 1)
cell_case("dual_k3", "dual_random_walk", 8, 16, 3, "relu", 3)
cell_case("rw_k2", "random_walk", 12, 32, 2, "tanh", 2)
np.savez_compressed(os.path.join(HERE, "golden_k_v1.npz"), **G)
print(len(G), "arrays")
