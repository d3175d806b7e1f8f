#!/usr/bin/env python3
"""Golden GRADIENTS of the stand-alone diffusion convolution: the GENUINE reference DiffusionGraphConv
(imported from /root/reference, build container only) is run forward + backward on the closed-form inputs of
tests/cases.py DCONV_CASES; only outputs are stored -> golden_dconv_grad_v1.npz
(dconv/<tag>/{out,dx,ds,d_weight,d_biases}; the upstream gradient is the closed-form fill cf(..., 0.291, 0.4)).
Run once, here:  python tests/golden/make_golden_dconv_grad.py"""
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

from model.cell import DiffusionGraphConv  # noqa: E402
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


def supports(filt, b, batched):
    if filt == "laplacian":
        s = torch.FloatTensor(LAP)
        return [s.unsqueeze(0).repeat(b, 1, 1)] if batched else [s]
    s1, s2 = [], []
    for i in range(b):
        a = keep_topk(cf_adjacency(N, phase=0.3 + 1.7 * i), top_k=3, directed=True)
        s1.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a).T.toarray()))
        s2.append(torch.FloatTensor(ref_utils.calculate_random_walk_matrix(a.T).T.toarray()))
    return [torch.stack(s1), torch.stack(s2)]


def dconv_case(tag, filt, din, h, o, b, batched=True):
    ns = 2 if filt == "dual_random_walk" else 1
    mod = DiffusionGraphConv(num_supports=ns, input_dim=din, hid_dim=h, num_nodes=N, max_diffusion_step=2,
                             output_dim=o, filter_type=filt)
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict({k: T(v) for k, v in cf_params(shapes, base_phase=0.5).items()})
    x = T(cf((b, N * din), scale=1.0, freq=0.371, phase=0.1)).requires_grad_(True)
    s = T(cf((b, N * h), scale=0.8, freq=0.533, phase=0.7)).requires_grad_(True)
    out = mod(supports(filt, b, batched), x, s, o)
    (out * T(cf((b, N * o), scale=1.0, freq=0.291, phase=0.4))).sum().backward()
    G[f"dconv/{tag}/out"] = out.detach().numpy()
    G[f"dconv/{tag}/dx"], G[f"dconv/{tag}/ds"] = x.grad.numpy(), s.grad.numpy()
    G[f"dconv/{tag}/d_weight"], G[f"dconv/{tag}/d_biases"] = mod.weight.grad.numpy(), mod.biases.grad.numpy()


# must mirror tests/cases.py DCONV_CASES
dconv_case("lap_small", "laplacian", 8, 16, 32, 3)
dconv_case("lap_small_unbatched", "laplacian", 8, 16, 32, 3, batched=False)
dconv_case("dual_small", "dual_random_walk", 8, 16, 32, 3)
dconv_case("lap_default", "laplacian", 100, 64, 128, 2)
dconv_case("dual_default", "dual_random_walk", 100, 64, 128, 2)
np.savez_compressed(os.path.join(HERE, "golden_dconv_grad_v1.npz"), **G)
print(len(G), "arrays")
