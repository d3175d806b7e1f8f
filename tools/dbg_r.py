import sys, os, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
import numpy as np, cases
from eeg_gnn_ssl_amd import DCGRUCell, _lib, ops
lib = _lib.get_lib()
adj = np.load(ROOT + "/tests/golden/adj_mx_3d.npy")
dev = "cuda"
def run(h, din, T, B, generic):
    lib.call("eeg_dcrnn_set_tuning", 3, 1 if generic else 0)
    torch.manual_seed(0)
    cell = DCGRUCell(din, h, 2, 19).to(dev)
    sup = [s.to(dev) for s in cases.supports_for("laplacian", adj, B)]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, B, 19, din, generator=g).to(dev).requires_grad_(True)
    h0 = (0.5 * torch.randn(B, 19 * h, generator=g)).to(dev).requires_grad_(True)
    w = torch.randn(T, B, 19 * h, generator=g).to(dev)
    P, pb = ops.hop_polys(sup, 2, B)
    hseq, hsel = cell.run_sequence(x, h0, P, pb)
    (hseq * w).sum().backward()
    return hseq.detach(), x.grad.clone(), h0.grad.clone(), [p.grad.clone() for p in cell.parameters()]
for (h, din, T, B) in [(16, 8, 1, 2), (16, 8, 3, 2), (32, 8, 2, 2), (64, 8, 2, 2)]:
    a = run(h, din, T, B, True); b = run(h, din, T, B, False)
    print(f"H={h} T={T}: fwd {float((a[0]-b[0]).abs().max()):.2e} dx {float((a[1]-b[1]).abs().max()):.2e} dh0 {float((a[2]-b[2]).abs().max()):.2e} dW", [f"{float((p-q).abs().max()):.1e}" for p, q in zip(a[3], b[3])])
    d = (a[2] - b[2]).abs().view(B, 19, h)
    bad = (d > 1e-4).nonzero()
    if len(bad):
        print("   dh0 bad entries:", len(bad), "nodes", sorted(set(bad[:, 1].tolist())), "cols", sorted(set(bad[:, 2].tolist()))[:20])
