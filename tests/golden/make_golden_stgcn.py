"""Golden vectors for the downstream STGCN predictor, generated from the REFERENCE implementation:
    python tests/golden/make_golden_stgcn.py          (build container only: needs /root/reference)
Imports reference model/STGCN/stgcn.py (STGCN), model/STGCN/args.py (scaled_laplacian, cheb_poly_approx — loaded without its
config-file side effects) and model/Model.py's Fusion on the CPU; saves a seeded state_dict, an input, the output and the gradients of a
scalar loss, plus the graph helpers' outputs on a small adjacency.  Only data is written (tests/golden/stgcn_small.npz)."""
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "model"))
sys.path.insert(0, REF)

from STGCN.stgcn import STGCN                    # noqa: E402  (reference, read-only import)


def ref_graph_fns():
    """scaled_laplacian / cheb_poly_approx of the reference args.py, without importing its dataset helpers (lib.predifineGraph)."""
    src = open(os.path.join(REF, "model/STGCN/args.py")).read()
    src = re.sub(r"^from lib\.predifineGraph import.*$", "", src, flags=re.M)
    src = re.sub(r"^import pandas as pd$", "", src, flags=re.M)
    m = types.ModuleType("stgcn_args_ref")
    if not hasattr(np, "mat"):                     # the reference targets numpy < 2 (np.mat was removed): same function under its new name
        np.mat = np.asmatrix
    exec(compile(src, "stgcn_args_ref", "exec"), m.__dict__)
    return m.scaled_laplacian, m.cheb_poly_approx


def fusion_ref():
    src = open(os.path.join(REF, "model/Model.py")).read()
    src = src.replace("from Pretrain_model.GPTST import GPTST_Model", "")
    m = types.ModuleType("model_ref")
    exec(compile(src, "model_ref", "exec"), m.__dict__)
    return m.Fusion


def main():
    torch.manual_seed(1234)
    np.random.seed(1234)
    N, B, T, dim_in, dim_out = 20, 2, 12, 64, 1
    rng = np.random.RandomState(7)
    A = (rng.rand(N, N) < 0.2).astype(np.float32)
    np.fill_diagonal(A, 0)
    A[np.arange(N), (np.arange(N) + 1) % N] = 1                      # connected ring underneath
    sl, cheb = ref_graph_fns()
    L = np.asarray(sl(A.copy()))
    Lk = np.asarray(cheb(L, 3, N))
    ap = types.SimpleNamespace(Ks=3, Kt=3, num_nodes=N, G=torch.FloatTensor(Lk), blocks1=[64, 32, 128], drop_prob=0, outputl_ks=3)
    model = STGCN(ap, "cpu", dim_in, dim_out)
    x = torch.randn(B, T, N, dim_in, requires_grad=True)
    w = torch.randn(B, T, N, dim_out)
    y = model(x)
    (y * w).sum().backward()
    out = {"A": A, "L": L.astype(np.float64), "Lk": Lk.astype(np.float64), "x": x.detach().numpy(), "w": w.numpy(), "y": y.detach().numpy(),
           "dx": x.grad.numpy()}
    for k, v in model.state_dict().items():
        out["sd." + k] = v.numpy()
    for k, p in model.named_parameters():
        out["grad." + k] = p.grad.numpy()
    Fusion = fusion_ref()
    fu = Fusion(dim_in)
    a, b = torch.randn(B, T, N, dim_in), torch.randn(B, T, N, dim_in)
    out["fu.a"], out["fu.b"], out["fu.y"] = a.numpy(), b.numpy(), fu(a, b).detach().numpy()
    for k, v in fu.state_dict().items():
        out["fu.sd." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "stgcn_small.npz"), **out)
    print("wrote stgcn_small.npz:", len(out), "arrays, y", y.shape, float(y.abs().mean()))


if __name__ == "__main__":
    main()
