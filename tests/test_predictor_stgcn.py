"""Downstream STGCN predictor against golden vectors generated from the reference implementation
(tests/golden/make_golden_stgcn.py: reference model/STGCN/stgcn.py, STGCN/args.py graph helpers, Model.py Fusion)."""
import os
import types

import numpy as np
import pytest
import torch

from gptst_amd import graph
from gptst_amd.enhance import Fusion
from gptst_amd.predictors import STGCN

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stgcn_small.npz"))


def _model(device="cpu"):
    N = G["A"].shape[0]
    ap = types.SimpleNamespace(Ks=3, Kt=3, num_nodes=N, G=graph.stgcn_graph(G["A"]), blocks1=[64, 32, 128], drop_prob=0, outputl_ks=3)
    m = STGCN(ap, device, 64, 1).to(device)
    sd = {k[3:]: torch.tensor(G[k]) for k in G.files if k.startswith("sd.")}
    m.load_state_dict(sd, strict=True)                   # same parameter tree as the reference: its checkpoints load unchanged
    return m


def test_graph_helpers_match_reference():
    L = graph.scaled_laplacian(G["A"])
    assert np.abs(L - G["L"]).max() < 1e-6
    Lk = graph.cheb_polynomials(L, 3)
    assert Lk.shape == G["Lk"].shape and np.abs(Lk - G["Lk"]).max() < 1e-6
    assert np.abs(graph.cheb_polynomials(L, 1)[0] - np.identity(L.shape[0])).max() == 0


def test_adjacency_from_distance_csv(tmp_path):
    p = tmp_path / "d.csv"
    p.write_text("from,to,cost\n0,1,3.5\n2,0,1.0\nbad,row\n3,3,0.0\n")
    A = graph.adjacency_from_distance_csv(str(p), 4)
    ref = np.zeros((4, 4), dtype=np.float32); ref[0, 1] = ref[2, 0] = ref[3, 3] = 1
    assert np.array_equal(A, ref)


def _check(m, dev, tol):
    x = torch.tensor(G["x"], device=dev, requires_grad=True)
    y = m(x)
    assert y.shape == (2, 12, 20, 1)
    assert float((y.detach().cpu() - torch.tensor(G["y"])).abs().max()) < tol * float(np.abs(G["y"]).max())
    (y * torch.tensor(G["w"], device=dev)).sum().backward()
    assert float((x.grad.cpu() - torch.tensor(G["dx"])).abs().max()) < tol * float(np.abs(G["dx"]).max())
    for k, p in m.named_parameters():
        ref = torch.tensor(G["grad." + k])
        assert float((p.grad.cpu() - ref).abs().max()) <= tol * float(ref.abs().max()) + 1e-7, k


def test_stgcn_forward_and_gradients_match_reference_cpu():
    _check(_model("cpu"), "cpu", 2e-5)


def test_fusion_matches_reference():
    fu = Fusion(64)
    fu.load_state_dict({k[6:]: torch.tensor(G[k]) for k in G.files if k.startswith("fu.sd.")})
    y = fu(torch.tensor(G["fu.a"]), torch.tensor(G["fu.b"]))
    assert float((y - torch.tensor(G["fu.y"])).abs().max()) < 1e-5


@pytest.mark.gpu
def test_stgcn_forward_and_gradients_match_reference_gpu():
    _check(_model("cuda:0"), "cuda:0", 1e-4)
