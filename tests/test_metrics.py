"""Evaluation metrics: the oracle restatement against golden rows from the reference's lib/metrics.All_Metrics (CPU), and the HIP
accumulation kernel + host finalisation against the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))
CASES = ["pems", "nyc", "thr"]


def _thr(name):
    a, b = FX[name + ".thr"]
    return (None if np.isnan(a) else float(a)), float(b)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_rows(name):
    mae_t, mape_t = _thr(name)
    rows = MO.test_report(torch.from_numpy(FX[name + ".pred"]), torch.from_numpy(FX[name + ".true"]), mae_t, mape_t)
    np.testing.assert_allclose(rows.numpy(), FX[name + ".rows"], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_metrics_match_oracle(name):
    from gptst_amd import ops
    dev = "cuda:0"
    mae_t, mape_t = _thr(name)
    y_pred, y_true = torch.from_numpy(FX[name + ".pred"]), torch.from_numpy(FX[name + ".true"])
    B, T, N, D = y_true.shape
    # feed the kernel normalised tensors + the visibility mask, as Trainer.test has them; here sigma = 1, mu = 0 and the
    # fixtures are already (value * masked), so vis = None reproduces them exactly
    sums_t, sums_tn = ops.metrics_new(T, N, dev)
    half = B // 2
    for lo, hi in ((0, half), (half, B)):                                  # two batches accumulate
        out = y_pred[lo:hi].reshape(-1, D).contiguous().to(dev)
        src = y_true[lo:hi].contiguous().to(dev)
        ops.metrics_accum(out, src, D, None, 1.0, 0.0, mae_t, mape_t, hi - lo, T, N, D, sums_t, sums_tn)
    got = ops.metrics_report(sums_t, sums_tn)
    want = MO.test_report(y_pred, y_true, mae_t, mape_t)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4)      # fp32 (oracle: torch.mean in fp32) vs double accumulation


@pytest.mark.gpu
def test_hip_metrics_masking_and_scaler():
    """p = (out*m)*sigma+mu, y = (label*m)*sigma+mu with m = 1 - vis, label = first D channels of the (B,T,N,D+2) input."""
    from gptst_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(4)
    B, T, N, D = 6, 12, 7, 2
    src = torch.randn(B, T, N, D + 2, generator=g)
    out = src[..., :D] + 0.1 * torch.randn(B, T, N, D, generator=g)
    vis = (torch.rand(B, T, N, D, generator=g) > 0.25).float()
    sigma, mu = 146.0, 230.0
    sums_t, sums_tn = ops.metrics_new(T, N, dev)
    ops.metrics_accum(out.reshape(-1, D).to(dev), src.to(dev), D + 2, vis.reshape(-1).to(dev), sigma, mu, None, 0.001, B, T, N, D, sums_t, sums_tn)
    got = ops.metrics_report(sums_t, sums_tn)
    m = 1 - vis
    want = MO.test_report((out * m) * sigma + mu, (src[..., :D] * m) * sigma + mu, None, 0.001)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4)      # fp32 (oracle: torch.mean in fp32) vs double accumulation
