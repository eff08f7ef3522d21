#!/usr/bin/env python
"""Golden vectors for the evaluation metrics from the REFERENCE's lib/metrics.All_Metrics (torch branch), called as
model/BasicTrainer.py:241-248 does (per horizon, then over all horizons).  Writes tests/golden/metrics.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from lib.metrics import All_Metrics    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for name, (B, T, N, D, mae_t, mape_t) in {"pems": (37, 12, 9, 1, None, 0.0), "nyc": (20, 12, 6, 2, None, 0.001), "thr": (16, 12, 5, 1, 40.0, 40.0)}.items():
    g = torch.Generator().manual_seed(len(name) + B)
    true = torch.rand(B, T, N, D, generator=g) * 300
    pred = true + torch.randn(B, T, N, D, generator=g) * 20
    mask = (torch.rand(B, T, N, D, generator=g) < 0.3).float()             # pretrain: only masked cells carry signal (:229-232)
    y_true, y_pred = true * mask, pred * mask
    y_true[:, :, 0] = 7.0                                                     # a constant node: true_std == 0 is skipped by CORR
    rows = []
    for t in range(T):
        mae, rmse, mape, _, corr = All_Metrics(y_pred[:, t, ...], y_true[:, t, ...], mae_t, mape_t)
        rows.append([float(mae), float(rmse), float(mape), float(corr)])
    mae, rmse, mape, _, corr = All_Metrics(y_pred, y_true, mae_t, mape_t)
    rows.append([float(mae), float(rmse), float(mape), float(corr)])
    out[name + ".pred"], out[name + ".true"] = y_pred.numpy(), y_true.numpy()
    out[name + ".thr"] = np.array([np.nan if mae_t is None else mae_t, mape_t])
    out[name + ".rows"] = np.array(rows)
np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
print("wrote metrics.npz", sum(v.nbytes for v in out.values()) // 1024, "KB")
