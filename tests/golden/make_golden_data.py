#!/usr/bin/env python
"""Golden vectors for the input pipeline (gpt-st_amd/data.py) from the REFERENCE's own functions, run here on CPU:
lib/load_dataset.time_add, lib/dataloader.{split_data_by_ratio, normalize_dataset}, lib/add_window.Add_Window_Horizon applied
in the order of lib/dataloader.get_dataloader to a small synthetic PEMS08-/NYC-shaped series.  Writes tests/golden/data_pipeline.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from lib.add_window import Add_Window_Horizon                      # noqa: E402
from lib.dataloader import normalize_dataset, split_data_by_ratio   # noqa: E402
from lib.load_dataset import time_add                               # noqa: E402

import configparser                                                # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for name, (ds, L, N, F, first_only, week_start, interval, base) in {"pems": ("PEMS08", 1500, 4, 3, True, 5, 5, 1),
                                                                    "nyc": ("NYC_TAXI", 500, 3, 2, False, 5, 30, 2),
                                                                    "metr": ("METR_LA", 1300, 5, 0, False, 4, 5, 1)}.items():
    cp = configparser.ConfigParser()
    cp.read("/root/reference/conf/GPTST_pretrain/%s.conf" % ds)
    val_ratio, test_ratio = float(cp["data"]["val_ratio"]), float(cp["data"]["test_ratio"])
    rng = np.random.RandomState(len(name))
    raw = rng.gamma(2.0, 60.0, size=(L, N, F) if F else (L, N))
    data = raw[:, :, 0] if first_only else raw
    day, week, _ = time_add(data if data.ndim == 2 else data[..., 0], week_start, interval=interval, weekday_only=False, holiday_list=[])
    if data.ndim == 2:
        data = np.expand_dims(data, -1)
    data = np.concatenate([data, np.expand_dims(day, -1).astype(int), np.expand_dims(week, -1).astype(int)], -1)
    tr, va, te = split_data_by_ratio(data, val_ratio, test_ratio)
    xs = [Add_Window_Horizon(p, 12, 12, False) for p in (tr, va, te)]
    _, s_d, s_day, s_week, _ = normalize_dataset(tr, "std", base, False)
    out[name + ".raw"] = raw
    out[name + ".day"], out[name + ".week"] = day[:, 0].astype(np.int64), week[:, 0].astype(np.int64)
    out[name + ".stats"] = np.array([s_d.mean, s_d.std, s_day.mean, s_day.std, s_week.mean, s_week.std])
    out[name + ".lens"] = np.array([len(tr), len(va), len(te)] + [x[0].shape[0] for x in xs])
    for tag, (x, y) in zip(("tr", "va", "te"), xs):
        xn = np.concatenate([s_d.transform(x[..., :base]), s_day.transform(x[..., base:base + 1]), s_week.transform(x[..., base + 1:base + 2])], -1)
        yn = np.concatenate([s_d.transform(y[..., :base]), s_day.transform(y[..., base:base + 1]), s_week.transform(y[..., base + 1:base + 2])], -1)
        pick = np.array([0, 1, x.shape[0] // 2, x.shape[0] - 1])
        out["%s.%s.idx" % (name, tag)] = pick
        out["%s.%s.x" % (name, tag)] = xn[pick].astype(np.float32)          # TensorFloat(X)
        out["%s.%s.y" % (name, tag)] = yn[pick].astype(np.float32)
    out[name + ".zeros"] = np.array([s_d.transform(0)])
np.savez_compressed(os.path.join(HERE, "data_pipeline.npz"), **out)
print("wrote data_pipeline.npz", sum(v.nbytes for v in out.values()) // 1024, "KB")
