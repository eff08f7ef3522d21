"""Helpers to read the compact golden fixtures written by tests/golden/make_golden.py."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRIDE = 13


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def has(fx, key):
    return key in fx.files or (key + "::sub") in fx.files


def t(fx, key):
    return torch.from_numpy(np.asarray(fx[key]))


def check(fx, key, value, rtol=1e-5, atol=1e-6, what=""):
    """Compare tensor ``value`` with fixture entry ``key`` (full or compact form)."""
    value = value.detach().cpu()
    if key in fx.files:
        ref = t(fx, key)
        if ref.numel() == 0:
            assert value is None or value.numel() == 0 or float(value.abs().max()) == 0.0, key
            return
        torch.testing.assert_close(value.reshape(ref.shape).to(ref.dtype), ref, rtol=rtol, atol=atol, msg=lambda m: "%s %s: %s" % (what, key, m))
        return
    sub, stats = t(fx, key + "::sub"), fx[key + "::stats"]
    f = value.reshape(-1)
    assert f.numel() == int(stats[2]), (key, f.numel(), stats[2])
    torch.testing.assert_close(f[::STRIDE].to(sub.dtype), sub, rtol=rtol, atol=atol, msg=lambda m: "%s %s(sub): %s" % (what, key, m))
    asum = float(f.double().abs().sum())
    assert abs(asum - stats[1]) <= rtol * 10 * max(stats[1], 1.0) + atol * f.numel(), (key, asum, stats[1])
    assert abs(float(f.double().sum()) - stats[0]) <= rtol * 10 * max(stats[1], 1.0) + atol * f.numel(), (key, "sum")


def cfg_args(fx, tag, make_args, **extra):
    cfg = json.loads(str(fx[tag + ".cfg"]))
    return make_args("PEMS08", **cfg, **extra)
