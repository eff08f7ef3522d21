"""Input pipeline (gpt-st_amd/data.py) against golden vectors produced by the reference's own loader functions
(tests/golden/make_golden_data.py): time-index channels, split, window gather, z-score statistics, fp32 cast."""
import os

import numpy as np
import pytest
import torch

from gptst_amd import data as D
from gptst_amd.config import make_args

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz"))
CASES = {"pems": ("PEMS08", 1), "nyc": ("NYC_TAXI", 2), "metr": ("METR_LA", 1)}


@pytest.mark.parametrize("name", list(CASES))
def test_pipeline_matches_reference(name):
    ds, base = CASES[name]
    args = make_args(ds, batch_size=7)
    assert args.input_base_dim == base
    raw = FX[name + ".raw"]
    _, week_start, interval, _ = D.DATASETS[ds]
    day, week = D.time_add(raw.shape[0], week_start, interval)
    assert np.array_equal(day, FX[name + ".day"]) and np.array_equal(week, FX[name + ".week"])
    tr, va, te, s_d, s_day, s_week = D.get_dataloader(args, raw=raw)
    np.testing.assert_allclose([s_d.mean, s_d.std, s_day.mean, s_day.std, s_week.mean, s_week.std], FX[name + ".stats"], rtol=1e-13)
    lens = FX[name + ".lens"]
    assert [tr.series.shape[0], va.series.shape[0], te.series.shape[0], tr.n, va.n, te.n] == list(lens)
    assert float(s_d.transform(0)) == float(FX[name + ".zeros"][0])
    for tag, ld in (("tr", tr), ("va", va), ("te", te)):
        x, y = ld.windows(torch.from_numpy(FX["%s.%s.idx" % (name, tag)]))
        assert x.dtype == torch.float32 and x.shape == (4, 12, raw.shape[1], base + 2)
        assert np.array_equal(x.numpy(), FX["%s.%s.x" % (name, tag)])          # bit-exact: same float64 arithmetic, same cast
        assert np.array_equal(y.numpy(), FX["%s.%s.y" % (name, tag)])


def test_loader_iteration_shuffle_and_ragged_tail():
    args = make_args("PEMS08", batch_size=64)
    raw = FX["pems.raw"]
    tr, va, te, *_ = D.get_dataloader(args, raw=raw, generator=torch.Generator().manual_seed(3))
    sizes = [x.shape[0] for x, _ in tr]
    assert sum(sizes) == tr.n and all(s == 64 for s in sizes[:-1]) and sizes[-1] == tr.n % 64     # drop_last=False
    assert len(tr) == len(sizes)
    first_a = next(iter(tr))[0]
    first_b = next(iter(tr))[0]
    assert not torch.equal(first_a, first_b)                                   # reshuffled every epoch
    xs = torch.cat([x for x, _ in va])                                         # validation: in order, not shuffled
    want, _ = va.windows(torch.arange(va.n))
    assert torch.equal(xs, want)
