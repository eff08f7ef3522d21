"""Input pipeline of ``-mode pretrain``: ``<dataset>.npz`` -> day / week index channels -> split ->
sliding windows -> z-score, as the reference does it (lib/load_dataset.py:4-105, lib/dataloader.py:8-158,
lib/add_window.py:3-27, lib/normalization.py:11-27), but MI355X-resident: the normalised SERIES (L, N, base+2) lives on
the GPU once (fp32) and a batch of windows is gathered from it on demand — the reference materialises every window on the
device (12x the bytes).  Values are identical: statistics and the normalisation are computed in float64 on the host
exactly as the reference does, then cast to fp32 (the reference's ``TensorFloat(X)``).
"""
import os

import numpy as np
import torch

# dataset -> (file, week_start, interval [min], first-channel-only)      reference lib/load_dataset.py:44-90
DATASETS = {
    "PEMS08": ("PEMS08/PEMS08.npz", 5, 5, True),
    "METR_LA": ("METR_LA/metr_la.npz", 4, 5, False),
    "NYC_BIKE": ("NYC_BIKE/NYC_BIKE.npz", 5, 30, False),
    "NYC_TAXI": ("NYC_TAXI/NYC_TAXI.npz", 5, 30, False),
}


def time_add(length, week_start, interval=5, week_max=7, day_start=0, hour_of_day=24):
    """Day-slot index (1..slots, restarting every day) and weekday index (week_start.., wrapping after week_max) per time step —
    lib/load_dataset.py:4-40, vectorised (the reference loops over the time steps)."""
    slots = hour_of_day * 60 // interval
    idx = np.arange(length)
    day = (idx % slots) + 1 + day_start
    week = (week_start - 1 + idx // slots) % week_max + 1
    return day.astype(np.int64), week.astype(np.int64)


def load_st_dataset(dataset, args, root="../data", raw=None):
    """(L, N, base+2) float64: the data channels followed by the day and week indices — lib/load_dataset.py:43-105.
    ``raw`` overrides the file (tests, synthetic series)."""
    if dataset not in DATASETS:
        raise ValueError(dataset)
    fname, week_start, interval, first_only = DATASETS[dataset]
    if raw is None:
        raw = np.load(os.path.join(root, fname))["data"]
    data = raw[:, :, 0] if (first_only and raw.ndim == 3) else raw
    args.interval, args.week_day = interval, 7
    day, week = time_add(data.shape[0], week_start, interval)
    if data.ndim == 2:
        data = data[..., None]
    N = data.shape[1]
    day = np.broadcast_to(day[:, None, None], (data.shape[0], N, 1))
    week = np.broadcast_to(week[:, None, None], (data.shape[0], N, 1))
    return np.concatenate([data, day, week], axis=-1)


def split_data_by_ratio(data, val_ratio, test_ratio):                     # lib/dataloader.py:84-89
    n = data.shape[0]
    test = data[-int(n * test_ratio):]
    val = data[-int(n * (test_ratio + val_ratio)):-int(n * test_ratio)]
    train = data[:-int(n * (test_ratio + val_ratio))]
    return train, val, test


def split_data_by_days(data, val_days, test_days, interval=60):           # lib/dataloader.py:71-82
    T = int((24 * 60) / interval)
    return data[:-T * (test_days + val_days)], data[-T * (test_days + val_days):-T * test_days], data[-T * test_days:]


def window_count(length, window, horizon):                                # lib/add_window.py:10-11
    return max(length - horizon - window + 1, 0)


class StandardScaler:                                                     # lib/normalization.py:11-27
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def transform(self, data):
        return (data - self.mean) / self.std

    def inverse_transform(self, data):
        return data * self.std + self.mean


class WindowLoader:
    """Batches of (x, y) windows gathered from a device-resident series; the reference's DataLoader(shuffle, drop_last=False):
    x = series[i : i+lag], y = series[i+lag : i+lag+horizon] (single=False) or the single step i+lag+horizon-1 (single=True)."""

    def __init__(self, series, lag, horizon, batch_size, shuffle, single=False, generator=None):
        self.series, self.lag, self.horizon, self.bs, self.shuffle, self.single = series, lag, horizon, batch_size, shuffle, single
        self.n = window_count(series.shape[0], lag, horizon)
        self.gen = generator
        dev = series.device
        self._tx = torch.arange(lag, device=dev)
        self._ty = torch.arange(horizon - 1, horizon, device=dev) + lag if single else torch.arange(horizon, device=dev) + lag

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def windows(self, idx):
        idx = idx.to(self.series.device)
        return self.series[idx[:, None] + self._tx[None, :]], self.series[idx[:, None] + self._ty[None, :]]

    def _order(self):
        return torch.randperm(self.n, generator=self.gen) if self.shuffle else torch.arange(self.n)

    def __iter__(self):
        order = self._order()
        for i in range(0, self.n, self.bs):
            yield self.windows(order[i:i + self.bs])

    def iter_x(self, rank=0, world=1, limit=None):
        """Input windows only (pretraining never reads the y windows, BasicTrainer.py:74-76): one gather per batch instead of two.
        rank / world: data parallelism — the permutation of the epoch is drawn in full (every rank draws the same one), but only batches
        rank, rank + world, ... (of the first `limit`) are gathered; the others cost nothing."""
        order = self._last_order = self._order()
        for k, i in enumerate(range(0, self.n, self.bs)):
            if (limit is not None and k >= limit) or k % world != rank:
                continue
            idx = order[i:i + self.bs].to(self.series.device)
            yield self.series[idx[:, None] + self._tx[None, :]]

    def tail_rounds(self, first, world):
        """Batch indices of the padded rounds behind the first `first` batches under `world`-way data parallelism: the remaining FULL batches
        as one round (fewer than `world` of them), then the ragged last batch as a round of its own (other batch size)."""
        full, nb = self.n // self.bs, (self.n + self.bs - 1) // self.bs
        if first < full:
            yield list(range(first, full))
        if nb > full:
            yield [full]

    def iter_tail(self, first, world):
        """-> per padded round the list of its batches' input windows (iter_x's permutation: call it BEHIND iter_x of the same epoch, which
        drew the permutation — the order is cached per epoch)."""
        order = self._last_order
        for ks in self.tail_rounds(first, world):
            out = []
            for k in ks:
                idx = order[k * self.bs:(k + 1) * self.bs].to(self.series.device)
                out.append(self.series[idx[:, None] + self._tx[None, :]])
            yield out


def get_dataloader(args, root="../data", device="cpu", raw=None, single=False, generator=None):
    """-> train, val, test loaders, scaler_data, scaler_day, scaler_week (lib/dataloader.py:100-158, normalizer 'std',
    column_wise False: one mean/std per channel group computed on the TRAIN split)."""
    if getattr(args, "normalizer", "std") != "std" or getattr(args, "column_wise", False):
        raise NotImplementedError("only the reference's pretrain setting normalizer=std, column_wise=False is implemented")
    data = load_st_dataset(args.dataset, args, root, raw=raw)
    if args.test_ratio > 1:
        parts = split_data_by_days(data, args.val_ratio, args.test_ratio)
    else:
        parts = split_data_by_ratio(data, args.val_ratio, args.test_ratio)
    b = args.input_base_dim
    tr = parts[0]
    sc = [StandardScaler(tr[..., 0:b].mean(), tr[..., 0:b].std()), StandardScaler(tr[..., b:b + 1].mean(), tr[..., b:b + 1].std()),
          StandardScaler(tr[..., b + 1:b + 2].mean(), tr[..., b + 1:b + 2].std())]
    loaders = []
    for k, part in enumerate(parts):
        norm = np.concatenate([sc[0].transform(part[..., 0:b]), sc[1].transform(part[..., b:b + 1]), sc[2].transform(part[..., b + 1:b + 2])], -1)
        series = torch.from_numpy(norm.astype(np.float32)).to(device)
        ld = WindowLoader(series, args.lag, args.horizon, args.batch_size, shuffle=(k == 0), single=single, generator=generator)
        loaders.append(ld if ld.n > 0 else None)
    return loaders[0], loaders[1], loaders[2], sc[0], sc[1], sc[2]


def epoch_batches(train, batch_size, dp=None):
    """-> (batches, batches_per_epoch): `batches(epoch)` yields THIS rank's batches of one epoch for trainer.Trainer — shuffle=True,
    drop_last=False (lib/dataloader.py:152).  Under data parallelism (dp: dist.DataParallel) every rank draws the same permutation and takes
    every world-th batch of the whole groups of `world` full batches; the tail of the epoch is KEPT as padded rounds (r04; it was dropped):
    the full batches beyond the last whole group, then the ragged last batch, each as a round in which a rank without a batch of its own
    steps on a copy of a real one with rank weight 0 — yielded as (batch, weight) on EVERY rank of such a round (trainer.py / step.py::_allreduce)."""
    full = train.n // batch_size
    if dp is None:
        def batches(epoch):
            for x in train.iter_x():
                yield x.contiguous()                                 # incl. the ragged last batch
        return batches, len(train)
    usable = (full // dp.world) * dp.world

    def batches(epoch):
        for x in train.iter_x(rank=dp.rank, world=dp.world, limit=usable):     # only this rank's batches are gathered
            yield x.contiguous()
        for xs in train.iter_tail(usable, dp.world):
            yield (xs[min(dp.rank, len(xs) - 1)].contiguous(), 1.0 if dp.rank < len(xs) else 0.0)
    return batches, usable // dp.world + len(list(train.tail_rounds(usable, dp.world)))

