"""Synthetic PEMS08/METR_LA/NYC-shaped pretrain batches (SURVEY.md §8d "Synthetic inputs").

The raw datasets are absent from the reference mount, so every measurement and
parity case uses these tensors.  Layout matches what reference lib/dataloader.py:36-53
hands to the model: ``source`` (B, T, N, base+2) float32 with z-scored flow channels
followed by the standardised day-slot index and weekday index (identical for all nodes,
reference lib/load_dataset.py:4-40).
"""
import math

import torch

SCALER_MEAN = 230.0   # synthetic flow scaler (plays reference Run.py:63 scaler_data)
SCALER_STD = 146.0


def scaler_zeros():
    """Value written into masked cells: scaler.transform(0) (reference Run.py:67)."""
    return (0.0 - SCALER_MEAN) / SCALER_STD


def make_batch(B, T, N, base, interval=5, seed=2024, start_slot=0, device="cpu"):
    """One synthetic batch.  Sample b covers slots start_slot+b .. start_slot+b+T-1."""
    g = torch.Generator().manual_seed(seed)
    S = 24 * 60 // interval
    src = torch.empty(B, T, N, base + 2, dtype=torch.float32)
    src[..., :base] = torch.randn(B, T, N, base, generator=g)
    slot = start_slot + torch.arange(B).view(B, 1) + torch.arange(T).view(1, T)      # absolute 5-min slot
    day = (slot % S + 1).to(torch.float32)
    week = ((slot // S) % 7 + 1).to(torch.float32)
    day = (day - (S + 1) / 2.0) / math.sqrt((S * S - 1) / 12.0)
    week = (week - 4.0) / 2.0
    src[..., base] = day.view(B, T, 1)
    src[..., base + 1] = week.view(B, T, 1)
    return src.to(device)


def make_noise(numel, seed):
    """Mask noise from a host generator (reference draws it with rand_like on the device
    generator, GPTST.py:316,389,400 — not reproducible across devices, so it is injected)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(numel, generator=g, dtype=torch.float32)


def class_order(HS, seed):
    """Shuffled class list (reference uses python random.shuffle, GPTST.py:357-358)."""
    import random
    lst = list(range(HS))
    random.Random(seed).shuffle(lst)
    return lst


def make_series(num_nodes, channels, interval=5, days=14, seed=10):
    """A learnable synthetic raw series (L, N, channels) in flow units: per-node daily sinusoid with random phase plus noise, clipped
    at 0 — the stand-in Run.py uses when the dataset file is absent, and the input of the loss-curve parity run (a pure-noise
    series gives chaotic, non-contracting training dynamics: two fp32 runs decorrelate after ~25 steps)."""
    import numpy as np
    S = 24 * 60 // interval
    L = days * S
    rng = np.random.RandomState(seed)
    t = np.arange(L)[:, None]
    base = 230 + 120 * np.sin(2 * np.pi * t / S + rng.uniform(0, 6.28, (1, num_nodes)))
    return np.maximum(base[..., None] + rng.normal(0, 30, (L, num_nodes, channels)), 0.0)
