#!/usr/bin/env python
"""``python Run.py -dataset PEMS08 -mode pretrain [-key value ...]`` — same flags as the reference model/Run.py for the
pretrain mode (reference Run.py:35,49,55-58,63-69,72-74,79-85,115-117,132-143,153-156).  Data: ``-data <path.npz>`` with
``['data']`` of shape (L, N, F) as in the reference datasets; without it a synthetic PEMS08-shaped series is used (the
reference's data zips are not redistributable).  Multi-GPU: launch with torch.distributed.run (one process per GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np   # noqa: E402
import torch         # noqa: E402

from gptst_amd import synth                                  # noqa: E402
from gptst_amd.config import parse_args                      # noqa: E402
from gptst_amd.model import GPTST_Model, init_seed, xavier_init_   # noqa: E402
from gptst_amd.trainer import Trainer                        # noqa: E402


def load_windows(path, args, device):
    """reference lib/load_dataset.py:4-40,92-100 + lib/dataloader.py:85-99 + lib/add_window.py:3-27 (train split only)."""
    raw = np.load(path)["data"].astype(np.float64)
    if raw.ndim == 2:
        raw = raw[..., None]
    raw = raw[..., :args.input_base_dim]
    L, N = raw.shape[0], raw.shape[1]
    S = 24 * 60 // args.interval
    day = (np.arange(L) % S + 1).astype(np.float64)
    week = ((np.arange(L) // S + 4) % 7 + 1).astype(np.float64)
    ntr = int(L * (1 - args.val_ratio - args.test_ratio))
    mean, std = raw[:ntr].mean(), raw[:ntr].std()
    x = np.concatenate([(raw - mean) / std,
                        np.broadcast_to(((day - day[:ntr].mean()) / day[:ntr].std())[:, None, None], (L, N, 1)),
                        np.broadcast_to(((week - week[:ntr].mean()) / week[:ntr].std())[:, None, None], (L, N, 1))], -1)
    T = args.lag
    idx = np.arange(0, ntr - 2 * T + 1)
    win = np.stack([x[i:i + T] for i in idx]).astype(np.float32)
    return torch.from_numpy(win).to(device), float(mean), float(std)


def main():
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    args = parse_args(str(dev))
    if args.mode != "pretrain":
        raise SystemExit("gpt-st_amd implements -mode pretrain (the encoder is reusable through GPTST_Model(mode='eval'))")
    extra = [a for a in sys.argv[1:]]
    data_path = extra[extra.index("-data") + 1] if "-data" in extra else None
    dp = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from gptst_amd.dist import DataParallel
        dp = DataParallel("nccl")
    init_seed(args.seed)
    args.log_dir = os.path.join(os.path.dirname(os.path.realpath(__file__)), "SAVE", args.dataset)
    if data_path:
        windows, mean, std = load_windows(data_path, args, dev)
    else:
        nwin = 64 * args.batch_size
        windows = torch.cat([synth.make_batch(args.batch_size, args.lag, args.num_nodes, args.input_base_dim, interval=args.interval,
                                              seed=100 + i, start_slot=args.batch_size * i) for i in range(nwin // args.batch_size)]).to(dev)
        mean, std = synth.SCALER_MEAN, synth.SCALER_STD
    args.scaler_zeros = (0.0 - mean) / std
    model = GPTST_Model(args)
    if args.xavier:
        xavier_init_(model)
    model = model.to(dev)
    if dp is not None:
        dp.broadcast_(model.flat)
        windows = windows[dp.rank::dp.world]
    g = torch.Generator().manual_seed(args.seed)

    def batches(epoch):
        perm = torch.randperm(windows.shape[0], generator=g).to(dev)            # shuffle=True, drop_last=False
        for i in range(0, perm.numel(), args.batch_size):
            yield windows[perm[i:i + args.batch_size]].contiguous()

    Trainer(model, args, batches, mean, std, args.batch_size, dp=dp).train()


if __name__ == "__main__":
    main()
