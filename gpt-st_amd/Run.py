#!/usr/bin/env python
"""``python Run.py -dataset PEMS08 -mode pretrain [-key value ...]`` (and ``-mode eval -model STGCN``: the downstream predictor on the
enhanced embedding) — same flags as the reference model/Run.py for the
pretrain mode (reference Run.py:35,49,55-58,63-69,72-74,79-85,115-117,132-143,153-156).  Data: the reference's layout
``<root>/<DATASET>/<file>.npz`` with ``['data']`` of shape (L, N, F) under ``-data_root`` (default ../data, as in the reference),
loaded by gptst_amd.data (time indices, split, windows, z-score: parity-tested against the reference's loader functions);
when the file is absent a synthetic series of the dataset's shape is used (the data zips are not redistributable).
Multi-GPU: launch with torch.distributed.run (one process per GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np   # noqa: E402
import torch         # noqa: E402

from gptst_amd import data as gdata, synth                   # noqa: E402
from gptst_amd.config import apply_predictor_overrides, parse_args, predictor_args   # noqa: E402
from gptst_amd.model import GPTST_Model, init_seed, xavier_init_   # noqa: E402
from gptst_amd.trainer import Trainer                        # noqa: E402


def load_series(args, dev):
    """-> (data_root, loaders + scalers of gdata.get_dataloader): the dataset under -data_root, or — when its file is absent — a synthetic
    series of the dataset's shape (METR_LA's file holds (L, N) without a channel axis, lib/load_dataset.py:57-60)."""
    extra = [a for a in sys.argv[1:]]
    data_root = extra[extra.index("-data_root") + 1] if "-data_root" in extra else "../data"
    fname = gdata.DATASETS[args.dataset][0]
    raw = None
    if not os.path.exists(os.path.join(data_root, fname)):                       # synthetic stand-in of the dataset's shape
        F = 3 if args.dataset == "PEMS08" else args.input_base_dim
        raw = synth.make_series(args.num_nodes, F, interval=gdata.DATASETS[args.dataset][2], seed=args.seed)
        if args.dataset == "METR_LA":
            raw = raw[..., 0]
        print("gpt-st_amd: %s not found -> synthetic %s-shaped series %s" % (os.path.join(data_root, fname), args.dataset, raw.shape))
    g = torch.Generator().manual_seed(args.seed)
    return data_root, gdata.get_dataloader(args, root=data_root, device=dev, raw=raw, generator=g)


def main():
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    args = parse_args(str(dev))
    if args.mode == "eval":
        return main_eval(args, dev)
    if args.mode != "pretrain":
        raise SystemExit("gpt-st_amd implements -mode pretrain and -mode eval -model STGCN")
    dp = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from gptst_amd.dist import DataParallel
        # per-step collectives on the C-ABI communicator (captured inside the step's hipGraph); GPTST_NATIVE_COMM=0: torch.distributed
        dp = DataParallel("nccl", native=os.environ.get("GPTST_NATIVE_COMM", "1") == "1" and os.environ.get("GPTST_DIST_BACKEND", "nccl") == "nccl")
    init_seed(args.seed)
    args.log_dir = os.path.join(os.path.dirname(os.path.realpath(__file__)), "SAVE", args.dataset)
    _, (train, val, test, scaler, _, _) = load_series(args, dev)
    mean, std = float(scaler.mean), float(scaler.std)
    args.scaler_zeros = float(scaler.transform(0))                               # Run.py:67
    model = GPTST_Model(args)
    if args.xavier:
        xavier_init_(model)
    model = model.to(dev)
    if dp is not None:
        dp.broadcast_(model.flat)

    batches, nb = gdata.epoch_batches(train, args.batch_size, dp)      # under data parallelism the epoch's tail is kept as padded rounds
    Trainer(model, args, batches, mean, std, args.batch_size, dp=dp, batches_per_epoch=nb).train()
    if dp is not None:
        import torch.distributed as dist
        dp.barrier()
        dist.destroy_process_group()


def main_eval(args, dev):
    """``-mode eval -model STGCN``: train the STGCN predictor on the enhanced embedding of the frozen pretrained encoder (reference Run.py
    mode 'eval', model/Model.py:20-107, model/STGCN/args.py:52-88 with the values of conf/STGCN/<dataset>.conf)."""
    from types import SimpleNamespace
    from gptst_amd import graph
    from gptst_amd.enhance import EnhanceFrontEnd
    from gptst_amd.eval_trainer import EvalTrainer
    from gptst_amd.predictors import STGCN
    if str(args.model) != "STGCN":
        raise SystemExit("gpt-st_amd -mode eval implements -model STGCN (SURVEY.md §8f rank 4: one baseline predictor)")
    pargs = predictor_args(args.dataset, str(args.model), [a for a in sys.argv[1:] if a.startswith("--") or not a.startswith("-")])
    apply_predictor_overrides(args, pargs)       # reference Run.py:36-43: the predictor's schedule replaces the pretrain conf's (epochs 100, ...)
    init_seed(args.seed)
    args.log_dir = os.path.join(os.path.dirname(os.path.realpath(__file__)), "SAVE", args.dataset)
    os.makedirs(args.log_dir, exist_ok=True)
    data_root, (train, val, test, scaler, _, _) = load_series(args, dev)
    args.scaler_zeros = float(scaler.transform(0))
    csv_path = os.path.join(data_root, args.dataset, args.dataset + ".csv")
    A = (graph.adjacency_from_distance_csv(csv_path, args.num_nodes) if os.path.exists(csv_path)
         else graph.synthetic_adjacency(args.num_nodes, seed=args.seed))
    ap = SimpleNamespace(Ks=pargs.Ks, Kt=pargs.Kt, num_nodes=args.num_nodes, G=graph.stgcn_graph(A), blocks1=list(pargs.blocks1),
                         drop_prob=pargs.drop_prob, outputl_ks=pargs.outputl_ks)
    model = EnhanceFrontEnd(args, predictor=STGCN(ap, dev, args.hidden_dim, args.output_dim)).to(dev)
    ckpt = args.log_dir + str(args.load_pretrain_path)                           # model/Model.py:92 (plain concatenation: '/GPTST_ada.pth')
    if os.path.exists(ckpt):
        model.load_pretrained_model(ckpt)                                        # model/Model.py:91-94
    elif "-demo_random_encoder" not in sys.argv and os.environ.get("GPTST_DEMO_RANDOM_ENCODER", "0") != "1":
        # the reference's load_pretrained_model raises on a missing file: "enhanced" results on a random frozen encoder are meaningless
        raise FileNotFoundError("no pretrained encoder at %s (run -mode pretrain first; -demo_random_encoder runs the plumbing on a "
                                "randomly initialised frozen encoder)" % ckpt)
    else:
        print("gpt-st_amd: no pretrained encoder at %s -> Xavier-initialised encoder (demo run)" % ckpt)
        for p_ in model.pretrain_model.parameters():                             # frozen (requires_grad False): xavier_init_ would skip them
            torch.nn.init.xavier_uniform_(p_) if p_.dim() > 1 else torch.nn.init.uniform_(p_)
    EvalTrainer(model, args, train, val, test, float(scaler.mean), float(scaler.std)).train()


if __name__ == "__main__":
    main()
