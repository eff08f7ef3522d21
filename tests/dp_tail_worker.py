"""Worker of tests/test_gpu_multiproc.py::test_trainer_keeps_the_tail_of_a_data_parallel_epoch — one rank of a 2-rank gloo job on ONE GPU:
one epoch of gptst_amd.trainer.Trainer over a tiny synthetic series whose number of batches is odd and whose last batch is ragged.
Rank 0 prints ONE JSON line: optimiser steps taken, batches per epoch, whether both ranks hold the same weights, the averaged loss."""
import hashlib
import json
import logging
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gptst_amd import data as D, synth          # noqa: E402
from gptst_amd.config import make_args         # noqa: E402
from gptst_amd.dist import DataParallel         # noqa: E402
from gptst_amd.model import GPTST_Model, init_seed, xavier_init_   # noqa: E402
from gptst_amd.trainer import Trainer           # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dp = DataParallel("gloo", native=False)
args = make_args("PEMS08", num_nodes=20, embed_dim=4, HS=5, HT=6, device=str(dev), batch_size=8, debug=True, epochs=4, change_epoch=1,
                 steps_per_replay=2)
args.log_dir = sys.argv[1] + "/rank%d" % dp.rank
raw = synth.make_series(20, 3, interval=5, days=8, seed=3)[:-5]
train, _, _, scaler, _, _ = D.get_dataloader(args, raw=raw, device=dev, generator=torch.Generator().manual_seed(5))
args.scaler_zeros = float(scaler.transform(0))
init_seed(3)
model = xavier_init_(GPTST_Model(args)).to(dev)
dp.broadcast_(model.flat)
batches, nb = D.epoch_batches(train, 8, dp)
tr = Trainer(model, args, batches, float(scaler.mean), float(scaler.std), 8, dp=dp, batches_per_epoch=nb)
tr.logger.setLevel(logging.WARNING)
losses = [tr.train_epoch(e) for e in (1, 2)]            # random-mask phase, then adaptive + KL
extra = {}
if len(sys.argv) > 2 and sys.argv[2] == "train":        # Trainer.train() end to end — its closing evaluation walks the padded tail rounds too
    best = tr.train()
    extra = {"trained": best is not None, "eval_samples": tr.eval_samples}
torch.cuda.synchronize()
h = hashlib.sha256(model.flat.detach().cpu().numpy().tobytes()).hexdigest()
hs = [None, None]
dist.all_gather_object(hs, h)
full = train.n // 8
if dp.rank == 0:
    print(json.dumps({"tA": tr.step.tA, "nb": nb, "full": full, "n": train.n, "same_weights": hs[0] == hs[1], "losses": losses,
                      "tail_rounds": [len(r) for r in train.tail_rounds(full // 2 * 2, 2)], **extra}))
dp.barrier()
dist.destroy_process_group()
