"""GPU box: write a checkpoint with the PRODUCT's training loop — gptst_amd.trainer.Trainer.train() on the HIP path, two epochs of a tiny
synthetic series (crossing change_epoch, ragged last batch) — to gpurun_out/trainer_ckpt.pth (+ the arguments it was trained with):

    python tests/golden/make_trainer_ckpt.py            # on an MI355X (gpurun); the file is merged back into gpurun_out/

tests/golden/check_ckpt_in_reference.py (build container) then loads that very file into the REFERENCE's GPTST_Model.  The checkpoint is
committed as tests/golden/trainer_ckpt.pth (data: the drop-in consumer's input, model/Model.py:95-98)."""
import json
import logging
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gptst_amd import data as D, synth                      # noqa: E402
from gptst_amd.config import make_args                       # noqa: E402
from gptst_amd.model import GPTST_Model, init_seed, xavier_init_   # noqa: E402
from gptst_amd.trainer import Trainer                        # noqa: E402

ARGS = dict(num_nodes=12, embed_dim=4, embed_dim_spa=2, HS=4, HT=4, HT_Tem=3, num_route=2, epochs=2, change_epoch=1, lr_decay_step="1,2",
            batch_size=8, debug=True)


def main():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros(), **ARGS)
    args.log_dir = out
    args.save_pretrain_path = "trainer_ckpt.pth"
    raw = synth.make_series(ARGS["num_nodes"], 3, interval=5, days=8, seed=3)[:-5]
    train, _, _, scaler, _, _ = D.get_dataloader(args, raw=raw, device="cuda:0", generator=torch.Generator().manual_seed(5))
    args.scaler_zeros = float(scaler.transform(0))
    init_seed(3)
    model = xavier_init_(GPTST_Model(args)).to("cuda:0")
    tr = Trainer(model, args, lambda epoch: (x.contiguous() for x in train.iter_x()), float(scaler.mean), float(scaler.std), 8,
                 batches_per_epoch=len(train))
    tr.logger.setLevel(logging.WARNING)
    tr.train()
    path = os.path.join(out, "trainer_ckpt.pth")
    ck = torch.load(path, map_location="cpu")
    json.dump(dict(args=ARGS, scaler_zeros=args.scaler_zeros, steps=int(tr.step.tA), keys=len(ck), bytes=os.path.getsize(path)),
              open(os.path.join(out, "trainer_ckpt.json"), "w"))
    print("wrote", path, os.path.getsize(path), "bytes,", len(ck), "keys,", tr.step.tA, "optimiser steps")


if __name__ == "__main__":
    main()
