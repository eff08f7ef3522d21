"""Build container only (needs /root/reference): the checkpoint file the product's trainer wrote on the MI355X (tests/golden/trainer_ckpt.pth,
made by tests/golden/make_trainer_ckpt.py) goes through the drop-in consumer's own code path — reference model/Model.py:95-98:

    pretrain_model = GPTST_Model(args); pretrain_model.load_state_dict(torch.load(path))        # strict=True

and one CPU forward of the REFERENCE on it (eval mode = what Enhance_model calls, GPTST.py:485-487; and pretrain mode with the drawn mask
noise recorded) equals the oracle's forward of the same file.  Writes tests/golden/ckpt_in_reference.json (checked by
tests/test_oracle_golden.py::test_trainer_checkpoint_was_accepted_by_the_reference).

    python tests/golden/check_ckpt_in_reference.py
"""
import hashlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import Recorder, ref, ROOT          # noqa: E402  (loads the reference with 'cuda:0' -> 'cpu' in memory)
from gptst_amd.config import make_args               # noqa: E402
from gptst_amd import synth                          # noqa: E402
from oracle import gptst_oracle as O                 # noqa: E402


def main():
    path = os.path.join(HERE, "trainer_ckpt.pth")
    meta = json.load(open(os.path.join(HERE, "trainer_ckpt.json")))
    res = dict(file="tests/golden/trainer_ckpt.pth", sha256=hashlib.sha256(open(path, "rb").read()).hexdigest(), written_by="gptst_amd.trainer.Trainer.train "
               "(tests/golden/make_trainer_ckpt.py on an MI355X, %d optimiser steps)" % meta["steps"])
    out = {}
    for mode in ("eval", "pretrain"):
        args = make_args("PEMS08", mode=mode, scaler_zeros=meta["scaler_zeros"], **meta["args"])
        args.device = "cpu"
        m = ref.GPTST_Model(args)                                            # the REFERENCE module (model/Pretrain_model/GPTST.py:459-478)
        sd = torch.load(path, map_location="cpu")                            # model/Model.py:96
        missing = m.load_state_dict(sd, strict=True)                         # raises on any missing / unexpected key or shape mismatch
        assert not missing.missing_keys and not missing.unexpected_keys
        assert list(sd.keys()) == list(m.state_dict().keys())                # same order as the reference registers them
        src = synth.make_batch(3, 12, meta["args"]["num_nodes"], 1, seed=77)
        with torch.no_grad():
            if mode == "eval":
                got = m(src, None)                                           # forward -> forward_fune: the embedding, five times (GPTST.py:485-487)
                want = O.forward_eval(sd, args, src)
                assert all(torch.equal(got[0], g) for g in got[1:])
                err = float((got[0] - want).abs().max() / want.abs().max())
            else:
                torch.manual_seed(5)
                with Recorder() as rec:
                    got = m(src, None, None, 1)                              # random-mask phase (epoch 1 <= change_epoch)
                (outs, _) = O.forward_pretrain(sd, args, src, 1, noise=rec.noise[0].reshape(-1))
                assert torch.equal(got[2], outs[2].to(got[2].dtype)), "mask"
                err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip((got[0], got[1], got[3], got[4]), (outs[0], outs[1], outs[3], outs[4])))
        out[mode] = err
        assert err < 2e-6, (mode, err)
    res.update(strict_load="ok", keys=len(sd), forward_max_rel_err_vs_oracle=out)
    json.dump(res, open(os.path.join(HERE, "ckpt_in_reference.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
