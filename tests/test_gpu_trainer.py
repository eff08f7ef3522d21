"""GPU: the input pipeline on the device (window gather from a device-resident series vs the reference loader's golden vectors), the
ragged last batch of an epoch (reference DataLoader(drop_last=False), lib/dataloader.py:152) through the steppers that share one
optimiser state, and the training loop (gpt-st_amd/trainer.py = model/BasicTrainer.py:67-197 pretrain branches): MultiStepLR,
best-state tracking, checkpoint written in the reference's format."""
import logging
import os

import numpy as np
import pytest
import torch

from gptst_amd import data as D
from gptst_amd import synth
from gptst_amd.config import make_args
from oracle import gptst_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_pipeline.npz"))


@pytest.mark.parametrize("name,ds,base", [("pems", "PEMS08", 1), ("nyc", "NYC_TAXI", 2), ("metr", "METR_LA", 1)])
def test_device_window_gather_matches_reference_loader(name, ds, base):
    """WindowLoader.windows on a series resident in HBM == the reference's Add_Window_Horizon + normalisation (fixtures written by
    tests/golden/make_golden_data.py from lib/dataloader.py / lib/add_window.py), bit for bit; x-only iteration == x of (x, y)."""
    args = make_args(ds, batch_size=7)
    raw = FX[name + ".raw"]
    tr, va, te, s_d, _, _ = D.get_dataloader(args, raw=raw, device=DEV, generator=torch.Generator().manual_seed(1))
    assert tr.series.is_cuda
    for tag, ld in (("tr", tr), ("va", va), ("te", te)):
        x, y = ld.windows(torch.from_numpy(FX["%s.%s.idx" % (name, tag)]))
        assert x.is_cuda and x.dtype == torch.float32 and x.shape == (4, 12, raw.shape[1], base + 2)
        assert np.array_equal(x.cpu().numpy(), FX["%s.%s.x" % (name, tag)])
        assert np.array_equal(y.cpu().numpy(), FX["%s.%s.y" % (name, tag)])
    xs = torch.cat([x for x in va.iter_x()])
    want, _ = va.windows(torch.arange(va.n))
    assert torch.equal(xs, want)
    sizes = [x.shape[0] for x in tr.iter_x()]
    assert sum(sizes) == tr.n and sizes[-1] == (tr.n % 7 or 7)            # ragged tail kept (drop_last=False)


def _args(**kw):
    return make_args("PEMS08", num_nodes=20, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(), epochs=6,
                     change_epoch=2, lr_decay_step="2,4", **kw)


def test_ragged_last_batch_shares_optimizer_state_with_the_graph_stepper(parity):
    """An epoch of batches [4, 4, 3] x 3 epochs (random phase, then adaptive + KL, one LR decay): the full batches run through the
    hipGraph stepper, the ragged one through an eager stepper that shares m / v / step counts (Trainer._stepper_for) — losses equal
    the oracle's (torch.optim.Adam + MultiStepLR) on the same batches and injected noise."""
    from gptst_amd.model import GPTST_Model
    from gptst_amd.trainer import Trainer
    args = _args()
    args.log_dir = "/tmp/gptst_test_ragged"
    sd = O.init_state_dict(args, 4)
    model = GPTST_Model(args); model.load_state_dict(sd); model = model.to(DEV)
    tr = Trainer(model, args, lambda e: [], synth.SCALER_MEAN, synth.SCALER_STD, batch_size=4)
    tr.logger.setLevel(logging.WARNING)
    ora = O.Stepper(sd, args, synth.SCALER_MEAN, synth.SCALER_STD)
    sched = torch.optim.lr_scheduler.MultiStepLR(ora.opt, milestones=[2, 4], gamma=args.lr_decay_rate)        # Run.py:141
    k = 0
    for epoch in (1, 2, 3):
        for B in (4, 4, 3):
            k += 1
            src = synth.make_batch(B, 12, 20, 1, seed=100 + k, start_slot=7 * k)
            M = B * 12 * 20
            if epoch <= args.change_epoch:
                kw = dict(noise=synth.make_noise(M, k))
            else:
                kw = dict(noise_a=synth.make_noise(M, k), noise_r=synth.make_noise(M, 50 + k), list_c=synth.class_order(5, k))
            ref = ora.step(src, epoch, **kw)
            st = tr._stepper_for(B)
            forced = (1 - ref[3][2].float()).to(DEV) if epoch > args.change_epoch else None       # adaptive phase: teacher-forced mask
            st.step(src.to(DEV), epoch, forced_mask=forced, **{a: (v.to(DEV) if torch.is_tensor(v) else v) for a, v in kw.items()})
            if st is not tr.step:
                tr.step.tA, tr.step.tB = st.tA, st.tB
            got = st.losses()
            e = abs(got[1] - ref[1]) / abs(ref[1])
            parity("flow_loss_rel_step%d" % k, e)
            assert e < (2e-4 if k <= 4 else 5e-3), (k, B, got, ref[:3])        # round-off grows with the step count (Adam, lr 3e-3)
        if epoch in tr.lr_steps:
            tr.step.lr *= args.lr_decay_rate
        sched.step()
        assert abs(tr.step.lr - ora.opt.param_groups[0]["lr"]) < 1e-12
    assert (tr.step.tA, tr.step.tB) == (9, 3)


def test_trainer_train_saves_reference_format_checkpoint(tmp_path):
    """Trainer.train over 6 epochs of a tiny synthetic dataset whose size is not a multiple of the batch size: crosses change_epoch and
    both lr_decay steps; the loss falls, the checkpoint holds the 159 reference keys and loads into a fresh model."""
    from gptst_amd.model import GPTST_Model, init_seed, xavier_init_
    from gptst_amd.trainer import Trainer
    args = _args(batch_size=8, debug=True)
    args.log_dir = str(tmp_path)
    raw = synth.make_series(20, 3, interval=5, days=8, seed=3)[:-5]              # PEMS08 file layout: (L, N, 3), flow = channel 0; > 1 week so every time channel varies
    train, _, _, scaler, _, _ = D.get_dataloader(args, raw=raw, device=DEV, generator=torch.Generator().manual_seed(5))
    assert train.n % 8 != 0
    args.scaler_zeros = float(scaler.transform(0))
    init_seed(3)
    model = xavier_init_(GPTST_Model(args)).to(DEV)
    w0 = model.flat.clone()
    tr = Trainer(model, args, lambda epoch: (x.contiguous() for x in train.iter_x()), float(scaler.mean), float(scaler.std), 8,
                 batches_per_epoch=len(train))
    tr.logger.setLevel(logging.WARNING)
    first = tr.train_epoch(1)
    best = tr.train()
    assert tr.step.tA == 7 * len(train) and tr.step.tB == 4 * len(train)                 # every batch stepped, ragged tail included
    assert abs(tr.step.lr - args.lr_init * args.lr_decay_rate ** 2) < 1e-12
    assert not torch.equal(w0, model.flat)
    path = os.path.join(args.log_dir, args.save_pretrain_path)
    ck = torch.load(path, map_location="cpu")
    assert list(ck.keys()) == list(O.init_state_dict(args, 1).keys()) and len(ck) == 159
    fresh = GPTST_Model(args)
    fresh.load_state_dict(ck)
    assert all(torch.equal(ck[k], best[k].cpu()) for k in ck)
    last = tr.train_epoch(6)
    assert last < first and last == last


def test_epoch_average_counts_every_step_when_a_group_falls_back(monkeypatch):
    """when PretrainStep.step_group() cannot capture K steps in one graph it runs them one by one — the trainer's flush() must still
    account K loss triples (it averaged every K-th step before), so the epoch average that feeds best-model selection equals the one of
    a run with one replay per step."""
    from gptst_amd.model import GPTST_Model
    from gptst_amd.trainer import Trainer
    monkeypatch.setenv("GPTST_DETERMINISTIC", "1")
    avgs = []
    for broken in (False, True):
        args = _args()
        args.log_dir = "/tmp/gptst_test_fallback"
        args.steps_per_replay = 4 if broken else 1
        sd = O.init_state_dict(args, 4)
        model = GPTST_Model(args); model.load_state_dict(sd); model = model.to(DEV)
        batches = [synth.make_batch(4, 12, 20, 1, seed=300 + k, start_slot=5 * k).to(DEV) for k in range(8)]
        tr = Trainer(model, args, lambda e: iter(batches), synth.SCALER_MEAN, synth.SCALER_STD, batch_size=4)
        tr.logger.setLevel(logging.WARNING)
        if broken:
            def boom(*a, **k):
                raise RuntimeError("operation not permitted when stream is capturing")
            monkeypatch.setattr(tr.step, "_capture_group", boom)
        avgs.append(tr.train_epoch(1))
        assert tr.step.tA == 8
        assert bool(getattr(tr.step, "_group_failed", False)) == broken
    assert abs(avgs[0] - avgs[1]) <= 1e-5 * abs(avgs[0]), avgs
