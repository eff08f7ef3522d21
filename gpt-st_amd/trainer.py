"""Pretraining loop with the reference's behaviour (model/BasicTrainer.py:125-197 train, :67-123 train_epoch, pretrain
branches only): per-epoch MultiStepLR, best-on-train-flow-loss tracking with the up_epoch resets, the loss>1e6 abort, the
log line formats, and the final ``torch.save(best_state_dict, SAVE/<dataset>/<save_pretrain_path>)``.
The step itself is the fused HIP step (step.py); the host only reads the loss statistics when it has to log."""
import copy
import logging
import os
import time

import torch

from .step import PretrainStep


def get_logger(log_dir, name="GPTST", debug=True):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        fmt = logging.Formatter("%(asctime)s: %(message)s", "%Y-%m-%d %H:%M")
        ch = logging.StreamHandler()
        ch.setLevel(logging.DEBUG if debug else logging.INFO)
        ch.setFormatter(fmt)
        logger.addHandler(ch)
        if not debug:                                   # reference lib/logger.py:18-32: file only when debug=False
            fh = logging.FileHandler(os.path.join(log_dir, "run.log"), mode="w")
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    return logger


class Trainer:
    def __init__(self, model, args, batches, scaler_mean, scaler_std, batch_size, dp=None, use_graph=True, batches_per_epoch=None):
        """batches: a callable epoch -> iterable of (B,T,N,base+2) device tensors (ragged last batch allowed), consumed lazily — the
        windowed dataset is 12x the series and is never materialised; batches_per_epoch: the 'i/n' of the log line when known."""
        self.model, self.args, self.batches = model, args, batches
        self.dp, self.nb_epoch = dp, batches_per_epoch
        self.step = PretrainStep(model, args, scaler_mean, scaler_std, batch_size, use_graph=use_graph, dp=dp, seed=args.seed)
        self.ragged = {}
        self.scaler = (scaler_mean, scaler_std)
        self.use_graph = use_graph
        os.makedirs(args.log_dir, exist_ok=True)
        self.logger = get_logger(args.log_dir, name=str(args.model), debug=args.debug)
        if dp is not None and dp.rank != 0:              # one log stream per job (losses are global: every rank would print the same lines)
            self.logger.setLevel(logging.WARNING)
        self.best_path = os.path.join(args.log_dir, args.save_pretrain_path)
        self.lr_steps = [int(i) for i in str(args.lr_decay_step).split(",")] if args.lr_decay else []
        self.up_epoch = [int(i) for i in str(args.up_epoch).split(",")]

    def _stepper_for(self, B, tail=False):
        """tail: an eager stepper even for the full batch size — the rounds of a data-parallel epoch in which some ranks step on padding
        with rank_weight 0 (a weight cannot change inside a captured graph)."""
        if B == self.step.B and not tail:
            return self.step
        if B not in self.ragged:                         # drop_last=False in the reference: the last batch is smaller
            s = PretrainStep(self.model, self.args, self.scaler[0], self.scaler[1], B, use_graph=False, dp=self.dp, seed=self.args.seed)
            s.m, s.v = self.step.m, self.step.v          # one optimiser state
            self.ragged[B] = s
        s = self.ragged[B]
        s.tA, s.tB, s.lr = self.step.tA, self.step.tB, self.step.lr
        return s

    def train_epoch(self, epoch):
        a = self.args
        tot = tot_f = tot_s = 0.0
        nb = 0
        G = max(int(getattr(a, "steps_per_replay", 1)), 1)
        if G > 1 and not self.step.group_ok(epoch):
            G = 1
        pend = []                                        # (batch index, group input buffer) of full-size batches not yet enqueued

        def account(bi, loss, lf, ls):
            nonlocal tot, tot_f, tot_s, nb
            tot += loss; tot_f += lf; tot_s += ls; nb += 1
            if bi % a.log_step == 0:
                self.logger.info("Train Epoch {}: {}/{} Loss: {:.6f}".format(epoch, bi, self.nb_epoch if self.nb_epoch is not None else "?", loss))

        def flush():
            """G pending batches: ONE graph replay (PretrainStep.step_group); fewer (end of the epoch): single steps.  One host sync per flush."""
            if not pend:
                return
            if len(pend) == G:
                self.step.step_group([s_ for _, s_ in pend], epoch)
                ls_ = self.step.losses_group()
            else:
                ls_ = []
                for _, s_ in pend:
                    self.step.step(s_, epoch)
                    ls_.append(self.step.losses())
            for (bi_, _), l_ in zip(pend, ls_):
                account(bi_, *l_)
            del pend[:]

        for bi, src in enumerate(self.batches(epoch)):
            # (batch, rank weight): a TAIL round of a data-parallel epoch (Run.py batches()) — some rank steps on padding with weight 0.  Every
            # rank gets the tuple form in such a round (weight 1 for a real batch): all of them take the eager stepper, so the sequence of
            # collectives stays the same on every rank.
            weight, tail = 1.0, isinstance(src, tuple)
            if tail:
                src, weight = src
            if src.shape[0] != self.step.B or tail:
                flush()                                  # before the ragged stepper takes over the optimiser counters
            st = self._stepper_for(src.shape[0], tail=tail)
            st.rank_weight = weight
            if G > 1 and st is self.step:
                buf = self.step.group_sources(G)[len(pend)]
                buf.copy_(src, non_blocking=True)        # the loader may reuse its batch buffer: keep a copy until the group is enqueued
                pend.append((bi, buf))
                if len(pend) == G:
                    flush()
                continue
            flush()
            st.step(src, epoch)
            st.rank_weight = 1.0
            if st is not self.step:
                self.step.tA, self.step.tB = st.tA, st.tB
            account(bi, *st.losses())                    # the reference syncs every step too (BasicTrainer.py:98-103)
        flush()
        if nb == 0:
            raise RuntimeError("epoch %d: this rank received no batch (dataset smaller than world_size x batch_size?)" % epoch)
        self.logger.info("**********Train Epoch {}: averaged Loss: {:.6f} averaged Loss_s: {:.6f}".format(epoch, tot_f / nb, tot_s / nb))
        if a.lr_decay and epoch in self.lr_steps:        # MultiStepLR (Run.py:141), stepped per epoch (BasicTrainer.py:117-118)
            self.step.lr *= a.lr_decay_rate
        return tot_f / nb

    def train(self):
        a = self.args
        best_loss, best_state, not_improved = float("inf"), None, 0
        t0 = time.time()
        for epoch in range(1, a.epochs + 1):
            loss = self.train_epoch(epoch)
            if epoch in self.up_epoch:                   # BasicTrainer.py:138-139
                best_loss = float("inf")
            if loss > 1e6:                               # :166-168
                self.logger.warning("Gradient explosion detected. Ending...")
                break
            if loss < best_loss:
                best_loss, not_improved = loss, 0
                best_state = copy.deepcopy(self.model.state_dict())       # :177-180
                self.logger.info("*********************************Current best model saved!")
            else:
                not_improved += 1
            if a.early_stop and not_improved == a.early_stop_patience:    # :171-175
                self.logger.info("Validation performance didn't improve for {} epochs. Training stops.".format(a.early_stop_patience))
                break
        self.logger.info("Total training time: {:.4f}min, best loss: {:.6f}".format((time.time() - t0) / 60, best_loss))
        if a.debug and best_state is not None and (self.dp is None or self.dp.rank == 0):   # :187-189 (flag is inverted in the reference too)
            torch.save(best_state, self.best_path)
            self.logger.info("Saving current best model to " + self.best_path)
        if best_state is not None:                       # :193-195: pretrain mode evaluates on the TRAIN loader
            # data parallel: every rank evaluates ITS share of the batches and the metric sums are all-reduced (test()), so the
            # report covers what the job trained on, as the reference's covers its whole loader
            last = copy.deepcopy(self.model.state_dict())
            self.model.load_state_dict(best_state)
            self.test(self.batches(a.epochs))
            self.model.load_state_dict(last)
        return best_state

    def test(self, batches):
        """Trainer.test of the reference in pretrain mode (model/BasicTrainer.py:209-248): forward at epoch = args.epochs (adaptive
        masks), y_true = label*mask, y_pred = output*mask, inverse transform, per-horizon MAE / RMSE / MAPE / CORR and their average.
        The sums are accumulated on the device by gptst_metrics_accum — no concatenated prediction tensor."""
        from . import ops
        a, model = self.args, self.model
        base = a.input_base_dim
        sums = None
        with torch.no_grad():
            for src in batches:
                if isinstance(src, tuple):               # tail round of a data-parallel epoch (data.epoch_batches): (batch, rank weight)
                    src, weight = src
                    if weight == 0.0:                    # padding: a real rank's batch repeated — it is counted where it is real
                        continue
                src = src.contiguous()
                B, T, N, _ = src.shape
                out, _, masked, _, _ = model(src, None, None, a.epochs)
                if sums is None:
                    sums = ops.metrics_new(T, N, src.device)
                vis = (1 - masked).to(torch.float32).reshape(-1).contiguous()
                ops.metrics_accum(out.reshape(-1, base).contiguous(), src, base + 2, vis, self.scaler[1], self.scaler[0],
                                  getattr(a, "mae_thresh", None), a.mape_thresh, B, T, N, base, *sums)
        if self.dp is not None and self.dp.world > 1:
            import torch.distributed as dist
            if sums is None:                             # a rank whose whole share was padding still joins the reduction
                sums = ops.metrics_new(self.step.T, a.num_nodes, self.step.dev)
            for t_ in sums:                              # float64 sums over this rank's batches -> over the job's
                dist.all_reduce(t_, op=dist.ReduceOp.SUM)
        self.eval_samples = float(sums[1][0, 0, 0]) / base      # (batch, channel) entries per (horizon, node) cell = samples evaluated x base
        rows = ops.metrics_report(*sums)
        for t in range(rows.shape[0] - 1):
            mae, rmse, mape, corr = (float(v) for v in rows[t])
            self.logger.info("Horizon {:02d}, MAE: {:.2f}, RMSE: {:.2f}, MAPE: {:.4f}, CORR:{:.4f}%".format(t + 1, mae, rmse, mape * 100, corr))
        mae, rmse, mape, corr = (float(v) for v in rows[-1])
        self.logger.info("Average Horizon, MAE: {:.2f}, RMSE: {:.2f}, MAPE: {:.4f}%, CORR:{:.4f}".format(mae, rmse, mape * 100, corr))
        return rows
