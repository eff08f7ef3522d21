"""``-mode eval`` training loop (reference model/BasicTrainer.py:38-123 non-pretrain branches, :125-197, :209-248): a downstream predictor
is trained on the enhanced embedding — frozen HIP encoder -> Fusion -> predictor (enhance.EnhanceFrontEnd) — with the masked-MAE loss on
de-normalised values (Run.py:91-101, lib/metrics.py:11-18), gradient clipping and Adam + MultiStepLR (Run.py:134-141); the best model
on the validation loss is kept (early stopping as in the reference) and tested per horizon on the test loader, the metric sums
accumulated on the device by gptst_metrics_accum.  The trainable part is ordinary torch autograd (downstream plumbing, not the hot path)."""
import copy
import time

import torch

from .trainer import get_logger


def masked_mae(pred, true, mean, std, thresh):
    """scaler_mae_loss without a mask (Run.py:91-101): de-normalise, keep cells with true > thresh, mean |true - pred|."""
    p, y = pred * std + mean, true * std + mean
    keep = y > thresh
    return (y - p).abs()[keep].mean()


class EvalTrainer:
    def __init__(self, model, args, train, val, test, scaler_mean, scaler_std):
        self.model, self.args = model, args
        self.train_loader, self.val_loader, self.test_loader = train, val, test
        self.mean, self.std = float(scaler_mean), float(scaler_std)
        params = [p for p in model.parameters() if p.requires_grad]
        self.opt = torch.optim.Adam(params, lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False)      # Run.py:134-135
        steps = [int(i) for i in str(args.lr_decay_step).split(",")] if args.lr_decay else []
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=steps, gamma=args.lr_decay_rate) if steps else None
        self.logger = get_logger(args.log_dir, name=str(args.model), debug=args.debug)
        self.nin = args.input_base_dim + args.input_extra_dim

    def _loss(self, out, target):
        return masked_mae(out, target[..., :self.args.output_dim], self.mean, self.std, self.args.mape_thresh)

    def train_epoch(self, epoch):                                             # BasicTrainer.py:67-123, mode != 'pretrain'
        a = self.args
        self.model.train()
        total, nb = 0.0, len(self.train_loader)
        for i, (data, target) in enumerate(self.train_loader):
            self.opt.zero_grad()
            out = self.model(data[..., :self.nin].contiguous(), target[..., :self.nin])[0]
            loss = self._loss(out, target)
            loss.backward()
            if a.grad_norm:
                torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.requires_grad], a.max_grad_norm)
            self.opt.step()
            total += float(loss)
            if i % a.log_step == 0:
                self.logger.info("Train Epoch {}: {}/{} Loss: {:.6f}".format(epoch, i, nb, float(loss)))
        self.logger.info("**********Train Epoch {}: averaged Loss: {:.6f}".format(epoch, total / max(nb, 1)))
        if self.sched is not None:
            self.sched.step()
        return total / max(nb, 1)

    def val_epoch(self, epoch, loader):                                       # :38-65
        self.model.eval()
        total = 0.0
        with torch.no_grad():
            for data, target in loader:
                loss = self._loss(self.model(data[..., :self.nin].contiguous(), None)[0], target)
                if not torch.isnan(loss):
                    total += float(loss)
        val = total / max(len(loader), 1)
        self.logger.info("**********Val Epoch {}: average Loss: {:.6f}".format(epoch, val))
        return val

    def train(self):                                                          # :125-197
        a = self.args
        best, best_loss, not_improved = None, float("inf"), 0
        up_epoch = [int(i) for i in str(a.up_epoch).split(",")]
        t0 = time.time()
        for epoch in range(1, a.epochs + 1):
            tr = self.train_epoch(epoch)
            if epoch in up_epoch:
                best_loss = float("inf")
            val = self.val_epoch(epoch, self.val_loader if self.val_loader is not None else self.test_loader)
            if val < best_loss:
                best_loss, not_improved = val, 0
                self.logger.info("*********************************Current best model saved!")
                best = copy.deepcopy(self.model.state_dict())
            else:
                not_improved += 1
            if tr > 1e6:
                self.logger.warning("Gradient explosion detected. Ending...")
                break
            if a.early_stop and not_improved == a.early_stop_patience:
                self.logger.info("Validation performance didn't improve for {} epochs. Training stops.".format(a.early_stop_patience))
                break
        self.logger.info("Total training time: {:.4f}min, best loss: {:.6f}".format((time.time() - t0) / 60, best_loss))
        if best is not None:
            self.model.load_state_dict(best)
        return best, self.test(self.test_loader)

    def test(self, loader):                                                   # :209-248, mode != 'pretrain'
        from . import ops
        a = self.args
        base = a.output_dim
        self.model.eval()
        sums = None
        with torch.no_grad():
            for data, target in loader:
                out = self.model(data[..., :self.nin].contiguous(), None)[0]
                B, T, N, _ = out.shape
                if sums is None:
                    sums = ops.metrics_new(T, N, out.device)
                lab = target[..., :base].contiguous()
                ops.metrics_accum(out.reshape(-1, base).contiguous(), lab, base, None, self.std, self.mean, getattr(a, "mae_thresh", None),
                                  a.mape_thresh, B, T, N, base, *sums)
        rows = ops.metrics_report(*sums)
        for t in range(rows.shape[0] - 1):
            mae, rmse, mape, corr = (float(v) for v in rows[t])
            self.logger.info("Horizon {:02d}, MAE: {:.2f}, RMSE: {:.2f}, MAPE: {:.4f}, CORR:{:.4f}%".format(t + 1, mae, rmse, mape * 100, corr))
        mae, rmse, mape, corr = (float(v) for v in rows[-1])
        self.logger.info("Average Horizon, MAE: {:.2f}, RMSE: {:.2f}, MAPE: {:.4f}%, CORR:{:.4f}".format(mae, rmse, mape * 100, corr))
        return rows
