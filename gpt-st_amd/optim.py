"""``ClipAdam``: clip_grad_norm_ + Adam of the reference loop (model/BasicTrainer.py:95-97, Run.py:134) as the fused HIP optimiser, behind the
``torch.optim.Optimizer`` interface (r06).

    optimizer = torch.optim.Adam(params=model.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False)      # Run.py:134
 -> optimizer = gptst_amd.optim.ClipAdam(model.parameters(), lr=args.lr_init, eps=1.0e-8, max_grad_norm=args.max_grad_norm)

The drop-in ``GPTST_Model`` keeps its parameters in ONE flat buffer and hands autograd gradients that are views of ONE flat gradient buffer, so the
reference's ~470 per-tensor optimiser launches (or torch's multi-tensor lists over 155 tensors) become the two launches of ``gptst_clip_adam``:
global-norm clip (when ``max_grad_norm`` > 0: the loop's own ``clip_grad_norm_`` line then finds a norm <= max_norm and scales by 1) and Adam with
torch's arithmetic (lerp / addcmul / addcdiv forms, bias corrections per parameter GROUP OF FIRST GRADIENT: parameters whose gradient is ``None``
are skipped and keep no state, as in torch — the KL-path parameters before ``change_epoch``, the never-trained decoder time features).
MultiStepLR and friends work: the learning rate is read from ``param_groups`` every step.  Gradients that are not the model's flat views (a foreign
backward) are gathered into a flat buffer first.
"""
import math

import torch

from . import ops
from .model import GPTST_Model


class ClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, max_grad_norm=0.0):
        if weight_decay != 0 or amsgrad:
            raise ValueError("ClipAdam: weight_decay / amsgrad are not part of the reference's optimiser (Run.py:134)")
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, max_grad_norm=float(max_grad_norm)))
        ps = [q for gr in self.param_groups for q in gr["params"]]
        model = GPTST_Model.owner_of(ps[0])
        if model is None or len(self.param_groups) != 1 or len(ps) != len(model._named) or any(a is not b for a, (_, b) in zip(ps, model._named)):
            raise ValueError("ClipAdam steps ALL parameters of ONE gptst_amd GPTST_Model, in model.parameters() order")
        self.model = model
        dev = model.flat.device
        n = model.flat.numel()
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.gflat = None                                   # gather buffer (only when the gradients are not flat views)
        self.stats = torch.zeros(8, device=dev)
        self.stats_out = torch.zeros(8, device=dev)
        self.hyper = torch.zeros(16, device=dev)
        self._ring = [dict(h=torch.zeros(16).pin_memory(), ev=None) for _ in range(4)]
        self._ring_i = 0
        self.tA = self.tB = 0
        self._first_key = min((k for k, _ in model._named), key=lambda k: model._offs[k])      # first tensor of the reconstruction-path segment (offset 0)
        self._first = dict(model._named)[self._first_key]
        self._firstB = next((t for k, t in model._named if model.nA <= model._offs[k] < model.nA + model.nB), None)

    def _flat_grad(self):
        """the flat gradient buffer behind the parameters' .grad views — or a gathered copy"""
        mdl = self.model
        g0, last = self._first.grad, getattr(mdl, "_last_gflat", None)
        # (autograd keeps a returned view as .grad DETACHED — no ._base — so the model remembers the flat buffer its last backward node wrote)
        if last is not None and g0 is not None and g0.data_ptr() == last.data_ptr() + 4 * mdl._offs[self._first_key]:
            kB = self._firstB
            if kB is None or kB.grad is None or (last.data_ptr() <= kB.grad.data_ptr() < last.data_ptr() + 4 * last.numel()):
                return last
        if self.gflat is None:
            self.gflat = torch.zeros_like(mdl.flat)
            self._gviews = mdl.views_of(self.gflat)
        with torch.no_grad():
            have = [(self._gviews[k], t.grad) for k, t in mdl._named if t.grad is not None]
            torch._foreach_copy_([a for a, _ in have], [b for _, b in have])
        return self.gflat

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._first.grad is None:
            return loss
        gr = self.param_groups[0]
        b1, b2 = gr["betas"]
        hasB = self._firstB is not None and self._firstB.grad is not None
        self.tA += 1
        if hasB:
            self.tB += 1
        tA, tB, lr = self.tA, self.tB, float(gr["lr"])
        sl = self._ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        if sl["ev"] is not None:
            sl["ev"].synchronize()
        sl["h"].numpy()[:13] = (lr / (1 - b1 ** tA), math.sqrt(1 - b2 ** tA), lr / (1 - b1 ** tB) if tB else 0.0, math.sqrt(1 - b2 ** tB) if tB else 1.0,
                                b1, b2, gr["eps"], gr["max_grad_norm"], 1.0 if hasB else 0.0, 0.0, 1.0, 1 - b1, 1 - b2)
        self.hyper.copy_(sl["h"], non_blocking=True)
        if sl["ev"] is None:
            sl["ev"] = torch.cuda.Event()
        sl["ev"].record()
        mdl = self.model
        ops.clip_adam(mdl.flat, self._flat_grad(), self.m, self.v, mdl.nA, mdl.nB, self.hyper, self.stats, stats_out=self.stats_out)
        return loss

    def state_dict(self):
        """torch's dictionary + the flat moment buffers and step counts (the per-parameter `state` of torch.optim.Adam has no counterpart here)"""
        d = super().state_dict()
        d["gptst_flat"] = dict(exp_avg=self.m.clone(), exp_avg_sq=self.v.clone(), step=self.tA, step_kl=self.tB)
        return d

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop("gptst_flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self.m.copy_(flat["exp_avg"]); self.v.copy_(flat["exp_avg_sq"])
            self.tA, self.tB = int(flat["step"]), int(flat["step_kl"])

    def grad_norm(self):
        """total gradient norm of the last step, before clipping (what clip_grad_norm_ returns) — synchronises"""
        return float(self.stats_out[4].sqrt())
