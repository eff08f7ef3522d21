"""hipGraph replays under the drop-in ``GPTST_Model`` (r06).

The reference's training loop (model/BasicTrainer.py:72-103) calls ``model(source, label, None, epoch)``, builds its loss with torch ops, calls
``loss.backward()``, clips and steps a torch optimiser.  Behind the one-line import swap of INTEGRATION.md that loop was host-bound: ~55 C-ABI
enqueues for the forward and ~50 for the backward through ctypes, per step (162-217 steps/s against 885 for the fused stepper).  Shapes are static
in that loop, so the autograd node's forward and backward are captured ONCE per (input shape, masking phase) and replayed:

  forward graph   guide classifier -> mask generation (torch.rand noise, the class order / budgets of the call in a small device buffer) -> encoder ->
                  decoder, on the kernels the fused stepper uses (low-rank first layers, forward chains)
  backward graph  d_out [, d_prob] -> every parameter gradient in ONE flat buffer, on the dPre chain (backward pairs, one-launch cap backward)

Semantics kept: the returned tensors are fresh (clones of the graph's static outputs), parameter gradients are ordinary ``.grad`` tensors (views of
a flat buffer, as in the eager path), KL-path parameters get ``None`` gradients when ``prob`` did not enter the loss, and every situation the
static buffers cannot serve falls back to the eager node: gradients disabled, injected mask inputs, a second forward before the first one's
backward (eager node); ``flow_decode`` entering the loss (the backward body enqueued eagerly with the extra term); accumulated ``.grad``s that alias
the static buffer (detached from it first); a capture that fails (the model then stays on the eager node).  ``GPTST_MODULE_GRAPHS=0`` turns it off.
"""
import os
import random
import weakref

import torch

from . import engine, ops

ENABLED = os.environ.get("GPTST_MODULE_GRAPHS", "1") == "1"


_CACHE = weakref.WeakKeyDictionary()        # model -> {(input shape, masking phase, flat buffer address): GraphedPretrain}


class _Unavailable:
    """placeholder of a (shape, phase) whose capture failed: the model keeps its eager node for it"""
    busy = True


def graphs_of(model, key):
    d = _CACHE.setdefault(model, {})
    gp = d.get(key)
    if gp is None:
        if len(d) >= 8:                         # (ragged last batches, re-flattened buffers: bounded)
            d.clear()
        try:
            gp = GraphedPretrain(model, key[0], key[1])
        except Exception as e:                  # noqa: BLE001  (out of memory for the static buffers, a runtime that cannot capture, ...)
            import sys
            print("gpt-st_amd: capturing the module's forward failed (%s: %s) -> eager autograd node for input shape %s"
                  % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "", tuple(key[0])), file=sys.stderr)
            torch.cuda.synchronize()
            gp = _Unavailable()
        d[key] = gp
    return gp


class _Token:
    """dies with the autograd node of a graphed forward: the static buffers are free again"""
    __slots__ = ("__weakref__",)


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gp, gen_id, *params):
        ctx.gp, ctx.gen_id = gp, gen_id
        ctx.token = _Token()
        weakref.finalize(ctx.token, gp._release, gen_id)
        ctx.set_materialize_grads(False)                 # an output that did not enter the loss arrives as None, not as zeros
        out, dec, prob, c1 = gp.out.clone(), gp.dec.clone(), gp.prob.clone(), gp.c1.clone()
        ctx.mark_non_differentiable(c1)
        return out, dec, prob, c1

    @staticmethod
    def backward(ctx, d_out, d_dec, d_prob, _dc):
        gp = ctx.gp
        if ctx.gen_id != gp.gen_id:
            raise RuntimeError("gpt-st_amd: backward of a graphed forward whose static buffers were reused by a later forward "
                               "(set GPTST_MODULE_GRAPHS=0 for this calling pattern)")
        grads = gp.backward(d_out, d_dec, d_prob)
        gp._release(ctx.gen_id)
        return (None, None) + grads


class GraphedPretrain:
    """forward / backward graphs of one (input shape, masking phase) of a GPTST_Model in pretrain mode"""

    def __init__(self, model, shape, phase):
        self._model, self.shape, self.phase = weakref.ref(model), tuple(shape), phase      # (weak: the cache is keyed by the model)
        B, T, N, F = shape
        self.dev = model.flat.device
        self.base, self.HS = model.input_base_dim, model.HS
        self.dims = (B, T, N, model.hidden_dim)
        self.M = B * T * N
        self.src = torch.zeros(*shape, device=self.dev)
        self.ctrl = torch.zeros(self.HS + 2, dtype=torch.int32, device=self.dev)             # [class order | adaptive budget, random budget]
        self._ring = [dict(h=torch.zeros(self.HS + 2, dtype=torch.int32).pin_memory(), ev=None) for _ in range(4)]
        self._ring_i = 0
        self.arena_f, self.arena_b = engine.ZeroArena(self.dev), engine.ZeroArena(self.dev)
        self.gflat = torch.zeros_like(model.flat)
        self.g = model.views_of(self.gflat)
        self.d_out = torch.zeros(self.M, self.base, device=self.dev)
        self.d_prob = torch.zeros(self.M, self.HS, device=self.dev)
        self.gen_id, self.busy = 0, False
        from .model import _segment
        self._segs = [_segment(k) for k in model.param_keys]
        self.gf, self.gb = None, {}
        self._capture_fwd()

    @property
    def model(self):
        return self._model()

    # ---- bodies (the same engine calls as step.PretrainStep, minus the fused loss heads: the loss belongs to the caller) -------------------------
    def _fwd_body(self):
        m, base, dims = self.model, self.base, self.dims
        p = m.param_views()
        src = self.src
        engine.CTX.ARENA = self.arena_f
        self.arena_f.begin(zero=True)
        try:
            tidx = m._tidx(src)
            gen = engine.gen_all(p, tidx, dims)
            lowrank = engine.chain_ok(dims)
            prob, sv_g = engine.guide_fwd(p, src, tidx, dims, base, gen=gen["guide"], lowrank_in=lowrank)
            if self.phase == 0:                                                                # GPTST.py:314-323
                mask = ops.mask_random(torch.rand(self.M * base, device=self.dev), int(self.M * base * m.mask_ratio))
            else:                                                                              # :344-413
                label, counts = ops.mask_labels(prob.reshape(self.M, -1))
                mask = ops.mask_adaptive(label, counts, self.ctrl[:self.HS], self.ctrl[self.HS:], torch.rand(self.M, device=self.dev),
                                         torch.rand(self.M, device=self.dev), m.ada_type == "all", base)[2]
            if engine.chain_fwd_ok(dims):
                emb, c1, tidx, sv_e, dec_head = engine.model_fwd(p, src, mask, dims, base, m.num_route, m.scaler_zeros, gen=gen[engine.ENC], tidx=tidx,
                                                                 dec_gen=gen[engine.DEC], lowrank_in=lowrank)
            else:
                emb, c1, tidx, sv_e = engine.model_fwd(p, src, mask, dims, base, m.num_route, m.scaler_zeros, gen=gen[engine.ENC], tidx=tidx,
                                                       lowrank_in=lowrank)
                dec_head = None
            out, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, m.num_route, gen=gen[engine.DEC], dec_head=dec_head)
        finally:
            engine.CTX.ARENA = None
        B, T, N, C = dims
        self.out, self.dec, self.prob, self.c1, self.mask = out.view(B, T, N, base), dec.view(B, T, N, C), prob.view(B, T, N, -1), c1, mask
        self.saved = (tidx, gen, sv_g, sv_e, sv_d, dec, prob)

    def _bwd_body(self, has_kl, d_dec=None):
        m, base, dims = self.model, self.base, self.dims
        B, T, N, C = dims
        p, g = m.param_views(), self.g
        tidx, gen, sv_g, sv_e, sv_d, dec, prob = self.saved
        engine.CTX.ARENA = self.arena_b
        self.arena_b.begin(zero=True)
        try:
            self.gflat.zero_()
            red = engine.Reductions()
            chain = engine.chain_ok(dims)
            wo = "decoder.dim_flow_out."
            dd = ops.lin_in(self.d_out, base, base, p[wo + "weight"], None, C, wlayout=1)            # backward of dim_flow_out (GPTST.py:455)
            ops.rowouter(self.d_out, base, base, dec, g[wo + "weight"], 1, asum=g[wo + "bias"])
            if d_dec is not None:                                                                   # (eager only: flow_decode entered the loss)
                dd = dd + d_dec
            if chain:                                                                               # dPre chain: times lrelu'(dec), dec = a LeakyReLU output
                dd = dd * torch.where(dec > 0, 1.0, 0.01)
            engine.model_bwd(p, g, self.src, self.mask, tidx, sv_e, sv_d, dec, None, None, dims, base, m.scaler_zeros, red, dd=dd, chain=chain)
            if has_kl:      # softmax backward: dlogit = prob * (d_prob - sum(d_prob * prob)), then MLP_RL.ln3 (GPTST.py:33)
                dp = self.d_prob
                dlogit = (prob * (dp - (dp * prob).sum(-1, keepdim=True))).contiguous()
                mm = "encoder.MLP_RL."
                h2, HS = sv_g[3], self.HS
                dh2 = ops.lin_in(dlogit, HS, HS, p[mm + "ln3.weight"], None, C, wlayout=1)
                ops.rowouter(dlogit, HS, HS, h2, g[mm + "ln3.weight"], 1, asum=g[mm + "ln3.bias"])
                if chain:
                    dh2 = dh2 * torch.where(h2 > 0, 1.0, 0.01)
                engine.guide_bwd(p, g, self.src, tidx, sv_g, None, dims, base, red, dh2=dh2, chain=chain)
            red.flush(tidx)
            self._bwd_keep = red                                 # (the buffers its jobs point at live in the graph's pool)
        finally:
            engine.CTX.ARENA = None

    # ---- capture -----------------------------------------------------------------------------------------------------------------------------------
    def _warm(self, fn):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def _capture_fwd(self):
        with torch.no_grad():
            self.ctrl.copy_(torch.tensor(list(range(self.HS)) + [self.M // 8, self.M // 8], dtype=torch.int32))      # plausible budgets for the warm-up
            self._warm(self._fwd_body)
            self.gf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gf, capture_error_mode="thread_local"):
                self._fwd_body()

    def _capture_bwd(self, has_kl):
        with torch.no_grad():
            self._warm(lambda: self._bwd_body(has_kl))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.gf.pool(), capture_error_mode="thread_local"):
                self._bwd_body(has_kl)
        self.gb[has_kl] = g

    # ---- one call ------------------------------------------------------------------------------------------------------------------------------------
    def _release(self, gen_id):
        if gen_id == self.gen_id:
            self.busy = False

    def forward(self, source, epoch):
        m = self.model
        self.src.copy_(source, non_blocking=True)
        if self.phase == 1:
            sl = self._ring[self._ring_i]
            self._ring_i = (self._ring_i + 1) % len(self._ring)
            if sl["ev"] is not None:
                sl["ev"].synchronize()
            list_c = list(range(self.HS))
            random.shuffle(list_c)                                                             # GPTST.py:357-358
            ada, rnd = m.adaptive_counts(self.M, epoch)
            sl["h"].numpy()[:] = list_c + [ada, rnd]
            self.ctrl.copy_(sl["h"], non_blocking=True)
            if sl["ev"] is None:
                sl["ev"] = torch.cuda.Event()
            sl["ev"].record()
        self.gen_id += 1
        self.busy = True
        self.gf.replay()
        params = [t for _, t in m._named]
        out, dec, prob, c1 = _GraphedFn.apply(self, self.gen_id, *params)
        B, T, N, _ = self.shape
        mask_i = self.mask.view(B, T, N, self.base).to(torch.int64)
        hs1 = c1.view(B, T, self.HS, N).transpose(-1, -2)                                      # :424
        return out, dec, 1 - mask_i, prob, hs1

    def backward(self, d_out, d_dec, d_prob):
        m = self.model
        has_kl = d_prob is not None                      # (also in the random-mask phase, should a caller put the guide probabilities into its loss there)
        if d_out is None:
            self.d_out.zero_()
        else:
            self.d_out.copy_(d_out.reshape(self.M, self.base))
        if has_kl:
            self.d_prob.copy_(d_prob.reshape(self.M, self.HS))
        if d_dec is None and has_kl not in self.gb:
            try:
                self._capture_bwd(has_kl)
            except Exception as e:                       # noqa: BLE001  (the backward then runs eagerly on the same kernels)
                import sys
                print("gpt-st_amd: capturing the module's backward failed (%s: %s) -> enqueued eagerly"
                      % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else ""), file=sys.stderr)
                torch.cuda.synchronize()
                self.gb[has_kl] = None
        # .grad tensors that still alias the static gradient buffer (a loop that accumulates, or zero_grad(set_to_none=False)) are detached from it
        # first: the replay rewrites that memory
        lo = self.gflat.data_ptr()
        hi = lo + 4 * self.gflat.numel()
        for _, t in (m._named[0], m._named[-1]):
            if t.grad is not None and lo <= t.grad.data_ptr() < hi:
                for _, q in m._named:
                    if q.grad is not None and lo <= q.grad.data_ptr() < hi:
                        q.grad = q.grad.clone()
                break
        if d_dec is not None or self.gb.get(has_kl) is None:       # flow_decode entered the loss (or no graph): the same body, enqueued eagerly
            with torch.no_grad():
                self._bwd_body(has_kl, d_dec=None if d_dec is None else d_dec.reshape(self.M, -1).contiguous())
        else:
            self.gb[has_kl].replay()
        m._last_gflat = self.gflat
        g = m.views_of(self.gflat)                       # fresh view objects: autograd keeps them as .grad without a copy
        return tuple(g[k] if (sg == 0 or (sg == 1 and has_kl)) else None for k, sg in zip(m.param_keys, self._segs))
