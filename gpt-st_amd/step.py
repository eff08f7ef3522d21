"""One fused pretraining step = the body of reference BasicTrainer.train_epoch (model/BasicTrainer.py:72-103):
zero_grad -> forward (mask generation included) -> masked-MAE (+0.1 KL) -> backward -> clip_grad_norm_(5) -> Adam,
entirely as HIP kernels enqueued on one stream and captured in a hipGraph (two graphs: random-mask phase and
adaptive-mask + KL phase).  Nothing in the step synchronises with the host: loss scalars are read from a device
statistics block whenever the caller asks.  Data-parallel: one process per GPU, one RCCL all-reduce (sum) of
[flat gradient | loss statistics] between backward and the optimiser (dist.py).
"""
import math
import os
import random

import torch

from . import engine, ops

# mask generation on the 24-bit noise lattice (gptst_mask_*_u24, r04): the step's Philox noise and torch.rand fixtures are k * 2^-24, so the radix
# select takes two 12-bit digits on the integers instead of three float-bit digits: 6 launches instead of 8 in the adaptive phase.  (The whole
# generation as ONE single-workgroup launch was measured too: ~120 us against ~45 us at 65 280 cells — one CU's bandwidth; it serves <= 8192 cells.)
DEFER_GEN = os.environ.get("GPTST_DEFER_GEN", "1") == "1"      # the STHCNs' parameter generation inside the cooperative mask launch (r05)
U24 = os.environ.get("GPTST_MASK_U24", "1") == "1"


class PretrainStep:
    RING = 8                                  # pinned host slots in flight (see __init__)

    def __init__(self, model, args, scaler_mean, scaler_std, batch_size, use_graph=True, dp=None, seed=0, global_mask=True,
                 deterministic=None):
        """``seed`` drives the class-order shuffle and (data parallel, global masks) the mask noise: it must be the SAME on every
        rank, because every rank regenerates the selection over the global batch and keeps its rows (dist.py)."""
        self.model, self.args = model, args
        self.mean, self.std = float(scaler_mean), float(scaler_std)
        self.B, self.T, self.N = batch_size, args.lag, args.num_nodes
        self.C, self.base, self.HS = args.hidden_dim, args.input_base_dim, args.HS
        self.dims = (self.B, self.T, self.N, self.C)
        self.dev = model.flat.device
        assert self.dev.type == "cuda", "PretrainStep needs the model on an MI355X (no CPU fallback)"
        self.use_graph, self.dp = use_graph, dp
        # bit-reproducible steps (GPTST_DETERMINISTIC=1): single-owner, fixed-order variants of the two reductions that end in float
        # atomics by default (ops.set_deterministic); with injected mask noise two runs of a step sequence are then bit-identical
        self.deterministic = (os.environ.get("GPTST_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        n = model.flat.numel()
        self.gbuf = torch.zeros(n + 8, device=self.dev)             # [flat gradient | stats] -> one collective
        self.gflat, self.stats = self.gbuf[:n], self.gbuf[n:]
        self.g = model.views_of(self.gflat)
        self.m, self.v = torch.zeros(n, device=self.dev), torch.zeros(n, device=self.dev)
        M = self.B * self.T * self.N
        self.src = torch.zeros(self.B, self.T, self.N, self.base + 2, device=self.dev)
        self.noise = torch.zeros(M * self.base, device=self.dev)
        self.noise_ar = torch.zeros(2 * M, device=self.dev)          # adaptive phase: [noise_a | noise_r], drawn by ONE launch
        self.noise_a, self.noise_r = self.noise_ar[:M], self.noise_ar[M:]
        # per-step host scalars in ONE device buffer / ONE H2D copy: [hyper (16 fp32) | list_c (HS int32) | adaptive_num, random_num | rng seed, step]
        self.hc = torch.zeros(16 + self.HS + 4, dtype=torch.int32, device=self.dev)
        self.hyper = self.hc[:16].view(torch.float32)
        self.ctrl = self.hc[16:16 + self.HS + 2]
        self.rng_words = self.hc[16 + self.HS + 2:]                  # Philox key of the step's mask noise (step_begin)
        self.noise_seed = 1234567 + seed
        # Per-step host scalars travel through a RING of pinned slots, each guarded by an event recorded behind its H2D copies:
        # step() never synchronises, so with a single pinned buffer the host could rewrite the Adam bias corrections / class
        # order of step k+j before the DMA of step k has read them (hundreds of steps are queued back to back by bench.py).
        self._ring = [self._views(torch.zeros(16 + self.HS + 4, dtype=torch.int32).pin_memory()) for _ in range(self.RING)]
        self._ring_i = 0
        self.phase_kl = False                                        # phase of the last enqueued step (losses())
        self._gK, self._g_last, self._group_failed = 0, None, False  # step_group(): group size set up, (K, phase) of the last group, capture refused
        self.stats_out = torch.zeros(8, device=self.dev)            # snapshot of stats after the step (graph output)
        self.mask_buf = torch.ones(M * self.base, device=self.dev)   # teacher-forced mask (parity runs)
        self.last_mask = None
        self.force_mask = False
        self.tA = self.tB = 0
        self.lr = args.lr_init
        self.rng = random.Random(seed)
        self.graphs = {}
        self.inject_noise = False
        self.global_count_scale = False      # set by subclasses that all-reduce [gradient | statistics] themselves (shard.py)
        # r05, lost in-launch hand-offs (a bounded wait of the pair / role launches expired: the GPU was time-sliced away from the producer for
        # seconds): the optimiser skips the update of such a step (gptst_clip_adam's guard; stats_out[5] > 0), losses() / losses_group() notice,
        # switch this stepper to the launches WITHOUT hand-offs (safe_mode: graphs re-captured) and re-run the skipped steps from the untouched weights
        self.safe_mode = False
        self.lost_steps = 0                  # steps re-run so far
        self.lost_batches = 0                # ... and steps that were skipped with a later one and could not be repeated (their batches were gone)
        self._unseen = []                    # phase of every step enqueued since the host last read the statistics
        self._last_call = None               # (epoch, list_c) of the last plain step()
        self._g_list_cs = None               # class orders of the last group
        self.arena = engine.ZeroArena(self.dev)
        self.fused_tails = os.environ.get("GPTST_FUSED_TAILS", "1") == "1" and engine.fused_tails_ok(model.param_views(), self.C, self.base, self.HS)
        # ---- data parallel: masks over the GLOBAL batch (dist.py) ----
        self.W = dp.world if dp is not None else 1
        self.gmask = bool(global_mask) and dp is not None and self.W > 1
        if self.gmask:
            Mg = M * self.W
            self.noise_g = torch.zeros(Mg * self.base, device=self.dev)
            self.noise_ar_g = torch.zeros(2 * Mg, device=self.dev)
            self.noise_a_g, self.noise_r_g = self.noise_ar_g[:Mg], self.noise_ar_g[Mg:]
            self.label_l = torch.zeros(M, dtype=torch.int32, device=self.dev)
            self.label_g = torch.zeros(Mg, dtype=torch.int32, device=self.dev)
        # r04, data parallel on the capturable communicator: the gradient leaves in BUCKETS — the decoder's parameters are complete when the decoder's
        # backward ends, so their reductions + all-reduce run on a forked branch UNDER the encoder's backward; only the encoder / KL bucket
        # is exchanged behind the chain (one 4.15 MB all-reduce behind everything before).  GPTST_DP_OVERLAP=0: the single all-reduce.
        # (the bucket's reductions stay on the chain's stream, only its all-reduce is forked: forking the reductions too cost 5 % at one rank)
        self.always_guide = os.environ.get("GPTST_ALWAYS_GUIDE", "0") == "1"      # run the guide classifier in the random-mask phase too (as the reference does)
        # r06: OFF by default.  Measured with one forced-DP rank (profiles/r06_dp_world1.txt): 854.9 steps/s with the overlap, 885.6 without, 891.1 without
        # data parallelism — the early bucket needs its own reduction flush in the middle of the backward (gram_bwd + job table + time-feature launch:
        # ~50 us) and keeps the reductions out of the routing backward's idle slots, i.e. it costs ~41 us per step on EVERY rank to hide part of a
        # 4.15 MB all-reduce that takes about that long un-hidden on eight xGMI-connected GPUs.  GPTST_DP_OVERLAP=1 turns it on.
        self.dp_overlap = (os.environ.get("GPTST_DP_OVERLAP", "0") != "0" and dp is not None and getattr(dp, "capturable", False)
                           and (self.W > 1 or os.environ.get("GPTST_FORCE_DP", "0") == "1"))
        self.dec_lo, self.dec_hi = self._decoder_bucket(model)
        self.rank_weight = 1.0                          # 0.0: this rank steps on padding (eager tail round of a data-parallel epoch, see _allreduce)
        self.fork_side = torch.cuda.Stream() if self.dp_overlap else None

    # ---- the enqueued work ---------------------------------------------------------------------------------------
    # A step is enqueued in two parts: part 1 ends with the guide classifier (the cluster labels of the local rows), part 2 starts
    # with the mask.  Under data parallelism with global masks the adaptive phase exchanges the labels BETWEEN the two parts (one
    # all-gather; the class histogram is taken from the gathered labels) — as two hipGraphs sharing one memory pool, so that the
    # guide forward is computed once and no collective sits inside a graph.  Everything else runs the two parts back to back.
    def _needs_exchange(self, phase):
        return self.gmask and phase == 1 and not self.force_mask

    @staticmethod
    def _decoder_bucket(model):
        """[lo, hi) of the decoder's trained parameters in the flat buffer (they are contiguous: segment A is in registration order,
        encoder first; the decoder's never-trained time features live in the last segment)."""
        offs = [(o, model.state_dict()[k].numel()) for k, o in model._offs.items() if k.startswith("decoder.") and o < model.nA]
        lo = min(o for o, _ in offs)
        hi = max(o + (n + 3) // 4 * 4 for o, n in offs)
        assert hi == model.nA and all(o >= lo for o, _ in offs), (lo, hi, model.nA)
        assert not any(lo <= o < hi for k, o in model._offs.items() if not k.startswith("decoder.")), "decoder bucket is not contiguous"
        return lo, hi

    def _allreduce(self, sl):
        """All-reduce of a slice of the packed [gradient | statistics] buffer.  rank_weight = 0 (an eager tail step of a data-parallel epoch:
        this rank has no batch of its own in the last, incomplete round and steps on padding) zeroes the rank's contribution first — gradient
        AND statistics, so the global kept-cell count the optimiser divides by does not see the padding either."""
        if self.rank_weight != 1.0:
            assert not self.use_graph, "a rank weight is baked into a captured graph: tail rounds use eager steppers"
            lost = self.stats[5:6].clone()              # the hand-off expiry count rides in the statistics block: a padding rank's expiry must reach its
            sl.mul_(self.rank_weight)                   # peers unscaled, or it alone skips the update and re-runs a step of collectives nobody joins
            self.stats[5:6].copy_(lost)
        self.dp.allreduce_(sl)

    def _bucket_ready(self, k):
        """engine.Reductions callback (inside the side-stream fork): bucket 0 = the decoder's gradient is final -> its all-reduce starts now"""
        if k == 0:
            self._allreduce(self.gbuf[self.dec_lo:self.dec_hi])
            self._dec_reduced = True

    def _dp_in_graph(self):
        """Data parallel on a capturable communicator (dist.DataParallel(native=True)): label gather, gradient all-reduce and optimiser are
        enqueued as part of the step body — one hipGraph per phase holds everything, as in the single-GPU case."""
        return self.dp is not None and getattr(self.dp, "capturable", False) and not getattr(self, "_graph_comm_failed", False)

    def _part1(self, phase):
        p, dims, base = self.model.param_views(), self.dims, self.base
        engine.CTX.ARENA = self.arena
        src = self.src
        # zero_grad + the step's zero scratch + the time index of node 0: one launch
        noise = None                                              # the step's mask noise is drawn by the same launch (unless injected / forced)
        if not self.inject_noise and not self.force_mask:
            noise = (self.noise_g if phase == 0 else self.noise_ar_g) if self.gmask else (self.noise if phase == 0 else self.noise_ar)
        tidx = ops.step_begin(self.gbuf, self.arena.begin(zero=False), src, base, noise=noise, rng=self.rng_words)
        # the guide classifier (GPTST.py:325-332) feeds the adaptive mask and the KL term only: the random-mask phase of the FUSED step neither
        # generates its parameters nor runs it (the reference computes and discards the logits there; GPTST_Model.forward still returns them)
        need_guide = phase == 1 or self.always_guide
        # time embeddings + every generated parameter: 3 launches; r05: the STHCNs' forward jobs wait for the mask's launch (_part2_impl)
        gen = engine.gen_all(p, tidx, dims, guide=need_guide, defer=DEFER_GEN)
        red = engine.Reductions()
        red.no_carry = self.deterministic
        self._dec_reduced = False
        if self.dp_overlap and self._dp_in_graph():
            red.on_bucket = self._bucket_ready
            red.bucket_inline, red.fork_side = True, self.fork_side
        lowrank = self.fused_tails and engine.chain_ok(dims)      # the guide's backward (KL path) is the dPre chain: its first layers may run low-rank
        prob, sv_g = engine.guide_fwd(p, src, tidx, dims, base, gen=gen["guide"], lowrank_in=lowrank) if need_guide else (None, None)
        if self._needs_exchange(phase):
            self.label_l.copy_(sv_g[4])                           # this rank's cluster labels -> all-gather (_exchange_labels)
        return dict(tidx=tidx, gen=gen, red=red, prob=prob, sv_g=sv_g)

    def _part2(self, phase, ctx):
        if not self.safe_mode:
            return self._part2_impl(phase, ctx)
        with engine.no_handoffs():              # hyperTem backward per layer, cross-time backward as a prologue: no in-launch waits
            return self._part2_impl(phase, ctx)

    def _part2_impl(self, phase, ctx):
        mdl, p, g, dims, base = self.model, self.model.param_views(), self.g, self.dims, self.base
        a = self.args
        M = self.B * self.T * self.N
        src, tidx, gen, red, prob, sv_g = self.src, ctx["tidx"], ctx["gen"], ctx["red"], ctx["prob"], ctx["sv_g"]
        engine.CTX.ARENA = self.arena
        pend = gen.pop("pending", None)        # the STHCNs' generated-parameter jobs: inside the mask's launch where that is the cooperative one
        if self.gmask:
            mask = self._global_mask(phase, jobs=pend)
        else:
            if self.force_mask:
                mask = self.mask_buf
            elif phase == 0:
                mask = ops.mask_random(self.noise, int(M * base * a.mask_ratio), ws=self._mask_ws(), u24=U24, jobs=pend)       # Philox / torch.rand noise: k * 2^-24
            else:
                label, counts = ops.labels_and_counts(prob, sv_g[4])
                mask = ops.mask_adaptive(label, counts, self.ctrl[:self.HS], self.ctrl[self.HS:], self.noise_a, self.noise_r,
                                         a.ada_type == "all", base, ws=self._mask_ws(), u24=U24, jobs=pend)[2]
        if pend is not None:
            pend.launch()                      # (forced mask: nothing carried them; a no-op after a mask call)
        self.last_mask = mask
        dec_head = None
        lowrank = self.fused_tails and engine.chain_ok(dims)      # the backward below is the dPre chain: the low-rank first layer may run
        if engine.chain_fwd_ok(dims):          # the decoder's first hyperTem layer rides on the encoder's last chain launch
            emb, c1, tidx, sv_e, dec_head = engine.model_fwd(p, src, mask, dims, base, mdl.num_route, mdl.scaler_zeros, gen=gen[engine.ENC], tidx=tidx,
                                                             dec_gen=gen[engine.DEC], lowrank_in=lowrank)
        else:
            emb, c1, tidx, sv_e = engine.model_fwd(p, src, mask, dims, base, mdl.num_route, mdl.scaler_zeros, gen=gen[engine.ENC], tidx=tidx,
                                                   lowrank_in=lowrank)
        if self.fused_tails:
            # output head + masked MAE + their backward: one pass over dec (the mean's 1/#kept is applied by the optimiser)
            _, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, mdl.num_route, gen=gen[engine.DEC], head=False, dec_head=dec_head)
            sws = self.arena.zeros(ops.tail_parts(M), 4)                       # per-workgroup loss statistics of the two heads
            chain = engine.chain_ok(dims)                                      # dPre chain: no backward kernel re-reads its layer's output

            def kl_path():
                dh2 = engine.kl_head(p, g, sv_g, prob, c1, self.N, 0.1, sws, red, chain=chain)
                engine.guide_bwd(p, g, src, tidx, sv_g, None, dims, base, red, dh2=dh2, chain=chain)
            out, dd = engine.loss_tail(p, g, dec, src, mask, base, self.std, self.mean, a.mape_thresh, sws, red, chain=chain)
            engine.model_bwd(p, g, src, mask, tidx, sv_e, sv_d, dec, None, None, dims, base, mdl.scaler_zeros, red, dd=dd, chain=chain)
            if phase == 1:
                kl_path()
            if self.dp is None and not self.global_count_scale:
                self._sws = sws                                                # folded by the optimiser's first launch (one launch less)
            else:
                ops.stats_fold(sws, self.stats)                                # ordered sum -> stats[0..2]: the all-reduce must see them
        else:
            out, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, mdl.num_route, gen=gen[engine.DEC], dec_head=dec_head)
            ops.mae_fwd(out, src, base + 2, mask, self.std, self.mean, a.mape_thresh, M, base, self.stats)
            d_out = ops.mae_bwd(out, src, base + 2, mask, self.std, self.mean, a.mape_thresh, M, base, self.stats, normalize=False)
            engine.model_bwd(p, g, src, mask, tidx, sv_e, sv_d, dec, d_out, None, dims, base, mdl.scaler_zeros, red)
            if phase == 1:
                dlogit = ops.kl(prob, c1, self.N, 0.1, self.stats)
                engine.guide_bwd(p, g, src, tidx, sv_g, dlogit, dims, base, red)
        red.flush(tidx)                                           # all parameter-gradient reductions: 3 launches
        engine.CTX.ARENA = None
        if self.dp is None:
            self._optim()
        elif self._dp_in_graph():           # gradient all-reduce + optimiser as the last nodes of the step's graph (no host gap behind the replay)
            if self._dec_reduced:           # the decoder bucket went out under the encoder's backward: what is left is [encoder] and [KL path | never | statistics]
                self._allreduce(self.gbuf[:self.dec_lo])
                self._allreduce(self.gbuf[self.dec_hi:])
            else:
                self._allreduce(self.gbuf)
            self._optim()

    def _mask_ws(self):
        """The selections' histogram scratch comes zeroed out of the step's arena (cleared by the step's first launch): no zeroing launch."""
        return self.arena.zeros(ops.mask_ws_floats())

    def _global_mask(self, phase, jobs=None):
        """Mask of this rank's rows cut out of the selection over the global batch (identical on every rank)."""
        a, base, M = self.args, self.base, self.B * self.T * self.N
        Mg = M * self.W
        if self.force_mask:
            return self.mask_buf
        if phase == 0:
            mask_g = ops.mask_random(self.noise_g, int(Mg * base * a.mask_ratio), ws=self._mask_ws(), u24=U24, jobs=jobs)
        else:                                              # label_g was gathered by _exchange_labels(); class histogram taken inside
            mask_g = ops.mask_adaptive(self.label_g, None, self.ctrl[:self.HS], self.ctrl[self.HS:], self.noise_a_g,
                                       self.noise_r_g, a.ada_type == "all", base, ws=self._mask_ws(), u24=U24, jobs=jobs)[2]
        self.last_mask_global = mask_g
        return self.dp.rows_of(mask_g, M * base)

    def _exchange_labels(self):
        """Adaptive phase under DP: ONE all-gather of the int32 cluster labels (261 KB per 32-sample rank), between the two parts."""
        self.dp.gather_labels(self.label_l, out=self.label_g)

    def _optim(self):
        sws, self._sws = getattr(self, "_sws", None), None
        ops.clip_adam(self.model.flat, self.gflat, self.m, self.v, self.model.nA, self.model.nB, self.hyper, self.stats,
                      stats_out=self.stats_out, sws=sws)

    def _body(self, phase):
        """eager: part 1 [-> label exchange] -> part 2 (+ optimiser when there is no gradient all-reduce in between)"""
        ops.set_deterministic(self.deterministic)           # thread-local launch mode of the library (captured into the graph)
        try:
            ctx = self._part1(phase)
            if self._needs_exchange(phase):
                self._exchange_labels()
            self._part2(phase, ctx)
        finally:
            ops.set_deterministic(False)

    # ---- host side of one step -------------------------------------------------------------------------------------
    def _views(self, hc):
        """numpy views of one pinned host-scalar record (written without a torch dispatch per element)"""
        HS = self.HS
        n = hc.numpy()
        return dict(hc=hc, hyper=n[:16].view("float32"), ctrl=n[16:16 + HS + 2], rng=n[16 + HS + 2:], ev=None)

    def _slot(self):
        """Next pinned slot of the ring; waits (host side) until the copies that last used it have run."""
        sl = self._ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % self.RING
        if sl["ev"] is not None:
            sl["ev"].synchronize()
        return sl

    def _host_prepare(self, phase, epoch, list_c):
        sl = self._slot()
        self._fill(sl, phase, epoch, list_c)
        self.hc.copy_(sl["hc"], non_blocking=True)
        if sl["ev"] is None:
            sl["ev"] = torch.cuda.Event()
        sl["ev"].record()

    def _fill(self, sl, phase, epoch, list_c):
        """Advance the optimiser counters by one step and write that step's host scalars into the pinned views of `sl`."""
        a = self.args
        self.tA += 1
        if phase == 1:
            self.tB += 1
        self.phase_kl = phase == 1
        b1, b2 = 0.9, 0.999
        tA, tB = self.tA, self.tB
        sl["hyper"][11:13] = (1 - b1, 1 - b2)           # as the host rounds them (torch passes python's 1 - beta): 1.f - 0.999f is 1.3e-5 low
        sl["hyper"][:11] = (self.lr / (1 - b1 ** tA), math.sqrt(1 - b2 ** tA),
                            self.lr / (1 - b1 ** tB) if tB else 0.0, math.sqrt(1 - b2 ** tB) if tB else 1.0,
                            b1, b2, 1e-8, float(a.max_grad_norm) if a.grad_norm else 0.0, 1.0 if phase == 1 else 0.0,
                            1.0,          # the backward carries the gradient of the SUM loss: the optimiser divides path A by the (global) kept count
                            1.0)
        sl["rng"][:2] = (self.noise_seed & 0x7FFFFFFF, tA & 0x7FFFFFFF)                      # same key on every rank: same global noise
        if phase == 1:
            if list_c is None:
                list_c = list(range(self.HS))
                self.rng.shuffle(list_c)                                   # GPTST.py:357-358
            ada, rnd = self.model.adaptive_counts(self.B * self.T * self.N * (self.W if self.gmask else 1), epoch)
            ada, rnd = self._budgets(ada, rnd, epoch)
            sl["ctrl"][:] = [int(v) for v in list_c] + [int(ada), int(rnd)]
        self._filled_list_c = [int(v) for v in list_c] if phase == 1 else None

    def _budgets(self, ada, rnd, epoch):
        """Hook: subclasses whose masks cover more cells than this rank's batch (node sharding) replace the budgets."""
        return ada, rnd

    def step(self, source, epoch, noise=None, noise_a=None, noise_r=None, list_c=None, forced_mask=None):
        """Enqueue one optimisation step on ``source`` (B,T,N,base+2).  Never synchronises.
        ``forced_mask`` (fp32, 1 = visible) teacher-forces the mask: used by loss-curve parity runs in the adaptive phase,
        where an fp32-level argmax flip of the cluster classifier would otherwise change which cells are masked."""
        phase = 0 if epoch <= self.args.change_epoch else 1
        self._g_last = None
        self._g_fallback = None
        if source is not self.src:
            self.src.copy_(source, non_blocking=True)
        inject = noise is not None or noise_a is not None
        if inject:                                      # with global masks the injected noise covers the GLOBAL batch
            n0, na, nr = (self.noise_g, self.noise_a_g, self.noise_r_g) if self.gmask else (self.noise, self.noise_a, self.noise_r)
            if phase == 0:
                n0.copy_(noise.reshape(-1), non_blocking=True)
            else:
                na.copy_(noise_a.reshape(-1), non_blocking=True)
                nr.copy_(noise_r.reshape(-1), non_blocking=True)
        if forced_mask is not None:
            self.mask_buf.copy_(forced_mask.reshape(-1), non_blocking=True)
        self._host_prepare(phase, epoch, list_c)
        self._last_call = (epoch, self._filled_list_c, self.rank_weight) if not inject and forced_mask is None else None
        self._unseen.append(phase)                      # steps enqueued since the host last looked at the statistics (losses())
        if len(self._unseen) > 4096:
            del self._unseen[:2048]
        key = (phase, inject, forced_mask is not None)
        if not self.use_graph:
            self.inject_noise, self.force_mask = inject, forced_mask is not None
            self._body(phase)
        else:
            if key not in self.graphs:
                self._capture(key)
            g1, g2 = self.graphs[key]
            g1.replay()
            if g2 is not None:                          # global masks, adaptive phase: labels are exchanged between the two graphs
                self._exchange_labels()
                g2.replay()
        if self.dp is not None and not self._dp_in_graph():
            self._allreduce(self.gbuf)
            self._optim()

    def _capture(self, key):
        """Capture the step of `key` = (phase, injected noise, forced mask).  If capturing the collectives of a data-parallel step fails
        (a runtime that cannot record RCCL kernels into a hipGraph raises at capture time, identically on every rank), the step falls back
        to collectives BETWEEN graph replays — the pre-round-3 form — instead of aborting the run."""
        try:
            return self._capture_impl(key)
        except Exception as e:                                    # noqa: BLE001
            if not self._dp_in_graph():
                raise
            import sys
            print("gpt-st_amd: capturing the step with its collectives failed (%s: %s) -> collectives between graph replays"
                  % (type(e).__name__, str(e).splitlines()[0][:200]), file=sys.stderr)
            self._graph_comm_failed = True
            self.graphs.clear()                         # graphs captured earlier hold the in-graph all-reduce + optimiser: replaying them next to
            getattr(self, "_g_graphs", {}).clear()      # the eager collectives would reduce and step twice
            torch.cuda.synchronize()
            return self._capture_impl(key)

    def _capture_impl(self, key):
        phase, inject, forced = key
        self.inject_noise, self.force_mask = inject, forced
        keep = (self.model.flat.clone(), self.m.clone(), self.v.clone())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                      # warm-up on a side stream (allocator, lazy kernel attributes)
            for _ in range(2):
                self._body(phase)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ops.set_deterministic(self.deterministic)
        try:
            g1 = torch.cuda.CUDAGraph()
            if self._needs_exchange(phase) and not self._dp_in_graph():
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):      # other threads (RCCL watchdog) may call the runtime meanwhile
                    ctx = self._part1(phase)
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                    self._part2(phase, ctx)
                self._ctx_keep = getattr(self, "_ctx_keep", []) + [ctx]            # part 1's tensors are inputs of graph 2
            else:
                g2 = None
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                    ctx = self._part1(phase)
                    if self._needs_exchange(phase):                                # capturable communicator: the label gather is a graph node
                        self._exchange_labels()
                    self._part2(phase, ctx)
        finally:
            ops.set_deterministic(False)
            engine.CTX.ARENA = None
            self.model.flat.copy_(keep[0]); self.m.copy_(keep[1]); self.v.copy_(keep[2])   # undo the warm-up updates (also when the capture failed)
            torch.cuda.synchronize()
        self.graphs[key] = (g1, g2)

    # ---- several steps per graph replay -----------------------------------------------------------------------------
    # Between two replays of the step graph the device idles for ~19 us (tools/timeline.py on profiles/r03z: 8.8 us from the optimiser's last
    # kernel to the H2D copy of the next step's host scalars, 4.1 us for that copy, 5.8 us until the graph's first kernel) — 1.3 % of a
    # 1.42 ms step.  step_group() enqueues K consecutive optimisation steps (K batches, K optimiser updates, each with its own host
    # scalars, source buffer and statistics snapshot) as ONE graph replay behind ONE H2D copy: the reference's batch loop
    # (BasicTrainer.py:72-103) unrolled K times.  Results are those of K step() calls (tests/test_gpu_step.py).
    def group_ok(self, epoch):
        """A group runs as one graph when the whole step is one graph (no host-side collective inside the step)."""
        phase = 0 if epoch <= self.args.change_epoch else 1
        if self._group_failed:
            return False
        return self.use_graph and (self.dp is None or self._dp_in_graph()) and not (self._needs_exchange(phase) and not self._dp_in_graph())

    def _group_init(self, K):
        if self._gK == K:
            return
        W = self.hc.numel()
        self._gK = K
        self._g_hc = torch.zeros(K, W, dtype=torch.int32, device=self.dev)
        self._g_src = [torch.zeros_like(self.src) for _ in range(K)]
        self._g_stats = torch.zeros(K, 8, device=self.dev)
        self._g_ring = []
        for _ in range(self.RING):
            hc = torch.zeros(K, W, dtype=torch.int32).pin_memory()
            self._g_ring.append(dict(hc=hc, rows=[self._views(hc[j]) for j in range(K)], ev=None))
        self._g_ring_i = 0
        self._g_graphs = {}

    def group_sources(self, K):
        """The K device input buffers (B,T,N,base+2) of a group: fill them (or pass other tensors to step_group, which copies)."""
        self._group_init(K)
        return self._g_src

    def _sub(self, j):
        """Point the per-step buffers of the body at sub-step j of the group (host scalars, source, statistics snapshot)."""
        HS = self.HS
        r = self._g_hc[j]
        self.src, self.stats_out = self._g_src[j], self._g_stats[j]
        self.hyper, self.ctrl, self.rng_words = r[:16].view(torch.float32), r[16:16 + HS + 2], r[16 + HS + 2:]

    def step_group(self, sources, epoch, list_cs=None):
        """Enqueue len(sources) consecutive optimisation steps of the same epoch.  Never synchronises.  Falls back to a loop over step()
        where a step is not one graph (eager mode, host-side collectives).  losses_group() returns the K loss triples."""
        K = len(sources)
        if K == 1 or not self.group_ok(epoch):
            snaps = []                                  # every step's statistics (device copies, no sync): losses_group() returns K triples here too
            for j, src in enumerate(sources):
                self.step(src, epoch, list_c=None if list_cs is None else list_cs[j])
                snaps.append((self.stats_out.clone(), bool(self.tB and self.phase_kl), src, self._filled_list_c, epoch))
            self._g_last = None
            self._g_fallback = snaps
            return
        phase = 0 if epoch <= self.args.change_epoch else 1
        self._group_init(K)
        for j, src in enumerate(sources):
            if src is not self._g_src[j]:
                self._g_src[j].copy_(src, non_blocking=True)
        if phase not in self._g_graphs:
            # capture BEFORE any per-step state is advanced: if the runtime cannot record this group (e.g. collectives of K steps in one
            # graph), every rank fails alike, the group falls back to K single steps and later groups do not try again
            keep = (self.src, self.stats_out, self.hyper, self.ctrl, self.rng_words)
            try:
                self._capture_group(phase, K, epoch)
            except Exception as e:                                    # noqa: BLE001
                import sys
                print("gpt-st_amd: capturing %d steps in one graph failed (%s: %s) -> one replay per step"
                      % (K, type(e).__name__, str(e).splitlines()[0][:200] if str(e) else ""), file=sys.stderr)
                self._group_failed = True
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            finally:
                self.src, self.stats_out, self.hyper, self.ctrl, self.rng_words = keep
            if self.dp is not None and getattr(self.dp, "world", 1) > 1 and hasattr(self.dp, "max_over_ranks"):
                # ... and so is the outcome of the capture itself: one rank that could not record the group takes every rank to single steps
                if self.dp.max_over_ranks(1.0 if self._group_failed else 0.0) > 0 and not self._group_failed:
                    self._group_failed = True
                    self._g_graphs.pop(phase, None)
            if self._group_failed:
                return self.step_group(sources, epoch, list_cs)
        sl = self._g_ring[self._g_ring_i]
        self._g_ring_i = (self._g_ring_i + 1) % self.RING
        if sl["ev"] is not None:
            sl["ev"].synchronize()
        self._g_list_cs = []
        for j in range(K):
            self._fill(sl["rows"][j], phase, epoch, None if list_cs is None else list_cs[j])
            self._g_list_cs.append(self._filled_list_c)
        self._g_epoch = epoch
        self._g_hc.copy_(sl["hc"], non_blocking=True)
        if sl["ev"] is None:
            sl["ev"] = torch.cuda.Event()
        sl["ev"].record()
        self._g_graphs[phase].replay()
        self._g_last = (K, phase)

    def _capture_group(self, phase, K, epoch):
        self.inject_noise, self.force_mask = False, False
        keep = (self.model.flat.clone(), self.m.clone(), self.v.clone())
        # the warm-up runs need plausible host scalars in the device table: fill it as the first group would, then put the counters back
        st = (self.tA, self.tB, self.phase_kl, self.rng.getstate())
        for j in range(K):
            self._fill(self._g_ring[0]["rows"][j], phase, epoch, None)
        self._g_hc.copy_(self._g_ring[0]["hc"])
        torch.cuda.synchronize()
        self.tA, self.tB, self.phase_kl = st[:3]
        self.rng.setstate(st[3])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        with torch.cuda.stream(s):                      # warm-up on a side stream (allocator, lazy kernel attributes)
            for j in range(min(K, 2)):
                self._sub(j)
                self._body(phase)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # Activations of the K sub-steps are NOT shared inside one capture (the graph-private pool grew K-fold at N = 4096, C = 128:
        # 4 x 56 GB): groups are for the shapes where the idle time between replays matters, i.e. small steps
        # (decided from the device's TOTAL memory, not from what happens to be free: every rank of a data-parallel job must take the same
        # branch, or their collective sequences diverge)
        per_step = torch.cuda.max_memory_allocated() - base
        total = torch.cuda.get_device_properties(self.dev).total_memory
        if self.dp is not None and getattr(self.dp, "world", 1) > 1 and hasattr(self.dp, "max_over_ranks"):
            # the group / no-group decision is COLLECTIVE: the largest per-step peak any rank measured and the smallest device decide
            # for all of them — ranks whose allocators differ must not take different branches (their collective sequences would diverge)
            per_step = int(self.dp.max_over_ranks(per_step))
            total = -int(self.dp.max_over_ranks(-total))
        if per_step * K > 0.25 * total:
            self.model.flat.copy_(keep[0]); self.m.copy_(keep[1]); self.v.copy_(keep[2])
            torch.cuda.synchronize()
            raise RuntimeError("a group of %d steps would need ~%.0f GB of activations (device: %.0f GB)" % (K, per_step * K / 2**30, total / 2**30))
        ops.set_deterministic(self.deterministic)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for j in range(K):
                    self._sub(j)
                    ctx = self._part1(phase)
                    if self._needs_exchange(phase):
                        self._exchange_labels()
                    self._part2(phase, ctx)
        finally:
            ops.set_deterministic(False)
            engine.CTX.ARENA = None
            self.model.flat.copy_(keep[0]); self.m.copy_(keep[1]); self.v.copy_(keep[2])   # undo the warm-up updates
            torch.cuda.synchronize()
        self._g_graphs[phase] = g

    def losses_group(self):
        """[(loss, loss_flow, loss_s)] of the steps of the last step_group() — synchronises."""
        if self._g_last is None:
            if getattr(self, "_g_fallback", None):      # step_group() fell back to single steps: one triple per step, as the grouped path
                snaps, self._g_fallback = self._g_fallback, None
                rows = [(st.cpu(), kl) for st, kl, _, _, _ in snaps]
                out = [self._stats_row(st, kl) for st, kl in rows]
                lost = [j for j, (st, _) in enumerate(rows) if float(st[5]) > 0]
                del self._unseen[:]
                if lost:                                # a hand-off expired in step j0: its update and every later one were skipped —
                    j0 = lost[0]                        # take them back and re-run them, as the grouped path does
                    self._enter_safe_mode(len(snaps) - j0, sum(1 for _, kl, _, _, _ in snaps[j0:] if kl))
                    for j in range(j0, len(snaps)):
                        _, kl, src, lc, ep = snaps[j]
                        self.step(src, ep, list_c=lc)
                        out[j] = self._stats_row(self.stats_out.cpu(), kl)
                    del self._unseen[:]
                return out
            return [self.losses()]
        K, phase = self._g_last
        st = self._g_stats.cpu()
        del self._unseen[:]
        rerun = {}
        if float(st[:, 5].max()) > 0:           # a hand-off expired in sub-step j0: its update and every later one were skipped
            j0 = int((st[:, 5] > 0).float().argmax())
            # (groups enqueued BEFORE this one without a look in between — bench loops — were skipped too if the expiry is older: the device's count of
            #  skipped updates says how many; their batches are gone, their optimiser steps are taken back with this group's)
            extra = max(0, int(st[K - 1, 6]) - (K - j0))
            if extra:
                import sys
                self.lost_batches += extra
                print("gpt-st_amd: %d step(s) of earlier groups were skipped too and cannot be repeated (their batches are gone): the optimiser counters "
                      "were taken back, the batches were not trained on" % extra, file=sys.stderr)
            self._enter_safe_mode(K - j0 + extra, (K - j0 + extra) if phase == 1 else 0)
            srcs, lcs, epoch = [t.clone() for t in self._g_src[j0:]], self._g_list_cs[j0:], self._g_epoch
            for j, (src, lc) in enumerate(zip(srcs, lcs)):
                self.step(src, epoch, list_c=lc)
                rerun[j0 + j] = self._stats_row(self.stats_out.cpu(), phase == 1)
            del self._unseen[:]
            self._g_last = (K, phase)            # (step() cleared it: the group's triples stay readable)
        out = []
        for j in range(K):
            out.append(rerun[j] if j in rerun else self._stats_row(st[j], phase == 1))
        return out

    @staticmethod
    def _stats_row(st, kl):
        lf = float(st[0] / max(float(st[1]), 1.0))
        ls = float(st[2]) * 0.1 if kl else 0.0
        return (lf + ls, lf, ls)

    def _enter_safe_mode(self, nsteps, nkl):
        """A hand-off expiry is on record: the last `nsteps` optimiser updates (nkl of them with the KL path) were skipped on the device.  Rewind the
        host's step counters, drop every captured graph (they hold the hand-off launches), clear the record, and from now on enqueue the step
        without in-launch hand-offs.  Collective under data parallelism by construction: the count travels in the all-reduced statistics."""
        import sys
        from . import _C
        torch.cuda.synchronize()
        print("gpt-st_amd: an in-launch hand-off expired (the GPU was shared?) — %d skipped step(s) are re-run without hand-off launches; "
              "this stepper stays in that mode" % nsteps, file=sys.stderr)
        self.tA -= nsteps
        self.tB -= nkl
        self.lost_steps += nsteps
        self.safe_mode = True
        self.graphs.clear()
        getattr(self, "_g_graphs", {}).clear()
        _C.lib().call("gptst_handoff_reset")

    # ---- results ---------------------------------------------------------------------------------------------------
    def losses(self):
        """(loss, loss_flow, loss_s) of the last step — synchronises (reference BasicTrainer.py:98-103 does so every step)."""
        if self._g_last is not None:
            return self.losses_group()[-1]
        st = self.stats_out.cpu()
        unseen, self._unseen = self._unseen, []
        if float(st[5]) > 0:                     # the update of this step was skipped (a hand-off expired): re-run it from the untouched weights
            if self._last_call is None:
                raise RuntimeError("an in-launch hand-off expired in a step with injected mask inputs: call _enter_safe_mode() and repeat the step")
            # The guard skips EVERY update while the record is up, and the host may have enqueued several steps since it last looked (bench loops): the
            # device counts them (stats_out[6]).  All of them are taken back; only the last one can be repeated — the earlier batches are gone.
            nskip = max(1, min(int(st[6]), len(unseen))) if unseen else 1
            epoch, lc, weight = self._last_call
            self._enter_safe_mode(nskip, sum(unseen[-nskip:]) if unseen else (1 if self.phase_kl else 0))
            if nskip > 1:
                import sys
                self.lost_batches += nskip - 1
                print("gpt-st_amd: %d step(s) enqueued before the last one were skipped with it and cannot be repeated (their batches are gone): "
                      "the optimiser counters were taken back, the batches were not trained on" % (nskip - 1), file=sys.stderr)
            keep_w, self.rank_weight = self.rank_weight, weight          # the same weighted step (a padding rank of a tail round re-runs as padding)
            try:
                self.step(self.src, epoch, list_c=lc)
            finally:
                self.rank_weight = keep_w
            st = self.stats_out.cpu()
            del self._unseen[:]
        return self._stats_row(st, bool(self.tB and self.phase_kl))
