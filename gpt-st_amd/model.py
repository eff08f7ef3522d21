"""Drop-in ``GPTST_Model`` for ``-mode pretrain`` (reference model/Pretrain_model/GPTST.py:459-493).

Same constructor (an ``args`` namespace), same ``forward(source, label, batch_seen=None, epoch=None)`` 5-tuple, same
``state_dict`` keys / shapes / order and — on CPU construction — bit-identical initial values for a given seed, because the
modules register the same parameters with the same initialisers in the same order (including the throw-away ``hyperguide1``
draw, GPTST.py:305).  All compute runs in the HIP kernels (engine.py); autograd sees one Function for the whole network.
Parameters are views into one flat fp32 buffer laid out [reconstruction path | KL path | never-trained] for the fused
clip+Adam kernel.  There is no CPU execution path: calling forward on CPU tensors raises.
"""
import random
from collections import OrderedDict

import torch
import torch.nn as nn

from . import engine, module_graph, ops


# --------------------------------------------------------------------------------------------------------------------
# module tree (parameter containers only — forward lives in engine.py)
# --------------------------------------------------------------------------------------------------------------------
class TimeFeature(nn.Module):                      # GPTST.py:187-196 / 204-213
    def __init__(self, embed_dim, in_dim=1):
        super().__init__()
        self.ln_day = nn.Linear(in_dim, embed_dim)
        self.ln_week = nn.Linear(in_dim, embed_dim)
        self.ln1 = nn.Linear(embed_dim, embed_dim)
        self.ln2 = nn.Linear(embed_dim, embed_dim)
        self.ln = nn.Linear(embed_dim, embed_dim)


class HyperTem(nn.Module):                         # GPTST.py:144-152
    def __init__(self, timesteps, dim_in, dim_out, embed_dim, HT_Tem):
        super().__init__()
        self.adj = nn.Parameter(torch.randn(embed_dim, HT_Tem, timesteps), requires_grad=True)
        self.weights_pool = nn.Parameter(torch.FloatTensor(embed_dim, dim_in, dim_out))
        self.bias_pool = nn.Parameter(torch.FloatTensor(embed_dim, dim_out))


class Cap(nn.Module):                              # GPTST.py:79-98
    def __init__(self, dim, num_nodes, timesteps, embed_dim, embed_dim_spa, HS, HT):
        super().__init__()
        self.ln_p = nn.Linear(dim, dim)
        self.t_adj = nn.Parameter(torch.randn(embed_dim_spa, HT, HS * timesteps), requires_grad=True)
        self.adj = nn.Parameter(torch.randn(embed_dim_spa, HS, num_nodes), requires_grad=True)
        self.weights_spa = nn.Parameter(torch.FloatTensor(embed_dim, dim, dim))
        self.bias_spa = nn.Parameter(torch.FloatTensor(embed_dim, dim))
        self.register_buffer("mask_template", torch.linspace(1, timesteps, steps=timesteps) / 12.0)


class MlpRl(nn.Module):                            # GPTST.py:6-19
    def __init__(self, dim_in, dim_out, hidden_dim, embed_dim):
        super().__init__()
        self.ln1 = nn.Linear(dim_in, hidden_dim)
        self.ln3 = nn.Linear(hidden_dim, dim_out)
        self.weights_pool_spa = nn.Parameter(torch.FloatTensor(embed_dim, hidden_dim, hidden_dim))
        self.bias_pool_spa = nn.Parameter(torch.FloatTensor(embed_dim, hidden_dim))
        self.weights_pool_tem = nn.Parameter(torch.FloatTensor(embed_dim, hidden_dim, hidden_dim))
        self.bias_pool_tem = nn.Parameter(torch.FloatTensor(embed_dim, hidden_dim))


class STHCN(nn.Module):                            # GPTST.py:221-251
    def __init__(self, a):
        super().__init__()
        N, C, d, ds, T = a.num_nodes, a.hidden_dim, a.embed_dim, a.embed_dim_spa, a.horizon
        self.node_embeddings = nn.Parameter(torch.randn(N, d), requires_grad=True)
        self.node_embeddings_spg = nn.Parameter(torch.randn(N, d), requires_grad=True)
        self.hyperTem1 = HyperTem(T, C, C, d, a.HT_Tem)
        self.hyperTem2 = HyperTem(T, C, C, d, a.HT_Tem)
        self.hyperTem3 = HyperTem(T, C, C, d, a.HT_Tem)
        self.hyperTem4 = HyperTem(T, C, C, d, a.HT_Tem)
        self.time_feature1 = TimeFeature(d)
        self.time_feature1_ = TimeFeature(ds)
        self.time_feature2 = TimeFeature(ds, in_dim=12)
        self.cap1 = Cap(C, N, T, d, ds, a.HS, a.HT)
        self.cap2 = Cap(C, N, T, d, ds, a.HS, a.HT)


class HypergraphEncoder(nn.Module):                # GPTST.py:276-310
    def __init__(self, a):
        super().__init__()
        self.dim_in_flow = nn.Linear(a.input_base_dim, a.hidden_dim, bias=True)
        self.STHCN_encode = STHCN(a)
        torch.randn(a.hidden_dim, a.lag, a.HS, a.num_nodes)     # reference's unused hyperguide1 consumes the RNG (:305)
        self.MLP_RL = MlpRl(a.input_base_dim, a.HS, a.hidden_dim, a.embed_dim)
        self.teb4mask = TimeFeature(a.embed_dim)
        self.neb4mask = nn.Parameter(torch.randn(a.num_nodes, a.embed_dim), requires_grad=True)


class HypergraphDecoder(nn.Module):                # GPTST.py:429-451
    def __init__(self, a):
        super().__init__()
        self.time_feature1_ = TimeFeature(a.embed_dim_spa)      # never used in forward, part of the checkpoint
        self.time_feature2_ = TimeFeature(a.embed_dim_spa)
        self.STHCN_decode = STHCN(a)
        self.dim_flow_out = nn.Linear(a.hidden_dim, a.input_base_dim, bias=True)


def _segment(key):
    if key.startswith("encoder.MLP_RL.") or key.startswith("encoder.teb4mask.") or key == "encoder.neb4mask":
        return 1          # KL path only (gradients appear after change_epoch)
    if key.startswith("decoder.time_feature1_.") or key.startswith("decoder.time_feature2_."):
        return 2          # never trained
    return 0


class _PretrainFn(torch.autograd.Function):
    """Whole-network autograd node: forward/backward are the hand-written HIP pipelines of engine.py."""

    @staticmethod
    def forward(ctx, model, source, mask, pre, *params):
        p = model.param_views()
        dims, base = model._dims(source), model.input_base_dim
        # pre: (tidx, gen, prob, sv_g) of forward_pretrain — the generated parameters and the guide classifier's forward, which the mask
        # needed first, are computed ONCE per call (r03 review: the guide ran twice)
        tidx, gen, prob, sv_g = pre
        emb, c1, tidx, sv_e = engine.model_fwd(p, source, mask, dims, base, model.num_route, model.scaler_zeros, gen=gen[engine.ENC], tidx=tidx)
        out, dec, sv_d = engine.decoder_fwd(p, tidx, emb, dims, model.num_route, gen=gen[engine.DEC])
        ctx.model, ctx.saved = model, (source, mask, tidx, sv_g, sv_e, sv_d, dec, prob, dims)
        B, T, N, C = dims
        ctx.mark_non_differentiable(c1)
        return out.view(B, T, N, base), dec.view(B, T, N, C), prob.view(B, T, N, -1), c1

    @staticmethod
    def backward(ctx, d_out, d_dec, d_prob, _dc):
        model = ctx.model
        source, mask, tidx, sv_g, sv_e, sv_d, dec, prob, dims = ctx.saved
        B, T, N, C = dims
        base = model.input_base_dim
        p = model.param_views()
        gflat = model._grad_buffer()
        model._last_gflat = gflat                     # (optim.ClipAdam: the flat buffer behind the .grad views autograd is about to keep)
        g = model.views_of(gflat)
        d_out = d_out.contiguous().view(-1, base)
        d_dec2 = None if d_dec is None or not bool(d_dec.any()) else d_dec.contiguous().view(-1, C)
        red = engine.Reductions()
        engine.model_bwd(p, g, source, mask, tidx, sv_e, sv_d, dec, d_out, d_dec2, dims, base, model.scaler_zeros, red)
        has_kl = d_prob is not None and bool(d_prob.any())
        if has_kl:      # softmax backward: dlogit = prob * (d_prob - sum(d_prob * prob))
            dp = d_prob.contiguous().view(-1, prob.shape[1])
            dlogit = (prob * (dp - (dp * prob).sum(-1, keepdim=True))).contiguous()
            engine.guide_bwd(p, g, source, tidx, sv_g, dlogit, dims, base, red)
        red.flush(tidx)
        grads = []
        for k in model.param_keys:
            seg = _segment(k)
            grads.append(g[k] if seg == 0 or (seg == 1 and has_kl) else None)
        return (None, None, None, None) + tuple(grads)


class GPTST_Model(nn.Module):
    _named = None                                # [(key, Parameter)] in module order, walked once per _flatten()

    def __init__(self, args):
        super().__init__()
        self.num_node = args.num_nodes
        self.input_base_dim = args.input_base_dim
        self.input_extra_dim = args.input_extra_dim
        self.hidden_dim = args.hidden_dim
        self.output_dim = args.output_dim
        self.horizon = args.horizon
        self.embed_dim = args.embed_dim
        self.embed_dim_spa = args.embed_dim_spa
        self.HS, self.HT, self.HT_Tem = args.HS, args.HT, args.HT_Tem
        self.num_route = args.num_route
        self.mode = args.mode
        self.model = getattr(args, "model", None)
        self.scaler_zeros = float(getattr(args, "scaler_zeros", 0.0))
        self.mask_ratio, self.ada_mask_ratio, self.ada_type = args.mask_ratio, args.ada_mask_ratio, args.ada_type
        self.change_epoch, self.epochs = args.change_epoch, args.epochs
        self.encoder = HypergraphEncoder(args)
        self.decoder = HypergraphDecoder(args)
        self.param_keys = [k for k, _ in self.named_parameters()]
        self.flat = None
        self._views = None
        self._inject = None
        self._flatten()

    # ---- flat parameter storage ------------------------------------------------------------------------------------
    def _flatten(self):
        """(Re)build the flat buffer on the parameters' current device; segment order [path A | KL path | never]."""
        self._named = None                       # (walk the module tree afresh)
        named = OrderedDict(self.named_parameters())
        dev = next(iter(named.values())).device
        order = sorted(named.keys(), key=lambda k: (_segment(k), self.param_keys.index(k)))
        offs, n, seg_end = {}, 0, [0, 0, 0]
        for k in order:
            offs[k] = n
            n += (named[k].numel() + 3) // 4 * 4          # 16-byte aligned tensors
            seg_end[_segment(k)] = n
        flat = torch.zeros(n, device=dev, dtype=torch.float32)
        for k in order:
            t = named[k]
            flat[offs[k]:offs[k] + t.numel()].copy_(t.detach().reshape(-1))
            t.data = flat[offs[k]:offs[k] + t.numel()].view(t.shape)
        self.flat, self._offs = flat, offs
        import weakref
        GPTST_Model._OWNERS[flat.untyped_storage().data_ptr()] = weakref.ref(self)
        self.nA, self.nB = seg_end[0], seg_end[1] - seg_end[0]
        self._views = None
        # the parameter objects in module order, walked ONCE: named_parameters() traverses ~640 modules, and the eager path asked for it four times per
        # step (1.2 ms of the reference-style loop's 6.7 ms, tools/experiments/prof_module_path.py)
        self._named = list(named.items())
        self._vmeta = None
        self._shapes = {k: (t.numel(), t.shape) for k, t in self._named}

    _OWNERS = {}                                 # address of a flat parameter buffer -> weakref of its model (optim.ClipAdam finds the layout by it)

    @staticmethod
    def owner_of(param):
        """the GPTST_Model whose flat buffer `param` is a view of (None: not one of ours)"""
        base = param._base if param._base is not None else param
        ref = GPTST_Model._OWNERS.get(base.untyped_storage().data_ptr())
        m = ref() if ref is not None else None
        return m if m is not None and m.flat.untyped_storage().data_ptr() == base.untyped_storage().data_ptr() else None

    # The reference's loop asks for model.parameters() every step (clip_grad_norm_, BasicTrainer.py:95-96): nn.Module walks ~640 modules for it,
    # 0.35 ms of a 3.8 ms step (tools/experiments/prof_module_path.py).  The module tree is fixed after construction: serve the cached list.
    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        if self._named is not None and recurse and prefix == "" and remove_duplicate:
            return iter(self._named)
        return super().named_parameters(prefix=prefix, recurse=recurse, remove_duplicate=remove_duplicate)

    def parameters(self, recurse=True):
        if self._named is not None and recurse:
            return iter([t for _, t in self._named])
        return super().parameters(recurse=recurse)

    def _apply(self, fn, *a, **k):               # .to(device) / .cuda(): re-establish the flat views afterwards
        super()._apply(fn, *a, **k)
        self._flatten()
        return self

    def load_state_dict(self, sd, strict=True):
        r = super().load_state_dict(sd, strict)  # copy_ into the views keeps the flat layout
        self._views = None
        return r

    def _grad_buffer(self):
        """A flat gradient buffer for ONE backward node of the autograd path, zeroed.  Fresh per call: the node returns views of it, and autograd
        may still hold those views (a second forward of the same model inside one backward pass, tensors returned by torch.autograd.grad) when the
        next node runs — a reused buffer was overwritten under them.  The caching allocator makes this one memset, as zero_() was."""
        return torch.zeros_like(self.flat)

    def views_of(self, flat):
        """{key: view of `flat` with the parameter's shape} — one as_strided per tensor (a slice + a view were two dispatches each: 310 per backward)"""
        if self._vmeta is None:
            self._vmeta = []
            for k, o in self._offs.items():
                shp = tuple(self._shapes[k][1])
                st, acc = [], 1
                for d in reversed(shp):
                    st.append(acc); acc *= d
                self._vmeta.append((k, shp, tuple(reversed(st)), o))
        a = torch.as_strided
        return {k: a(flat, shp, st, o) for k, shp, st, o in self._vmeta}

    def param_views(self):
        k0, t0 = self._named[0]
        if self._views is None or self._views[k0].data_ptr() != t0.data_ptr():
            v = {k: t.data for k, t in self._named}
            for k, b in self.named_buffers():
                v[k] = b
            self._views = v
        return self._views

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _dims(self, source):
        B, T, N, _ = source.shape
        return (B, T, N, self.hidden_dim)

    def _tidx(self, source):
        b = self.input_base_dim
        return source[:, :, 0, b:b + 2].contiguous()

    def set_mask_inputs(self, noise=None, noise_a=None, noise_r=None, list_c=None, forced_mask=None):
        """Inject the random inputs of mask generation for the next forward (the reference draws them from the device /
        python RNG, GPTST.py:316,358,389,400).  Without injection they are drawn with torch.rand / random.shuffle."""
        self._inject = dict(noise=noise, noise_a=noise_a, noise_r=noise_r, list_c=list_c, forced_mask=forced_mask)

    def adaptive_counts(self, numel_btn, epoch):
        tp = ((epoch - self.change_epoch) / (self.epochs - self.change_epoch)) * self.ada_mask_ratio     # :348-350
        if tp > 1:
            tp = 1
        total = int(numel_btn * self.mask_ratio)                                                         # :351
        ada = int(total * tp)                                                                            # :352
        return ada, total - ada

    def make_mask(self, source, prob, epoch):
        """fp32 visibility mask (B*T*N*base,), 1 = visible — GPTST.py:314-323 / 344-413, all on the device."""
        B, T, N, _ = source.shape
        base, dev = self.input_base_dim, source.device
        inj = self._inject or {}
        self._inject = None
        if inj.get("forced_mask") is not None:
            return inj["forced_mask"].to(dev, torch.float32).reshape(-1).contiguous()
        M = B * T * N
        if epoch <= self.change_epoch:
            noise = inj.get("noise")
            noise = torch.rand(M * base, device=dev) if noise is None else noise.to(dev).reshape(-1).contiguous()
            return ops.mask_random(noise, int(M * base * self.mask_ratio))
        label, counts = ops.mask_labels(prob.reshape(M, -1))
        ada, rnd = self.adaptive_counts(M, epoch)
        list_c = inj.get("list_c")
        if list_c is None:
            list_c = list(range(self.HS))
            random.shuffle(list_c)                                                                       # :357-358
        na, nr = inj.get("noise_a"), inj.get("noise_r")
        na = torch.rand(M, device=dev) if na is None else na.to(dev).reshape(-1).contiguous()
        nr = torch.rand(M, device=dev) if nr is None else nr.to(dev).reshape(-1).contiguous()
        lc = torch.tensor(list(list_c), dtype=torch.int32, device=dev)
        nums = torch.tensor([ada, rnd], dtype=torch.int32, device=dev)
        return ops.mask_adaptive(label, counts, lc, nums, na, nr, self.ada_type == "all", base)[2]

    # ---- reference API ---------------------------------------------------------------------------------------------
    def forward_pretrain(self, source, label, batch_seen=None, epoch=None):
        if not source.is_cuda:
            raise RuntimeError("gpt-st_amd runs on MI355X only (no CPU fallback): move the model and inputs to a GPU")
        source = source.contiguous().float()
        B, T, N, _ = source.shape
        base = self.input_base_dim
        # r06: the training loop's call — gradients on, nothing injected, static shape — replays the captured forward; its backward is a second graph
        # (module_graph.py).  Anything else (no_grad, injected mask inputs, a forward while the previous one still waits for its backward) runs eagerly.
        if module_graph.ENABLED and torch.is_grad_enabled() and self._inject is None and epoch is not None and self.hidden_dim == 64 \
                and not torch.cuda.is_current_stream_capturing():
            key = (tuple(source.shape), 0 if epoch <= self.change_epoch else 1, self.flat.data_ptr())
            gp = module_graph.graphs_of(self, key)          # (kept outside the module: copy.deepcopy(model), BasicTrainer.py:180, must not meet a hipGraph)
            if not gp.busy:
                return gp.forward(source, epoch)
        p = self.param_views()
        with torch.no_grad():
            # the mask depends on the guide probabilities only through argmax (no gradient), so the guide classifier runs first — once: its
            # outputs and saved activations are handed to the autograd node together with every generated parameter of the step
            dims, tidx = self._dims(source), self._tidx(source)
            gen = engine.gen_all(p, tidx, dims)
            prob0, sv_g = engine.guide_fwd(p, source, tidx, dims, base, gen=gen["guide"])
            mask = self.make_mask(source, prob0, epoch)
        params = [t for _, t in self._named]
        out, dec, prob, c1 = _PretrainFn.apply(self, source, mask, (tidx, gen, prob0, sv_g), *params)
        mask_i = mask.view(B, T, N, base).to(torch.int64)
        hs1 = c1.view(B, T, self.HS, N).transpose(-1, -2)                                                  # :424
        return out, dec, 1 - mask_i, prob, hs1

    def forward_fune(self, source, label):
        source = source.contiguous().float()
        with torch.no_grad():
            emb, _, _, _ = engine.model_fwd(self.param_views(), source, None, self._dims(source), self.input_base_dim,
                                            self.num_route, self.scaler_zeros)
        B, T, N, _ = source.shape
        e = emb.view(B, T, N, self.hidden_dim)
        return e, e, e, e, e

    def forward(self, source, label, batch_seen=None, epoch=None):
        if self.mode == "pretrain":
            return self.forward_pretrain(source, label, batch_seen, epoch)
        return self.forward_fune(source, label)


def xavier_init_(model):
    """Reference Run.py:79-85: every parameter with dim > 1 -> xavier_uniform_, else uniform_(0,1)."""
    for p_ in model.parameters():
        if p_.requires_grad:
            if p_.dim() > 1:
                nn.init.xavier_uniform_(p_)
            else:
                nn.init.uniform_(p_)
    return model


def init_seed(seed):
    """Reference lib/TrainInits.py:5-16."""
    import numpy as np
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)
