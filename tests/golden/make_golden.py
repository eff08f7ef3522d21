"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py

The reference's model/Pretrain_model/GPTST.py hard-codes 'cuda:0'; it is loaded here with
the in-memory substitution 'cuda:0' -> 'cpu'  — nothing of the reference is
copied into the repo; only inputs and its outputs (data) are saved.  The fixtures pin
oracle/gptst_oracle.py (tests/test_oracle_golden.py); the GPU parity tests then compare the
HIP path with the pinned oracle.
"""
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from gptst_amd.config import make_args          # noqa: E402
from gptst_amd import synth                     # noqa: E402


def load_reference():
    src = open(os.path.join(REF, "model/Pretrain_model/GPTST.py")).read().replace("'cuda:0'", "'cpu'")
    m = types.ModuleType("gptst_ref")
    exec(compile(src, "gptst_ref", "exec"), m.__dict__)
    return m


ref = load_reference()
from lib.metrics import MAE_torch               # noqa: E402  (reference lib, read-only import)
from lib.normalization import StandardScaler    # noqa: E402
from lib.TrainInits import init_seed            # noqa: E402


class Recorder:
    """Records what the reference draws from torch.rand_like / random.shuffle during a call."""

    def __enter__(self):
        self.noise, self.orders = [], []
        self._rl, self._sh = torch.rand_like, random.shuffle

        def rl(*a, **k):
            out = self._rl(*a, **k)
            self.noise.append(out.detach().clone())
            return out

        def sh(lst, *a, **k):
            self._sh(lst, *a, **k)
            self.orders.append(list(lst))

        torch.rand_like, random.shuffle = rl, sh
        return self

    def __exit__(self, *exc):
        torch.rand_like, random.shuffle = self._rl, self._sh


def sd_hash(sd):
    import hashlib
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode()); h.update(v.detach().numpy().tobytes())
    return h.hexdigest()


def xavier(params):
    for p in params:                                  # reference Run.py:79-85
        if p.requires_grad:
            if p.dim() > 1:
                torch.nn.init.xavier_uniform_(p)
            else:
                torch.nn.init.uniform_(p)


def build_ref_model(args, seed):
    init_seed(seed, True)
    m = ref.GPTST_Model(args)
    xavier(m.parameters())
    return m


def tie_free(noise, k):
    s = torch.sort(noise, descending=True)[0]
    return k == 0 or k >= s.numel() or bool(s[k - 1] != s[k])


COMPACT_OVER = 8192   # tensors larger than this are stored as a strided subsample + checksums
COMPACT_STRIDE = 13


def put(A, key, t, compact=True):
    """Store tensor t under key; big float tensors as '<key>::sub' (flat[::13]) + '<key>::stats' [sum, abs-sum, numel]."""
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if compact and t.is_floating_point() and t.numel() > COMPACT_OVER:
            f = t.reshape(-1)
            A[key + "::sub"] = f[::COMPACT_STRIDE].clone()
            A[key + "::stats"] = np.array([float(f.double().sum()), float(f.double().abs().sum()), f.numel()])
            return
    A[key] = t


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, "%.1f KB" % (os.path.getsize(os.path.join(HERE, name)) / 1024))


# --------------------------------------------------------------------------------------------
def gen_init_kat():
    import hashlib
    kat = {}
    for ds, seed in (("PEMS08", 12), ("METR_LA", 0), ("NYC_TAXI", 12)):
        args = make_args(ds)
        m = build_ref_model(args, seed)
        sd = m.state_dict()
        h = hashlib.sha256()
        for k, v in sd.items():
            h.update(k.encode()); h.update(v.numpy().tobytes())
        kat[ds] = dict(seed=seed, sha256=h.hexdigest(), keys=list(sd.keys()),
                       shapes=[list(v.shape) for v in sd.values()],
                       psum=float(sum(p.double().sum() for p in m.parameters())),
                       pabs=float(sum(p.double().abs().sum() for p in m.parameters())),
                       nparams=int(sum(p.numel() for p in m.parameters())),
                       neb4mask_0_4=[float(x) for x in sd["encoder.neb4mask"][0, :4]])
    with open(os.path.join(HERE, "init_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("wrote init_kat.json", {k: v["sha256"][:12] for k, v in kat.items()})


def _grads(out, gout, leaves):
    for t in leaves:
        t.grad = None
    (out * gout).sum().backward(retain_graph=True)
    return [t.grad.detach().clone() if t.grad is not None else torch.zeros_like(t) for t in leaves]


def gen_modules():
    """Per-module forward + gradients (G2)."""
    g = torch.Generator().manual_seed(4242)
    rn = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    T = 12
    A = {}
    # ---- squash (incl. a zero vector)
    x = rn(3, 5, 64); x[0, 0] = 0
    A["squash.x"], A["squash.y"] = x, ref.squash(x)
    # ---- time_feature / time_feature_spg
    for e in (16, 4):
        tf = ref.time_feature(e); xavier(tf.parameters())
        inp = rn(2, T, 2).requires_grad_()
        out = tf(inp); go = rn(*out.shape)
        leaves = [inp] + list(tf.parameters())
        gr = _grads(out, go, leaves)
        A["tf%d.in" % e], A["tf%d.out" % e], A["tf%d.gout" % e], A["tf%d.gin" % e] = inp, out, go, gr[0]
        for (k, p), gp in zip(tf.named_parameters(), gr[1:]):
            A["tf%d.p.%s" % (e, k)], A["tf%d.g.%s" % (e, k)] = p, gp
    tf = ref.time_feature_spg(4); xavier(tf.parameters())
    inp = rn(2, T, 2).requires_grad_()
    out = tf(inp); go = rn(*out.shape)
    gr = _grads(out, go, [inp] + list(tf.parameters()))
    A["tfs.in"], A["tfs.out"], A["tfs.gout"], A["tfs.gin"] = inp, out, go, gr[0]
    for (k, p), gp in zip(tf.named_parameters(), gr[1:]):
        A["tfs.p.%s" % k], A["tfs.g.%s" % k] = p, gp
    # ---- hyperTem
    for tag, (B, N, C, d, Hm) in {"ht_a": (2, 20, 64, 4, 8), "ht_b": (3, 17, 64, 8, 5)}.items():
        m = ref.hyperTem(T, N, C, C, d, Hm); xavier(m.parameters())
        x = rn(B, T, N, C).requires_grad_(); ne = rn(N, d).requires_grad_(); te = rn(B, T, d).requires_grad_()
        out = m(x, ne, te); go = rn(*out.shape)
        leaves = [x, ne, te] + list(m.parameters())
        gr = _grads(out, go, leaves)
        for nm, t, gt in zip(["x", "ne", "te"], leaves[:3], gr[:3]):
            A["%s.%s" % (tag, nm)], A["%s.g.%s" % (tag, nm)] = t, gt
        A[tag + ".out"], A[tag + ".gout"] = out, go
        for (k, p), gp in zip(m.named_parameters(), gr[3:]):
            A["%s.p.%s" % (tag, k)], A["%s.g.%s" % (tag, k)] = p, gp
    # ---- cap
    for tag, (B, N, C, d, ds, HS, HT, R) in {"cap_a": (2, 20, 64, 4, 4, 5, 6, 3), "cap_b": (2, 17, 64, 8, 4, 10, 16, 2)}.items():
        m = ref.cap(C, N, T, d, ds, HS, HT, R); xavier(m.parameters())
        x = (0.5 * rn(B, T, N, C)).requires_grad_(); ne = rn(N, d).requires_grad_()
        tes = rn(B, ds).requires_grad_(); teb = rn(B, T, ds).requires_grad_()
        out, c, dyn = m(x, ne, tes, teb); go = rn(*out.shape)
        leaves = [x, ne, tes, teb] + list(m.parameters())
        gr = _grads(out, go, leaves)
        for nm, t, gt in zip(["x", "ne", "tes", "teb"], leaves[:4], gr[:4]):
            A["%s.%s" % (tag, nm)], A["%s.g.%s" % (tag, nm)] = t, gt
        A[tag + ".out"], A[tag + ".gout"], A[tag + ".c"], A[tag + ".dyn"] = out, go, c, dyn
        A[tag + ".R"] = np.int64(R)
        A[tag + ".mask_template"] = m.mask_template
        for (k, p), gp in zip(m.named_parameters(), gr[4:]):
            A["%s.p.%s" % (tag, k)], A["%s.g.%s" % (tag, k)] = p, gp
    # ---- MLP_RL
    for tag, (B, N, C, d, base, HS) in {"mlp_a": (2, 20, 64, 4, 1, 5), "mlp_b": (2, 17, 64, 8, 2, 10)}.items():
        m = ref.MLP_RL(base, HS, C, d, "cpu"); xavier(m.parameters())
        eb = rn(B, T, N, base).requires_grad_(); te = rn(B, T, d).requires_grad_(); ne = rn(N, d).requires_grad_()
        out = m(eb, te, ne); go = rn(*out.shape)
        leaves = [eb, te, ne] + list(m.parameters())
        gr = _grads(out, go, leaves)
        for nm, t, gt in zip(["eb", "te", "ne"], leaves[:3], gr[:3]):
            A["%s.%s" % (tag, nm)], A["%s.g.%s" % (tag, nm)] = t, gt
        A[tag + ".out"], A[tag + ".gout"] = out, go
        for (k, p), gp in zip(m.named_parameters(), gr[3:]):
            A["%s.p.%s" % (tag, k)], A["%s.g.%s" % (tag, k)] = p, gp
    # outputs / gradients (reference results) are stored compactly; inputs and parameters in full
    B2 = {}
    for k, v in A.items():
        is_result = (".g." in k) or k.endswith(".out") or k.endswith(".gin")
        put(B2, k, v, compact=is_result)
    npz("modules.npz", **B2)


def small_args(**kw):
    a = make_args("PEMS08", num_nodes=20, embed_dim=8, HS=5, HT=6, num_route=2, scaler_zeros=synth.scaler_zeros(),
                  epochs=30, change_epoch=3)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def run_ref_forward(model, source, epoch, seed):
    """One reference pretrain forward with recorded noise; re-draws until top-k boundaries are tie-free."""
    while True:
        torch.manual_seed(seed); random.seed(seed)
        with Recorder() as rec:
            outs = model(source, source, None, epoch)
        ok = True
        a = model.encoder
        M = source[..., 0:a.input_base_dim].numel()
        if epoch <= a.change_epoch:
            ok = tie_free(rec.noise[0], int(M * a.mask_ratio))
        if ok:
            return outs, rec
        seed += 1000


def gen_small_forward():
    """Full forward + all parameter gradients on a small model, both phases, both ada types, base 1 and 2 (G3+G4 small)."""
    A = {}
    cases = {"s_rand": dict(epoch=2), "s_ada_all": dict(epoch=20), "s_ada_half": dict(epoch=20, ada_type="half"),
             "s_ada_full": dict(epoch=30, ada_mask_ratio=1.0),
             "s_base2": dict(epoch=20, input_base_dim=2, output_dim=2, num_nodes=17, ada_type="half")}
    for tag, kw in cases.items():
        epoch = kw.pop("epoch")
        args = small_args(**kw)
        model = build_ref_model(args, 7)
        B, T, N, base = 3, 12, args.num_nodes, args.input_base_dim
        src = synth.make_batch(B, T, N, base, seed=99)
        (out, dec, mask, prob, hs1), rec = run_ref_forward(model, src, epoch, 5)
        # loss as in reference Run.py:92-100 + BasicTrainer.py:83-88
        sc = StandardScaler(synth.SCALER_MEAN, synth.SCALER_STD)
        p = sc.inverse_transform(out) * mask; y = sc.inverse_transform(src[..., :base]) * mask
        lf, _ = MAE_torch(pred=p, true=y, mask_value=args.mape_thresh)
        loss = lf
        ls = torch.zeros(())
        if epoch > args.change_epoch:
            ls = torch.nn.KLDivLoss(reduction="sum")(prob.log(), hs1) * 0.1
            loss = lf + ls
        model.zero_grad(); loss.backward()
        A[tag + ".src"] = src
        A[tag + ".epoch"] = np.int64(epoch)
        A[tag + ".out"], A[tag + ".mask"], A[tag + ".prob"], A[tag + ".hs1"] = out, mask.to(torch.int8), prob, hs1
        put(A, tag + ".dec", dec)
        A[tag + ".loss"] = np.array([float(loss), float(lf), float(ls)])
        for i, nz in enumerate(rec.noise):
            A["%s.noise%d" % (tag, i)] = nz
        if rec.orders:
            A[tag + ".list_c"] = np.array(rec.orders[0], dtype=np.int64)
        A[tag + ".sd_seed"] = np.int64(7)
        A[tag + ".sd_hash"] = np.array(sd_hash(model.state_dict()))
        for k, pp in model.named_parameters():
            put(A, "%s.grad.%s" % (tag, k), pp.grad if pp.grad is not None else torch.zeros(0))
        A[tag + ".cfg"] = np.array(json.dumps({k: getattr(args, k) for k in (
            "num_nodes", "input_base_dim", "output_dim", "hidden_dim", "embed_dim", "embed_dim_spa", "HS", "HT", "HT_Tem",
            "num_route", "mask_ratio", "ada_mask_ratio", "ada_type", "change_epoch", "epochs", "mape_thresh")}))
    npz("forward_small.npz", **A)


def gen_full_forward():
    """Full PEMS08-dims forward 5-tuple, B=2, epochs {1, 11, 200, 300} + eval mode (G4). Init = seed 12 (KAT)."""
    args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros())
    model = build_ref_model(args, 12)
    src = synth.make_batch(2, 12, 170, 1, seed=1234)
    A = {"src": src}
    for epoch in (1, 11, 200, 300):
        (out, dec, mask, prob, hs1), rec = run_ref_forward(model, src, epoch, 99)
        t = "e%d." % epoch
        A[t + "out"], A[t + "mask"], A[t + "prob"], A[t + "hs1"] = out, mask.to(torch.int8), prob, hs1
        A[t + "dec_sub"] = dec[:, :, ::7, ::5]
        A[t + "dec_stats"] = np.array([float(dec.double().sum()), float(dec.double().abs().mean())])
        for i, nz in enumerate(rec.noise):
            A["%snoise%d" % (t, i)] = nz
        if rec.orders:
            A[t + "list_c"] = np.array(rec.orders[0], dtype=np.int64)
    eargs = make_args("PEMS08", scaler_zeros=synth.scaler_zeros(), mode="eval")
    emodel = ref.GPTST_Model(eargs)
    emodel.load_state_dict(model.state_dict())
    emb = emodel(src, None)[0]
    A["eval.emb_sub"] = emb[:, :, ::7, ::5]
    A["eval.emb_stats"] = np.array([float(emb.double().sum()), float(emb.double().abs().mean())])
    npz("forward_full.npz", **A)


def gen_steps():
    """Optimiser-step sequence (G5): 6 random-phase + 6 adaptive-phase steps, B=4, small model, reference
    model + torch Adam + clip_grad_norm_ exactly as BasicTrainer.py:79-97."""
    args = small_args()
    model = build_ref_model(args, 3)
    opt = torch.optim.Adam(params=model.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False)
    sc = StandardScaler(synth.SCALER_MEAN, synth.SCALER_STD)
    A = {}
    A["sd_seed"] = np.int64(3)
    A["sd0_hash"] = np.array(sd_hash(model.state_dict()))
    losses = []
    B, T, N, base = 4, 12, args.num_nodes, 1
    step = 0
    for epoch in (1, 1, 1, 2, 2, 3, 4, 4, 10, 10, 30, 30):
        src = synth.make_batch(B, T, N, base, seed=500 + step, start_slot=17 * step)
        opt.zero_grad()
        (out, dec, mask, prob, hs1), rec = run_ref_forward(model, src, epoch, 100 + step)
        p = sc.inverse_transform(out) * mask; y = sc.inverse_transform(src[..., :base]) * mask
        lf, _ = MAE_torch(pred=p, true=y, mask_value=args.mape_thresh)
        ls = torch.zeros(())
        loss = lf
        if epoch > args.change_epoch:
            ls = torch.nn.KLDivLoss(reduction="sum")(prob.log(), hs1) * 0.1
            loss = lf + ls
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), args.max_grad_norm)
        opt.step()
        losses.append([float(loss), float(lf), float(ls)])
        A["st%d.epoch" % step] = np.int64(epoch)
        A["st%d.mask" % step] = mask.to(torch.int8)
        for i, nz in enumerate(rec.noise):
            A["st%d.noise%d" % (step, i)] = nz
        if rec.orders:
            A["st%d.list_c" % step] = np.array(rec.orders[0], dtype=np.int64)
        step += 1
    A["losses"] = np.array(losses)
    for k, v in model.state_dict().items():
        put(A, "sdN." + k, v.clone())
    A["adam_steps"] = np.array([int(opt.state[p]["step"]) if p in opt.state else 0 for p in model.parameters()])
    npz("steps.npz", **A)
    print("losses", np.array(losses)[:, 0])


class Injector:
    """Feeds prepared tensors to the reference's torch.rand_like calls and a prepared order to random.shuffle (GPTST.py:316,358,
    389,400), so that a long run is reproducible from seeds alone."""

    def __init__(self, noises, order):
        self.noises, self.order = list(noises), order

    def __enter__(self):
        self._rl, self._sh = torch.rand_like, random.shuffle

        def rl(x, *a, **k):
            return self.noises.pop(0).view_as(x).to(x.dtype)

        def sh(lst, *a, **k):
            lst[:] = self.order

        torch.rand_like, random.shuffle = rl, sh
        return self

    def __exit__(self, *exc):
        torch.rand_like, random.shuffle = self._rl, self._sh


def gen_curve(nsteps=200, per_epoch=50, lr=None, name="curve_c2.npz", perturb_seed=None, perturb_rel=None):
    """Loss curve at the BASELINE configs[1] shape (G5b): `nsteps` optimiser steps of the REFERENCE model (B=32, N=170, C=64) with a
    shortened schedule (epochs=4, change_epoch=2: random-mask phase, then adaptive mask + KL), mask noise / class order derived
    from seeds (synth.make_noise / synth.class_order).  Stored per step: epoch, seeds, the reference's mask (bit-packed), losses."""
    from gptst_amd import data as gdata
    args = make_args("PEMS08", epochs=nsteps // per_epoch, change_epoch=nsteps // per_epoch // 2, batch_size=32)
    if lr is not None:
        args.lr_init = lr
    raw = synth.make_series(args.num_nodes, 3, interval=5, seed=10)          # learnable series (see synth.make_series)
    train, _, _, scaler, _, _ = gdata.get_dataloader(args, raw=raw)
    args.scaler_zeros = float(scaler.transform(0))
    model = build_ref_model(args, 12)
    if perturb_seed is not None:          # envelope run: every initial parameter moved by one ulp (or by perturb_rel, relative) randomly
        g = torch.Generator().manual_seed(perturb_seed)
        with torch.no_grad():
            for p in model.parameters():
                if perturb_rel is None:
                    up = torch.rand(p.shape, generator=g) < 0.5
                    p.copy_(torch.where(up, torch.nextafter(p, torch.full_like(p, float("inf"))), torch.nextafter(p, torch.full_like(p, -float("inf")))))
                else:
                    p.mul_(1.0 + perturb_rel * torch.randn(p.shape, generator=g))
    opt = torch.optim.Adam(params=model.parameters(), lr=args.lr_init, eps=1.0e-8, weight_decay=0, amsgrad=False)
    sc = StandardScaler(float(scaler.mean), float(scaler.std))
    B, T, N, base, HS = 32, 12, args.num_nodes, 1, args.HS
    M = B * T * N
    A = {"sd_seed": np.int64(12), "sd0_hash": np.array(sd_hash(model.state_dict())), "epochs": np.int64(args.epochs),
         "change_epoch": np.int64(args.change_epoch), "scaler": np.array([float(scaler.mean), float(scaler.std)]),
         "lr": np.float64(args.lr_init)}
    losses, seeds, masks, orders, epochs = [], [], [], [], []
    import time
    t0 = time.time()
    for step in range(nsteps):
        epoch = step // per_epoch + 1
        src = train.windows((torch.arange(B) * 7 + 13 * step) % train.n)[0]       # deterministic, spread over the series
        s0 = 50000 + 10 * step
        if epoch <= args.change_epoch:
            while not tie_free(synth.make_noise(M * base, s0), int(M * base * args.mask_ratio)):
                s0 += 1
            noises, sd3 = [synth.make_noise(M * base, s0)], (s0, 0, 0)
        else:
            noises, sd3 = [synth.make_noise(M, s0), synth.make_noise(M, s0 + 1)], (0, s0, s0 + 1)
        order = synth.class_order(HS, 900 + step)
        opt.zero_grad()
        with Injector(noises, order):
            out, dec, mask, prob, hs1 = model(src, src, None, epoch)
        p = sc.inverse_transform(out) * mask; y = sc.inverse_transform(src[..., :base]) * mask
        lf, _ = MAE_torch(pred=p, true=y, mask_value=args.mape_thresh)
        ls = torch.zeros(())
        loss = lf
        if epoch > args.change_epoch:
            ls = torch.nn.KLDivLoss(reduction="sum")(prob.log(), hs1) * 0.1
            loss = lf + ls
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), args.max_grad_norm)
        opt.step()
        losses.append([float(loss), float(lf), float(ls)]); seeds.append(sd3); orders.append(order); epochs.append(epoch)
        masks.append(np.packbits(mask.reshape(-1).to(torch.uint8).numpy()))
        if step % 20 == 0:
            print("step", step, "epoch", epoch, losses[-1], "%.0fs" % (time.time() - t0), flush=True)
    if perturb_seed is not None:
        return np.array(losses)
    A["losses"], A["seeds"], A["orders"], A["epoch"] = np.array(losses), np.array(seeds), np.array(orders), np.array(epochs)
    A["masks"] = np.stack(masks)                       # (nsteps, M/8) uint8: the reference's `1 - final_mask` (1 = masked cell)
    np.savez_compressed(os.path.join(HERE, name), **A)
    print("wrote", name)


def gen_curve_envelope(K=8, name="curve_c2_env.npz", perturb_rel=None):
    """fp32 envelope of the reference's own loss curve: K more runs of gen_curve's schedule, each from an initial state that differs
    from the golden run's by one ulp per parameter (free-running masks: the same noise and class orders, the run's own cluster labels).
    Two correct fp32 implementations of the step can differ by this much and no less: tests/test_gpu_curve.py asks the HIP curve to
    stay inside [min, max] of these runs (plus the golden one) at every step."""
    runs = [gen_curve(perturb_seed=100 + k, perturb_rel=perturb_rel) for k in range(K)]
    np.savez_compressed(os.path.join(HERE, name), losses=np.stack(runs), perturb_seeds=np.arange(100, 100 + K),
                        perturb_rel=np.float64(perturb_rel if perturb_rel is not None else 0.0))
    print("wrote", name)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["init", "modules", "small", "full", "steps"]
    if "init" in which:
        gen_init_kat()
    if "modules" in which:
        gen_modules()
    if "small" in which:
        gen_small_forward()
    if "full" in which:
        gen_full_forward()
    if "steps" in which:
        gen_steps()
    if "curve" in which:            # not in the default list: ~3 min of reference CPU time each
        gen_curve()
    if "curve_env" in which:        # K reference runs from 1-ulp perturbed initial states (~25 min)
        gen_curve_envelope()
    if "curve_env_1e-6" in which:   # ... and from initial states perturbed by 1e-6 relative: the size of the difference between two fp32
        gen_curve_envelope(name="curve_c2_env_rel1e-6.npz", perturb_rel=1e-6)     # implementations of ONE forward pass (measured)
    if "curve_lowlr" in which:      # the same run at a tenth of the learning rate: contracting enough to be compared pointwise
        gen_curve(lr=3e-4, name="curve_c2_lr3e-4.npz")
