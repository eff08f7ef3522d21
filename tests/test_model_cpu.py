"""CPU-side checks of the drop-in module (no compute): state_dict format, init KAT, C-ABI exports."""
import json
import os
import re

import pytest
import torch

from golden_util import GOLDEN
from gptst_amd.config import make_args


def test_state_dict_format_and_init_kat():
    from gptst_amd.model import GPTST_Model, init_seed, xavier_init_
    from oracle import gptst_oracle as O
    kat = json.load(open(os.path.join(GOLDEN, "init_kat.json")))
    for ds, k in kat.items():
        args = make_args(ds)
        init_seed(k["seed"])
        m = xavier_init_(GPTST_Model(args))
        sd = m.state_dict()
        assert list(sd.keys()) == k["keys"], ds
        assert [list(v.shape) for v in sd.values()] == k["shapes"]
        assert O.state_hash(sd) == k["sha256"], "init must be bit-identical to the reference for seed %d" % k["seed"]
        assert sum(p.numel() for p in m.parameters()) == k["nparams"]


def test_flat_views_and_load_state_dict():
    from gptst_amd.model import GPTST_Model
    from oracle import gptst_oracle as O
    args = make_args("PEMS08", num_nodes=20, embed_dim=4)
    m = GPTST_Model(args)
    sd = O.init_state_dict(args, 5)
    m.load_state_dict(sd)
    for k, p in m.named_parameters():
        assert torch.equal(p.detach(), sd[k])
        o = m._offs[k]
        assert p.data_ptr() == m.flat[o:].data_ptr() and o % 4 == 0
    assert m.nA + m.nB <= m.flat.numel()
    # every KL-path parameter lies in [nA, nA+nB), never-trained ones after
    for k in m.param_keys:
        o = m._offs[k]
        if k.startswith("encoder.MLP_RL.") or k.startswith("encoder.teb4mask.") or k == "encoder.neb4mask":
            assert m.nA <= o < m.nA + m.nB
        elif k.startswith("decoder.time_feature"):
            assert o >= m.nA + m.nB
        else:
            assert o < m.nA
    assert O.state_hash(m.state_dict()) == O.state_hash(sd)


def test_cpu_forward_fails_loudly():
    from gptst_amd.model import GPTST_Model
    import pytest
    args = make_args("PEMS08", num_nodes=20, embed_dim=4)
    m = GPTST_Model(args)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 12, 20, 3), None, None, 1)


def test_c_abi_exports_every_header_symbol():
    from gptst_amd import _C
    txt = open(_C.HEADER).read() + open(_C.TESTING_HEADER).read()          # the C ABI + the test / benchmark hooks
    names = set(re.findall(r"\bint\s+(gptst_\w+)\s*\(", txt))
    lib = _C.lib()
    assert names == set(lib.protos) and len(names) >= 25
    import ctypes
    dll = ctypes.CDLL(_C.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), n
    assert lib.value("gptst_abi_version") == _C.header_abi_version()


def test_library_exports_exactly_the_header():
    """The C-ABI library exports every symbol include/gptst_hip.h declares and NOTHING else (no debug / tuning scaffolding:
    those exist only in -DGPTST_DEBUG builds; cross-file helpers have hidden visibility)."""
    import shutil
    import subprocess
    from gptst_amd import _C
    hdr = set(_C.parse_header()) | set(_C.parse_header(_C.TESTING_HEADER))      # the C ABI + the test / benchmark hooks
    assert not {"gptst_tune", "gptst_mask_force_multi"} & set(_C.parse_header()), "experiment knobs do not belong in the product header"
    lib = _C.lib()
    assert all(hasattr(lib, "_raw_" + n) for n in hdr)
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _C.LIB_PATH], capture_output=True, text=True, check=True).stdout
    # EVERY defined dynamic symbol, no prefix filter (r03: 378 kernel stubs / handles were exported next to the API): -fvisibility=hidden +
    # default visibility on the header's declarations leaves exactly the API (and the linker's own _init / _fini where present)
    exp = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TDBRVW"} - {"_init", "_fini"}
    assert exp == hdr, (sorted(exp - hdr)[:20], sorted(hdr - exp))


def test_integration_stub_matches_header():
    """The ctypes stub a maintainer would copy out of INTEGRATION.md (section 2) declares the prototype include/gptst_hip.h declares: same
    entry point, same number and kinds of arguments, and the call passes that many values (a stale stub corrupts memory, it does not fail)."""
    import ctypes
    from gptst_amd import _C
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", txt, flags=re.S)
    stub = [b for b in blocks if "argtypes" in b]
    assert len(stub) == 1
    stub = stub[0]
    m = re.search(r"lib\.(gptst_\w+)\.argtypes\s*=\s*(.+)", stub)
    name, expr = m.group(1), m.group(2)
    declared = eval(expr, {"ctypes": ctypes})
    want = _C.parse_header()[name]
    assert [t is ctypes.c_void_p for t in declared] == [t is ctypes.c_void_p for t in want], (declared, want)
    assert all(d is w for d, w in zip(declared, want))
    call = re.search(r"lib\.%s\((.*?)\)\nassert" % name, stub, flags=re.S).group(1)
    depth, nargs = 0, 1
    for ch in call:
        depth += ch in "([{"
        depth -= ch in ")]}"
        nargs += (ch == "," and depth == 0)
    assert nargs == len(want), (nargs, len(want))


def test_predictor_conf_overrides_the_pretrain_schedule():
    """-mode eval (reference Run.py:36-43): every attribute the predictor's argument set also has replaces the pretrain conf's value —
    epochs 100, lr_decay_step 25,50,75, early_stop_patience 25, batch_size 64, debug / xavier False, seed from conf/STGCN/<dataset>.conf."""
    from gptst_amd.config import apply_predictor_overrides, predictor_args
    for ds, seed, n in (("PEMS08", 12, 170), ("METR_LA", 0, 207), ("NYC_TAXI", 12, 266), ("NYC_BIKE", 12, 250)):
        a = make_args(ds, mode="eval")
        assert a.epochs == 300 and a.early_stop_patience >= 80                       # the pretrain conf's schedule
        p = predictor_args(ds, "STGCN", ["--lr_init", "0.01"])
        apply_predictor_overrides(a, p)
        assert (a.epochs, a.lr_decay_step, a.early_stop_patience, a.batch_size) == (100, "25, 50, 75", 25, 64)
        assert (a.debug, a.xavier, a.seed, a.lr_init, a.num_nodes) == (False, False, seed, 0.01, n)
        assert (p.Ks, p.Kt, p.blocks1, p.drop_prob, p.outputl_ks) == (3, 3, [64, 32, 128], 0, 3)
    with pytest.raises(ValueError):
        predictor_args("PEMS08", "GWN")


def test_window_loader_rank_slices_partition_the_epoch():
    """iter_x(rank, world, limit): the ranks' batches are disjoint, together they are the first `limit` batches of the one permutation."""
    import numpy as np
    from gptst_amd.data import WindowLoader
    series = torch.arange(200 * 3, dtype=torch.float32).view(200, 3, 1)
    mk = lambda: WindowLoader(series, 12, 12, 8, shuffle=True, generator=torch.Generator().manual_seed(5))
    full = [x for x in mk().iter_x()]
    parts = [[x for x in mk().iter_x(rank=r, world=3, limit=18)] for r in range(3)]
    assert [len(p) for p in parts] == [6, 6, 6]
    for k in range(18):
        assert torch.equal(parts[k % 3][k // 3], full[k])


def test_window_loader_tail_rounds_keep_every_batch():
    """Data parallelism keeps the tail of an epoch as padded rounds: the batches iter_x hands to the ranks plus the batches of the tail rounds
    are exactly the batches of the single-process epoch (same permutation), the ragged last batch in a round of its own."""
    from gptst_amd.data import WindowLoader
    series = torch.arange(200 * 3, dtype=torch.float32).view(200, 3, 1)
    for world in (2, 3, 4):
        mk = lambda: WindowLoader(series, 12, 12, 8, shuffle=True, generator=torch.Generator().manual_seed(5))
        full_epoch = [x for x in mk().iter_x()]
        full = mk().n // 8
        usable = full // world * world
        ld = mk()
        per_rank = [[x for x in ld.iter_x(rank=r, world=world, limit=usable)] for r in range(1)]   # (rank 0 draws the permutation of the epoch)
        rounds = list(ld.tail_rounds(usable, world))
        assert sum(len(r) for r in rounds) == len(full_epoch) - usable
        assert all(len(r) < world or len(r) == 1 for r in rounds)
        got = [x for xs in ld.iter_tail(usable, world) for x in xs]
        for k, x in zip(range(usable, len(full_epoch)), got):
            assert torch.equal(x, full_epoch[k])
        if mk().n % 8:
            assert got[-1].shape[0] == mk().n % 8 and rounds[-1] == [full]
        assert torch.equal(per_rank[0][0], full_epoch[0])


def test_downstream_gate_has_no_cpu_path():
    from gptst_amd.enhance import Fusion
    from gptst_amd.fusion import fusion_gate
    with pytest.raises(RuntimeError):
        fusion_gate(torch.zeros(2, 3, 4, 64), torch.zeros(2, 3, 4, 3), Fusion(64), torch.nn.Linear(1, 64), 1)


def test_parameters_are_served_from_the_cached_walk_and_owner_lookup():
    """r06: GPTST_Model.parameters() / named_parameters() return the list of the last _flatten() (the reference loop asks every step), in the module
    tree's order; owner_of() finds the model behind a parameter of its flat buffer (optim.ClipAdam needs the layout)."""
    import torch.nn as nn
    from gptst_amd.config import make_args
    from gptst_amd.model import GPTST_Model
    args = make_args("PEMS08", num_nodes=12, embed_dim=4, HS=3, HT=4)
    m = GPTST_Model(args)
    fast = list(m.named_parameters())
    slow = list(nn.Module.named_parameters(m))
    assert [k for k, _ in fast] == [k for k, _ in slow] and all(a is b for (_, a), (_, b) in zip(fast, slow))
    assert [id(p) for p in m.parameters()] == [id(p) for _, p in slow]
    assert [k for k, _ in m.named_parameters(prefix="x")] == ["x." + k for k, _ in slow]       # non-default calls take nn.Module's walk
    assert GPTST_Model.owner_of(fast[3][1]) is m and GPTST_Model.owner_of(nn.Parameter(torch.zeros(3))) is None
    m2 = GPTST_Model(args)
    assert GPTST_Model.owner_of(next(m2.parameters())) is m2
    assert len(m.state_dict()) == len(slow) + 4          # 4 mask_template buffers
