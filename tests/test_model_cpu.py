"""CPU-side checks of the drop-in module (no compute): state_dict format, init KAT, C-ABI exports."""
import json
import os
import re

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
    txt = open(_C.HEADER).read()
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
    hdr = set(_C.parse_header())
    lib = _C.lib()
    assert all(hasattr(lib, "_raw_" + n) for n in hdr)
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _C.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exp = {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("gptst_")}
    assert exp == hdr, (sorted(exp - hdr), sorted(hdr - exp))
