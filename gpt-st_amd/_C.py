"""ctypes binding of ``lib/libgptst_hip.so`` — prototypes are parsed from ``include/gptst_hip.h`` so the header
is the single source of truth.  There is NO fallback: a missing library raises ``ImportError`` (the product
path must fail loudly without its HIP extension).  ``import torch`` must come first so the library binds to the
HIP runtime torch already loaded (one runtime per process)."""
import ctypes
import os
import re

import torch  # noqa: F401  (loads libamdhip64 first)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPTST_LIB") or os.path.join(HERE, "lib", "libgptst_hip.so")     # GPTST_LIB: another build of the library (A/B runs on one box)
HEADER = os.path.join(os.path.dirname(HERE), "include", "gptst_hip.h")
TESTING_HEADER = os.path.join(os.path.dirname(HERE), "include", "gptst_hip_testing.h")      # test / benchmark hooks, not part of the C ABI

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
       "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "size_t": ctypes.c_size_t}


def parse_header(path=HEADER):
    """-> {name: [ctype per argument]} for every ``int gptst_*(...);`` declaration."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = {}
    for m in re.finditer(r"\bint\s+(gptst_\w+)\s*\(([^)]*)\)\s*;", txt):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", "").split()
                    types.append(_CT[base[0]])
        protos[name] = types
    return protos


def header_abi_version(path=HEADER):
    """GPTST_ABI_VERSION of include/gptst_hip.h: bumped whenever an exported signature or a workspace size changes."""
    return int(re.search(r"#define\s+GPTST_ABI_VERSION\s+(\d+)", open(path).read()).group(1))


class GptstError(RuntimeError):
    code = 0


ESHAPE = -2          # "this shape is not served by this kernel" (e.g. the (b,t) capsule matrix does not fit LDS)


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError("gpt-st_amd: %s not built — run `python -m gptst_amd.build` (no CPU fallback exists)" % LIB_PATH)
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        self.protos.update(parse_header(TESTING_HEADER))
        for name, types in self.protos.items():
            fn = getattr(self._dll, name)          # AttributeError if the header declares a missing symbol
            fn.argtypes = types
            fn.restype = ctypes.c_int
            setattr(self, "_raw_" + name, fn)
        want, got = header_abi_version(), self._raw_gptst_abi_version()
        if got != want:             # a stale .so called with this header's prototypes would get shifted arguments, not an error
            raise ImportError("gpt-st_amd: %s has ABI version %d, include/gptst_hip.h declares %d — rebuild (python -m gptst_amd.build)"
                              % (LIB_PATH, got, want))
        for kv in filter(None, os.environ.get("GPTST_TUNE", "").split(",")):      # experiments: GPTST_TUNE="5=1,2=4" -> gptst_tune(id, value)
            k, v = kv.split("=")
            self._raw_gptst_tune(int(k), int(v))

    def call(self, name, *args):
        rc = getattr(self, "_raw_" + name)(*args)
        if rc != 0:
            e = GptstError("%s failed with code %d" % (name, rc))
            e.code = rc
            raise e

    def value(self, name, *args):
        return getattr(self, "_raw_" + name)(*args)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def ptr(t):
    """Device pointer of a contiguous fp32/int tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "gpt-st_amd kernels need contiguous tensors"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """the current HIP stream of the current device as an integer handle (every C-ABI call takes it).  torch.cuda.current_stream() builds a Stream
    object per call (10 us, 1288 times in 46 eager steps: tools/experiments/prof_module_path.py); the raw getter is what torch's own code generators use"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
