"""Build the gfx950 C-ABI library ``lib/libgptst_hip.so`` in-tree with hipcc (no JIT cache).

    python -m gptst_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Objects are rebuilt only when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgptst_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result"] + os.environ.get("GPTST_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("gptst_hip.h", "gptst_hip_testing.h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs if os.path.exists(h))
    cc = _hipcc()
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append([cc] + FLAGS + ["-I", os.path.join(os.path.dirname(HERE), "include"), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r

    with ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs) or 1))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        # exports: the gptst_* entry points the headers declare and nothing else — -fvisibility=hidden covers functions, but hipcc gives the
        # kernel handle objects default visibility whatever the flag says (r03: 378 of them next to the API), so the linker drops the rest
        vmap = os.path.join(LIBDIR, "exports.map")
        with open(vmap, "w") as f:
            f.write("{ global: gptst_*; local: *; };\n")
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vmap, "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
