#!/usr/bin/env python
"""Build a VARIANT of the C-ABI library for same-box A/B runs: the named sources are recompiled with extra hipcc flags (usually -D switches),
every other object comes from the regular in-tree build.  The result is gpt-st_amd/lib/libgptst_<name>.so (git-ignored; travels with the gpurun
snapshot); select it with GPTST_LIB=$PWD/gpt-st_amd/lib/libgptst_<name>.so (tools/ab_lib.sh, tools/mb_*.py).

    python tools/build_variant.py occ6 "-DPJ_OCC=6 -DPJ_UC=4" poolgen.hip
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gptst_amd import build as B      # noqa: E402


def main():
    name, flags, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    B.build()
    cc = B._hipcc()
    vdir = os.path.join(B.LIBDIR, "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for s in sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip")):
        if s in files:
            obj = os.path.join(vdir, s[:-4] + ".o")
            cmd = [cc] + B.FLAGS + flags + ["-I", os.path.join(ROOT, "include"), "-c", os.path.join(B.CSRC, s), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stderr[-4000:])
        else:
            obj = os.path.join(B.LIBDIR, s[:-4] + ".o")
        objs.append(obj)
    out = os.path.join(B.LIBDIR, "libgptst_%s.so" % name)
    vmap = os.path.join(B.LIBDIR, "exports.map")
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + vmap, "-o", out] + objs + ["-ldl"], check=True)
    print(out)


if __name__ == "__main__":
    main()
