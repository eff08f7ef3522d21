#!/usr/bin/env python3
"""List loops whose body contains a global load followed by a full vmcnt(0) wait (= one serialised memory round trip per trip).
usage: tools/isa_loops.py file.hip ...   (compiles each to gfx950 ISA with hipcc -S and scans the basic blocks)"""
import re, subprocess, sys, os, tempfile
def scan(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-Igpt-st_amd/csrc", "-S",
                        "--cuda-device-only", "-o", out, path], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    # kernels -> list of (label or None, instruction)
    kerns, cur = {}, None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        m2 = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m: cur = kerns.setdefault(m.group(1), [])
        elif cur is not None and m2: cur.append((m2.group(1), None))
        elif cur is not None and ln.startswith("\t") and not ln.startswith("\t.") and not ln.strip().startswith(";"):
            ins = ln.strip()
            cur.append((None, ins))
            if ins.startswith("s_endpgm"): cur = None
    for kern, items in kerns.items():
        pos = {lab: i for i, (lab, _) in enumerate(items) if lab}
        for i, (lab, ins) in enumerate(items):
            if not ins: continue
            m = re.match(r"s_c?branch\w* (\.LBB\d+_\d+)", ins)
            if m and m.group(1) in pos and pos[m.group(1)] < i:
                body = [x for _, x in items[pos[m.group(1)]:i] if x]
                nload = sum(1 for x in body if x.startswith("global_load") or x.startswith("buffer_load"))
                nwait0 = sum(1 for x in body if "vmcnt(0)" in x)
                nmfma = sum(1 for x in body if "mfma" in x)
                if nload and nwait0:
                    print(f"{os.path.basename(path):14s} {kern[3:50]:47s} {m.group(1):9s} loads={nload:3d} vmcnt0={nwait0} mfma={nmfma:3d} len={len(body)}")
for p in sys.argv[1:]: scan(p)
