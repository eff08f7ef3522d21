#!/usr/bin/env python
"""Register / LDS / scratch footprint of every kernel of a .hip source, from the gfx950 ISA metadata (hipcc -S).
usage: python tools/kernel_regs.py gpt-st_amd/csrc/hypertem.hip ["-DHT_OCC=4 ..."]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
extra = sys.argv[2].split() if len(sys.argv) > 2 else []
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-Igpt-st_amd/csrc", "-S", "--cuda-device-only"] + extra + ["-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = g("name").group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*$", "", dem)
    print("%-62s vgpr %3s agpr %3s spill %3s scratch %4s lds %6s" % (dem[:62], g("vgpr_count").group(1), blk.split()[0], g("vgpr_spill_count").group(1),
                                                                     g("private_segment_fixed_size").group(1), g("group_segment_fixed_size").group(1)))
