#!/bin/bash
# usage (GPU box, repo root): tools/pmc_calibrate.sh  -> gpurun_out/pmc_calibration.txt
# Known-byte kernels (tools/experiments/pmc_calibration.hip) under separate FETCH_SIZE / WRITE_SIZE passes: bytes moved per counter unit, per pattern.
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
bin=$root/tools/experiments/pmc_calibration
[ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $bin $root/tools/experiments/pmc_calibration.hip
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/pmccal_$ctr && rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmccal_$ctr -o cal -- $bin ) > $root/gpurun_out/pmccal_$ctr.log 2>&1
done
python - > $root/gpurun_out/pmc_calibration.txt <<'PY'
import sqlite3, re, glob
KNOWN = {  # kernel name prefix -> (bytes read, bytes written) per launch
    "void read_stream<HIP_vector_type<float, 4": (2**30, 0), "void read_stream<HIP_vector_type<float, 2": (2**30, 0), "void read_stream<float>": (2**30, 0),
    "read_seg256": (2**30, 0), "read_l2_resident": (2**20, 0), "void write_stream<HIP_vector_type<float, 4": (0, 2**30), "void write_stream<float>": (0, 2**30),
    "write_seg64": (0, 2**28)}
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(glob.glob("/tmp/pmccal_%s/*.db" % ctr)[0])
    for n, v in db.execute("select kernel_name, value from counters_collection where counter_name='%s'" % ctr):
        for k in KNOWN:
            if n.startswith(k):
                a = res.setdefault(k, {}).setdefault(ctr, [0, 0.0]); a[0] += 1; a[1] += v
print("# bytes moved per launch vs the counters (KB units; average of 3 launches).  factor = known bytes / (counter * 1024)")
print("%-46s %12s %14s %8s %14s %8s" % ("kernel", "known MiB", "FETCH_SIZE KB", "factor", "WRITE_SIZE KB", "factor"))
for k, (rd, wr) in KNOWN.items():
    f = res.get(k, {}).get("FETCH_SIZE", [1, 0.0]); w = res.get(k, {}).get("WRITE_SIZE", [1, 0.0])
    fk, wk = f[1] / max(f[0], 1), w[1] / max(w[0], 1)
    print("%-46s %12.1f %14.0f %8s %14.0f %8s" % (k[:46], (rd + wr) / 2**20, fk, ("%.3f" % (rd / (fk * 1024))) if rd and fk else "-", wk, ("%.3f" % (wr / (wk * 1024))) if wr and wk else "-"))
PY
cat $root/gpurun_out/pmc_calibration.txt
