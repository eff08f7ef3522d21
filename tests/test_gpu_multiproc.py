"""The multi-process entry of bench.py (what the driver launches for N > 1) exercised on ONE GPU: two ranks share cuda:0 and talk
over gloo instead of RCCL.  Covers process-group setup, the [gradient | statistics] all-reduce, the label exchange of the
adaptive-mask phase, the node-sharded mode, max-over-ranks timing and the single JSON line on rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ, GPTST_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "4"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,scaling,par", [([], "weak", "dp2"), (["--epoch", "1"], "weak", "dp2"), (["--shard", "nodes"], "strong", "nodes2")])
def test_bench_two_ranks(extra, scaling, par):
    out = _run(extra, 29610 + len(extra) * 7 + os.getpid() % 50)
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["config"]["parallelism"] == par
    assert out["value"] > 0 and out["steps"] == 4 and out["higher_is_better"] is True
    assert out["last_loss"] == out["last_loss"] and out["last_loss"] > 0          # finite
