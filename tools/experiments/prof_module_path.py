import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo')
import torch, bench
from gptst_amd.config import make_args
from gptst_amd import synth
args = make_args("PEMS08", scaler_zeros=synth.scaler_zeros(), device="cuda:0")
dev = torch.device("cuda", 0)
CA = os.environ.get("CLIP_ADAM", "1") == "1"
bench.module_path(args, 32, 200, dev, steps=10, clip_adam=CA)
pr = cProfile.Profile(); pr.enable()
r = bench.module_path(args, 32, 200, dev, steps=40, clip_adam=CA)
pr.disable()
print(r["steps_per_s"], r["ms_per_step"])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(60); print(s.getvalue()[:9000])
