"""Does a consumer launch read a producer's output out of the 256 MB Infinity Cache?  For buffer sizes 32 MB .. 1 GB: `dst.copy_(src)` (producer writes dst)
directly followed by `dst.sum()` (consumer reads dst), against the same sum after 2 GB of unrelated traffic (cold).  usage (GPU box): python tools/experiments/mall_reuse.py"""
import torch

dev = torch.device("cuda", 0)
big = torch.empty(512 * 2 ** 20, device=dev)        # 2 GB flush buffer


def timed(fn, reps=10, pre=None):
    ts = []
    for _ in range(reps):
        if pre is not None:
            pre()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb * 2 ** 20 // 4
    src = torch.randn(n, device=dev)
    dst = torch.empty_like(src)
    out = torch.empty((), device=dev)
    t_copy = timed(lambda: dst.copy_(src), pre=lambda: big.zero_())
    t_cold = timed(lambda: torch.sum(dst, dim=0, out=out), pre=lambda: big.zero_())
    t_hot = timed(lambda: torch.sum(dst, dim=0, out=out), pre=lambda: (big.zero_(), dst.copy_(src)))
    t_hot2 = timed(lambda: torch.sum(dst, dim=0, out=out), pre=lambda: (big.zero_(), torch.sum(dst, dim=0, out=out)))
    print("%5d MB: copy %7.1f us (%.2f TB/s r+w)   sum cold %7.1f us (%.2f TB/s)   sum right after the copy %7.1f us (%.2f TB/s)   sum after a sum %7.1f us (%.2f TB/s)" % (
        mb, t_copy, 2 * mb * 2 ** 20 / t_copy / 1e6, t_cold, mb * 2 ** 20 / t_cold / 1e6, t_hot, mb * 2 ** 20 / t_hot / 1e6, t_hot2, mb * 2 ** 20 / t_hot2 / 1e6))
