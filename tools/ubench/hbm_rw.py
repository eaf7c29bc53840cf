"""HBM streaming rates seen by simple kernels (torch fill / copy / sum) -- the write rate bounds the epilogue of the low-K tap-GEMMs."""
import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for gb in (0.5, 2.0, 4.0):
    n = int(gb * (1 << 30) / 4)
    x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
    print(f"{gb} GB: fill {gb*1.0737/t(lambda: x.fill_(1.0)):.2f} GB/ms?", end=" ")
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x)); ts = t(lambda: x.sum())
    print(f"fill {n*4/tf/1e12:.2f} TB/s written; copy {n*4/tc/1e12:.2f} TB/s read + same written ({2*n*4/tc/1e12:.2f} total); sum {n*4/ts/1e12:.2f} TB/s read")
