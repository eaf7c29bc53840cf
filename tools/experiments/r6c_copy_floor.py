"""Round 6: what a plain streaming pass over rb_stream6's tensors costs on this box (torch's own elementwise kernels: copy, ELU) --
the practical HBM floor of a kernel that reads 1.97 GB and writes 1.97 GB."""
import torch, time
x = torch.randn(64, 120000, 64, device="cuda")
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 4 * 2 / 1e9
for name, fn in (("copy_", lambda: y.copy_(x)), ("elu", lambda: torch.nn.functional.elu(x, out=y) if False else torch.ops.aten.elu.out(x, out=y)), ("fill (write only)", lambda: y.fill_(1.0)), ("sum (read only)", lambda: x.sum())):
    ms = t(fn)
    b = gb if name in ("copy_", "elu") else gb / 2
    print(f"{name}: {ms:.3f} ms  {b / ms:.2f} TB/s", flush=True)
