"""The 1024^3 fp32 products of the PPO step by operand layout, with and without the bias + ReLU epilogue, default heuristics and
the committed TunableOp solutions: is another weight layout worth keeping for the forward pass?  (back-to-back launches, gap included)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brl_amd import tuned
dev = "cuda"
x = torch.randn(1024, 1024, device=dev)
W = torch.randn(1024, 1024, device=dev) / 32      # [out, in]
Wt = W.t().contiguous()                            # [in, out]
b = torch.randn(1024, device=dev)
out = torch.empty(1024, 1024, device=dev)


def timeit(f, n=300):
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for label, en in (("default heuristics", False), ("committed TunableOp solutions", True)):
    if en:
        tuned.enable()
    print(label)
    print("  addmm_activation x @ W.t() (the forward as it is)      %.2f us" % timeit(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False, out=out)))
    print("  addmm_activation x @ Wt    (weights kept transposed)   %.2f us" % timeit(lambda: torch._addmm_activation(b, x, Wt, use_gelu=False, out=out)))
    print("  mm x @ W.t()  (no epilogue)                            %.2f us" % timeit(lambda: torch.mm(x, W.t(), out=out)))
    print("  mm x @ Wt     (no epilogue = the dh product)           %.2f us" % timeit(lambda: torch.mm(x, Wt, out=out)))
    print("  mm x.t() @ Wt (the dW product's layout)                %.2f us" % timeit(lambda: torch.mm(x.t(), Wt, out=out)))
