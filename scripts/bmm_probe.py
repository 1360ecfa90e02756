"""Is ONE batched launch of the three hidden-layer weight-gradient products (dW_l = dz_l^T h_{l-1}, 1024^3 each) faster than three?"""
import torch
dev = torch.device("cuda", 0)
dz = torch.randn(3, 1024, 1024, device=dev)
hp = torch.randn(3, 1024, 1024, device=dev)
out = torch.empty(3, 1024, 1024, device=dev)


def sep():
    for l in range(3):
        torch.mm(dz[l].t(), hp[l], out=out[l])


def bat():
    torch.bmm(dz.transpose(1, 2), hp, out=out)


for name, fn in (("3 x mm", sep), ("bmm", bat)):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us for the three products")
