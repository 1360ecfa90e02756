"""Forward hidden layers of the minibatch step: torch._addmm_activation with and without out= (does the fused bias + ReLU epilogue
survive out=?), and the dW products batched (bmm) when activations live in one stacked buffer."""
import torch
dev = torch.device("cuda", 0)
B, H = 1024, 1024
x = torch.randn(B, H, device=dev)
W = [torch.randn(H, H, device=dev) / 32 for _ in range(3)]
b = [torch.randn(H, device=dev) for _ in range(3)]
hs = torch.empty(3, B, H, device=dev)


def plain():
    y = x
    for l in range(3):
        y = torch._addmm_activation(b[l], y, W[l].t(), use_gelu=False)
    return y


def with_out():
    y = x
    for l in range(3):
        y = torch._addmm_activation(b[l], y, W[l].t(), use_gelu=False, out=hs[l])
    return y


def addmm_relu_out():
    y = x
    for l in range(3):
        y = torch.addmm(b[l], y, W[l].t(), out=hs[l]).relu_()
    return y


for name, fn in (("_addmm_activation", plain), ("_addmm_activation(out=)", with_out), ("addmm(out=).relu_()", addmm_relu_out)):
    for _ in range(10):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us for three layers (in a graph)")
assert torch.allclose(plain(), with_out(), atol=1e-4)
