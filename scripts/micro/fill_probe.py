"""What the chip sustains for plain device fills of the launch's byte count (148 MB), rotating over 3 buffers:
torch fill_ (a vectorised store kernel) and hipMemsetAsync.  The floor k_rollout_fs is compared with."""
import torch, ctypes as C
N = 148_341_545 // 16 * 16
bufs = [torch.empty(N, dtype=torch.uint8, device="cuda") for _ in range(3)]
s = torch.cuda.current_stream()
hip = C.CDLL("libamdhip64.so")
def run(fn, reps=400):
    for i in range(50): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rep in range(2):
    t_fill = run(lambda i: bufs[i % 3].fill_(1))
    v32 = [b.view(torch.int32) for b in bufs]
    t_fill32 = run(lambda i: v32[i % 3].fill_(7))
    t_memset = run(lambda i: hip.hipMemsetAsync(C.c_void_p(bufs[i % 3].data_ptr()), 1, C.c_size_t(N), C.c_void_p(s.cuda_stream)))
    print(f"fill_ u8 {t_fill:.2f} us ({N / t_fill / 1e6:.2f} TB/s)  fill_ i32 {t_fill32:.2f} us ({N / t_fill32 / 1e6:.2f} TB/s)  hipMemsetAsync {t_memset:.2f} us ({N / t_memset / 1e6:.2f} TB/s)")
