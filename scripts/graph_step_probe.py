"""Would a hipGraph of [rollout, GAE] x 3 (one node per launch, rotating buffers) beat stream launches for the bench's step?
Timing experiment only: the captured rollouts re-use one draw index (their outputs are not checked)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, torch
import brl_amd
from brl_amd import _capi
from brl_amd.roll_out import alloc_transition
from bench import synthetic_lut
N, T, NB = 8192, 32, 3
env = brl_amd.BridgeBidding(lut=synthetic_lut(100000, 0))
dev = env.device
st = env.init(0, num_envs=N)
trajs = [alloc_transition(T, N, dev) for _ in range(NB)]
advs = [torch.empty((T, N), device=dev) for _ in range(NB)]; tgts = [torch.empty((T, N), device=dev) for _ in range(NB)]
lv = torch.zeros(N, device=dev); lo = torch.empty((N, 480), dtype=torch.bool, device=dev); lm = torch.empty((N, 38), dtype=torch.bool, device=dev)
tc = torch.zeros(1, dtype=torch.int64, device=dev)
ptrs = []
for tr in trajs:
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, getattr(tr, f).data_ptr())
    ptrs.append(p)
L, h = _capi.lib(), env._h
def step(i, stream):
    k = i % NB
    _capi.check(L.brl_rollout_random(h, st.packed.data_ptr(), N, T, 1, (i * T) & 0xFFFFFFFF, 7600.0, C.byref(ptrs[k]), lo.data_ptr(), lm.data_ptr(), tc.data_ptr(), stream))
    tr = trajs[k]
    _capi.check(L.brl_gae(h, tr.done.data_ptr(), tr.value.data_ptr(), tr.reward.data_ptr(), lv.data_ptr(), 1.0, 0.95, T, N, advs[k].data_ptr(), tgts[k].data_ptr(), stream))
s = torch.cuda.current_stream()
for i in range(30): step(i, s.cuda_stream)
torch.cuda.synchronize()
def timeit(fn, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for r in range(reps): fn(r)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t_stream = timeit(lambda r: [step(3 * r + j, s.cuda_stream) for j in range(3)], 200) / 3
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cs = torch.cuda.current_stream().cuda_stream
    for j in range(3): step(j, cs)
for _ in range(10): g.replay()
t_graph = timeit(lambda r: g.replay(), 200) / 3
g12 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g12):
    cs = torch.cuda.current_stream().cuda_stream
    for j in range(12): step(j, cs)
for _ in range(5): g12.replay()
t_graph12 = timeit(lambda r: g12.replay(), 60) / 12
print("per step: stream launches %.2f us, graph of 3 steps %.2f us, graph of 12 steps %.2f us" % (t_stream * 1e6, t_graph * 1e6, t_graph12 * 1e6))
