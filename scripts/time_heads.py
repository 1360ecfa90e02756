"""Times the head kernels of the fused PPO minibatch step (brl_ppo_heads_loss_split / _bwd / brl_ppo_stats_gram / brl_bias_finalize_ex)
at minibatch 1024, hidden 1024: 200 launches between one HIP-event pair each."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd import _capi  # noqa: E402

L, dev = _capi.lib(), torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
B, H = 1024, 1024
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)  # noqa: E731
h = rn(B, H).relu_()
Wh, bh = rn(39, H) / 32, rn(39) * 0.1
mask = (torch.rand(B, 38, device=dev, generator=g) < 0.6).to(torch.uint8)
mask[:, 0] = 1
action = torch.multinomial(mask.float(), 1, generator=g)[:, 0].to(torch.int32)
old_v, old_lp, gae, tgt = rn(B) * 0.3, -rn(B).abs() - 0.1, rn(B), rn(B) * 0.3
groups, lgroups, nsplit = B // 16, B // 4, B // 64
dheads = torch.empty(B, 39, device=dev)
partials, gram_p = torch.empty(lgroups, 8, device=dev), torch.empty(lgroups, 1444, device=dev)
dwp, dbp = torch.empty(nsplit, 39 * H, device=dev), torch.empty(nsplit, 39, device=dev)
dh, ts = torch.empty(B, H, device=dev), torch.empty(groups, H, device=dev)
out = torch.zeros(8, device=dev)
row = torch.zeros(1, dtype=torch.int32, device=dev)
ssum, gsum, rows_out = torch.zeros(2560, 8, device=dev), torch.zeros(2560, 1444, device=dev), torch.zeros(2560, 8, device=dev)
hparts = torch.empty(4, B, 39, device=dev)
gW, gb, gbias = torch.empty(39, H, device=dev), torch.empty(39, device=dev), torch.empty(H, device=dev)
parts = (C.c_void_p * 3)(dwp.data_ptr(), dbp.data_ptr(), ts.data_ptr())
cols, tiles = (C.c_int64 * 3)(39 * H, 39, H), (C.c_int64 * 3)(nsplit, nsplit, groups)
outs = (C.c_void_p * 3)(gW.data_ptr(), gb.data_ptr(), gbias.data_ptr())


def loss():
    _capi.check(L.brl_ppo_heads_loss_split(0, h.data_ptr(), H, Wh.data_ptr(), bh.data_ptr(), H, mask.data_ptr(), action.data_ptr(),
                                           old_v.data_ptr(), old_lp.data_ptr(), gae.data_ptr(), tgt.data_ptr(), B, 0.2, 0.5, 0.001, 1, 1, 0,
                                           None, dheads.data_ptr(), partials.data_ptr(), gram_p.data_ptr(), hparts.data_ptr(), 4, s))


def bwd():
    _capi.check(L.brl_ppo_heads_bwd(0, dheads.data_ptr(), h.data_ptr(), H, Wh.data_ptr(), B, H, 0, nsplit, dwp.data_ptr(), dbp.data_ptr(),
                                    dh.data_ptr(), ts.data_ptr(), partials.data_ptr(), gram_p.data_ptr(), lgroups, row.data_ptr(),
                                    ssum.data_ptr(), gsum.data_ptr(), s))


def stats():
    _capi.check(L.brl_ppo_stats_gram(0, partials.data_ptr(), lgroups, B, gram_p.data_ptr(), lgroups, 0.5, 0.001, 0.0, out.data_ptr(), None, None, s))


def stats_rows():
    _capi.check(L.brl_ppo_stats_rows(0, ssum.data_ptr(), gsum.data_ptr(), 2560, B, 0.5, 0.001, 0.0, rows_out.data_ptr(), s))


def fin():
    _capi.check(L.brl_bias_finalize_ex(0, 3, parts, cols, tiles, outs, s))


def act_bwd():
    _capi.check(L.brl_act_bwd_colsum(0, dh.data_ptr(), h.data_ptr(), B, H, H, 0, ts.data_ptr(), s))


for name, fn in (("brl_ppo_heads_loss_split (ksplit 4)", loss), ("brl_ppo_heads_bwd", bwd), ("brl_ppo_stats_gram", stats), ("brl_ppo_stats_rows (2560 rows)", stats_rows), ("brl_bias_finalize_ex (3 segments)", fin),
                 ("brl_act_bwd_colsum", act_bwd)):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 200 * 1e3:7.2f} us per launch (back to back, launch gap included)")
