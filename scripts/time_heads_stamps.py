"""k_heads_loss in-kernel stamps (a -DHD_TIMING build of the library: brl_amd/lib/variants/libbrl_hip_hdtiming.so)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd import _capi
_capi.LIB_PATH = os.path.join(os.path.dirname(_capi.LIB_PATH), "variants", "libbrl_hip_hdtiming.so")
L, dev = _capi.lib(), torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
B, H = 1024, 1024
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)
h = rn(B, H).relu_(); Wh, bh = rn(39, H) / 32, rn(39) * 0.1
mask = (torch.rand(B, 38, device=dev, generator=g) < 0.6).to(torch.uint8); mask[:, 0] = 1
action = torch.multinomial(mask.float(), 1, generator=g)[:, 0].to(torch.int32)
old_v, old_lp, gae, tgt = rn(B) * 0.3, -rn(B).abs() - 0.1, rn(B), rn(B) * 0.3
groups, lgroups = B // 16, B // 4
dheads, heads = torch.empty(B, 39, device=dev), torch.zeros(B + 64, 39, device=dev)
partials, gram_p = torch.empty(lgroups, 8, device=dev), torch.empty(lgroups, 1444, device=dev)
for _ in range(400):
    _capi.check(L.brl_ppo_heads_loss(0, h.data_ptr(), H, Wh.data_ptr(), bh.data_ptr(), H, mask.data_ptr(), action.data_ptr(),
                                     old_v.data_ptr(), old_lp.data_ptr(), gae.data_ptr(), tgt.data_ptr(), B, 0.2, 0.5, 0.001, 1, 1, 0,
                                     heads.data_ptr(), dheads.data_ptr(), partials.data_ptr(), gram_p.data_ptr(), s))
torch.cuda.synchronize()
st = heads[B:B + 64, 0:5].cpu()
print("cycles since kernel entry (mean over 64 workgroups): loads+MFMA %.0f | exchange %.0f | own loss %.0f | all losses %.0f | end %.0f" % tuple(st.mean(0).tolist()))
print("max over workgroups:", st.max(0).values.tolist())
