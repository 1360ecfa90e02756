"""Why two correct fp32 implementations of the PPO step drift apart over a few steps (and why a float64 reference can disagree with
either by a percent in ONE bias gradient): the "DeepMind_6" MLP's first steps on tests/test_update_cpu.py::fake_batch — the step's
d(loss)/d(heads), top-layer dz and bias gradient against float64 autograd, and the ReLU gates that differ between the fp32
activations and the float64 ones (pre-activations of ~1e-9: a kink).  Round 5: step 1 has two such units in the top layer; the
top bias gradient then differs by 1.6 %, every other quantity by 1e-7.  (profiles/r05/r05n_relu_kink_probe.txt)"""
import os, sys; sys.path.insert(0, '/root/repo')
import torch, numpy as np
from brl_amd.models import make_forward_pass
from brl_amd.update import FusedMinibatch, make_optimizer, ppo_loss
from brl_amd.roll_out import Transition
from tests.test_update_cpu import CFG, fake_batch
dev = torch.device("cuda", 0)
model, mbs = "DeepMind_6", 256
tb, adv, tgt = fake_batch(4, mbs, seed=3)
B = 4 * mbs
flat = Transition(*[x.reshape((B,) + x.shape[2:]).cuda() for x in tb])
advf, tgtf = adv.reshape(-1).cuda(), tgt.reshape(-1).cuda()
net = make_forward_pass("relu", model).init(11, device=dev)
cfg = dict(CFG, minibatch_size=mbs, update_epochs=1)
opt = make_optimizer(cfg, net)["opt"]
fm = FusedMinibatch(cfg, net, opt, mbs, dev)
gen = torch.Generator(device=dev).manual_seed(5)
perms = [torch.randperm(B, device=dev, generator=gen) for _ in range(2)]
fm.begin_update(flat, advf, tgtf, perms)
for k in range(3):
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    mb = (fm.x0.clone(), fm.mask.clone(), fm.action.clone(), fm.old_v.clone(), fm.adv.clone(), fm.tgt.clone(), fm.old_lp.clone())
    fm.run_steps(1)
    torch.cuda.synchronize()
    ref = make_forward_pass("relu", model).init(11, device=dev).double()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            p.copy_(before[n].double())
    x = mb[0].double()
    hs = []
    for lin in ref.body:
        x = torch.relu(lin(x)); hs.append(x)
    logits, value = ref.actor(x), ref.critic(x).squeeze(-1)
    logits.retain_grad(); value.retain_grad()
    t = Transition(None, mb[2], mb[3].double(), None, mb[6].double(), None, mb[1].bool())
    total, _ = ppo_loss(dict(CFG), logits, value, t, mb[4].double(), mb[5].double())
    total.backward()
    want = torch.cat([logits.grad, value.grad[:, None]], 1).float()
    got = fm.dheads
    d = (want - got).abs()
    rows = torch.nonzero(d.max(1).values > 1e-7 + 1e-4 * want.abs().max()).reshape(-1)
    print("step", k, "dheads max diff", float(d.max()), "of", float(want.abs().max()), "rows off:", rows.tolist()[:10])
    hk = fm.h[-1]
    print("   h_top max diff", float((hk.double() - hs[-1]).abs().max()), " value diff", float((fm.head_parts.sum(0)[:, 38] + fm.bh[38] - value.detach().float()).abs().max()))
    Wh_before = torch.cat([before["actor.weight"], before["critic.weight"]], 0)
    dz_want = (got.double() @ Wh_before.double()) * (fm.h[-1] > 0)
    top = fm.nl - 1
    print("   dz_top max diff", float((fm.dzs[top].double() - dz_want).abs().max()), "of", float(dz_want.abs().max()),
          "| db_top (flat buffer) vs column sums of dz_top:", float((fm.Gb[top].double() - fm.dzs[top].double().sum(0)).abs().max()), "of", float(fm.Gb[top].abs().max()),
          "| vs autograd:", float((fm.Gb[top].double() - ref.body[top].bias.grad).abs().max()))
    ts = fm.tile_sums[top].view(-1, fm.H)[: (mbs + 15) // 16]
    print("   tile sums vs dz rows:", float((ts.double().sum(0) - fm.dzs[top].double().sum(0)).abs().max()),
          " autograd db_top vs column sums of the reference dz:", float((ref.body[top].bias.grad - dz_want.sum(0)).abs().max()))
    for l in range(fm.nl):
        g32, g64 = fm.h[l] > 0, hs[l] > 0
        flips = torch.nonzero(g32 != g64)
        if flips.numel():
            zs = [(int(i), int(j), float(fm.h[l][i, j]), float(hs[l][i, j])) for i, j in flips[:4].tolist()]
            print("   layer", l, "gate flips fp32 vs float64:", flips.shape[0], zs)
    for r in rows.tolist()[:3]:
        v = float(value[r]); ov = float(mb[3][r]); tg = float(mb[5][r])
        print("   row", r, "value", v, "old_v", ov, "v-old", v - ov, "tgt", tg, "want dv", float(want[r, 38]), "got dv", float(got[r, 38]),
              "max dlogit diff", float(d[r, :38].max()))
fm.end_update()
