"""Fused vs eager PPO update: per-parameter max abs difference after a few minibatch steps (debug helper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from brl_amd.models import make_forward_pass
from brl_amd.update import make_update_step, FusedMinibatch
from tests.test_update_cpu import CFG, fake_batch
tb, adv, tgt = fake_batch(4, 256, seed=3)
tb = type(tb)(*[x.cuda() for x in tb]); adv, tgt = adv.cuda(), tgt.cuda()
fp = make_forward_pass("relu", "DeepMind")
for epochs in (2,):
    nets = []
    for mode in ("eager", "graph-autograd", "fused"):
        net = fp.init(11, device="cuda")
        cfg = dict(CFG, minibatch_size=256, update_epochs=epochs, graph_update=mode != "eager", fused_update=mode == "fused")
        rs, (total, aux) = make_update_step(cfg, fp)((net, None, None, None, 0, 5), tb, adv, tgt)
        print(mode, type(rs[1].get("graphed")).__name__, rs[1].get("graph_error"), "total", [round(x, 7) for x in total.flatten().tolist()], "kl", [round(x, 9) for x in aux[3].flatten().tolist()])
        nets.append(net)
    for name, other in (("graph-autograd", nets[1]), ("fused", nets[2])):
        for (n, a), (_, b) in zip(nets[0].named_parameters(), other.named_parameters()):
            d = (a - b).abs()
            print(f"  epochs {epochs} eager vs {name:15s} {n:16s} max|diff| {float(d.max()):.3e}  mean {float(d.mean()):.3e}  max|p| {float(a.abs().max()):.3f}")
