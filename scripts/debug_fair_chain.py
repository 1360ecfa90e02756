"""FusedFair with / without brl_fair_chain on the same minibatch (the numpy test's set-up): where do they part?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brl_amd.models import make_forward_pass
from brl_amd.roll_out import Transition
from brl_amd.update import make_update_step
from tests.test_update_cpu import CFG, fake_batch

tb, adv, tgt = fake_batch(4, 256, seed=2)
B = 1024
out = {}
for chain in (False, True):
    cfg = dict(CFG, minibatch_size=B, update_epochs=1, lr=1e-3, fair_chain=chain)
    fp = make_forward_pass("relu", "FAIR")
    net = fp.init(4, device="cuda")
    rs, (total, aux) = make_update_step(cfg, fp)((net, None, None, None, 0, 9), Transition(*[x.cuda() for x in tb]), adv.cuda(), tgt.cuda())
    fm = rs[1]["graphed"]
    d = {"total": total.cpu().numpy(), "aux": [a.cpu().numpy() for a in aux], "x4": fm.t["x4"].cpu().numpy(), "x0": fm.x0.cpu().numpy(),
         "stat": fm.stat_sums[:2].cpu().numpy(), "inp": fm.inp.cpu().numpy(), "dzs": fm.dzs.cpu().numpy(),
         "dh": (fm.dheads[:, :39] if chain else torch.cat([fm.dlogits, fm.dvalue[:, None]], 1)).cpu().numpy(),
         "adv": fm.adv.cpu().numpy(), "mask": fm.mask.cpu().numpy()}
    if chain:
        d["cpart"] = fm.cpartials.cpu().numpy().sum(0)
    else:
        d["cpart"] = fm.partials.cpu().numpy().sum(0)
    out[chain] = d
a, b = out[False], out[True]
print("total", a["total"], b["total"])
print("aux", [float(x[0, 0]) for x in a["aux"]], [float(x[0, 0]) for x in b["aux"]])
print("stat rows", a["stat"], b["stat"])
print("partials sums", a["cpart"], b["cpart"])
for k in ("x0", "adv", "mask", "x4", "dh"):
    print(k, float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()))
print("inp", [float(np.abs(a["inp"][i] - b["inp"][i]).max()) for i in range(9)])
print("dzs", [float(np.abs(a["dzs"][i] - b["dzs"][i]).max()) for i in range(9)])
