"""Timing build (-DBRL_TIMING): per-wave total vs barrier-wait cycles of k_rollout_ws."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libbrl_timing.so"
SRC = sorted(__import__("glob").glob(os.path.join(ROOT, "brl_amd/csrc/*.hip")))   # (every unit of the library)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                       "-DBRL_TIMING"] + os.environ.get("FLAGS", "").split() + ["-o", so] + SRC, stderr=subprocess.DEVNULL)
from brl_amd import _capi
_capi.LIB_PATH = so
import numpy as np, torch, ctypes as C
import brl_amd
from brl_amd.roll_out import alloc_transition
from brl_amd.bridge_bidding import _stream
from bench import synthetic_lut
N, T = 8192, 32
keys, values = synthetic_lut(100000, 0)
for cfg in os.environ.get("CFGS", "32x16").split(","):
    tpb, nw = map(int, cfg.split("x"))
    nw_dump = int(os.environ.get("NW_DUMP", nw))  # waves per workgroup of the kernel actually launched (pipe: +NP)
    os.environ["BRL_ROLLOUT_WS"] = cfg
    os.environ["BRL_DEBUG"] = os.environ.get("DBG", "0")
    env = brl_amd.BridgeBidding(lut=(keys, values))
    NB = int(os.environ.get("NBUF", "1"))
    trajs = [alloc_transition(T, N, env.device) for _ in range(NB)]
    st = env.init(0, num_envs=N)
    ps = []
    for traj in trajs:
        p = _capi.TransitionPtrs()
        for f in _capi.TransitionPtrs._names:
            setattr(p, f, _capi.ptr(getattr(traj, f)))
        ps.append(p)
    nblk = (N + tpb - 1) // tpb
    dump = torch.zeros(nblk * nw_dump * 2 + nblk * nw_dump * 32, dtype=torch.int64, device=env.device)
    for i in range(5 if NB == 1 else 4 * NB + 1):
        _capi.check(_capi.lib().brl_rollout_random(env._h, _capi.ptr(st.packed), N, T, 1, i * T, 7600.0, C.byref(ps[i % NB]), None, None, _capi.ptr(dump), _stream()))
    torch.cuda.synchronize()
    full = dump.cpu().numpy()
    nw = nw_dump
    d = full[:nblk * nw * 2].reshape(nblk, nw, 2)
    if int(os.environ.get('DBG', '0')) & 256:
        tl = full[nblk * nw * 2:].reshape(nblk, nw, 16, 2)
        wg = 5
        print('timeline of workgroup', wg, '(cycles since wave start): arrive->release per barrier')
        for w in range(nw):
            print(f'  wave {w:2d}: ' + ' '.join(f'{int(a)//100:4d}>{int(r)//100:4d}' for a, r in tl[wg, w] if r))
        print('  total', d[wg, :, 0] // 100)
        sc = full[nblk * nw * 2:].reshape(nblk, nw, 32)[:, 2, 29:31].mean(0)
        print('scorer wave: pass 1 %.0f cycles, pass 2 %.0f cycles per launch (rest of its work: pass 3 + bookkeeping)' % (sc[0], sc[1]))
        if int(os.environ.get('DBG', '0')) & 1024:
            pr = full[nblk * nw * 2:].reshape(nblk, nw, 32)[:, 3:, 29:32].sum((0, 1))
            print('emit waves, per sub-step: command read + image update + deals %.0f cycles, copy + bookkeeping %.0f cycles, n=%d' % (pr[0] / pr[2], pr[1] / pr[2], pr[2]))
        continue
    tot, wait = d[..., 0].mean(0), d[..., 1].mean(0)
    per_block = d[..., 0].max(1)
    print("per-workgroup launch length (slowest wave): mean %.0f  p50 %.0f  p95 %.0f  max %.0f" % (
        per_block.mean(), np.percentile(per_block, 50), np.percentile(per_block, 95), per_block.max()))
    print(cfg, "cycles(100MHz ticks?) per wave role: total / barrier-wait / work")
    if int(os.environ.get("DBG", "0")) & 64:
        seg01 = d[..., 0].mean(0); seg2 = (d[..., 1] & 0xFFFFFFFF).mean(0); seg3 = (d[..., 1] >> 32).mean(0)
        for w in range(3, nw):
            print(f"  emit wave {w:2d}: apply {seg01[w]:9.0f}  obs-copy {seg2[w]:9.0f}  mask {seg3[w]:9.0f}")
        continue
    for w in range(nw):
        print(f"  wave {w:2d}: total {tot[w]:9.0f} wait {wait[w]:9.0f} work {tot[w]-wait[w]:9.0f}")
