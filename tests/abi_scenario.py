"""One parity scenario written against the C-ABI of include/brl_hip.h ONLY (ctypes, raw pointers) — SURVEY §8b: the CPU
oracle exports the same symbols (oracle/brl_shim.c -> oracle/_build/liboracle_brl.so, host pointers), the product
library takes device pointers.  `run(lib, mem)` drives either through init / step / auto-reset / observe / get_fields /
duplicate_step / fused random rollout / GAE / IMP and returns every output as numpy arrays; the tests compare the two."""
import ctypes as C

import numpy as np

from brl_amd._capi import Fields, TableInfoPtrs, TransitionPtrs

i64, i32, u32, u64, f32, vp = C.c_int64, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_void_p


class HostMem:
    """numpy arrays, host pointers (the oracle shim)"""

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype)

    def put(self, a):
        return np.ascontiguousarray(a)

    def ptr(self, a):
        return None if a is None else a.ctypes.data

    def get(self, a):
        return np.array(a)

    stream = None


class DeviceMem:
    """torch tensors on the GPU, device pointers (libbrl_hip.so)"""

    def __init__(self):
        import torch
        self.torch = torch

    def empty(self, shape, dtype):
        t = self.torch
        dt = {np.uint8: t.uint8, np.int32: t.int32, np.int64: t.int64, np.float32: t.float32, np.uint32: t.int32,
              np.uint64: t.int64}[dtype]
        return t.zeros(shape, dtype=dt, device="cuda")

    def put(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        return self.torch.from_numpy(a).cuda()

    def ptr(self, a):
        return None if a is None else a.data_ptr()

    def get(self, a):
        self.torch.cuda.synchronize()
        return a.cpu().numpy()

    @property
    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream


def declare(L):
    L.brl_last_error.restype = C.c_char_p
    L.brl_create.argtypes = [i32, vp, vp, i64, C.POINTER(vp)]
    L.brl_set_rng.argtypes = [vp, u64, u64]
    L.brl_destroy.argtypes = [vp]
    L.brl_init_random.argtypes = [vp, vp, i64, u32, vp]
    L.brl_init_from_deals.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.brl_step.argtypes = [vp, vp, vp, i64, vp, i32, vp, vp, vp, vp, vp, vp]
    L.brl_observe.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.brl_get_fields.argtypes = [vp, vp, i64, C.POINTER(Fields), vp]
    L.brl_rollout_random.argtypes = [vp, vp, i64, i32, i32, u32, f32, C.POINTER(TransitionPtrs), vp, vp, vp, vp]
    L.brl_gae.argtypes = [vp, vp, vp, vp, vp, f32, f32, i32, i64, vp, vp, vp]
    L.brl_imp_reward.argtypes = [vp, vp, vp, vp, i64, vp]
    L.brl_duplicate_step.argtypes = [vp, vp, vp, i64, vp, C.POINTER(TableInfoPtrs), C.POINTER(TableInfoPtrs), vp, vp, vp, vp, vp, vp]
    return L


def run(L, mem, keys, values, n=257, steps=60, seed=20240611):
    declare(L)
    out = {}

    def ok(rc):
        assert rc == 0, L.brl_last_error().decode()

    h = vp()
    ok(L.brl_create(0, keys.ctypes.data, values.ctypes.data, len(keys), C.byref(h)))
    ok(L.brl_set_rng(h, seed, 1000))
    st = mem.empty((n, 16), np.uint64)
    ok(L.brl_init_random(h, mem.ptr(st), n, 0, mem.stream))
    obs, mask = mem.empty((n, 480), np.uint8), mem.empty((n, 38), np.uint8)
    rew, term, cur = mem.empty((n, 4), np.float32), mem.empty((n,), np.uint8), mem.empty((n,), np.int32)
    ok(L.brl_observe(h, mem.ptr(st), n, None, mem.ptr(obs), mem.ptr(mask), mem.stream))
    out["obs0"], out["mask0"] = mem.get(obs), mem.get(mask)
    rng = np.random.default_rng(7)
    trace = []
    m = out["mask0"]
    for t in range(steps):  # random legal calls, auto-reset on; alternately in place and into a second state array
        act = (rng.random(m.shape) * m).argmax(1).astype(np.int32)
        act[rng.random(n) < 0.4] = 0
        a_d = mem.put(act)
        dst = st if t % 2 == 0 else mem.empty((n, 16), np.uint64)
        ok(L.brl_step(h, mem.ptr(st), mem.ptr(dst), n, mem.ptr(a_d), 1, mem.ptr(obs), mem.ptr(mask), mem.ptr(rew),
                      mem.ptr(term), mem.ptr(cur), mem.stream))
        st = dst
        m = mem.get(mask)
        trace.append((mem.get(obs), m, mem.get(rew), mem.get(term), mem.get(cur)))
    for k, name in enumerate(("obs", "mask", "rewards", "terminated", "current_player")):
        out["step_" + name] = np.stack([x[k] for x in trace])
    # every pgx-named field of the final state
    f = Fields()
    bufs = {}
    spec = {"current_player": ((n,), np.int32), "terminated": ((n,), np.uint8), "rewards": ((n, 4), np.float32),
            "step_count": ((n,), np.int32), "turn": ((n,), np.int32), "dealer": ((n,), np.int32), "vul_ns": ((n,), np.uint8),
            "vul_ew": ((n,), np.uint8), "shuffled_players": ((n, 4), np.int32), "last_bid": ((n,), np.int32),
            "last_bidder": ((n,), np.int32), "call_x": ((n,), np.uint8), "call_xx": ((n,), np.uint8),
            "pass_num": ((n,), np.int32), "first_denomination_ns": ((n, 5), np.int32),
            "first_denomination_ew": ((n, 5), np.int32), "hand": ((n, 52), np.int32), "tricks": ((n, 20), np.uint8),
            "lut_idx": ((n,), np.int32), "board_ctr": ((n,), np.uint32), "illegal": ((n,), np.uint8)}
    for name, (shape, dt) in spec.items():
        bufs[name] = mem.empty(shape, dt)
        setattr(f, name, mem.ptr(bufs[name]))
    ok(L.brl_get_fields(h, mem.ptr(st), n, C.byref(f), mem.stream))
    for name in spec:
        v = mem.get(bufs[name])
        out["field_" + name] = np.sort(v.reshape(n, 4, 13), 2).reshape(n, 52) if name == "hand" else v
    # observe from every player's point of view
    for p in range(4):
        pid = mem.put(np.full(n, p, np.int32))
        ok(L.brl_observe(h, mem.ptr(st), n, mem.ptr(pid), mem.ptr(obs), None, mem.stream))
        out[f"observe_p{p}"] = mem.get(obs)
    # fused random rollout (state carried on), then GAE over its columns
    T = 12
    tr = {"done": mem.empty((T, n), np.uint8), "action": mem.empty((T, n), np.int32), "value": mem.empty((T, n), np.float32),
          "reward": mem.empty((T, n), np.float32), "log_prob": mem.empty((T, n), np.float32),
          "obs": mem.empty((T, n, 480), np.uint8), "legal_action_mask": mem.empty((T, n, 38), np.uint8)}
    p = TransitionPtrs()
    for name in TransitionPtrs._names:
        setattr(p, name, mem.ptr(tr[name]))
    tc = mem.empty((1,), np.int64)
    lo, lm = mem.empty((n, 480), np.uint8), mem.empty((n, 38), np.uint8)
    ok(L.brl_rollout_random(h, mem.ptr(st), n, T, 1, 5, 7600.0, C.byref(p), mem.ptr(lo), mem.ptr(lm), mem.ptr(tc), mem.stream))
    for name in tr:
        out["rollout_" + name] = mem.get(tr[name])
    out["rollout_last_obs"], out["rollout_last_mask"], out["rollout_count"] = mem.get(lo), mem.get(lm), mem.get(tc)
    adv, tgt = mem.empty((T, n), np.float32), mem.empty((T, n), np.float32)
    lv = mem.put(rng.standard_normal(n).astype(np.float32))
    ok(L.brl_gae(h, mem.ptr(tr["done"]), mem.ptr(tr["value"]), mem.ptr(tr["reward"]), mem.ptr(lv), 0.99,
                 float(np.float32(0.99 * 0.95)), T, n, mem.ptr(adv), mem.ptr(tgt), mem.stream))
    out["gae_adv"], out["gae_tgt"] = mem.get(adv), mem.get(tgt)
    # duplicate pairing to completion on fresh boards, greedy-ish random calls
    ok(L.brl_init_random(h, mem.ptr(st), n, 50, mem.stream))
    info = []
    for _ in range(2):
        d = {"terminated": mem.empty((n,), np.uint8), "rewards": mem.empty((n, 4), np.float32),
             "last_bid": mem.put(np.full(n, -1, np.int32)), "last_bidder": mem.put(np.full(n, -1, np.int32)),
             "call_x": mem.empty((n,), np.uint8), "call_xx": mem.empty((n,), np.uint8)}
        q = TableInfoPtrs()
        for name in TableInfoPtrs._names:
            setattr(q, name, mem.ptr(d[name]))
        info.append((d, q))
    ok(L.brl_observe(h, mem.ptr(st), n, None, None, mem.ptr(mask), mem.stream))
    m = mem.get(mask)
    cum = np.zeros(n, np.float32)
    for it in range(400):
        act = (rng.random(m.shape) * m).argmax(1).astype(np.int32)
        act[rng.random(n) < 0.5] = 0
        a_d = mem.put(act)
        ok(L.brl_duplicate_step(h, mem.ptr(st), mem.ptr(st), n, mem.ptr(a_d), C.byref(info[0][1]), C.byref(info[1][1]),
                                None, mem.ptr(mask), mem.ptr(rew), mem.ptr(term), None, mem.stream))
        m = mem.get(mask)
        cum += mem.get(rew)[:, 0]
        if mem.get(term).all():
            break
    out["dup_iterations"], out["dup_cum_imp"] = np.array(it), cum
    for tag, (d, _) in zip("AB", info):
        for name in TableInfoPtrs._names:
            out[f"dup_{tag}_{name}"] = mem.get(d[name])
    a = mem.put((rng.integers(-760, 761, (n, 1)) * 10 * np.array([1, 1, -1, -1])).astype(np.float32))
    b = mem.put((rng.integers(-760, 761, (n, 1)) * 10 * np.array([1, 1, -1, -1])).astype(np.float32))
    o = mem.empty((n, 4), np.float32)
    ok(L.brl_imp_reward(h, mem.ptr(a), mem.ptr(b), mem.ptr(o), n, mem.stream))
    out["imp"] = mem.get(o)
    ok(L.brl_destroy(h))
    return out
