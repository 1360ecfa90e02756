"""-m gpu: the shards of ranks > 0 (BASELINE.json configs[4]: 65 536 tables = 8 ranks x 8192) on the ONE GPU of the box.

Rank r owns the global tables [r * num_envs, (r + 1) * num_envs) (`env_offset`, ppo.py:305,318 are the only init sites of the
reference): every board it deals and every action it draws comes from the counter-based stream of the table's
GLOBAL index, so a shard must equal the same slice of one big single-process rollout.  These tests run the device path at
`env_offset != 0` — incl. offsets >= 2^32, where the fourth Philox counter word (`eid >> 32`) is non-zero — against the oracle,
bit for bit, as tests/test_gpu_parity.py does for rank 0."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.gpu_util import assert_state_equal, to_np
from tests.test_gpu_parity import POLICY_CFG, make_env, replay_policy_rollout

pytestmark = pytest.mark.gpu

OFFSETS = [8192, 7 * 8192, 2 ** 32 + 5]
COLUMNS = ("obs", "legal_action_mask", "action", "done", "value", "reward", "log_prob")


def _env_at(dds, offset, k=4, ws=None, lut=None):
    e = make_env(dds, k, ws, lut=lut)
    e.env_offset = int(offset)     # (env.init re-keys the handle with its current env_offset: brl_set_rng)
    return e


@pytest.mark.parametrize("offset", OFFSETS)
@pytest.mark.parametrize("ws,substeps,n,T", [(None, 1, 2048, 32),    # k_rollout_fs
                                              (None, 1, 1000, 16),    # k_rollout_ws (n % 32 != 0)
                                              (None, 4, 512, 12),     # the competitive macro-step, every seat random
                                              ("0", 1, 515, 16)])     # k_rollout_random<K>
def test_fused_random_rollout_of_a_rank_shard_matches_oracle(dds, oracle, offset, ws, substeps, n, T):
    """test_fused_random_rollout_matches_oracle at env_offset = r * 8192 (ranks 1 and 7 of configs[4]) and beyond 2^32: two
    consecutive rollouts (state, re-deal counters and the draw counter carry over), all seven columns + the packed state."""
    import brl_amd
    env = _env_at(dds, offset, 4, ws)
    cfg = {"num_steps": T, "game_mode": "competitive" if substeps == 4 else "normal", "substeps": substeps, "reward_scale": 7600}
    roll = brl_amd.make_random_roll_out(cfg, env)
    seed = 2024
    st = env.init(seed, num_envs=n)
    ref = oracle.init_random(n, seed=seed, env_offset=offset)
    assert_state_equal(st, ref, where=f"offset {offset}: init")
    rs = (None, None, st, st.observation, 0, 0)
    draw = 0
    for call in range(2):
        rs, traj = roll(rs)
        want = oracle.rollout_random(ref, T, seed=seed, substeps=substeps, draw_base=draw, env_offset=offset)
        draw += T * substeps
        torch.cuda.synchronize()
        for name in COLUMNS:
            g, o = to_np(getattr(traj, name)), want[name]
            assert g.shape == o.shape and np.array_equal(g, o), f"offset {offset} ws={ws} sub={substeps} call {call}: {name}"
        assert_state_equal(rs[2], ref, where=f"offset {offset} ws={ws} sub={substeps} call {call}: final state")
        assert np.array_equal(to_np(rs[3]), ref["observation"])
    assert float(ref["board_ctr"].mean()) >= 1     # a re-deal per table on average: the shard's re-deal stream was exercised
    # ... and the shard is NOT rank 0's: the same call at offset 0 deals other boards
    ref0 = oracle.init_random(n, seed=seed)
    assert not np.array_equal(ref0["lut_idx"], oracle.init_random(n, seed=seed, env_offset=offset)["lut_idx"])


@pytest.mark.parametrize("offset,n,T", [(8192, 8192, 32), (7 * 8192, 1024, 32), (2 ** 32 + 5, 2048, 40), (2 ** 32 + 5, 96, 64)])
def test_rollout_random_gae_in_one_launch_of_a_rank_shard(dds, oracle, offset, n, T):
    """The bench's own entry point (brl_rollout_random_gae: Transition + calc_gae in one launch) on a rank > 0 shard, two
    consecutive launches: seven columns, last_obs / mask, packed state, terminated_count, advantages / targets vs the oracle."""
    from brl_amd import _capi
    from brl_amd.bridge_bidding import State
    from brl_amd.roll_out import alloc_transition
    e = _env_at(dds, offset)
    dev = e.device
    seed = 31
    st = e.init(seed, num_envs=n)
    ref = oracle.init_random(n, seed=seed, env_offset=offset)
    traj = alloc_transition(T, n, dev)
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, getattr(traj, f).data_ptr())
    lo = torch.empty((n, 480), dtype=torch.bool, device=dev); lm = torch.empty((n, 38), dtype=torch.bool, device=dev)
    tc = torch.zeros(1, dtype=torch.int64, device=dev)
    adv = torch.empty((T, n), device=dev); tgt = torch.empty((T, n), device=dev)
    rng = np.random.default_rng(n)
    last_val = rng.standard_normal(n).astype(np.float32)
    lv = torch.from_numpy(last_val).to(dev)
    gamma, lam = 0.99, 0.95
    gl = float(torch.tensor(gamma * lam, dtype=torch.float32))
    total = 0
    for step in range(2):
        _capi.check(_capi.lib().brl_rollout_random_gae(e._h, st.packed.data_ptr(), n, T, step * T, 7600.0, C.byref(p), lo.data_ptr(),
                                                       lm.data_ptr(), tc.data_ptr(), lv.data_ptr(), gamma, gl, adv.data_ptr(),
                                                       tgt.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        want = oracle.rollout_random(ref, T, seed=seed, draw_base=step * T, env_offset=offset)
        for name in COLUMNS:
            assert np.array_equal(to_np(getattr(traj, name)).astype(want[name].dtype), want[name]), f"offset {offset} step {step}: {name}"
        assert np.array_equal(to_np(lo), ref["observation"]) and np.array_equal(to_np(lm), ref["legal_action_mask"])
        assert_state_equal(State(e, st.packed), ref, where=f"offset {offset} step {step}: packed state")
        total += want["terminated_count"]
        assert int(tc.item()) == total
        wa, wt = oracle.gae(want["done"], want["value"], want["reward"], last_val, gamma, lam)
        assert np.array_equal(to_np(adv), wa) and np.array_equal(to_np(tgt), wt), f"offset {offset} step {step}: advantages / targets"
    assert total > 0


@pytest.mark.parametrize("offset,n,T,graph", [(8192, 2048, 32, True), (2 ** 32 + 5, 700, 9, False), (7 * 8192, 8192, 32, True)])
def test_policy_rollout_of_a_rank_shard_replays_through_oracle(dds, oracle, offset, n, T, graph):
    """configs[3] / [4]'s rollout (MLPs in the loop, hipGraph) on the shard of rank 1 / rank 7: the recorded actions replayed through
    the oracle's auto_reset(step) at the same env_offset — obs, mask, done, reward, final state, terminated_count bit-exact; the
    sampled actions legal.  (The re-deals inside the captured sub-steps read the key + offset from the device-resident mirror.)"""
    import brl_amd
    from brl_amd.models import make_forward_pass
    env = _env_at(dds, offset)
    cfg = dict(POLICY_CFG, num_steps=T, graph_rollout=graph)
    fp = make_forward_pass("relu", "DeepMind")
    actor, opp = fp.init(0, device="cuda"), fp.init(1, device="cuda")
    roll = brl_amd.make_roll_out(cfg, env, fp, fp)
    seed = 4242
    st = env.init(seed, num_envs=n)
    ref = oracle.init_random(n, seed=seed, env_offset=offset)
    rs = (actor, None, st, st.observation, 0, 0)
    total = 0
    for call in range(2 if n < 8192 else 1):
        rs, traj = roll(rs, opp)
        torch.cuda.synchronize()
        want = replay_policy_rollout(oracle, ref, traj, roll.sub_actions, seed, env_offset=offset)
        where = f"offset {offset} n={n} T={T} graph={graph} call {call}"
        for name in ("obs", "legal_action_mask", "done", "reward"):
            assert np.array_equal(to_np(getattr(traj, name)), want[name]), f"{where}: {name}"
        act = to_np(traj.action)
        assert np.take_along_axis(want["legal_action_mask"], act[..., None].astype(np.int64), 2).all(), f"{where}: sampled action illegal"
        assert_state_equal(rs[2], ref, where=f"{where}: final state")
        total += want["terminated_count"]
        assert int(rs[4].item()) == total
    assert total > 0


def test_config4_as_eight_shards_equals_one_rollout_of_65536_tables():
    """BASELINE.json configs[4] (num_envs = 65 536 sharded over 8 GPUs) BY SHARDS on one GPU: the eight shards r * 8192, r = 0..7,
    each through the bench's own entry point (brl_rollout_random_gae, 8192 x 32, the bench's 100 000-row table) one after another,
    laid side by side along N == ONE oracle rollout of 65 536 tables from one init: seven columns, advantages / targets, the packed
    state of every table, the summed terminated_count.  Two consecutive rollouts (draw counter and boards carry over)."""
    import brl_amd
    from bench import LUT_LEN, NUM_ENVS, NUM_STEPS, synthetic_lut as bench_lut
    from brl_amd import _capi
    from brl_amd.bridge_bidding import State
    from brl_amd.roll_out import alloc_transition
    from oracle import Oracle
    keys, values = bench_lut(LUT_LEN, 0)
    orc = Oracle(keys, values)
    world, n, T = 8, NUM_ENVS, NUM_STEPS
    seed = 0
    ref = orc.init_random(world * n, seed=seed)                     # ONE process, 65 536 tables
    envs, states = [], []
    for r in range(world):
        e = brl_amd.BridgeBidding(lut=(keys, values), env_offset=r * n)
        envs.append(e)
        states.append(e.init(seed, num_envs=n))
    dev = envs[0].device
    traj = alloc_transition(T, n, dev)
    p = _capi.TransitionPtrs()
    for f in _capi.TransitionPtrs._names:
        setattr(p, f, getattr(traj, f).data_ptr())
    lo = torch.empty((n, 480), dtype=torch.bool, device=dev); lm = torch.empty((n, 38), dtype=torch.bool, device=dev)
    adv = torch.empty((T, n), device=dev); tgt = torch.empty((T, n), device=dev)
    gl = float(torch.tensor(1.0 * 0.95, dtype=torch.float32))
    last_val = torch.zeros(n, dtype=torch.float32, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    total = 0
    for step in range(2):
        want = orc.rollout_random(ref, T, seed=seed, draw_base=step * T)             # [T, 65536, ...]
        wa, wt = orc.gae(want["done"], want["value"], want["reward"], np.zeros(world * n, np.float32), 1.0, 0.95)
        total += want["terminated_count"]
        for r in range(world):
            e, st = envs[r], states[r]
            _capi.check(_capi.lib().brl_rollout_random_gae(e._h, st.packed.data_ptr(), n, T, step * T, 7600.0, C.byref(p),
                                                           lo.data_ptr(), lm.data_ptr(), counts[r].data_ptr(), last_val.data_ptr(),
                                                           1.0, gl, adv.data_ptr(), tgt.data_ptr(),
                                                           torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            sl = slice(r * n, (r + 1) * n)
            for name in brl_amd.Transition._fields:
                w = want[name][:, sl]
                assert np.array_equal(to_np(getattr(traj, name)).astype(w.dtype), w), f"step {step} shard {r}: {name}"
            assert np.array_equal(to_np(adv), wa[:, sl]) and np.array_equal(to_np(tgt), wt[:, sl]), f"step {step} shard {r}: GAE"
            assert np.array_equal(to_np(lo), ref["observation"][sl]) and np.array_equal(to_np(lm), ref["legal_action_mask"][sl])
            assert_state_equal(State(e, st.packed), ref[sl], where=f"step {step} shard {r}: packed state")
        assert sum(int(c.item()) for c in counts) == total
    assert total > 2 * world * n


def test_sharded_evaluator_plays_its_global_boards(dds, oracle):
    """ppo.py:366-381 under a process group: rank 1 of 2 plays the global boards [501, 1001) of a 1001-board duplicate evaluation
    (`_Shard`: env_offset = 501).  Its recorded calls replayed through the oracle's duplicate_step from
    init_random(500, env_offset=501): both tables' snapshots equal.  (A world-1 gloo group stands in for the all-reduce.)"""
    import socket
    import torch.distributed as dist
    import brl_amd
    from brl_amd.evaluation import make_simple_duplicate_evaluate
    from brl_amd.models import make_forward_pass
    from oracle import Oracle
    env = brl_amd.BridgeBidding(lut=(dds["keys"], dds["values"]))
    fp = make_forward_pass("relu", "DeepMind")
    t1, t2 = fp.init(3, device="cuda"), fp.init(4, device="cuda")
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        calls = []
        ev = make_simple_duplicate_evaluate(env, "relu", "DeepMind", "relu", "DeepMind", 1001, shard=(1, 2), record_calls=calls)
        _, A, B = ev(t1, t2, 99)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    n, offset = 500, 501
    ref = oracle.init_random(n, seed=99, env_offset=offset)
    oA, oB = Oracle.table_info_from(ref), Oracle.table_info_from(ref)
    for a in calls:
        act = to_np(a)
        idle = act < 0
        keep = (ref[idle].copy(), oA[idle].copy(), oB[idle].copy())
        oracle.duplicate_step(ref, np.where(idle, 0, act).astype(np.int32), oA, oB)
        ref[idle], oA[idle], oB[idle] = keep
    assert ref["terminated"].all() and oA["terminated"].all() and oB["terminated"].all()
    for T_, oT in ((A, oA), (B, oB)):
        for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
            assert np.array_equal(to_np(getattr(T_, f)).astype(np.float64), oT[f].astype(np.float64)), f


def test_empty_batches_are_no_ops(dds, oracle):
    """n = 0 (an evaluator shard of a rank that got no board, a filtered batch): every handle-taking entry point returns OK without
    a launch — through the host mirror and through the raw C-ABI with NULL arrays — and leaves the handle usable."""
    import brl_amd
    from brl_amd import _capi
    env = brl_amd.BridgeBidding(lut=(dds["keys"], dds["values"]))
    st = env.init(5, num_envs=0)
    assert st.packed.shape == (0, 16) and st.observation.shape == (0, 480) and st.legal_action_mask.shape == (0, 38)
    st2 = env.step(st, torch.zeros(0, dtype=torch.int32), autoreset=True)
    assert st2.packed.shape == (0, 16) and st2.rewards.shape == (0, 4)
    L, h, s = _capi.lib(), env._h, torch.cuda.current_stream().cuda_stream
    p = _capi.TransitionPtrs()
    assert L.brl_init_random(h, None, 0, 0, s) == 0
    assert L.brl_step(h, None, None, 0, None, 1, None, None, None, None, None, s) == 0
    assert L.brl_observe(h, None, 0, None, None, None, s) == 0
    assert L.brl_rollout_random(h, None, 0, 32, 1, 0, 7600.0, C.byref(p), None, None, None, s) == 0
    assert L.brl_rollout_random_gae(h, None, 0, 32, 0, 7600.0, C.byref(p), None, None, None, None, 1.0, 0.95, None, None, s) == 0
    assert L.brl_gae(h, None, None, None, None, 1.0, 0.95, 32, 0, None, None, s) == 0
    assert L.brl_imp_reward(h, None, None, None, 0, s) == 0
    # ... and a negative n is an argument error with a message, not a launch
    assert L.brl_init_random(h, None, -1, 0, s) == -1 and b"n" in L.brl_last_error()
    ref = oracle.init_random(7, seed=5)
    assert_state_equal(env.init(5, num_envs=7), ref, where="after the empty calls")
