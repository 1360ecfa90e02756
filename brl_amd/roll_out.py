"""``src/roll_out.py`` on the GPU.

* ``make_random_roll_out`` — BASELINE config 1/2: uniform-random masked policy; the whole
  T-step scan is ONE kernel launch (``brl_rollout_random``) that streams the time-major
  Transition buffer to HBM.
* ``make_roll_out`` — the reference's signature (src/roll_out.py:23,49) with torch MLPs in the
  loop: per sub-step one GEMM forward (PyTorch-ROCm) + one ``brl_policy_step`` launch, no host
  synchronisation inside the scan.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch

from . import _capi
from ._capi import NUM_ACTIONS, OBS_SIZE, check, ptr
from .bridge_bidding import BridgeBidding, State, _stream
from .models import InferenceSnapshot
from .utils import MODE, SAMPLE, _pass_logits, policy_step


class Transition(NamedTuple):  # src/roll_out.py:13-20 ; all time-major [T,N,...]
    done: torch.Tensor               # bool
    action: torch.Tensor             # int32
    value: torch.Tensor              # f32
    reward: torch.Tensor             # f32  rewards[actor] / reward_scale
    log_prob: torch.Tensor           # f32
    obs: torch.Tensor                # bool [T,N,480]
    legal_action_mask: torch.Tensor  # bool [T,N,38]


def alloc_transition(T: int, n: int, device) -> Transition:
    e = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)  # noqa: E731
    return Transition(e((T, n), torch.bool), e((T, n), torch.int32), e((T, n), torch.float32),
                      e((T, n), torch.float32), e((T, n), torch.float32), e((T, n, OBS_SIZE), torch.bool),
                      e((T, n, NUM_ACTIONS), torch.bool))


def _count_tensor(terminated_count, device):
    if torch.is_tensor(terminated_count):
        return terminated_count.to(device=device, dtype=torch.int64).reshape(1)
    return torch.tensor([int(terminated_count)], dtype=torch.int64, device=device)


def make_random_roll_out(config, env: BridgeBidding):
    """roll_out with the uniform-random masked policy (logits = 0 -> masked Categorical, value = 0)
    and ``normal_step`` (config["game_mode"] == "normal", 1 env.step per scan step — BASELINE
    config 2) or the 4-sub-step competitive macro-step with every seat random ("competitive").
    ``runner_state[5]`` (the reference's PRNG key slot) is an int: the index of the next action draw."""
    T = int(config["num_steps"])
    substeps = int(config.get("substeps", 4 if config.get("game_mode", "normal") == "competitive" else 1))
    reward_scale = float(config.get("reward_scale", 7600))

    def roll_out(runner_state, out: Transition = None):
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        n = env_state.num_envs
        traj = out if out is not None else alloc_transition(T, n, env.device)
        tc = _count_tensor(terminated_count, env.device)
        p = _capi.TransitionPtrs()
        for name in _capi.TransitionPtrs._names:
            setattr(p, name, ptr(getattr(traj, name)))
        want_last = config.get("return_last_obs", True)
        last_obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=env.device) if want_last else None
        last_mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=env.device) if want_last else None
        check(_capi.lib().brl_rollout_random(env._h, ptr(env_state.packed), n, T, substeps, int(rng) & 0xFFFFFFFF,
                                             reward_scale, C.byref(p), ptr(last_obs), ptr(last_mask), ptr(tc), _stream()))
        cache = {"observation": last_obs, "legal_action_mask": last_mask} if want_last else None
        new_state = State(env, env_state.packed, cache)  # updated in place; last_obs written by the same launch
        return (params, opt_state, new_state, last_obs, tc, int(rng) + T * substeps), traj

    return roll_out


class _GraphedRollout:
    """The competitive policy-in-the-loop rollout with every macro-step captured ONCE in a hipGraph and replayed
    (opt-in, config["graph_rollout"]): the eager loop issues ~50 launches per macro-step and, with low-precision
    inference, is bound by the host's launch rate, not by the GPU.  One graph per scan step t (they differ only in the
    Transition rows they read and write), all sharing one memory pool.  Differences from the eager path that a caller
    can see: the returned Transition buffers are REUSED by the next call (consume them first — the PPO loop does),
    and the action-draw index lives in device memory (``brl_policy_step_at``).  Same outputs, bit for bit, as the
    eager path with the same inference dtype."""

    def __init__(self, env, n, T, reward_scale, infer_dtype, actor, opp):
        self.env, self.n, self.T, self.reward_scale = env, n, T, reward_scale
        dev = env.device
        self.traj = alloc_transition(T, n, dev)
        self.packed = torch.empty((n, 16), dtype=torch.int64, device=dev)
        self.cur = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(2)]
        self.scratch_obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        self.final_obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        self.final_mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=dev)
        self.racc = torch.empty((n, 4), dtype=torch.float32, device=dev)
        self.tacc = torch.empty(n, dtype=torch.bool, device=dev)
        self.draw = torch.zeros(1, dtype=torch.int32, device=dev)
        self.tc = torch.zeros(1, dtype=torch.int64, device=dev)
        self.snap_actor = InferenceSnapshot.make(actor, infer_dtype, env)
        self.snap_opp = InferenceSnapshot.make(opp, infer_dtype, env)
        self.graphs = None

    def _macro_step(self, t):
        env, traj, packed, cur = self.env, self.traj, self.packed, self.cur
        racc, tacc, T = self.racc, self.tacc, self.T
        actor = cur[t & 1]  # src/roll_out.py:72
        logits, value = self.snap_actor(traj.obs[t])  # :73-76
        traj.value[t].copy_(value)
        racc.zero_()
        tacc.zero_()
        policy_step(env, packed, packed, logits, SAMPLE, 0, True, action=traj.action[t], log_prob=traj.log_prob[t],
                    obs=self.scratch_obs, rewards_acc=racc, terminated_acc=tacc, draw_base=self.draw)
        last = t + 1 == T
        obs_out = self.final_obs if last else traj.obs[t + 1]
        mask_out = self.final_mask if last else traj.legal_action_mask[t + 1]
        for k in (1, 2, 3):  # opp, partner (actor params), opp — src/utils.py:78-120
            lg, _ = (self.snap_opp if k != 2 else self.snap_actor)(self.scratch_obs)
            fin = k == 3
            policy_step(env, packed, packed, lg, SAMPLE, k, True, obs=obs_out if fin else self.scratch_obs,
                        mask=mask_out if fin else None, rewards_acc=racc, terminated_acc=tacc,
                        current_player=cur[(t + 1) & 1] if fin else None, draw_base=self.draw)
        self.draw.add_(4)
        traj.done[t].copy_(tacc)  # G2
        traj.reward[t].copy_(racc.gather(1, actor.to(torch.int64)[:, None])[:, 0] / self.reward_scale)  # G1
        self.tc.add_(tacc.sum())

    def _capture(self):
        # warm-up on a side stream (allocator, hipBLASLt heuristics), then one capture per scan step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for t in range(min(self.T, 2)):
                self._macro_step(t)
        torch.cuda.current_stream().wait_stream(side)
        pool = torch.cuda.graph_pool_handle()
        self.graphs = []
        with torch.no_grad():
            for t in range(self.T):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    self._macro_step(t)
                self.graphs.append(g)

    def run(self, runner_state, opp_params):
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        env, n, T, traj = self.env, self.n, self.T, self.traj
        with torch.no_grad():
            if self.graphs is None:
                self.packed.copy_(env_state.packed)
                self.cur[0].copy_(env_state.current_player)
                traj.obs[0].copy_(env_state.observation if last_obs is None else last_obs)
                self._capture()
            self.snap_actor.refresh(params)
            self.snap_opp.refresh(opp_params)
            self.packed.copy_(env_state.packed)
            self.cur[0].copy_(env_state.current_player)
            traj.obs[0].copy_(env_state.observation if last_obs is None else last_obs)
            traj.legal_action_mask[0].copy_(env_state.legal_action_mask)
            self.tc.copy_(_count_tensor(terminated_count, env.device))
            self.draw.fill_(int(rng) & 0x7FFFFFFF)
            for g in self.graphs:
                g.replay()
            packed = self.packed.clone()  # the returned state owns its tables; the static buffers are reused
            final_obs, final_mask = self.final_obs.clone(), self.final_mask.clone()
            new_state = State(env, packed, {"observation": final_obs, "legal_action_mask": final_mask,
                                            "current_player": self.cur[T & 1].clone()})
            new_state = new_state.replace(rewards=self.racc.clone(), terminated=self.tacc.clone())  # src/utils.py:128
        return (params, opt_state, new_state, new_state.observation, self.tc.clone(), int(rng) + 4 * T), traj


def make_roll_out(config, env: BridgeBidding, actor_forward_pass, opp_forward_pass):
    """``make_roll_out(config, env, actor_forward_pass, opp_forward_pass)`` (src/roll_out.py:23).
    Returns ``roll_out(runner_state, opp_params) -> (runner_state, traj_batch)`` (src/roll_out.py:49).
    Only the masked policy (config["actor_illegal_action_mask"]) is on the hot path."""
    if not config.get("actor_illegal_action_mask", True):
        raise NotImplementedError("the unmasked / illegal-action-penalty policy is outside the hot path")
    T = int(config["num_steps"])
    reward_scale = float(config["reward_scale"])
    mode = config.get("game_mode", "competitive")
    if mode not in ("competitive", "free-run"):
        raise ValueError(mode)
    # Optional reduced-precision INFERENCE for the rollout forwards (MFMA bf16/fp16 GEMMs).  Default
    # fp32 like the reference; with bf16 the stored log_prob/value differ from the fp32 recomputation
    # in the PPO ratio by bf16 rounding (SURVEY §7 "Hard parts") — opt-in, never the default.
    infer_dtype = {None: None, "fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[config.get("inference_dtype")]

    snapshots = {}  # id(params) -> InferenceSnapshot (or None), rebuilt on every roll_out call

    def forward(fp, pr, obs_bool):
        snap = snapshots.get(id(pr), False)
        if snap is False:
            snap = snapshots[id(pr)] = InferenceSnapshot.make(pr, infer_dtype)  # (eager: host-bound, torch's cast launches faster)
        if snap is not None:  # "DeepMind" ReLU MLP: fused epilogues, merged heads (fp32 by default)
            return snap(obs_bool)
        if infer_dtype is None:
            return fp.apply(pr, obs_bool.to(torch.float32))
        with torch.autocast("cuda", dtype=infer_dtype):
            lg, v = fp.apply(pr, obs_bool.to(infer_dtype))
        return lg.float(), v.float()

    graphed = {}  # (n,) -> _GraphedRollout, built on first use when config["graph_rollout"] is set

    def roll_out(runner_state, opp_params):
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        n, dev = env_state.num_envs, env.device
        if config.get("graph_rollout") and mode == "competitive":
            gr = graphed.get(n)
            if gr is None and InferenceSnapshot.make(params, infer_dtype) is not None \
                    and InferenceSnapshot.make(opp_params, infer_dtype) is not None:
                gr = graphed[n] = _GraphedRollout(env, n, T, reward_scale, infer_dtype, params, opp_params)
            if gr is not None:
                return gr.run(runner_state, opp_params)
        snapshots.clear()  # the weights may have been updated since the last rollout
        traj = alloc_transition(T, n, dev)
        tc = _count_tensor(terminated_count, dev)
        packed = env_state.packed.clone()  # the caller's env_state stays valid, like a JAX pytree
        cur = [env_state.current_player.clone(), torch.empty(n, dtype=torch.int32, device=dev)]
        traj.obs[0].copy_(env_state.observation if last_obs is None else last_obs)
        traj.legal_action_mask[0].copy_(env_state.legal_action_mask)
        scratch_obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        final_obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        final_mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=dev)
        racc = torch.empty((n, 4), dtype=torch.float32, device=dev)
        tacc = torch.empty(n, dtype=torch.bool, device=dev)
        draw = int(rng)
        with torch.no_grad():
            for t in range(T):
                actor = cur[t & 1]  # src/roll_out.py:72
                logits, value = forward(actor_forward_pass, params, traj.obs[t])  # :73-76
                traj.value[t].copy_(value)
                racc.zero_()
                tacc.zero_()
                # sub-step 1: actor samples from the masked Categorical (src/roll_out.py:79-84)
                policy_step(env, packed, packed, logits, SAMPLE, draw, True, action=traj.action[t],
                            log_prob=traj.log_prob[t], obs=scratch_obs, rewards_acc=racc, terminated_acc=tacc)
                last = t + 1 == T
                obs_out = final_obs if last else traj.obs[t + 1]
                mask_out = final_mask if last else traj.legal_action_mask[t + 1]
                for k in (1, 2, 3):  # opp, partner (actor params), opp — src/utils.py:78-120
                    is_opp = k != 2
                    if mode == "free-run" and is_opp:
                        lg, m = _pass_logits(env, n), MODE
                    else:
                        fp, pr = (opp_forward_pass, opp_params) if is_opp else (actor_forward_pass, params)
                        lg, _ = forward(fp, pr, scratch_obs)
                        m = SAMPLE if mode == "competitive" else MODE
                    fin = k == 3
                    policy_step(env, packed, packed, lg, m, draw + k, True,
                                obs=obs_out if fin else scratch_obs, mask=mask_out if fin else None,
                                rewards_acc=racc, terminated_acc=tacc, current_player=cur[(t + 1) & 1] if fin else None)
                draw += 4
                traj.done[t].copy_(tacc)  # G2
                traj.reward[t].copy_(racc.gather(1, actor.to(torch.int64)[:, None])[:, 0] / reward_scale)  # G1
                tc += tacc.sum()
        new_state = State(env, packed, {"observation": final_obs, "legal_action_mask": final_mask,
                                        "current_player": cur[T & 1]})
        new_state = new_state.replace(rewards=racc, terminated=tacc)  # src/utils.py:128
        return (params, opt_state, new_state, new_state.observation, tc, draw), traj

    return roll_out
