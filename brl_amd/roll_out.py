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
from .utils import MODE, SAMPLE, UNMASKED, _pass_logits, policy_step


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


def make_random_roll_out_with_gae(config, env: BridgeBidding):
    """``roll_out`` followed by ``calc_gae`` (ppo.py's _update_step: src/roll_out.py:49-108, then src/gae.py:20-39) for the
    uniform-random policy in ONE launch (``brl_rollout_random_gae``): the value column of that policy is 0, so the scan
    needs only what the launch itself produces and ``last_val``.  normal_step only (1 env.step per scan step); one launch
    for ``num_steps <= 40`` and ``num_envs % 32 == 0``, the rollout launch(es) + ``brl_gae`` otherwise.  Returns ``(runner_state, traj_batch, advantages, targets)`` — the same
    bytes as ``make_random_roll_out`` + ``gae.gae_scan``."""
    T = int(config["num_steps"])
    reward_scale = float(config.get("reward_scale", 7600))
    gamma = float(config.get("gamma", 1.0))
    # config["gamma"] * config["gae_lambda"] is a Python-float product before it meets an array (as in gae.gae_scan)
    gl = float(torch.tensor(gamma * float(config.get("gae_lambda", 0.95)), dtype=torch.float32))

    def roll_out(runner_state, last_val=None, out: Transition = None, out_adv=None, out_tgt=None, out_last=None):
        """``out`` / ``out_adv`` / ``out_tgt`` / ``out_last = (last_obs, last_mask)``: caller-owned output buffers (a loop
        that rotates its buffers allocates nothing per call); default: fresh tensors, like a JAX function's results."""
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        n = env_state.num_envs
        traj = out if out is not None else alloc_transition(T, n, env.device)
        tc = terminated_count if (torch.is_tensor(terminated_count) and terminated_count.dtype == torch.int64
                                  and terminated_count.device == env.device and terminated_count.numel() == 1) \
            else _count_tensor(terminated_count, env.device)   # (a count tensor that is already in place is accumulated into)
        p = _capi.TransitionPtrs()
        for name in _capi.TransitionPtrs._names:
            setattr(p, name, ptr(getattr(traj, name)))
        if out_last is not None:
            last_obs, last_mask = out_last
        else:
            last_obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=env.device)
            last_mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=env.device)
        lv = (torch.zeros(n, dtype=torch.float32, device=env.device) if last_val is None
              else last_val.to(device=env.device, dtype=torch.float32).contiguous())
        adv = out_adv if out_adv is not None else torch.empty((T, n), dtype=torch.float32, device=env.device)
        tgt = out_tgt if out_tgt is not None else torch.empty_like(adv)
        check(_capi.lib().brl_rollout_random_gae(env._h, ptr(env_state.packed), n, T, int(rng) & 0xFFFFFFFF, reward_scale,
                                                 C.byref(p), ptr(last_obs), ptr(last_mask), ptr(tc), ptr(lv), gamma, gl,
                                                 ptr(adv), ptr(tgt), _stream()))
        new_state = State(env, env_state.packed, {"observation": last_obs, "legal_action_mask": last_mask})
        return (params, opt_state, new_state, last_obs, tc, int(rng) + T), traj, adv, tgt

    return roll_out


class _PolicyRollout:
    """``_env_step`` (src/roll_out.py:63-103) with torch MLPs in the loop — ONE implementation for the eager scan and
    for the hipGraph scan (config["graph_rollout"]: every macro-step captured once and replayed; the eager loop issues
    ~50 launches per macro-step and, with low-precision inference, is bound by the host's launch rate).

    Per macro-step: actor forward -> ``brl_policy_step_at`` (sample, log_prob, auto_reset(step)) -> 3 x (forward of the
    network whose turn it is -> ``brl_policy_step_at``) with rewards summed and ``terminated`` OR-ed on the device
    (src/utils.py:69-128).  The action-draw index lives in device memory in both modes.  In graph mode the Transition
    buffers are static and REUSED by the next call (consume them first — the PPO loop does); the captured launches read
    the handle's RNG key / LUT through the library's device-resident context, so ``env.seed`` / ``env.set_lut`` between
    calls are followed by the replays.  ``sub_actions`` [T,3,n] keeps the actions of sub-steps 2-4 (tests replay the
    whole rollout through the oracle)."""

    def __init__(self, env, n, T, reward_scale, game_mode, masked, infer_dtype, actor_fp, opp_fp, static, fuse_heads=True, gemm=None,
                 graph_steps=4):
        self.graph_steps = max(1, int(graph_steps))
        self.env, self.n, self.T, self.reward_scale = env, n, T, reward_scale
        self.fuse_heads = bool(fuse_heads)
        self.gemm = gemm      # config["inference_gemm"]: "bf16x3" = the hidden layers of the fp32 forwards on brl_mlp_gemm_x3 (models.InferenceSnapshot)
        self.game_mode, self.masked, self.infer_dtype, self.static = game_mode, masked, infer_dtype, static
        self.actor_fp, self.opp_fp = actor_fp, opp_fp
        self.graphs = None
        self.snap_actor = self.snap_opp = None
        if static:
            self._alloc()

    def _alloc(self):
        env, n, T = self.env, self.n, self.T
        dev = env.device
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)  # noqa: E731
        self.traj = alloc_transition(T, n, dev)
        self.packed = e((n, 16), torch.int64)
        self.cur = [e(n, torch.int32) for _ in range(2)]
        self.final_obs = e((n, OBS_SIZE), torch.bool)
        self.final_mask = e((n, NUM_ACTIONS), torch.bool)
        self.racc = e((n, 4), torch.float32)
        self.tacc = e(n, torch.bool)
        self.sub_actions = e((T, 3, n), torch.int32)
        self.draw = torch.zeros(1, dtype=torch.int32, device=dev)   # the rollout's first draw index (device: graph replays)
        self.tc = torch.zeros(1, dtype=torch.int64, device=dev)
        # the step kernels also write each new observation in the networks' input dtype (brl_macro_ext.obs_cast)
        self.xin = e((n, OBS_SIZE), self.infer_dtype or torch.float32)
        # a TENSOR divisor: torch turns `x / python_float` into x * (1 / float) on the GPU, which is not the correctly
        # rounded quotient `rewards / config["reward_scale"]` (src/roll_out.py:90) that the fused kernel and the oracle compute
        self.scale = torch.tensor(self.reward_scale, dtype=torch.float32, device=dev)

    # ---- the networks -------------------------------------------------------------------------------------------
    def _bind(self, params, opp_params):
        """(Re)build the inference views of the two networks for this call."""
        if self.static:
            if self.snap_actor is None:
                self.snap_actor = InferenceSnapshot.make(params, self.infer_dtype, self.env, gemm=self.gemm)
                self.snap_opp = InferenceSnapshot.make(opp_params, self.infer_dtype, self.env, gemm=self.gemm)
            else:  # weights are re-read INTO the tensors whose addresses the graphs hold
                self.snap_actor.refresh(params)
                self.snap_opp.refresh(opp_params)
        else:  # eager: host-bound, torch's own cast launches faster than brl_obs_cast (same layer kernels as the replayed form)
            self.snap_actor = InferenceSnapshot.make(params, self.infer_dtype, self.env, own_cast=False, gemm=self.gemm)
            self.snap_opp = self.snap_actor if opp_params is params \
                else InferenceSnapshot.make(opp_params, self.infer_dtype, self.env, own_cast=False, gemm=self.gemm)
        self.params, self.opp_params = params, opp_params
        # fp32 inference whose every forward runs on brl_linear_x3p (models.InferenceSnapshot.planes_for): the step kernels write each new
        # observation as bf16 instead of fp32 (0 / 1: exact) — the first layer then reads ONE plane and multiplies three products
        consumers = [self.snap_actor] + ([self.snap_opp] if self.game_mode == "competitive" else [])
        want = self.infer_dtype or torch.float32
        if self.infer_dtype is None and all(sn is not None and sn.planes_for(self.n) for sn in consumers):
            want = torch.bfloat16
        if getattr(self, "xin", None) is not None and self.xin.dtype != want:
            if self.graphs is not None:
                raise RuntimeError("the captured rollout holds the observation buffer in " + str(self.xin.dtype))
            self.xin = torch.empty((self.n, OBS_SIZE), dtype=want, device=self.env.device)

    _FMT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

    def _fused_heads(self, is_opp):
        """the snapshot whose heads the step kernel forms itself (16-bit inference, 4 tables per wave), or None"""
        import os
        snap = self.snap_opp if is_opp else self.snap_actor
        ok = snap is not None and self.fuse_heads and self.infer_dtype in (torch.bfloat16, torch.float16) \
            and os.environ.get("BRL_TABLES_PER_WAVE", "4") == "4"
        return snap if ok else None

    def _heads_input(self, snap, obs, x):
        """what the step kernel forms the heads from: partial products of the last hidden layer's launch (the library's own
        layer kernel) or, failing that, the last hidden layer itself"""
        parts = snap.head_parts(obs, x)
        return parts if parts is not None else snap.hidden(obs, x)

    def _head_ext(self, snap, h, **kw):
        if h.dim() == 3:   # [parts, n, ld] fp32 partial products (models.InferenceSnapshot.head_parts)
            return _capi.MacroExt(head_part=h.data_ptr(), head_part_stride=h.stride(0), head_part_ld=h.stride(1),
                                  head_nparts=h.shape[0], head_b=snap.head_bf.data_ptr(), **kw)
        return _capi.MacroExt(head_h=h.data_ptr(), head_ldh=h.stride(0), head_w=snap.head_wt.data_ptr(), head_b=snap.head_bf.data_ptr(),
                              head_hidden=h.shape[1], head_fmt=self._FMT[h.dtype], **kw)

    def _forward(self, is_opp, obs_bool, x=None):
        """-> f32 [n, 39] heads (38 logits + value; DeepMind ReLU nets: fused epilogues, merged heads) or, for the other
        architectures, (logits, value).  ``x``: obs_bool already cast by the step kernel."""
        snap = self.snap_opp if is_opp else self.snap_actor
        if snap is not None:   # low precision: the heads stay in the GEMM's dtype, the step kernel converts while reading
            out = snap.heads(obs_bool, x, raw=True)
            return out[:, :snap.n_actions], out[:, snap.n_actions]
        fp, pr = (self.opp_fp, self.opp_params) if is_opp else (self.actor_fp, self.params)
        if self.infer_dtype is None:
            return fp.apply(pr, obs_bool.to(torch.float32) if x is None else x)
        with torch.autocast("cuda", dtype=self.infer_dtype):
            lg, v = fp.apply(pr, obs_bool.to(self.infer_dtype) if x is None else x)
        return lg.float(), v.float()

    # ---- one scan step ------------------------------------------------------------------------------------------
    def _macro_step(self, t):
        """Per macro-step: 4 forwards + 4 ``brl_policy_step_ex`` launches and nothing else — value copy, accumulator reset,
        done / reward / terminated_count and the cast of every new observation ride on the sub-step launches."""
        env, traj, packed, cur = self.env, self.traj, self.packed, self.cur
        racc, tacc, T = self.racc, self.tacc, self.T
        competitive = self.game_mode == "competitive"
        MX = _capi.MacroExt
        fmt = self._FMT[self.xin.dtype]
        actor = cur[t & 1]                                              # src/roll_out.py:72
        # the first forward of a rollout reads the loaded observation; later ones the cast written by the previous launch
        mode1 = SAMPLE if self.masked else SAMPLE | UNMASKED
        snap = self._fused_heads(False)
        if snap is not None:   # the step kernel forms logits + value from the last hidden layer itself (no N = 39 GEMM)
            h = self._heads_input(snap, traj.obs[t], None if t == 0 else self.xin)
            policy_step(env, packed, packed, None, mode1, 4 * t, True, action=traj.action[t], log_prob=traj.log_prob[t],
                        rewards_acc=racc, terminated_acc=tacc, draw_base=self.draw,
                        ext=self._head_ext(snap, h, first=1, value_out=traj.value[t].data_ptr(), obs_cast=self.xin.data_ptr(),
                                           obs_fmt=fmt))
        else:
            logits, value = self._forward(False, traj.obs[t], None if t == 0 else self.xin)   # :73-76
            ifmt = self._FMT[logits.dtype]
            # sub-step 1: the actor samples from the (un)masked Categorical (src/roll_out.py:27-39,79-84)
            policy_step(env, packed, packed, logits, mode1, 4 * t, True,
                        action=traj.action[t], log_prob=traj.log_prob[t], rewards_acc=racc,
                        terminated_acc=tacc, draw_base=self.draw,
                        ext=MX(first=1, value_in=value.data_ptr(), value_stride=value.stride(0), value_out=traj.value[t].data_ptr(),
                               obs_cast=self.xin.data_ptr(), obs_fmt=fmt, in_fmt=ifmt))
        last = t + 1 == T
        obs_out = self.final_obs if last else traj.obs[t + 1]
        mask_out = self.final_mask if last else traj.legal_action_mask[t + 1]
        for k in (1, 2, 3):  # opp, partner (actor params), opp — src/utils.py:78-120; always masked
            is_opp = k != 2
            snap = None
            if not competitive and is_opp:                              # free-run: opponents pass (src/utils.py:205-246)
                lg, m = _pass_logits(env, self.n), MODE
            else:
                m = SAMPLE if competitive else MODE
                snap = self._fused_heads(is_opp)
                if snap is not None:
                    lg = None
                    hk = self._heads_input(snap, None, self.xin)
                else:
                    lg, _ = self._forward(is_opp, None, self.xin)  # (the observation comes as the cast the previous launch wrote)
            fin = k == 3
            ext = self._head_ext(snap, hk, obs_cast=self.xin.data_ptr(), obs_fmt=fmt) if snap is not None \
                else MX(obs_cast=self.xin.data_ptr(), obs_fmt=fmt, in_fmt=self._FMT[lg.dtype])
            if fin:                                                     # G2 / G1 / :85 by the same launch
                ext.last, ext.done_out, ext.reward_out = 1, traj.done[t].data_ptr(), traj.reward[t].data_ptr()
                # (terminated_count: NOT through the launch — 2048 waves adding to one address cost the last sub-step 20 us of
                #  its 31; `run` adds traj.done.sum() once per rollout instead: the same number, src/roll_out.py:85)
                ext.actor, ext.reward_scale = actor.data_ptr(), self.reward_scale
            policy_step(env, packed, packed, lg, m, 4 * t + k, True, action=self.sub_actions[t, k - 1],
                        obs=obs_out if fin else None, mask=mask_out if fin else None, rewards_acc=racc,
                        terminated_acc=tacc, current_player=cur[(t + 1) & 1] if fin else None, draw_base=self.draw, ext=ext)

    def _capture(self):
        # warm-up on a side stream (allocator, hipBLASLt heuristics), then one capture per scan step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for t in range(min(self.T, 2)):
                self._macro_step(t)
        torch.cuda.current_stream().wait_stream(side)
        pool = torch.cuda.graph_pool_handle()
        graphs = []
        from ._capture import graph_kwargs, quiet_gc
        gkw = graph_kwargs()   # (thread-local error mode only beside an RCCL watchdog thread: _capture.py)
        with torch.no_grad(), quiet_gc():   # (a graph freed by the collector mid-capture would abort the process: _capture.py)
            for t0 in range(0, self.T, self.graph_steps):   # (a replay boundary costs ~8 us of idle GPU: several scan steps per graph)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, **gkw):
                    for t in range(t0, min(t0 + self.graph_steps, self.T)):
                        self._macro_step(t)
                graphs.append(g)
        self.graphs = graphs

    def _load(self, env_state, last_obs, terminated_count, rng):
        traj = self.traj
        self.packed.copy_(env_state.packed)  # the caller's env_state stays valid, like a JAX pytree
        self.cur[0].copy_(env_state.current_player)
        traj.obs[0].copy_(env_state.observation if last_obs is None else last_obs)
        traj.legal_action_mask[0].copy_(env_state.legal_action_mask)
        self.tc.copy_(_count_tensor(terminated_count, self.env.device))
        # ONE convention for the action-draw counter everywhere: it wraps mod 2^32 (the fused kernels' uint32 arithmetic,
        # `int(rng) & 0xFFFFFFFF` at the C-ABI); the device word is the same 32 bits held in an int32 tensor
        d = int(rng) & 0xFFFFFFFF
        self.draw.fill_(d - (1 << 32) if d >= (1 << 31) else d)

    def run(self, runner_state, opp_params):
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        T = self.T
        with torch.no_grad():
            if not self.static:
                self._alloc()  # the returned Transition belongs to the caller
            self._bind(params, opp_params)
            if self.static and self.graphs is None:
                self._load(env_state, last_obs, terminated_count, rng)
                try:
                    self._capture()
                except Exception as e:   # capture is an optimisation: the same macro-steps run eagerly on the static buffers
                    import warnings
                    self.graphs = False
                    self.graph_error = repr(e)
                    warnings.warn(f"brl_amd.roll_out: hipGraph capture of the rollout failed ({e!r}); the macro-steps run "
                                  f"eagerly (host-launch-bound, slower).", RuntimeWarning)
            self._load(env_state, last_obs, terminated_count, rng)
            if self.static and self.graphs:
                for g in self.graphs:
                    g.replay()
            else:
                for t in range(T):
                    self._macro_step(t)
            self.tc += self.traj.done.sum(dtype=torch.int64)   # terminated_count += envs with `done`, per scan step (src/roll_out.py:85)
            own = (lambda x: x.clone()) if self.static else (lambda x: x)  # static buffers are reused by the next call
            new_state = State(self.env, own(self.packed), {"observation": own(self.final_obs),
                                                           "legal_action_mask": own(self.final_mask),
                                                           "current_player": own(self.cur[T & 1])})
            new_state = new_state.replace(rewards=own(self.racc), terminated=own(self.tacc))  # src/utils.py:128
        return (params, opt_state, new_state, new_state.observation, own(self.tc), int(rng) + 4 * T), self.traj


def make_roll_out(config, env: BridgeBidding, actor_forward_pass, opp_forward_pass):
    """``make_roll_out(config, env, actor_forward_pass, opp_forward_pass)`` (src/roll_out.py:23).
    Returns ``roll_out(runner_state, opp_params) -> (runner_state, traj_batch)`` (src/roll_out.py:49).
    ``config["actor_illegal_action_mask"]`` selects the masked policy, otherwise
    ``config["actor_illegal_action_penalty"]`` the unmasked one (src/roll_out.py:24-39: the actor may then draw an
    illegal call, which ends the board with pgx's penalty rewards).  After a call, ``roll_out.sub_actions`` holds the
    [T,3,n] actions of sub-steps 2-4."""
    masked = bool(config.get("actor_illegal_action_mask", True))
    if not masked and not config.get("actor_illegal_action_penalty", False):
        raise ValueError("set actor_illegal_action_mask or actor_illegal_action_penalty (src/roll_out.py:24-39)")
    T = int(config["num_steps"])
    reward_scale = float(config["reward_scale"])
    mode = config.get("game_mode", "competitive")
    if mode not in ("competitive", "free-run"):
        raise ValueError(mode)
    # Optional reduced-precision INFERENCE for the rollout forwards (MFMA bf16/fp16 GEMMs).  Default
    # fp32 like the reference; with bf16 the stored log_prob/value differ from the fp32 recomputation
    # in the PPO ratio by bf16 rounding (SURVEY §7 "Hard parts") — opt-in, never the default.
    infer_dtype = {None: None, "fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[config.get("inference_dtype")]
    engines = {}  # (n, static) -> _PolicyRollout
    if config.get("tuned_gemm", True):   # committed TunableOp solutions for the forwards' GEMM shapes (brl_amd/tuned): lookups only
        from . import tuned
        tuned.enable()

    def roll_out(runner_state, opp_params):
        params, env_state = runner_state[0], runner_state[2]
        n = env_state.num_envs
        static = bool(config.get("graph_rollout")) and mode == "competitive" \
            and InferenceSnapshot.covers(params) and InferenceSnapshot.covers(opp_params)
        eng = engines.get((n, static))
        if eng is None:
            eng = engines[(n, static)] = _PolicyRollout(env, n, T, reward_scale, mode, masked, infer_dtype,
                                                        actor_forward_pass, opp_forward_pass, static,
                                                        fuse_heads=config.get("fuse_heads", True), gemm=config.get("inference_gemm"),
                                                        graph_steps=config.get("rollout_graph_steps", 4))
        out = eng.run(runner_state, opp_params)
        roll_out.sub_actions = eng.sub_actions
        roll_out.engine = eng
        return out

    roll_out.sub_actions = None
    return roll_out
