"""``src/utils.py`` on the GPU: auto-reset and the 4-sub-step macro-steps.

Every sub-step is ONE kernel (``brl_policy_step``): masked categorical over the logits of the
network whose turn it is -> action -> ``auto_reset(env.step)`` -> next observation / mask, with
the rewards summed and ``terminated`` OR-ed across sub-steps on the device (src/utils.py:126-127).
"""
from __future__ import annotations

import torch

from . import _capi
from ._capi import NUM_ACTIONS, OBS_SIZE, check, ptr
from .bridge_bidding import BridgeBidding, State, _stream

SAMPLE, MODE = 0, 1  # pi.sample(seed) / pi.mode()
UNMASKED = 2         # OR-ed in: the categorical ranges over all 38 actions (src/roll_out.py:33-39)


def _env_of(step_fn) -> BridgeBidding:
    env = getattr(step_fn, "env", None) or getattr(step_fn, "__self__", None)
    if not isinstance(env, BridgeBidding):
        raise TypeError("expected env.step (or auto_reset(env.step, env.init)) of a brl_amd.BridgeBidding")
    return env


def auto_reset(step_fn, init_fn=None):
    """``auto_reset(env.step, env.init)`` (src/utils.py:9-58): the pre-clear, the step and the
    re-deal all happen inside the step kernel (autoreset=1)."""
    env = _env_of(step_fn)

    def wrapped_step_fn(state: State, action, inplace: bool = False):
        return env.step(state, action, autoreset=True, inplace=inplace)

    wrapped_step_fn.env = env
    wrapped_step_fn.autoreset = True
    return wrapped_step_fn


class SubstepBuffers:
    """Scratch reused across sub-steps: accumulators and the next observation."""

    def __init__(self, env: BridgeBidding, n: int):
        dev = env.device
        self.rewards_acc = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        self.terminated_acc = torch.zeros(n, dtype=torch.bool, device=dev)
        self.obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        self.mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=dev)
        self.current_player = torch.empty(n, dtype=torch.int32, device=dev)
        self.action = torch.empty(n, dtype=torch.int32, device=dev)
        self.log_prob = torch.empty(n, dtype=torch.float32, device=dev)


def policy_step(env: BridgeBidding, packed_in, packed_out, logits, mode: int, draw: int, autoreset: bool, *,
                action=None, log_prob=None, obs=None, mask=None, rewards_acc=None, terminated_acc=None,
                current_player=None, draw_base=None, ext=None):
    """One launch of ``brl_policy_step`` (include/brl_hip.h).  With ``draw_base`` (a 1-element int32/uint32 device
    tensor) the draw index is ``draw_base[0] + draw``, read on the device (``brl_policy_step_at``: hipGraph replays).
    ``ext`` (a ``_capi.MacroExt``): the macro-step bookkeeping of src/roll_out.py:72-103 done by the same launch
    (``brl_policy_step_ex``)."""
    n = packed_in.shape[0]
    if ext is not None and (ext.head_h or ext.head_part):   # the launch forms the logits itself (brl_macro_ext.head_h / head_part)
        import ctypes as C
        check(_capi.lib().brl_policy_step_ex(env._h, ptr(packed_in), ptr(packed_out), n, None, NUM_ACTIONS, int(mode), ptr(draw_base),
                                             int(draw) & 0xFFFFFFFF, int(bool(autoreset)), ptr(action), ptr(log_prob), ptr(obs),
                                             ptr(mask), ptr(rewards_acc), ptr(terminated_acc), ptr(current_player), C.byref(ext),
                                             _stream()))
        return
    if ext is None or not ext.in_fmt:
        logits = logits.to(torch.float32)
    assert logits.shape == (n, NUM_ACTIONS)
    strided = logits.stride(1) == 1 and logits.stride(0) >= NUM_ACTIONS  # e.g. the first 38 columns of a [n,39] matrix
    if not strided:
        logits = logits.contiguous()
    if ext is not None:
        import ctypes as C
        check(_capi.lib().brl_policy_step_ex(env._h, ptr(packed_in), ptr(packed_out), n, logits.data_ptr(),
                                             logits.stride(0), int(mode), ptr(draw_base), int(draw) & 0xFFFFFFFF,
                                             int(bool(autoreset)), ptr(action), ptr(log_prob), ptr(obs), ptr(mask),
                                             ptr(rewards_acc), ptr(terminated_acc), ptr(current_player), C.byref(ext),
                                             _stream()))
        return
    if draw_base is not None or logits.stride(0) != NUM_ACTIONS:
        check(_capi.lib().brl_policy_step_at(env._h, ptr(packed_in), ptr(packed_out), n, logits.data_ptr(),
                                             logits.stride(0), int(mode), ptr(draw_base), int(draw) & 0xFFFFFFFF,
                                             int(bool(autoreset)), ptr(action), ptr(log_prob), ptr(obs), ptr(mask),
                                             ptr(rewards_acc), ptr(terminated_acc), ptr(current_player), _stream()))
        return
    check(_capi.lib().brl_policy_step(env._h, ptr(packed_in), ptr(packed_out), n, ptr(logits), int(mode),
                                      int(draw) & 0xFFFFFFFF, int(bool(autoreset)), ptr(action), ptr(log_prob),
                                      ptr(obs), ptr(mask), ptr(rewards_acc), ptr(terminated_acc),
                                      ptr(current_player), _stream()))


_PASS_LOGITS = {}


def _pass_logits(env, n):
    key = (env.device, n)
    if key not in _PASS_LOGITS:
        lg = torch.zeros((n, NUM_ACTIONS), dtype=torch.float32, device=env.device)
        lg[:, 0] = 1.0  # arg-max = Pass, always legal
        _PASS_LOGITS[key] = lg
    return _PASS_LOGITS[key]


def _macro(step_fn, actor_forward_pass, actor_params, opp_forward_pass, opp_params, sub_modes, opp_passes):
    env = _env_of(step_fn)
    autoreset = bool(getattr(step_fn, "autoreset", False))

    def wrapped_step_fn(state: State, action, rng):
        n = state.num_envs
        buf = SubstepBuffers(env, n)
        draw = int(rng)
        with torch.no_grad():
            s = env.step(state, action, autoreset=autoreset)  # sub-step 1 (src/utils.py:70)
            buf.rewards_acc.copy_(s.rewards)
            buf.terminated_acc.copy_(s.terminated)
            packed = s.packed
            obs = s.observation
            for k, mode in enumerate(sub_modes):  # sub-steps 2..4: opp, actor(partner), opp
                is_opp = k % 2 == 0
                if is_opp and opp_passes:
                    logits, m = _pass_logits(env, n), MODE
                else:
                    fp, pr = (opp_forward_pass, opp_params) if is_opp else (actor_forward_pass, actor_params)
                    logits, _ = fp.apply(pr, obs.to(torch.float32))
                    m = mode
                rec = wrapped_step_fn.record_actions
                policy_step(env, packed, packed, logits, m, draw + k, autoreset, obs=buf.obs, mask=buf.mask,
                            rewards_acc=buf.rewards_acc, terminated_acc=buf.terminated_acc,
                            current_player=buf.current_player, action=buf.action if rec is not None else None)
                if rec is not None:
                    rec.append(buf.action.clone())
                obs = buf.obs
        out = State(env, packed, {"observation": buf.obs, "legal_action_mask": buf.mask,
                                  "current_player": buf.current_player})
        return out.replace(rewards=buf.rewards_acc, terminated=buf.terminated_acc)  # src/utils.py:128

    wrapped_step_fn.record_actions = None   # tests: a list that receives the calls of sub-steps 2..4 (oracle replays)
    return wrapped_step_fn


def single_play_step_two_policy_commpetitive(step_fn, actor_forward_pass, actor_params, opp_forward_pass, opp_params):
    """src/utils.py:61-130 — sub-steps 2-4 sample from the masked Categorical."""
    return _macro(step_fn, actor_forward_pass, actor_params, opp_forward_pass, opp_params, (SAMPLE,) * 3, False)


def single_play_step_two_policy_commpetitive_deterministic(step_fn, actor_forward_pass, actor_params,
                                                           opp_forward_pass, opp_params):
    """src/utils.py:133-202 — sub-steps 2-4 take ``pi.mode()``."""
    return _macro(step_fn, actor_forward_pass, actor_params, opp_forward_pass, opp_params, (MODE,) * 3, False)


def single_play_step_free_run(step_fn, actor_forward_pass, actor_params, opp_forward_pass=None, opp_params=None):
    """src/utils.py:205-246 — opponents always pass, the partner plays ``pi.mode()``."""
    return _macro(step_fn, actor_forward_pass, actor_params, None, None, (MODE,) * 3, True)


def normal_step(step_fn):
    """src/utils.py:249-254."""

    def wrapped_step_fn(state, action, rng=None):
        return step_fn(state, action)

    return wrapped_step_fn
