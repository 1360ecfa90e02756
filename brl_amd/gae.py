"""``src/gae.py`` on the GPU: one critic forward for ``last_val`` + the reverse scan kernel."""
from __future__ import annotations

import torch

from . import _capi
from ._capi import check, ptr
from .bridge_bidding import _stream


def gae_scan(env, done, value, reward, last_val, gamma: float, gae_lambda: float):
    """advantages, targets = reverse scan over T (src/gae.py:20-39).  Inputs time-major [T,N]."""
    T, N = done.shape
    done_u8 = done.view(torch.uint8) if done.dtype == torch.bool else done.to(torch.uint8)
    done_u8 = done_u8.contiguous()
    value = value.to(torch.float32).contiguous()
    reward = reward.to(torch.float32).contiguous()
    last_val = last_val.to(torch.float32).contiguous()
    adv = torch.empty((T, N), dtype=torch.float32, device=value.device)
    tgt = torch.empty_like(adv)
    # config["gamma"] * config["gae_lambda"] is a Python-float product before it meets an array
    gl = float(torch.tensor(float(gamma) * float(gae_lambda), dtype=torch.float32))
    check(_capi.lib().brl_gae(env._h, ptr(done_u8), ptr(value), ptr(reward), ptr(last_val), float(gamma), gl,
                              int(T), int(N), ptr(adv), ptr(tgt), _stream()))
    return adv, tgt


def make_calc_gae(config, actor_forward_pass, env=None):
    """``make_calc_gae(config, actor_forward_pass)`` (src/gae.py:5).  ``env`` is only needed when the
    runner state's env_state does not carry one (it always does for brl_amd States)."""

    def calc_gae(runner_state, traj_batch):
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        with torch.no_grad():
            _, last_val = actor_forward_pass.apply(params, last_obs.to(torch.float32))  # src/gae.py:16-18
        e = env if env is not None else env_state.env
        return gae_scan(e, traj_batch.done, traj_batch.value, traj_batch.reward, last_val,
                        config["gamma"], config["gae_lambda"])

    return calc_gae
