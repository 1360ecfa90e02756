"""``src/evaluation.py`` hot-path evaluators on the GPU.

``make_simple_duplicate_evaluate`` — BASELINE config 3: N boards, each played at table A and then,
seat-swapped, at table B (``duplicate_step``); greedy (``pi.mode()``) actions from two MLPs chosen
by ``current_player in {0,1}`` (src/evaluation.py:146-151); returns (mean IMP, standard error,
win rate) like src/evaluation.py:199-202.  ``make_simple_evaluate`` — the single-table
deterministic evaluator of src/evaluation.py:11-66.
"""
from __future__ import annotations

import torch

from .bridge_bidding import BridgeBidding
from .duplicate import Table_info, duplicate_step
from .models import InferenceSnapshot, make_forward_pass
from .utils import single_play_step_two_policy_commpetitive_deterministic

_NEG = torch.finfo(torch.float32).min


def masked_mode(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``Categorical(logits + finfo.min * ~mask).mode()``: arg-max over the legal actions."""
    return torch.where(mask, logits, torch.full_like(logits, _NEG)).argmax(dim=-1).to(torch.int32)


def make_simple_duplicate_evaluate(eval_env: BridgeBidding, team1_activation, team1_model_type, team2_activation,
                                   team2_model_type, num_eval_envs, sync_every: int = 16, record_actions=None):
    """src/evaluation.py:69-204.  ``sync_every``: the ``~state.terminated.all()`` loop condition is
    read back every that many iterations (finished boards keep receiving no-op steps, G9, so
    overshooting changes nothing).  ``record_actions``: optional list that receives each iteration's
    action tensor (tests replay them through the oracle)."""
    team1_forward_pass = make_forward_pass(team1_activation, team1_model_type)
    team2_forward_pass = make_forward_pass(team2_activation, team2_model_type)
    step_fn = duplicate_step(eval_env.step)

    def duplicate_evaluate(team1_params, team2_params, rng_key):
        with torch.no_grad():
            state = eval_env.init(rng_key, num_envs=num_eval_envs)  # src/evaluation.py:93-95
            table_a_info = Table_info.from_state(state)              # :96-103
            table_b_info = Table_info.from_state(state)              # :104-111
            cum_return = torch.zeros(num_eval_envs, dtype=torch.float32, device=eval_env.device)
            count = 0
            snap1, snap2 = InferenceSnapshot.make(team1_params), InferenceSnapshot.make(team2_params)
            while True:
                obs = state.observation
                if snap1 is None or snap2 is None:
                    obs = obs.to(torch.float32)
                elif snap1.dtype == snap2.dtype:
                    obs = snap1._input(obs)  # one conversion for both networks
                # G10: the reference evaluates both networks for every env and selects; so do we
                l1, _ = snap1(obs) if snap1 is not None else team1_forward_pass.apply(team1_params, obs)
                l2, _ = snap2(obs) if snap2 is not None else team2_forward_pass.apply(team2_params, obs)
                team1 = (state.current_player < 2)[:, None]          # players {0,1} are team 1 (:148)
                action = masked_mode(torch.where(team1, l1, l2), state.legal_action_mask)
                if record_actions is not None:
                    record_actions.append(action.clone())
                state, table_a_info, table_b_info = step_fn(state, action, table_a_info, table_b_info, inplace=True)
                cum_return += state.rewards[:, 0]                    # G8, :167-169
                count += 1
                if count % sync_every == 0 and bool(state.terminated.all()):
                    break
            n = float(num_eval_envs)
            std_error = cum_return.std(unbiased=True) / (n ** 0.5)   # :199
            win_rate = (cum_return > 0).sum() / n                    # :200
            log_info = (cum_return.mean(), std_error, win_rate)
        return log_info, table_a_info, table_b_info

    return duplicate_evaluate


def make_simple_evaluate(eval_env: BridgeBidding, team1_activation, team1_model_type, team2_activation,
                         team2_model_type, team2_params, num_eval_envs, sync_every: int = 8):
    """src/evaluation.py:11-66: actor (greedy) vs a fixed opponent (greedy) on single tables, no
    auto-reset; returns the mean total reward of the acting player.  ``team2_params`` replaces the
    reference's pickle path (model files are torch modules here)."""
    actor_forward_pass = make_forward_pass(team1_activation, team1_model_type)
    opp_forward_pass = make_forward_pass(team2_activation, team2_model_type)

    def simple_evaluate(actor_params, rng):
        step_fn = single_play_step_two_policy_commpetitive_deterministic(
            step_fn=eval_env.step, actor_forward_pass=actor_forward_pass, actor_params=actor_params,
            opp_forward_pass=opp_forward_pass, opp_params=team2_params)
        with torch.no_grad():
            state = eval_env.init(rng, num_envs=num_eval_envs)
            R = torch.zeros(num_eval_envs, dtype=torch.float32, device=eval_env.device)
            it = 0
            while True:
                actor = state.current_player.to(torch.int64)
                logits, _ = actor_forward_pass.apply(actor_params, state.observation.to(torch.float32))
                action = masked_mode(logits, state.legal_action_mask)
                state = step_fn(state, action, it * 4)
                R += state.rewards.gather(1, actor[:, None])[:, 0]   # src/evaluation.py:60
                it += 1
                if it % sync_every == 0 and bool(state.terminated.all()):
                    break
        return R.mean()

    return simple_evaluate
