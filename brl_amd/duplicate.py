"""``src/duplicate.py`` on the GPU: duplicate-table pairing and IMP conversion."""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch

from . import _capi
from ._capi import NUM_ACTIONS, OBS_SIZE, check, ptr
from .bridge_bidding import BridgeBidding, State, _stream

PASS_ACTION_NUM = 0      # src/duplicate.py:9
DOUBLE_ACTION_NUM = 1    # src/duplicate.py:10
REDOUBLE_ACTION_NUM = 2  # src/duplicate.py:11
BID_OFFSET_NUM = 3       # src/duplicate.py:12


class Table_info(NamedTuple):  # src/duplicate.py:138-144
    terminated: torch.Tensor   # bool  [N]
    rewards: torch.Tensor      # f32   [N,4]
    last_bid: torch.Tensor     # int32 [N]
    last_bidder: torch.Tensor  # int32 [N]
    call_x: torch.Tensor       # bool  [N]
    call_xx: torch.Tensor      # bool  [N]

    @staticmethod
    def from_state(state: State) -> "Table_info":
        """What src/evaluation.py:96-111 builds from a freshly initialised state: its own tensors, filled by ONE
        brl_get_fields launch (as six attribute reads + six clones an evaluator paid 24 launches for its two tables)."""
        n, dev = state.num_envs, state.packed.device
        f, out = _capi.Fields(), []
        for name in ("terminated", "rewards", "_last_bid", "_last_bidder", "_call_x", "_call_xx"):
            cname, dtype, shape = State._FIELDS[name]
            t = torch.empty((n,) + shape, dtype=dtype, device=dev)
            setattr(f, cname, ptr(t))
            out.append(t)
        check(_capi.lib().brl_get_fields(state.env._h, ptr(state.packed), n, C.byref(f), _stream()))
        return Table_info(*out)

    def _ptrs(self) -> _capi.TableInfoPtrs:
        p = _capi.TableInfoPtrs()
        for name in _capi.TableInfoPtrs._names:
            setattr(p, name, ptr(getattr(self, name)))
        return p


def _imp_reward(table_a_reward: torch.Tensor, table_b_reward: torch.Tensor, env: BridgeBidding = None) -> torch.Tensor:
    """``_imp_reward`` (src/duplicate.py:15-70), batched [N,4] (a single [4] vector is accepted)."""
    a = torch.as_tensor(table_a_reward)
    single = a.dim() == 1
    if env is None:
        raise ValueError("_imp_reward needs env= (the handle that owns the GPU)")
    a = a.to(device=env.device, dtype=torch.float32).reshape(-1, 4).contiguous()
    b = torch.as_tensor(table_b_reward).to(device=env.device, dtype=torch.float32).reshape(-1, 4).contiguous()
    out = torch.empty_like(a)
    check(_capi.lib().brl_imp_reward(env._h, ptr(a), ptr(b), ptr(out), a.shape[0], _stream()))
    return out[0] if single else out


def duplicate_init(state: State) -> State:
    """``duplicate_init`` (src/duplicate.py:132-135): same hands / dealer / vulnerabilities, seats
    permuted by [1,0,3,2], everything else back to defaults."""
    f = state
    return state.env.init_from_deals(f._hand, f._dealer, f._vul_NS, f._vul_EW,
                                     f._shuffled_players[:, [1, 0, 3, 2]], f._dds_tricks)


def duplicate_step(step_fn):
    """``duplicate_step(env.step)`` (src/duplicate.py:147-192).  Table_info tensors are updated IN
    PLACE and returned (the reference returns fresh pytrees)."""
    env = getattr(step_fn, "__self__", None)
    if not isinstance(env, BridgeBidding):
        raise TypeError("duplicate_step expects env.step of a brl_amd.BridgeBidding")

    def wrapped_step(state: State, action, table_a_info: Table_info, table_b_info: Table_info, inplace=False):
        n = state.num_envs
        dev = env.device
        action = torch.as_tensor(action, device=dev).to(torch.int32).contiguous()
        out = state.packed if inplace else env._new_packed(n)
        obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=dev)
        rewards = torch.empty((n, 4), dtype=torch.float32, device=dev)
        term = torch.empty(n, dtype=torch.bool, device=dev)
        cur = torch.empty(n, dtype=torch.int32, device=dev)
        pa, pb = table_a_info._ptrs(), table_b_info._ptrs()
        check(_capi.lib().brl_duplicate_step(env._h, ptr(state.packed), ptr(out), n, ptr(action), C.byref(pa),
                                             C.byref(pb), ptr(obs), ptr(mask), ptr(rewards), ptr(term), ptr(cur),
                                             _stream()))
        nxt = State(env, out, {"observation": obs, "legal_action_mask": mask, "rewards": rewards,
                               "terminated": term, "current_player": cur})
        return nxt, table_a_info, table_b_info

    return wrapped_step
