"""ctypes binding of include/brl_hip.h.  No torch types cross this boundary: raw device
pointers (tensor.data_ptr()) and the current HIP stream handle only."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libbrl_hip.so")

STATE_WORDS = 16
OBS_SIZE = 480
NUM_ACTIONS = 38

_vp = C.c_void_p


class Fields(C.Structure):
    _names = ["current_player", "terminated", "rewards", "step_count", "turn", "dealer", "vul_ns", "vul_ew",
              "shuffled_players", "last_bid", "last_bidder", "call_x", "call_xx", "pass_num",
              "first_denomination_ns", "first_denomination_ew", "hand", "tricks", "lut_idx", "board_ctr", "illegal"]
    _fields_ = [(n, _vp) for n in _names]


class TransitionPtrs(C.Structure):
    _names = ["done", "action", "value", "reward", "log_prob", "obs", "legal_action_mask"]
    _fields_ = [(n, _vp) for n in _names]


class TableInfoPtrs(C.Structure):
    _names = ["terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"]
    _fields_ = [(n, _vp) for n in _names]


class MacroExt(C.Structure):
    _fields_ = [("first", C.c_int32), ("last", C.c_int32), ("value_in", _vp), ("value_stride", C.c_int64), ("value_out", _vp),
                ("done_out", _vp), ("reward_out", _vp), ("actor", _vp), ("reward_scale", C.c_float), ("obs_fmt", C.c_int32),
                ("terminated_count", _vp), ("obs_cast", _vp), ("in_fmt", C.c_int32), ("reserved", C.c_int32),
                ("head_h", _vp), ("head_ldh", C.c_int64), ("head_w", _vp), ("head_b", _vp), ("head_hidden", C.c_int32),
                ("head_fmt", C.c_int32), ("head_part", _vp), ("head_part_stride", C.c_int64), ("head_part_ld", C.c_int32),
                ("head_nparts", C.c_int32)]


class EvalStatsPtrs(C.Structure):
    _names = ["illegal_prob_sum", "step_count", "pass_count", "bid_count"]
    _fields_ = [(n, _vp) for n in _names]


class MlpRef(C.Structure):
    """brl_mlp_ref: a "DeepMind" network by reference (pointers into the module's own parameters)"""
    _fields_ = [("nlayers", C.c_int32), ("act", C.c_int32), ("in_features", C.c_int64), ("hidden", C.c_int64),
                ("w", _vp * 8), ("b", _vp * 8), ("actor_w", _vp), ("actor_b", _vp), ("critic_w", _vp), ("critic_b", _vp)]


class ShardGeom(C.Structure):
    """brl_shard_geom: the bucketed flat buffers of the multi-rank PPO step (offsets / slice lengths in floats)"""
    _fields_ = [("nbuckets", C.c_int32), ("world", C.c_int32), ("nsub", C.c_int32), ("reserved", C.c_int32),
                ("off", C.c_int64 * 12), ("len", C.c_int64 * 12)]


class FairNet(C.Structure):
    """brl_fair_net: the FAIR network's parameters (nn.Linear layout), device pointers"""
    _fields_ = [("w", _vp * 11), ("b", _vp * 11), ("head_w", _vp), ("head_b", _vp)]


class FairWork(C.Structure):
    """brl_fair_work: what brl_fair_chain leaves for the weight-gradient products and brl_bias_finalize_rows"""
    _names = ("inp", "dzs", "gates", "cat6", "x4", "dz0", "dz6", "dheads", "tiles", "partials", "gram_partials")
    _fields_ = [(n, _vp) for n in _names]


EVAL_COUNTS = 231  # BRL_EVAL_COUNTS


class BrlError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Loads libbrl_hip.so.  Fails loudly — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BrlError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -m brl_amd.build` (needs hipcc, gfx950). brl_amd has no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    i64, i32, u32, u64, f32 = C.c_int64, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    L.brl_last_error.restype = C.c_char_p
    L.brl_version.restype = i32
    sigs = {
        "brl_create": [i32, _vp, _vp, i64, C.POINTER(_vp)],
        "brl_set_lut": [_vp, _vp, _vp, i64],
        "brl_destroy": [_vp],
        "brl_set_rng": [_vp, u64, u64],
        "brl_init_random": [_vp, _vp, i64, u32, _vp],
        "brl_init_from_deals": [_vp, _vp, i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_step": [_vp, _vp, _vp, i64, _vp, i32, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_observe": [_vp, _vp, i64, _vp, _vp, _vp, _vp],
        "brl_get_fields": [_vp, _vp, i64, C.POINTER(Fields), _vp],
        "brl_rollout_random": [_vp, _vp, i64, i32, i32, u32, f32, C.POINTER(TransitionPtrs), _vp, _vp, _vp, _vp],
        "brl_rollout_random_gae": [_vp, _vp, i64, i32, u32, f32, C.POINTER(TransitionPtrs), _vp, _vp, _vp, _vp, f32, f32, _vp, _vp,
                                   _vp],
        "brl_policy_step": [_vp, _vp, _vp, i64, _vp, i32, u32, i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_policy_step_at": [_vp, _vp, _vp, i64, _vp, i64, i32, _vp, u32, i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_policy_step_ex": [_vp, _vp, _vp, i64, _vp, i64, i32, _vp, u32, i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               C.POINTER(MacroExt), _vp],
        "brl_obs_cast": [_vp, _vp, i64, _vp, i32, _vp],
        "brl_obs_cast_rows": [_vp, _vp, _vp, i64, _vp, i32, _vp],
        "brl_live_index": [_vp, _vp, i64, _vp, _vp, i64, _vp],
        "brl_linear_act_heads": [_vp, _vp, i64, _vp, i64, _vp, _vp, i64, i64, i32, i32, i32, i32, _vp, i64, i32, _vp, i64, i64, _vp],
        "brl_linear_act": [_vp, _vp, i64, _vp, i64, _vp, _vp, i64, i64, i32, i32, i32, i32, _vp],
        "brl_gae": [_vp, _vp, _vp, _vp, _vp, f32, f32, i32, i64, _vp, _vp, _vp],
        "brl_imp_reward": [_vp, _vp, _vp, _vp, i64, _vp],
        "brl_duplicate_step": [_vp, _vp, _vp, i64, _vp, C.POINTER(TableInfoPtrs), C.POINTER(TableInfoPtrs),
                               _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_eval_step": [_vp, _vp, _vp, i64, _vp, i64, _vp, i64, C.POINTER(TableInfoPtrs), C.POINTER(TableInfoPtrs),
                          C.POINTER(EvalStatsPtrs), i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_eval_step_team": [_vp, _vp, _vp, i64, _vp, i64, i32, C.POINTER(TableInfoPtrs), C.POINTER(TableInfoPtrs),
                               C.POINTER(EvalStatsPtrs), i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_eval_reduce": [_vp, i64, C.POINTER(TableInfoPtrs), C.POINTER(TableInfoPtrs), _vp, _vp, _vp, _vp],
        "brl_ppo_loss": [i32, _vp, i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, i64, f32, f32, f32, i32, i32, _vp, _vp, _vp, _vp, _vp],
        "brl_ppo_stats": [i32, _vp, i64, _vp, f32, f32, _vp, _vp],
        "brl_ppo_heads_loss_split": [i32, _vp, i64, _vp, _vp, i64, _vp, _vp, _vp, _vp, _vp, _vp, i64, f32, f32, f32, i32, i32, i32, _vp, _vp, _vp, _vp, _vp, i32, _vp],
        "brl_ppo_heads_bwd": [i32, _vp, _vp, i64, _vp, i64, i64, i32, i32, _vp, _vp, _vp, _vp, _vp, _vp, i64, _vp, _vp, _vp, _vp],
        "brl_ppo_stats_rows": [i32, _vp, _vp, i64, i64, f32, f32, f32, _vp, _vp],
        "brl_mb_gather_bind": [i32, C.POINTER(TransitionPtrs), _vp, _vp, _vp, _vp, i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, i64, _vp, _vp],
        "brl_adam_clip_fin_gather": [i32, _vp, _vp, _vp, _vp, i64, _vp, f32, _vp, f32, f32, f32, f32, _vp, i64, _vp, _vp, _vp, i64,
                                     i32, _vp, _vp, _vp, _vp, _vp],
        "brl_mb_gather_dev": [i32, _vp, i64, _vp],
        "brl_ppo_stats_gram": [i32, _vp, i64, i64, _vp, i64, f32, f32, f32, _vp, _vp, _vp, _vp],
        "brl_ppo_illegal_grad": [i32, _vp, _vp, _vp, f32, i64, _vp, _vp],
        "brl_act_bwd_colsum": [i32, _vp, _vp, i64, i64, i64, i32, _vp, _vp],
        "brl_act_bwd_colsum_heads_dw": [i32, _vp, _vp, i64, i64, i64, i32, _vp, _vp, _vp, i64, i64, i64, i32, _vp, _vp, _vp, _vp, i64,
                                        _vp, _vp, _vp, _vp],
        "brl_bias_finalize_ex": [i32, i32, _vp, _vp, _vp, _vp, _vp],
        "brl_bias_finalize_rows": [i32, i32, _vp, _vp, _vp, _vp, i32, _vp, _vp],
        "brl_fair_forward": [i32, C.POINTER(FairNet), _vp, i64, i32, _vp, _vp, _vp],
        "brl_mlp_gemm_group": [i32, i32, i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_fair_chain": [i32, C.POINTER(FairNet), _vp, _vp, _vp, _vp, _vp, _vp, _vp, i64, f32, f32, f32, i32, i32, i32, i32,
                           C.POINTER(FairWork), _vp],
        "brl_mlp_gemm": [i32, i32, i32, _vp, i64, _vp, i64, _vp, i64, i64, i64, i64, i32, _vp, _vp, i64, _vp, _vp, _vp],
        "brl_mlp_gemm_x3_workspace": [i64, i64, i64, _vp],
        "brl_mlp_gemm_x3_group": [i32, i32, i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "brl_split_planes": [i32, _vp, i64, _vp, i64, _vp],
        "brl_linear_x3p": [i32, _vp, i32, i64, i64, _vp, i64, i64, _vp, i32, _vp, i64, _vp, i64, i64, i64, i64, i64, _vp],
        "brl_mlp_gemm_x3": [i32, i32, i32, _vp, i64, _vp, i64, _vp, i64, i64, i64, i64, i32, _vp, _vp, i64, _vp, _vp, i64, _vp],
        "brl_mlp_forward_rows": [i32, C.POINTER(MlpRef), _vp, _vp, i64, _vp, i64, _vp, i64, _vp],
        "brl_adam_shard_norm": [i32, _vp, C.POINTER(ShardGeom), i32, i32, f32, _vp, _vp, _vp, _vp],
        "brl_adam_shard_apply": [i32, _vp, _vp, _vp, _vp, C.POINTER(ShardGeom), i32, i32, _vp, _vp, f32, _vp, f32, f32, f32, f32, f32, _vp,
                                 _vp, i64, _vp],
        "brl_mlp_gemm_dh_heads_dw": [i32, _vp, i64, _vp, i64, _vp, i64, i64, i64, i64, i32, _vp, i64, _vp, _vp, _vp, i64, i64, i64, i32,
                                     _vp, _vp, _vp, _vp, i64, _vp, _vp, _vp, _vp],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    _lib = L
    return L


EXPORTS = ["brl_last_error", "brl_version", "brl_create", "brl_set_lut", "brl_destroy", "brl_set_rng",
           "brl_init_random", "brl_init_from_deals", "brl_step", "brl_observe", "brl_get_fields",
           "brl_rollout_random", "brl_policy_step", "brl_policy_step_at", "brl_obs_cast", "brl_obs_cast_rows", "brl_live_index", "brl_linear_act", "brl_linear_act_heads", "brl_gae", "brl_imp_reward", "brl_duplicate_step",
           "brl_eval_step", "brl_eval_reduce", "brl_ppo_loss", "brl_ppo_stats", "brl_policy_step_ex",
           "brl_eval_step_team", "brl_rollout_random_gae", "brl_ppo_heads_loss_split", "brl_adam_shard_norm", "brl_adam_shard_apply", "brl_ppo_heads_bwd", "brl_ppo_stats_gram",
           "brl_act_bwd_colsum", "brl_act_bwd_colsum_heads_dw", "brl_bias_finalize_ex", "brl_ppo_stats_rows", "brl_mb_gather_bind", "brl_mb_gather_dev", "brl_ppo_illegal_grad", "brl_adam_clip_fin_gather", "brl_mlp_gemm", "brl_mlp_gemm_dh_heads_dw", "brl_mlp_forward_rows",
           "brl_bias_finalize_rows", "brl_fair_chain", "brl_mlp_gemm_group", "brl_fair_forward", "brl_mlp_gemm_x3", "brl_mlp_gemm_x3_workspace", "brl_mlp_gemm_x3_group", "brl_split_planes", "brl_linear_x3p"]


def check(rc: int) -> None:
    if rc != 0:
        raise BrlError(f"libbrl_hip error {rc}: {lib().brl_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "brl_amd passes raw pointers: tensors must be contiguous"
    return t.data_ptr()
