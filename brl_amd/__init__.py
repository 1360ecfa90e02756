"""brl_amd — MI355X-native bridge-bidding environment + PPO rollout hot path.

A drop-in for ONE path of harukaki/brl: ``pgx.bridge_bidding.{init,step,observe}`` and the
``roll_out / calc_gae / duplicate_step`` surface of its ``ppo.py``.  Host code is Python; the
work is done by hand-written HIP kernels for gfx950 behind a C-ABI (``include/brl_hip.h``,
``brl_amd/lib/libbrl_hip.so``) bound with ctypes.  There is NO CPU fallback: importing the
environment without the built library raises.
"""
from .bridge_bidding import BridgeBidding, State, _observe, _player_position  # noqa: F401
from .duplicate import Table_info, duplicate_init, duplicate_step, _imp_reward  # noqa: F401
from .gae import make_calc_gae  # noqa: F401
from .roll_out import Transition, make_roll_out, make_random_roll_out, make_random_roll_out_with_gae  # noqa: F401

__all__ = [
    "BridgeBidding", "State", "_observe", "_player_position", "Table_info", "duplicate_init",
    "duplicate_step", "_imp_reward", "make_calc_gae", "Transition", "make_roll_out",
    "make_random_roll_out",
    "make_random_roll_out_with_gae",
]
