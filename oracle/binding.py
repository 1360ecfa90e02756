"""ctypes + numpy binding of oracle/bridge_oracle.c.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# mirrors `struct orc_state` field for field (C layout, natural alignment)
STATE_DTYPE = np.dtype(
    [
        ("current_player", "<i4"),
        ("terminated", "<i4"),
        ("truncated", "<i4"),
        ("step_count", "<i4"),
        ("turn", "<i4"),
        ("dealer", "<i4"),
        ("vul_ns", "<i4"),
        ("vul_ew", "<i4"),
        ("last_bid", "<i4"),
        ("last_bidder", "<i4"),
        ("call_x", "<i4"),
        ("call_xx", "<i4"),
        ("pass_num", "<i4"),
        ("illegal", "<i4"),
        ("mask_all", "<i4"),
        ("lut_idx", "<i4"),
        ("board_ctr", "<u4"),
        ("shuffled_players", "<i4", (4,)),
        ("first_denomination_ns", "<i4", (5,)),
        ("first_denomination_ew", "<i4", (5,)),
        ("rewards", "<f4", (4,)),
        ("hand", "<i4", (52,)),
        ("tricks", "u1", (20,)),
        ("legal_action_mask", "u1", (38,)),
        ("observation", "u1", (480,)),
        ("bidding_history", "<i2", (320,)),
    ],
    align=True,
)

TABLE_INFO_DTYPE = np.dtype(
    [
        ("terminated", "<i4"),
        ("rewards", "<f4", (4,)),
        ("last_bid", "<i4"),
        ("last_bidder", "<i4"),
        ("call_x", "<i4"),
        ("call_xx", "<i4"),
    ],
    align=True,
)


def build_dir() -> str:
    """oracle/_build, or oracle/_build_asan under BRL_ORACLE_BUILD=asan (the `make asan` objects: AddressSanitizer + UBSan,
    loaded into a python started with LD_PRELOAD=libasan.so:libubsan.so — scripts/cpu_sanitize.sh)"""
    return os.path.join(_HERE, "_build_asan" if os.environ.get("BRL_ORACLE_BUILD") == "asan" else "_build")


def lib_path() -> str:
    return os.path.join(build_dir(), "liboracle.so")


def shim_path() -> str:
    """the oracle behind the C symbols of include/brl_hip.h (oracle/brl_shim.c)"""
    return os.path.join(build_dir(), "liboracle_brl.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "bridge_oracle.c")
    out = lib_path()
    shim = shim_path()
    asan = os.environ.get("BRL_ORACLE_BUILD") == "asan"
    deps = [src, os.path.join(_HERE, "brl_shim.c"), os.path.join(_HERE, "Makefile"),
            os.path.join(os.path.dirname(_HERE), "include", "brl_hip.h")]
    # (checked here, without starting a process: a test session calls this once per Oracle(), often with the GPU runtime up)
    stale = force or not (os.path.exists(out) and os.path.exists(shim)) or \
        min(os.path.getmtime(out), os.path.getmtime(shim)) < max(os.path.getmtime(d) for d in deps)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "--no-print-directory", "-s"] + (["asan"] if asan else [])
                              + (["-B"] if force else []))
    return out


NATIVE_FLAGS = ["-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c11"]


def build_native():
    """bench.py's cpu_baseline leg: the same oracle compiled `-O3 -march=native -fopenmp` ON THE BOX that
    times it (a -march=native object must not travel between machines: it is rebuilt whenever the host CPU differs from
    the one recorded beside it).  Returns (path, flags string); falls back to the portable test build if gcc fails."""
    src = os.path.join(_HERE, "bridge_oracle.c")
    out = os.path.join(_HERE, "_build", "liboracle_native.so")
    tag = out + ".host"
    host = ""
    try:
        host = next(l for l in open("/proc/cpuinfo") if l.startswith("model name")).strip()
    except (OSError, StopIteration):
        pass
    try:
        fresh = os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src) and open(tag).read() == host
    except OSError:
        fresh = False
    if not fresh:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        try:
            subprocess.check_call([os.environ.get("CC", "gcc")] + NATIVE_FLAGS + ["-shared", "-o", out, src, "-lm"])
            with open(tag, "w") as f:
                f.write(host)
        except (subprocess.CalledProcessError, OSError):
            return build(), "-O2 -ffp-contract=off -fopenmp (portable test build: the native build failed)"
    return out, " ".join(NATIVE_FLAGS)


def _p(a, ctype=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Thin numpy-facing wrapper.  States are numpy structured arrays of STATE_DTYPE."""

    def __init__(self, lut_keys: np.ndarray | None = None, lut_values: np.ndarray | None = None, lib_file: str | None = None):
        build()
        self.lib = C.CDLL(lib_file or lib_path())
        L = self.lib
        assert L.orc_sizeof_state() == STATE_DTYPE.itemsize, (L.orc_sizeof_state(), STATE_DTYPE.itemsize)
        assert L.orc_sizeof_table_info() == TABLE_INFO_DTYPE.itemsize
        L.orc_score.restype = C.c_int
        L.orc_action_draw.restype = C.c_uint32
        L.orc_action_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_random_action.restype = C.c_int
        L.orc_card_to_obs_index.restype = C.c_int
        self.set_lut(lut_keys, lut_values)

    # ---- LUT -----------------------------------------------------------------------
    def set_lut(self, keys, values):
        if keys is None:
            self.lut_keys = self.lut_values = None
            self.lut_len = 0
            return
        self.lut_keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 4)
        self.lut_values = np.ascontiguousarray(values, dtype=np.int32).reshape(-1, 4)
        assert self.lut_keys.shape == self.lut_values.shape
        self.lut_len = self.lut_keys.shape[0]

    # ---- scalar helpers ------------------------------------------------------------
    def philox(self, ctr, key):
        c = np.asarray(ctr, dtype=np.uint32)
        k = np.asarray(key, dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        self.lib.orc_philox4x32(_p(c), _p(k), _p(out))
        return out

    def score(self, denomination, level, vul, call_x, call_xx, trick) -> int:
        return int(self.lib.orc_score(int(denomination), int(level), int(vul), int(call_x), int(call_xx), int(trick)))

    def card_to_obs_index(self, card: int) -> int:
        return int(self.lib.orc_card_to_obs_index(int(card)))

    def key_to_hand(self, key):
        k = np.ascontiguousarray(key, dtype=np.int32)
        hand = np.zeros(52, dtype=np.int32)
        self.lib.orc_key_to_hand(_p(k), _p(hand))
        return hand

    def hand_to_key(self, hand):
        h = np.ascontiguousarray(hand, dtype=np.int32)
        key = np.zeros(4, dtype=np.int32)
        self.lib.orc_hand_to_key(_p(h), _p(key))
        return key

    def value_to_tricks(self, value):
        v = np.ascontiguousarray(value, dtype=np.int32)
        t = np.zeros(20, dtype=np.uint8)
        self.lib.orc_value_to_tricks(_p(v), _p(t))
        return t

    def tricks_to_value(self, tricks):
        t = np.ascontiguousarray(tricks, dtype=np.uint8).reshape(20)
        v = np.zeros(4, dtype=np.int32)
        self.lib.orc_tricks_to_value(_p(t), _p(v))
        return v

    def imp_reward(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.zeros(4, dtype=np.float32)
        self.lib.orc_imp_reward(_p(a), _p(b), _p(out))
        return out

    # ---- env -----------------------------------------------------------------------
    def init_explicit(self, hand, dealer, vul_ns, vul_ew, shuffled, tricks):
        """Batched: hand [N,52], dealer [N], vul [N], shuffled [N,4], tricks [N,20]."""
        hand = np.ascontiguousarray(hand, dtype=np.int32).reshape(-1, 52)
        n = hand.shape[0]
        dealer = np.broadcast_to(np.asarray(dealer, dtype=np.int32), (n,))
        vul_ns = np.broadcast_to(np.asarray(vul_ns, dtype=np.int32), (n,))
        vul_ew = np.broadcast_to(np.asarray(vul_ew, dtype=np.int32), (n,))
        shuffled = np.ascontiguousarray(np.broadcast_to(np.asarray(shuffled, dtype=np.int32), (n, 4)))
        tricks = np.ascontiguousarray(np.broadcast_to(np.asarray(tricks, dtype=np.uint8).reshape(-1, 20), (n, 20)))
        st = np.zeros(n, dtype=STATE_DTYPE)
        for i in range(n):
            self.lib.orc_init_explicit(
                C.c_void_p(st.ctypes.data + i * STATE_DTYPE.itemsize),
                _p(hand[i]), int(dealer[i]), int(vul_ns[i]), int(vul_ew[i]), _p(shuffled[i]), _p(tricks[i]),
            )
        return st

    def init_random(self, n, seed, env_offset=0):
        st = np.zeros(n, dtype=STATE_DTYPE)
        self.lib.orc_init_random_batch(
            _p(st), C.c_int64(n), C.c_uint64(seed), C.c_uint64(env_offset),
            _p(self.lut_keys), _p(self.lut_values), C.c_int64(self.lut_len),
        )
        return st

    def step(self, st, action, autoreset=False, seed=0, env_offset=0):
        action = np.ascontiguousarray(action, dtype=np.int32)
        assert action.shape == (st.shape[0],)
        self.lib.orc_step_batch(
            _p(st), C.c_int64(st.shape[0]), _p(action), int(bool(autoreset)), C.c_uint64(seed),
            C.c_uint64(env_offset), _p(self.lut_keys), _p(self.lut_values), C.c_int64(self.lut_len),
        )
        return st

    def observe(self, st, player_id):
        n = st.shape[0]
        player_id = np.broadcast_to(np.asarray(player_id, dtype=np.int32), (n,))
        obs = np.zeros((n, 480), dtype=np.uint8)
        for i in range(n):
            self.lib.orc_observe(C.c_void_p(st.ctypes.data + i * STATE_DTYPE.itemsize), int(player_id[i]), _p(obs[i]))
        return obs

    def action_draw(self, seed, env_id, draw) -> int:
        return int(self.lib.orc_action_draw(seed, env_id, draw))

    def random_action(self, st_i, draw):
        """st_i: a length-1 slice of a state array."""
        n = C.c_int(0)
        a = self.lib.orc_random_action(_p(st_i), C.c_uint32(draw), C.byref(n))
        return int(a), int(n.value)

    def rollout_random(self, st, num_steps, seed, substeps=1, env_offset=0, draw_base=0, reward_scale=7600.0,
                       store_obs=True):
        n = st.shape[0]
        T = num_steps
        out = {
            "obs": np.zeros((T, n, 480), dtype=np.uint8) if store_obs else None,
            "legal_action_mask": np.zeros((T, n, 38), dtype=np.uint8),
            "action": np.zeros((T, n), dtype=np.int32),
            "log_prob": np.zeros((T, n), dtype=np.float32),
            "value": np.zeros((T, n), dtype=np.float32),
            "reward": np.zeros((T, n), dtype=np.float32),
            "done": np.zeros((T, n), dtype=np.uint8),
        }
        tc = C.c_int64(0)
        self.lib.orc_rollout_random(
            _p(st), C.c_int64(n), int(T), int(substeps), C.c_uint64(seed), C.c_uint64(env_offset),
            C.c_uint32(draw_base), _p(self.lut_keys), _p(self.lut_values), C.c_int64(self.lut_len),
            C.c_float(reward_scale),
            _p(out["obs"]), _p(out["legal_action_mask"]), _p(out["action"]), _p(out["log_prob"]),
            _p(out["value"]), _p(out["reward"]), _p(out["done"]), C.byref(tc),
        )
        out["terminated_count"] = int(tc.value)
        return out

    def gae(self, done, value, reward, last_val, gamma, gae_lambda):
        done = np.ascontiguousarray(done, dtype=np.uint8)
        value = np.ascontiguousarray(value, dtype=np.float32)
        reward = np.ascontiguousarray(reward, dtype=np.float32)
        last_val = np.ascontiguousarray(last_val, dtype=np.float32)
        T, N = done.shape
        adv = np.zeros((T, N), dtype=np.float32)
        tgt = np.zeros((T, N), dtype=np.float32)
        # config["gamma"] * config["gae_lambda"] is a Python-float product (src/gae.py:29)
        gl = np.float32(float(gamma) * float(gae_lambda))
        self.lib.orc_gae(_p(done), _p(value), _p(reward), _p(last_val), C.c_float(gamma), C.c_float(gl),
                         int(T), C.c_int64(N), _p(adv), _p(tgt))
        return adv, tgt

    def duplicate_step(self, st, action, A, B):
        action = np.ascontiguousarray(action, dtype=np.int32)
        self.lib.orc_duplicate_step_batch(_p(st), C.c_int64(st.shape[0]), _p(action), _p(A), _p(B))
        return st, A, B

    @staticmethod
    def table_info_from(st):
        t = np.zeros(st.shape[0], dtype=TABLE_INFO_DTYPE)
        for f in ("terminated", "rewards", "last_bid", "last_bidder", "call_x", "call_xx"):
            t[f] = st[f]
        return t
