"""Host mirror of ``pgx.bridge_bidding`` for the hot path.

Same names and argument meaning as the reference's call sites — ``BridgeBidding(path)``
(ppo.py:303), ``env.init(key)`` (ppo.py:305), ``env.step(state, action)`` (src/utils.py:44),
``env.observation_shape`` (ppo.py:241), ``_observe(state, player_id)`` (src/duplicate.py:134),
``State`` attribute names (src/utils.py:36-52, src/duplicate.py:113-127,170-173) — but
BATCHED: a ``State`` is N tables backed by torch tensors on one MI355X (there is no vmap),
and every operation is one HIP kernel launch through the C-ABI in ``include/brl_hip.h``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _capi
from ._capi import NUM_ACTIONS, OBS_SIZE, STATE_WORDS, check, ptr


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """the current HIP stream's handle (what every C-ABI call takes).  torch.cuda.current_stream() costs ~9 us per call (device
    lookups, a Stream object): ~30 us of an evaluator's 75 us host iteration; the raw accessor is the same handle in ~0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def load_dds_table(path: str):
    """``dds_results/*.npy`` as written for pgx: array (2, L, 4) int32 = (keys, values)
    (ppo.py:297-308).  [RECALL — the packing is restated from pgx 1.4.0, see DESIGN.md]"""
    arr = np.load(path)
    if arr.ndim != 3 or arr.shape[0] != 2 or arr.shape[2] != 4:
        raise ValueError(f"{path}: expected shape (2, L, 4), got {arr.shape}")
    return np.ascontiguousarray(arr[0], dtype=np.int32), np.ascontiguousarray(arr[1], dtype=np.int32)


class State:
    """N bridge tables.  ``packed`` is the opaque [N,16] int64 device tensor the kernels work on;
    every pgx ``State`` attribute is materialised on demand by one ``brl_get_fields`` launch."""

    _FIELDS = {
        # attribute -> (C field, dtype, trailing shape)
        "current_player": ("current_player", torch.int32, ()),
        "terminated": ("terminated", torch.bool, ()),
        "rewards": ("rewards", torch.float32, (4,)),
        "_step_count": ("step_count", torch.int32, ()),
        "_turn": ("turn", torch.int32, ()),
        "_dealer": ("dealer", torch.int32, ()),
        "_vul_NS": ("vul_ns", torch.bool, ()),
        "_vul_EW": ("vul_ew", torch.bool, ()),
        "_shuffled_players": ("shuffled_players", torch.int32, (4,)),
        "_last_bid": ("last_bid", torch.int32, ()),
        "_last_bidder": ("last_bidder", torch.int32, ()),
        "_call_x": ("call_x", torch.bool, ()),
        "_call_xx": ("call_xx", torch.bool, ()),
        "_pass_num": ("pass_num", torch.int32, ()),
        "_first_denomination_NS": ("first_denomination_ns", torch.int32, (5,)),
        "_first_denomination_EW": ("first_denomination_ew", torch.int32, (5,)),
        "_hand": ("hand", torch.int32, (52,)),
        "_dds_tricks": ("tricks", torch.uint8, (20,)),
        "_lut_idx": ("lut_idx", torch.int32, ()),
        "_board_count": ("board_ctr", torch.int32, ()),
        "_illegal": ("illegal", torch.bool, ()),
    }

    def __init__(self, env: "BridgeBidding", packed: torch.Tensor, cache: Optional[dict] = None):
        assert packed.dtype == torch.int64 and packed.dim() == 2 and packed.shape[1] == STATE_WORDS
        self.env = env
        self.packed = packed
        self._cache = dict(cache or {})

    @property
    def num_envs(self) -> int:
        return self.packed.shape[0]

    def __len__(self):
        return self.num_envs

    def _fetch(self, names):
        n = self.num_envs
        f = _capi.Fields()
        new = {}
        for name in names:
            cname, dtype, shape = self._FIELDS[name]
            t = torch.empty((n,) + shape, dtype=dtype, device=self.packed.device)
            setattr(f, cname, ptr(t))
            new[name] = t
        check(_capi.lib().brl_get_fields(self.env._h, ptr(self.packed), n, C.byref(f), _stream()))
        self._cache.update(new)

    def __getattr__(self, name):
        if name in State._FIELDS:
            if name not in self._cache:
                self._fetch([name])
            return self._cache[name]
        raise AttributeError(name)

    @property
    def truncated(self):
        return torch.zeros(self.num_envs, dtype=torch.bool, device=self.packed.device)

    @property
    def observation(self):
        if "observation" not in self._cache:
            self._observe_current()
        return self._cache["observation"]

    @property
    def legal_action_mask(self):
        if "legal_action_mask" not in self._cache:
            self._observe_current()
        return self._cache["legal_action_mask"]

    def _observe_current(self):
        obs, mask = self.env._observe_raw(self.packed, None)
        self._cache["observation"] = obs
        self._cache["legal_action_mask"] = mask

    def all_fields(self):
        """dict of every pgx attribute (one launch) — used by the parity tests."""
        self._fetch([k for k in State._FIELDS if k not in self._cache])
        out = {k: self._cache[k] for k in State._FIELDS}
        out["observation"] = self.observation
        out["legal_action_mask"] = self.legal_action_mask
        return out

    _BOARD_KEYS = ("_hand", "_dealer", "_vul_NS", "_vul_EW", "_shuffled_players")

    def replace(self, **kw):
        """``state.replace(**fields)`` — the call sites of the reference: ``rewards=, terminated=`` of the macro-step
        (src/utils.py:128) and of the evaluators (src/evaluation.py:591); the board fields ``_hand, _dealer, _vul_NS,
        _vul_EW, _shuffled_players`` that ``_duplicate_init`` / ``wb5`` set on a freshly initialised state
        (src/duplicate.py:120-128, wb5/utils.py:69-75, wb5/vis_pgx.py:51-56).  ``current_player`` and
        ``legal_action_mask`` are DERIVED here (``_shuffled_players[(dealer + turn) % 4]``; the rule mask), so a value
        passed for them must agree with the other fields (checked for ``current_player``); ``observation`` is recomputed.
        Replacing ``_hand`` looks the new deal up in the handle's double-dummy table (pgx does that at the end of the
        auction); a deal that is not in the table gets zero tricks.  Returns a new State sharing nothing with the old."""
        known = set(self._BOARD_KEYS) | {"rewards", "terminated", "current_player", "legal_action_mask", "observation"}
        unknown = set(kw) - known
        if unknown:
            raise NotImplementedError(f"State.replace({sorted(unknown)}): not a field the reference ever replaces")
        dev = self.packed.device
        packed = self.packed.clone()
        board = any(k in kw for k in self._BOARD_KEYS)
        keep = ("current_player",) if not board else ()
        if not board:
            keep += ("observation", "legal_action_mask")
        cache = {k: v for k, v in self._cache.items() if k in keep}
        n = self.num_envs
        sc = packed[:, 11]
        if "_dealer" in kw:
            d = torch.as_tensor(kw["_dealer"], device=dev).to(torch.int64).expand(n) & 3
            sc = (sc & ~3) | d
        if "_vul_NS" in kw:
            sc = (sc & ~(1 << 2)) | (torch.as_tensor(kw["_vul_NS"], device=dev).to(torch.int64).expand(n) & 1) << 2
        if "_vul_EW" in kw:
            sc = (sc & ~(1 << 3)) | (torch.as_tensor(kw["_vul_EW"], device=dev).to(torch.int64).expand(n) & 1) << 3
        if "_shuffled_players" in kw:
            sp = torch.as_tensor(kw["_shuffled_players"], device=dev).to(torch.int64).expand(n, 4)
            if not bool((sp.sort(dim=1).values == torch.arange(4, device=dev)).all()):
                raise ValueError("_shuffled_players rows must be permutations of 0..3")
            code = sp[:, 0] | (sp[:, 1] << 2) | (sp[:, 2] << 4) | (sp[:, 3] << 6)
            sc = (sc & ~(0xFF << 4)) | (code << 4)
        packed[:, 11] = sc
        if "_hand" in kw:
            hand = torch.as_tensor(kw["_hand"], device=dev).to(torch.int64).expand(n, 52)
            if not bool((hand.sort(dim=1).values == torch.arange(52, device=dev)).all()):
                raise ValueError("every _hand row must be a permutation of 0..51")
            # observation bit of a pgx card id: rank * 4 + suit with ranks 2..A and suits C,D,H,S (wb5/utils.py:18-19)
            idx = ((hand % 13 + 12) % 13) * 4 + (3 - hand // 13)
            words = (torch.ones_like(idx) << (idx + 4)).reshape(n, 4, 13).sum(dim=2)
            packed[:, 7:11] = words
            tricks, rows = self.env.lookup_deal(hand.cpu().numpy())
            t = torch.as_tensor(tricks, dtype=torch.int64, device=dev)           # [n,20] = [seat][C,D,H,S,NT]
            nib = torch.zeros((n, 20), dtype=torch.int64, device=dev)
            for seat in range(4):
                for den in range(5):
                    nib[:, seat * 5 + (4 - den)] = t[:, seat * 5 + den]         # nibble index of bridge_device.hpp
            lo = sum(nib[:, i] << (4 * i) for i in range(16))
            hi = sum(nib[:, 16 + i] << (4 * i) for i in range(4))
            packed[:, 13] = lo
            packed[:, 12] = (packed[:, 12] & 0xFFFFFFFF) | (hi << 32)
            packed[:, 14] = (packed[:, 14] & ~0xFFFFFFFF) | torch.as_tensor(rows, dtype=torch.int64, device=dev) & 0xFFFFFFFF
        if "rewards" in kw:
            r = torch.as_tensor(kw["rewards"], device=dev).to(torch.float32).round().to(torch.int64) & 0xFFFF
            packed[:, 15] = r[:, 0] | (r[:, 1] << 16) | (r[:, 2] << 32) | (r[:, 3] << 48)
            cache["rewards"] = torch.as_tensor(kw["rewards"], device=dev).to(torch.float32)
        if "terminated" in kw:
            term = torch.as_tensor(kw["terminated"], device=dev)
            packed[:, 11] = (packed[:, 11] & ~(1 << 25)) | (term.to(torch.int64) << 25)
            cache["terminated"] = term.to(torch.bool)
        out = State(self.env, packed, cache)
        if "current_player" in kw:
            want = torch.as_tensor(kw["current_player"], device=dev).to(torch.int32).expand(n)
            if not bool((out.current_player == want).all()):
                raise ValueError("current_player is derived: it must equal _shuffled_players[(_dealer + _turn) % 4]")
        return out


class BridgeBidding:
    """``pgx.bridge_bidding.BridgeBidding`` for one GPU.

    dds_results_table_path: a pgx ``dds_results/*.npy`` hash table, or pass ``lut=(keys, values)``
    (int32 [L,4] each) directly.  ``env_offset`` is the global index of this shard's table 0
    (rank * num_envs under torch.distributed) so that shards draw different boards.
    """

    observation_shape = (OBS_SIZE,)
    num_actions = NUM_ACTIONS
    num_players = 4

    def __init__(self, dds_results_table_path: Optional[str] = None, *, lut=None, device=None, env_offset: int = 0):
        L = _capi.lib()  # raises if the HIP extension is missing — no fallback
        if not torch.cuda.is_available():
            raise _capi.BrlError("brl_amd needs a ROCm GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if dds_results_table_path is not None:
            lut = load_dds_table(dds_results_table_path)
        self._h = C.c_void_p()
        self._lut_len = 0
        self._seed = 0
        self.env_offset = int(env_offset)
        keys, values = self._lut_arrays(lut)
        check(L.brl_create(self.device.index, keys.ctypes.data if len(keys) else None,
                           values.ctypes.data if len(values) else None, len(keys), C.byref(self._h)))
        self._lut_len = len(keys)
        self._lut_keys, self._lut_values, self._key_index = keys, values, None
        check(L.brl_set_rng(self._h, 0, self.env_offset))

    @staticmethod
    def _lut_arrays(lut):
        if lut is None:
            return np.zeros((0, 4), np.int32), np.zeros((0, 4), np.int32)
        keys = np.ascontiguousarray(lut[0], dtype=np.int32).reshape(-1, 4)
        values = np.ascontiguousarray(lut[1], dtype=np.int32).reshape(-1, 4)
        if keys.shape != values.shape:
            raise ValueError("lut keys / values must both be int32 [L,4]")
        return keys, values

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _capi.lib().brl_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- LUT rotation (ppo.py:525-549) --------------------------------------------------
    def set_lut(self, lut):
        keys, values = self._lut_arrays(lut)
        check(_capi.lib().brl_set_lut(self._h, keys.ctypes.data if len(keys) else None,
                                      values.ctypes.data if len(values) else None, len(keys)))
        self._lut_len = len(keys)
        self._lut_keys, self._lut_values, self._key_index = keys, values, None

    def lookup_deal(self, hand):
        """hand int [n,52] (13 pgx card ids per seat) -> (tricks uint8 [n,20], LUT row int64 [n], -1 if absent): the
        deal's entry in this handle's double-dummy table, found by its pgx key (host-side dictionary; not a hot path)."""
        hand = np.asarray(hand, dtype=np.int64).reshape(-1, 52)
        n = hand.shape[0]
        if self._key_index is None:
            self._key_index = {tuple(int(x) for x in k): i for i, k in enumerate(self._lut_keys)} if self._lut_len else {}
        owner = np.zeros((n, 52), np.int64)
        owner[np.arange(n)[:, None], hand] = np.repeat(np.arange(4), 13)[None, :]
        keys = (owner.reshape(n, 4, 13) * (4 ** np.arange(12, -1, -1, dtype=np.int64))).sum(-1).astype(np.int32)
        rows = np.array([self._key_index.get(tuple(int(x) for x in k), -1) for k in keys], np.int64)
        tricks = np.zeros((n, 20), np.uint8)
        found = rows >= 0
        if found.any():
            v = self._lut_values[rows[found]].astype(np.int64)                    # one word per declarer seat
            tricks[found] = ((v[:, :, None] >> (4 * np.arange(4, -1, -1))) & 15).reshape(-1, 20)
        return tricks, rows

    def seed(self, seed: int, env_offset: Optional[int] = None):
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if env_offset is not None:
            self.env_offset = int(env_offset)
        check(_capi.lib().brl_set_rng(self._h, self._seed, self.env_offset))

    def _new_packed(self, n):
        return torch.empty((n, STATE_WORDS), dtype=torch.int64, device=self.device)

    # ---- init --------------------------------------------------------------------------------
    def init(self, key, num_envs: Optional[int] = None) -> State:
        """``jax.vmap(env.init)(keys)`` (ppo.py:305,318).  ``key``: an int seed (then ``num_envs`` is
        required) or an integer array with one entry/row per env, whose first element seeds the
        counter-based generator.  JAX's threefry streams are not reproducible outside JAX; the
        generator is Philox4x32-10 keyed by (seed, global env index, board number)."""
        if isinstance(key, (int, np.integer)):
            seed = int(key)
            if num_envs is None:
                raise ValueError("env.init(seed) needs num_envs")
        else:
            k = torch.as_tensor(key).reshape(len(key), -1)
            num_envs = k.shape[0] if num_envs is None else num_envs
            seed = int(k[0].to(torch.int64).sum().item())
        self.seed(seed)
        packed = self._new_packed(num_envs)
        check(_capi.lib().brl_init_random(self._h, ptr(packed), num_envs, 0, _stream()))
        return State(self, packed)

    def init_from_deals(self, hand, dealer, vul_ns, vul_ew, shuffled_players, tricks) -> State:
        """Explicit deals: exactly the fields ``_duplicate_init`` copies (src/duplicate.py:120-128)
        plus the double-dummy tricks [N,20] ([declarer seat][C,D,H,S,NT]) used for the reward."""
        dev = self.device
        hand = torch.as_tensor(hand, dtype=torch.int32, device=dev).reshape(-1, 52).contiguous()
        n = hand.shape[0]
        dealer = torch.as_tensor(dealer, dtype=torch.int32, device=dev).expand(n).contiguous()
        vul_ns = torch.as_tensor(vul_ns, device=dev).to(torch.uint8).expand(n).contiguous()
        vul_ew = torch.as_tensor(vul_ew, device=dev).to(torch.uint8).expand(n).contiguous()
        sh = torch.as_tensor(shuffled_players, dtype=torch.int32, device=dev).expand(n, 4).contiguous()
        tricks = torch.as_tensor(tricks, device=dev).to(torch.uint8).reshape(-1, 20).expand(n, 20).contiguous()
        if not bool(((hand.sort(dim=1).values == torch.arange(52, device=dev, dtype=torch.int32)).all())):
            raise ValueError("every hand row must be a permutation of 0..51")
        if not bool((sh.sort(dim=1).values == torch.arange(4, device=dev, dtype=torch.int32)).all()):
            raise ValueError("shuffled_players rows must be permutations of 0..3")
        packed = self._new_packed(n)
        check(_capi.lib().brl_init_from_deals(self._h, ptr(packed), n, ptr(hand), ptr(dealer), ptr(vul_ns),
                                              ptr(vul_ew), ptr(sh), ptr(tricks), _stream()))
        return State(self, packed)

    # ---- step ----------------------------------------------------------------------------------
    def step(self, state: State, action, *, autoreset: bool = False, inplace: bool = False) -> State:
        """``env.step(state, action)`` (src/utils.py:44); ``autoreset=True`` is
        ``auto_reset(env.step, env.init)(state, action)`` (src/utils.py:9-58).  ``inplace`` reuses the
        state's storage (rollouts); the default allocates, like the reference's immutable pytrees."""
        n = state.num_envs
        action = torch.as_tensor(action, device=self.device).to(torch.int32).contiguous()
        if action.shape != (n,):
            raise ValueError(f"action must have shape ({n},)")
        out = state.packed if inplace else self._new_packed(n)
        obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=self.device)
        mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=self.device)
        rewards = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        term = torch.empty(n, dtype=torch.bool, device=self.device)
        cur = torch.empty(n, dtype=torch.int32, device=self.device)
        check(_capi.lib().brl_step(self._h, ptr(state.packed), ptr(out), n, ptr(action), int(autoreset), ptr(obs),
                                   ptr(mask), ptr(rewards), ptr(term), ptr(cur), _stream()))
        return State(self, out, {"observation": obs, "legal_action_mask": mask, "rewards": rewards,
                                 "terminated": term, "current_player": cur})

    # ---- observe -------------------------------------------------------------------------------
    def _observe_raw(self, packed, player_id):
        n = packed.shape[0]
        obs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=self.device)
        mask = torch.empty((n, NUM_ACTIONS), dtype=torch.bool, device=self.device)
        pid = None
        if player_id is not None:
            pid = torch.as_tensor(player_id, device=self.device).to(torch.int32).expand(n).contiguous()
        check(_capi.lib().brl_observe(self._h, ptr(packed), n, ptr(pid), ptr(obs), ptr(mask), _stream()))
        return obs, mask

    def observe(self, state: State, player_id=None):
        return self._observe_raw(state.packed, player_id)[0]


def _observe(state: State, player_id):
    """``pgx.bridge_bidding._observe(state, player_id)`` (src/duplicate.py:6,134)."""
    return state.env.observe(state, player_id)


def _player_position(player, state: State):
    """``pgx.bridge_bidding._player_position`` (src/duplicate.py:6): seat of a player id, -1 for -1."""
    sp = state._shuffled_players
    player = torch.as_tensor(player, device=sp.device).to(torch.int32).expand(sp.shape[0])
    pos = (sp == player[:, None]).to(torch.int32).argmax(dim=1).to(torch.int32)
    return torch.where(player < 0, torch.full_like(pos, -1), pos)
