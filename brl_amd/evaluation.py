"""``src/evaluation.py`` on the GPU.

* ``make_simple_duplicate_evaluate`` — BASELINE config 3 (src/evaluation.py:69-204): N boards, each played at table A
  and then, seat-swapped, at table B (``duplicate_step``); greedy (``pi.mode()``) actions from two MLPs chosen by
  ``current_player in {0,1}`` (:146-151); returns (mean IMP, standard error, win rate) like :199-202.
* ``make_evaluate`` — the evaluator with bidding statistics (src/evaluation.py:207-1032), duplicate or single table,
  and ``make_evaluate_log`` (:1035-1115), the flat ``eval/...`` dict ppo.py logs every ``num_eval_step`` iterations.
* ``make_simple_evaluate`` — the single-table deterministic evaluator of src/evaluation.py:11-66.

Every loop iteration is one or two (merged) GEMM forwards in PyTorch-ROCm plus ONE ``brl_eval_step`` launch: team selection,
masked arg-max, the step log (illegal-action probability mass, step / pass / bid counters), ``duplicate_step`` and the
return accumulators all happen in that kernel; the end-of-run histograms come from ``brl_eval_reduce`` as exact integer
counts.  The loop condition ``~state.terminated.all()`` is watched without stalling the GPU (``_DoneWatch``; finished boards
keep receiving no-op steps, G9, so overshooting changes nothing)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _capi
from ._capi import NUM_ACTIONS, OBS_SIZE, check, ptr
from .bridge_bidding import BridgeBidding, State, _stream
from .duplicate import Table_info
from .models import InferenceSnapshot, make_forward_pass
from .utils import single_play_step_two_policy_commpetitive_deterministic

_NEG = torch.finfo(torch.float32).min


class _Shard:
    """This rank's share of the ``num_eval_envs`` boards of one evaluation (ppo.py:366-381,461-484 runs 3-4 evaluators of
    10 000 boards per iteration; with every rank evaluating all of them a node would do the work 8 times).  Boards are
    GLOBAL indices: rank r plays the contiguous range [offset, offset + n) — the env handle's ``env_offset``, so each board
    is dealt exactly as in a single-process evaluation — and the results are sums that are all-reduced once
    (float64 / int64: IMPs, counts and wins are integers, so every rank gets the single-process numbers).
    ``shard``: None / False = no sharding; True = torch.distributed's (rank, world) when a process group exists;
    (rank, world) = explicit."""

    def __init__(self, n_global: int, shard):
        import torch.distributed as dist
        self.forced = shard is True    # (shard=True at world 1 under BRL_FORCE_DIST=1: one shard, the sums still all-reduced)
        if shard is True:
            shard = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else None
        self.rank, self.world = (0, 1) if not shard else (int(shard[0]), int(shard[1]))
        base, extra = divmod(int(n_global), self.world)
        self.n_global = int(n_global)
        self.n = base + (1 if self.rank < extra else 0)
        self.offset = self.rank * base + min(self.rank, extra)

    @property
    def active(self) -> bool:
        from .dist import distributed
        return self.world > 1 or (self.forced and distributed())

    def init(self, env: BridgeBidding, rng_key) -> State:
        if self.active:
            env.env_offset = self.offset      # (env.init re-keys the handle with its current env_offset)
        return env.init(rng_key, num_envs=self.n)

    def allsum(self, t: torch.Tensor) -> torch.Tensor:
        """element-wise sum over the ranks (float64 / int64), same result on every rank"""
        if self.active:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t


# rows up to which an evaluator's forward goes through brl_mlp_forward_rows (above, the library GEMMs' larger tiles win on the GPU
# and the host is not the bound); BRL_EVAL_OWN_ROWS=0 switches it off (A/B)
_OWN_FORWARD_ROWS = int(os.environ.get("BRL_EVAL_OWN_ROWS", "1024"))


_STEP_CASTS = os.environ.get("BRL_EVAL_STEP_CASTS", "1") != "0"   # (0: a cast launch in front of every full-batch forward: A/B)
_HOST_COUNT = os.environ.get("BRL_EVAL_HOST_COUNT", "1") != "0"   # (0: count to device memory + a copy launch, the earlier form: A/B)
_WORD_VISIBLE = [True]   # cleared for the process when a launch's store into pinned memory was never seen by the host (_DoneWatch.poll)


def masked_mode(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``Categorical(logits + finfo.min * ~mask).mode()``: arg-max over the legal actions."""
    return torch.where(mask, logits, torch.full_like(logits, _NEG)).argmax(dim=-1).to(torch.int32)


class _Forward:
    """logits of one team's network for a batch of observations: the fused inference snapshot when the architecture is
    covered (DeepMind / ReLU), else the module itself.  Returns a float32 [n, >=38] matrix whose first 38 columns are
    the logits (row stride may be 39: merged actor + critic heads)."""

    def __init__(self, forward_pass, params):
        self.fp, self.params = forward_pass, params
        self.snap = InferenceSnapshot.make(params, views=os.environ.get("BRL_SNAPSHOT_VIEWS", "1") != "0")
        self.ref = self._by_reference(params)

    @staticmethod
    def _by_reference(m):
        """brl_mlp_ref of a "DeepMind" fp32 network — pointers to the module's own parameters, for brl_mlp_forward_rows (the
        evaluators' small-batch iterations: one host call per forward) — or None where that entry point does not apply"""
        act = getattr(m, "act", None)
        if not str(getattr(m, "model", "")).startswith("DeepMind") or act not in (torch.relu, torch.tanh):
            return None
        vec = [t for lin in m.body for t in (lin.weight, lin.bias)] + [m.actor.weight, m.critic.weight]   # read 16 bytes at a time
        ts = vec + [m.actor.bias, m.critic.bias]   # (read one by one: inside FusedMinibatch's flat buffer critic.bias sits 8 bytes off)
        hidden = m.body[0].weight.shape[0]
        if (len(m.body) > 8 or hidden % 4 or hidden > 1024 or m.body[0].weight.shape[1] != OBS_SIZE
                or m.actor.weight.shape[0] != NUM_ACTIONS
                or any((not t.is_cuda) or t.dtype != torch.float32 or not t.is_contiguous() for t in ts)
                or any(t.data_ptr() % 16 for t in vec)
                or any(lin.weight.shape[0] != hidden for lin in m.body)):
            return None
        r = _capi.MlpRef()
        r.nlayers, r.act, r.in_features, r.hidden = len(m.body), 0 if act is torch.relu else 1, OBS_SIZE, hidden
        for i, lin in enumerate(m.body):
            r.w[i], r.b[i] = lin.weight.data_ptr(), lin.bias.data_ptr()
        r.actor_w, r.actor_b = m.actor.weight.data_ptr(), m.actor.bias.data_ptr()
        r.critic_w, r.critic_b = m.critic.weight.data_ptr(), m.critic.bias.data_ptr()
        return r

    def rows(self, obs_bool, idx, m, out, env):
        """logits (+ value) of the boards idx[0..m) written to their rows of ``out`` [n, >= 39]: ONE call into the library"""
        hidden = int(self.ref.hidden)
        need = m * (OBS_SIZE + 2 * hidden)
        if getattr(self, "_scratch", None) is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=out.device)
        di = out.device.index if out.device.index is not None else torch.cuda.current_device()
        check(_capi.lib().brl_mlp_forward_rows(di, C.byref(self.ref), ptr(obs_bool), ptr(idx), m, ptr(self._scratch),
                                               self._scratch.numel(), out.data_ptr(), out.stride(0), _stream()))

    def __call__(self, obs_bool, obs_f32):
        if self.snap is not None:
            # forwards of >= 4096 rows run on brl_linear_x3p (models.InferenceSnapshot.planes_for): it takes the 0/1 observation as bf16
            # (one plane) — from the bool rows, or already gathered + cast by the caller
            if obs_bool is not None and self.snap.planes_for(obs_bool.shape[0]):
                return self.snap.heads(obs_bool)
            return self.snap.heads(obs_f32)
        logits, _ = self.fp.apply(self.params, obs_f32)
        return logits.contiguous()


class EvalStats:
    """Device buffers behind ``brl_eval_stats`` (include/brl_hip.h)."""

    def __init__(self, n, device):
        self.illegal_prob_sum = torch.zeros((n, 2), dtype=torch.float32, device=device)
        self.step_count = torch.zeros((n, 2), dtype=torch.int32, device=device)
        self.pass_count = torch.zeros((n, 2), dtype=torch.int32, device=device)
        self.bid_count = torch.zeros((n, 2, 35), dtype=torch.int32, device=device)

    def ptrs(self):
        p = _capi.EvalStatsPtrs()
        for name in _capi.EvalStatsPtrs._names:
            setattr(p, name, ptr(getattr(self, name)))
        return p


class _DoneWatch:
    """The loop condition ``~state.terminated.all()`` (src/evaluation.py:120-122) without stalling the GPU and without running
    far past the end: after every iteration ONE launch (brl_live_index) counts the finished boards and stores the count, tagged
    with the iteration, straight into pinned host memory; before iteration i is launched the host waits — polling that word —
    for the count of iteration i - DEPTH, while the GPU is still busy with iteration i - 1.  The loop stops at most DEPTH
    iterations after the last board finished (finished boards take no-op calls, G9).  History: a blocking read every 16
    iterations idled the GPU for ~120 us per read and overshot by 8 iterations (of ~35); a copy to pinned memory + an event per
    iteration cost the stream ~10 us per iteration (the copy launch and the bubble behind the event's barrier packet).
    ``compact``: the same launch also leaves the indices of the boards still playing (no host round trip), so that the
    forwards can run on those rows only — boards only ever finish, so the list of iteration i - DEPTH
    is a superset of the boards playing at iteration i."""
    DEPTH, RING = 2, 4
    _epochs = 0

    def __init__(self, env, n, compact=False):
        self._h, self.n = env._h, n   # (the handle, not the environment: the environment keeps its watches — no reference cycle,
                                      #  both die by reference count, never at the collector's or the interpreter's whim)
        self.host = torch.zeros(self.RING, dtype=torch.int64).pin_memory()
        self.words = self.host.numpy()   # (the same memory: a plain load per poll)
        self.dev = torch.zeros(self.RING, dtype=torch.int64, device=env.device)
        self.events = [torch.cuda.Event() for _ in range(self.RING)]
        self.idx = None
        self.busy = False
        self.epoch = 0
        self.by_word = [False] * self.RING   # how slot k's count travels: the launch's own store into pinned memory, or copy + event
        self.last = None                     # event behind the last launch that writes this watch's memory (see release)
        if compact:   # (entries behind the live boards stay valid board indices: initialised with 0..n-1)
            self.idx = [torch.arange(n, dtype=torch.int64, device=env.device) for _ in range(self.RING)]

    @staticmethod
    def take(env, n, compact):
        """a watch of this environment that no loop is using — kept between evaluations (pinned memory, events and index lists
        cost ~10 launches to set up; the lists only ever hold valid board indices and a count is accepted only with the tag of
        the current loop, so nothing needs resetting); ``release`` hands it back"""
        cache = env.__dict__.setdefault("_done_watches", {})
        w = cache.get((n, compact))
        if w is None or w.busy:
            w = _DoneWatch(env, n, compact)
            cache.setdefault((n, compact), w)
        w.busy = True
        _DoneWatch._epochs += 1
        w.epoch = _DoneWatch._epochs & 0x7FFF   # (a loop's tags: epoch << 16 | iteration + 1 — never those of an earlier loop)
        return w

    def release(self):
        """hands the watch back; the launches of the loop's last iterations may still be in flight and WRITE this watch's pinned
        word: an event behind them is what `__del__` waits for before the memory goes away"""
        self.busy = False
        self.last = torch.cuda.Event()
        self.last.record()

    def __del__(self):
        try:
            if self.last is not None and not self.last.query():
                self.last.synchronize()
        except Exception:   # (interpreter shutdown, a capturing stream: never raise from a destructor)
            pass

    def _tag(self, i: int) -> int:
        return (self.epoch << 16) | ((i + 1) & 0xFFFF)

    def post(self, i: int, terminated: torch.Tensor):
        """after iteration i's launches: publish how many boards are finished (and which are not) — ONE launch"""
        k = i % self.RING
        live = ptr(self.idx[k]) if self.idx is not None else None
        self.by_word[k] = _HOST_COUNT and _WORD_VISIBLE[0]
        if self.by_word[k]:   # the launch stores tag | count in the pinned word itself (include/brl_hip.h: brl_live_index)
            check(_capi.lib().brl_live_index(self._h, ptr(terminated), self.n, live, self.host.data_ptr() + 8 * k, self._tag(i), _stream()))
            return
        check(_capi.lib().brl_live_index(self._h, ptr(terminated), self.n, live, ptr(self.dev[k:k + 1]), -1, _stream()))
        self.host[k:k + 1].copy_(self.dev[k:k + 1], non_blocking=True)
        self.events[k].record()

    def poll(self, i: int):
        """before iteration i's launches: (finished boards, indices of the others first) as of iteration i - DEPTH, or None"""
        j = i - self.DEPTH
        if j < 0:
            return None
        k = j % self.RING
        if self.by_word[k]:
            import time
            want, words, t0, spins = self._tag(j), self.words, None, 0
            while True:
                v = int(words[k])
                if (v >> 32) == want:
                    return v & 0xFFFFFFFF, (self.idx[k] if self.idx is not None else None)
                spins += 1
                if spins & 0xFF == 0:
                    time.sleep(0)             # (yield the GIL: other Python threads of the host process run while this one waits)
                if spins & 0x3FFF == 0:
                    t0 = t0 or time.perf_counter()
                    waited = time.perf_counter() - t0
                    # The launch has completed (the stream is idle) and its system-scope store is still not visible to the host:
                    # this host allocation is not coherent (HIP_HOST_COHERENT=0, coarse-grained pinned memory).  From here on the
                    # count travels by copy + event for the whole process; this iteration's count is simply unknown (the loop
                    # runs one more iteration: finished boards take no-op calls).
                    if waited > 2.0 and torch.cuda.current_stream().query() and (int(words[k]) >> 32) != want:
                        _WORD_VISIBLE[0] = False
                        import warnings
                        warnings.warn("brl_amd.evaluation: a count stored by a launch into pinned host memory never became "
                                      "visible (non-coherent host allocation?): falling back to copy + event per iteration.",
                                      RuntimeWarning)
                        return None
                    if waited > 60.0:         # (a launch that never completes must not hang the host for ever)
                        raise RuntimeError("brl_amd.evaluation: the finished-board count of an iteration never arrived (GPU fault?)")
        self.events[k].synchronize()
        return int(self.host[k]), (self.idx[k] if self.idx is not None else None)

    def finished(self, i: int) -> bool:
        r = self.poll(i)
        return r is not None and r[0] >= self.n


class _ActiveRows:
    """Forwards on the boards still playing only.  With 8192 boards and two DeepMind MLPs 53 % of the board-iterations of a
    duplicate evaluation belong to boards that are already finished (the last boards end at iteration ~30, half are done by
    iteration 16): their logits are never used (no-op calls, G9).  Rows to forward = the first `m` entries of `_DoneWatch`'s
    index list (m = the smallest step of a short ladder of batch sizes that holds the boards playing: the padding entries are
    valid board indices — at worst a board is forwarded twice, with identical results)."""

    def __init__(self, n):
        self.n, self.m, self.idx, self.full = n, n, None, None

    def update(self, polled):
        if polled is None or polled[1] is None:
            return
        live = self.n - polled[0]
        # a short ladder of batch sizes (n, 3n/4, n/2, 3n/8, n/4, n/8, .. >= 256, each rounded up to 256 rows): few distinct GEMM
        # shapes for the library's per-shape heuristics / TunableOp lookups to see (every new shape costs host time once)
        m, f = self.n, self.n
        while f > 256:
            for c in (3 * f // 4, f // 2):
                c = min(self.n, max(256, (c + 255) // 256 * 256))
                if live <= c < m:
                    m = c
            f //= 2
        if m < self.m:
            self.m, self.idx = m, polled[1][:m]

    def forward(self, fwd, obs, env, x=None):
        """``x``: the observation as float32 when the launch that produced it wrote that too (brl_eval_step_team.obs_f32)"""
        if getattr(fwd, "constant", False):   # (the same logits whatever the observation: nothing to compute)
            return fwd(obs, None)
        if self.idx is None or self.full is None:
            snap = getattr(fwd, "snap", None)
            if snap is not None and snap.planes_for(obs.shape[0]):      # (takes the bool rows themselves: _Forward.__call__)
                self.full = fwd(obs, None)
            else:
                self.full = fwd(obs, x if x is not None else obs.to(torch.float32))
            if getattr(fwd, "ref", None) is not None and self.full.stride(0) <= NUM_ACTIONS:
                # (the module's own [n, 38] logits: brl_mlp_forward_rows writes 38 + 1 numbers per row later on)
                wide = torch.empty((self.full.shape[0], NUM_ACTIONS + 2), dtype=torch.float32, device=self.full.device)
                wide[:, :NUM_ACTIONS] = self.full
                self.full = wide
            return self.full
        if self.m <= _OWN_FORWARD_ROWS and getattr(fwd, "ref", None) is not None:
            # few boards left: the iteration is bound by host launches — gather + cast, the layers, the heads and the scatter
            # back in one call (brl_mlp_forward_rows; rows of finished boards keep their last logits: never used)
            fwd.rows(obs, self.idx, self.m, self.full, env)
            return self.full
        snap = getattr(fwd, "snap", None)
        as_bf16 = snap is not None and snap.planes_for(self.m)      # (brl_linear_x3p: the observation as one bf16 plane)
        x = torch.empty((self.m, OBS_SIZE), dtype=torch.bfloat16 if as_bf16 else torch.float32, device=obs.device)
        check(_capi.lib().brl_obs_cast_rows(env._h, ptr(obs), ptr(self.idx), self.m, ptr(x), 1 if as_bf16 else 0, _stream()))   # gather + astype
        out = fwd(None, x)
        self.full[:, :out.shape[1]].index_copy_(0, self.idx, out)   # (rows of finished boards keep their last logits: never used)
        return self.full


def _eval_loop(env: BridgeBidding, state: State, fwd1: _Forward, fwd2: _Forward, tables, stats, bid_set, cum_return,
               rewards_sum, sync_every, record_actions=None, record_logits=None, by_turn=False, record_calls=None):
    """Runs ``brl_eval_step`` until every board is finished; ``state.packed`` is advanced in place.  (``sync_every`` — how often
    the loop condition used to be read back — is kept in the signatures and ignored: see ``_DoneWatch``.)"""
    n, dev = state.num_envs, env.device
    obs = state.observation
    term = torch.empty(n, dtype=torch.bool, device=dev)
    action = torch.empty(n, dtype=torch.int32, device=dev)
    pa = tables[0]._ptrs() if tables else None
    pb = tables[1]._ptrs() if tables else None
    ps = stats.ptrs() if stats is not None else None
    packed = state.packed
    count = 0
    # Two different networks: the reference evaluates BOTH on every board and selects by team (src/evaluation.py:146-151);
    # here the teams take turns — iteration i runs ONE forward (team i & 1's network) and only the boards whose player to
    # act is on that team make their call (brl_eval_step_team).  Calls are deterministic arg-maxes and a board's teams
    # alternate call by call, so every board plays exactly the same auction; it waits at most one iteration at its start
    # and one at the table switch.  Half the GEMM work per board.  (Recording runs — the tests' oracle replays — keep the
    # reference's lock-step loop.)
    alternate = (fwd2 is not fwd1) and record_actions is None and record_logits is None
    compact = record_actions is None and record_logits is None and os.environ.get("BRL_EVAL_COMPACT", "1") != "0"
    watch = _DoneWatch.take(env, n, compact)
    rows = [_ActiveRows(n), _ActiveRows(n)]   # (one logits buffer per team: their forwards alternate)
    obs_f32 = None
    while True:
        polled = watch.poll(count)
        if polled is not None and polled[0] >= n:
            watch.release()   # (an exception above leaves it marked busy: the next loop simply builds its own)
            break
        nobs = torch.empty((n, OBS_SIZE), dtype=torch.bool, device=dev)
        if by_turn:
            # the networks take turns by CALL, not by team: call i of every board is made by fwd1 (i even: the player who opened
            # and their partner) or fwd2 (i odd) — the macro-step of src/utils.py:133-202 one call at a time, every board in step;
            # ONE forward per call, on the boards still playing
            k = count & 1
            rows[k].update(polled)
            lg = rows[k].forward(fwd2 if k else fwd1, obs, env)
            check(_capi.lib().brl_eval_step(
                env._h, ptr(packed), ptr(packed), n, lg.data_ptr(), lg.stride(0), lg.data_ptr(), lg.stride(0),
                C.byref(pa) if pa is not None else None, C.byref(pb) if pb is not None else None,
                C.byref(ps) if ps is not None else None, int(bid_set),
                ptr(cum_return), ptr(rewards_sum), ptr(action), ptr(nobs), None, None, ptr(term), None, _stream()))
            obs = nobs
            if record_actions is not None:
                record_actions.append(action.clone())
            watch.post(count, term)
            count += 1
            continue
        if alternate:
            team = count & 1
            rows[team].update(polled)
            lg = rows[team].forward(fwd2 if team else fwd1, obs, env, obs_f32)
            # the next iteration's forward still takes every board: this launch writes its float32 input as well
            nxt, nfwd = rows[team ^ 1], (fwd1 if team else fwd2)
            nsnap = getattr(nfwd, "snap", None)
            obs_f32 = torch.empty((n, OBS_SIZE), dtype=torch.float32, device=dev) \
                if (_STEP_CASTS and nxt.idx is None and not getattr(nfwd, "constant", False)
                    and not (nsnap is not None and nsnap.planes_for(n))) else None
            check(_capi.lib().brl_eval_step_team(
                env._h, ptr(packed), ptr(packed), n, lg.data_ptr(), lg.stride(0), team,
                C.byref(pa) if pa is not None else None, C.byref(pb) if pb is not None else None,
                C.byref(ps) if ps is not None else None, int(bid_set),
                ptr(cum_return), ptr(rewards_sum), ptr(action), ptr(nobs), None, None, ptr(term), None, ptr(obs_f32), _stream()))
            if record_calls is not None:   # (the calls of THIS loop — -1: the board waited for its team's iteration — for oracle replays)
                record_calls.append(action.clone())
            obs = nobs
            watch.post(count, term)
            count += 1
            continue
        if compact:
            rows[0].update(polled)
            rows[1].update(polled)
            l1 = rows[0].forward(fwd1, obs, env)
            l2 = rows[1].forward(fwd2, obs, env) if fwd2 is not fwd1 else l1
        else:
            x = obs.to(torch.float32)
            l1 = fwd1(obs, x)
            l2 = fwd2(obs, x) if fwd2 is not fwd1 else l1   # G10: the reference evaluates both networks and selects
        if record_logits is not None:
            team1 = (State(env, packed).current_player < 2)[:, None]
            record_logits.append(torch.where(team1, l1[:, :NUM_ACTIONS], l2[:, :NUM_ACTIONS]).clone())
        check(_capi.lib().brl_eval_step(
            env._h, ptr(packed), ptr(packed), n, l1.data_ptr(), l1.stride(0), l2.data_ptr(), l2.stride(0),
            C.byref(pa) if pa is not None else None, C.byref(pb) if pb is not None else None,
            C.byref(ps) if ps is not None else None, int(bid_set),
            ptr(cum_return), ptr(rewards_sum), ptr(action), ptr(nobs), None, None, ptr(term), None, _stream()))
        obs = nobs
        if record_actions is not None:
            record_actions.append(action.clone())
        watch.post(count, term)
        count += 1
    return State(env, packed, {"observation": obs, "terminated": term}), count


def make_simple_duplicate_evaluate(eval_env: BridgeBidding, team1_activation, team1_model_type, team2_activation,
                                   team2_model_type, num_eval_envs, sync_every: int = 16, record_actions=None, shard=None,
                                   record_calls=None):
    """src/evaluation.py:69-204.  ``record_actions``: optional list that receives each iteration's action tensor
    (tests replay them through the oracle; the reference's lock-step loop runs).  ``record_calls``: the same for the DEFAULT
    loop (teams alternate, forwards on the boards still playing): one [n] tensor per iteration, -1 where a board waited.  ``shard`` (see _Shard): the boards are split over the ranks, (mean IMP,
    standard error, win rate) are those of all ``num_eval_envs`` boards on every rank; the Table_info are this rank's."""
    team1_forward_pass = make_forward_pass(team1_activation, team1_model_type)
    team2_forward_pass = make_forward_pass(team2_activation, team2_model_type)

    def duplicate_evaluate(team1_params, team2_params, rng_key):
        sh = _Shard(num_eval_envs, shard)
        with torch.no_grad():
            state = sh.init(eval_env, rng_key)                       # src/evaluation.py:93-95
            table_a_info = Table_info.from_state(state)              # :96-103
            table_b_info = Table_info.from_state(state)              # :104-111
            cum_return = torch.zeros(sh.n, dtype=torch.float32, device=eval_env.device)
            fwd1 = _Forward(team1_forward_pass, team1_params)
            fwd2 = fwd1 if team2_params is team1_params else _Forward(team2_forward_pass, team2_params)
            _eval_loop(eval_env, state, fwd1, fwd2, (table_a_info, table_b_info), None, 0, cum_return, None,
                       sync_every, record_actions, record_calls=record_calls)   # :120-197; cum_return += rewards[:, 0] (G8)
            n = float(num_eval_envs)
            if sh.active:   # sums over every rank's boards (IMPs are integers: exact in float64)
                x = cum_return.to(torch.float64)
                t = sh.allsum(torch.stack([x.sum(), (x * x).sum(), (x > 0).sum().to(torch.float64)]))
                mean = t[0] / n
                var = ((t[1] - n * mean * mean) / (n - 1.0)).clamp_min(0.0)
                log_info = (mean.to(torch.float32), (var.sqrt() / (n ** 0.5)).to(torch.float32), (t[2] / n).to(torch.float32))
                return log_info, table_a_info, table_b_info
            std_error = cum_return.std(unbiased=True) / (n ** 0.5)   # :199
            win_rate = (cum_return > 0).sum() / n                    # :200
            log_info = (cum_return.mean(), std_error, win_rate)
        return log_info, table_a_info, table_b_info

    return duplicate_evaluate


def make_evaluate(eval_env: BridgeBidding, team1_activation, team1_model_type, team2_activation, team2_model_type,
                  team2_params, num_eval_envs, game_mode="competitive", duplicate=False, sync_every: int = 16,
                  record_actions=None, record_logits=None, shard=None):
    """``make_evaluate`` (src/evaluation.py:207-1032).  ``team2_params`` replaces the reference's pickle path.
    Returns ``duplicate_evaluate(actor_params, rng_key) -> (log_info, table_a_info, table_b_info)`` (23-entry
    log_info, :985-1031) or, with ``duplicate=False``, ``evaluate(actor_params, rng_key) -> (state, log_info)``
    (19 entries, :583-605).  Feed ``log_info`` of the duplicate variant to ``make_evaluate_log``.

    game_mode "free-run": the opponents (players {2,3}) always pass (:243-247) — their "logits" are a constant row
    whose arg-max is Pass and whose softmax mass on illegal actions is what the reference's one-hot probs give."""
    actor_forward_pass = make_forward_pass(team1_activation, team1_model_type)
    opp_forward_pass = make_forward_pass(team2_activation, team2_model_type)
    if game_mode not in ("competitive", "free-run"):
        raise ValueError(game_mode)

    class _PassOnly:
        """free-run opponent: probs = one-hot(Pass) (src/evaluation.py:243-247); logits with softmax == that one-hot."""
        constant = True

        def __init__(self, n, device):
            self.lg = torch.full((n, NUM_ACTIONS), -1e30, dtype=torch.float32, device=device)
            self.lg[:, 0] = 0.0

        def __call__(self, obs_bool, obs_f32):
            return self.lg

    def run(actor_params, rng_key, dup):
        sh = _Shard(num_eval_envs, shard)   # (shard: the statistics below are those of all boards, on every rank; the
        n, dev = sh.n, eval_env.device      #  returned Table_info / State are this rank's boards)
        with torch.no_grad():
            state = sh.init(eval_env, rng_key)
            tables = (Table_info.from_state(state), Table_info.from_state(state)) if dup else None
            cum_return = torch.zeros(n, dtype=torch.float32, device=dev)
            rewards_sum = None if dup else torch.zeros((n, 4), dtype=torch.float32, device=dev)
            stats = EvalStats(n, dev)
            fwd1 = _Forward(actor_forward_pass, actor_params)
            fwd2 = _PassOnly(n, dev) if game_mode == "free-run" else _Forward(opp_forward_pass, team2_params)
            # the single-table evaluator marks a bid as made (.set(1), :349-358), the duplicate one counts it (:704-713)
            state, _ = _eval_loop(eval_env, state, fwd1, fwd2, tables, stats, 0 if dup else 1, cum_return, rewards_sum,
                                  sync_every, record_actions, record_logits)
            if not dup:  # make_terminated_log on the final state (:463-487): one "table" built from it
                f = State(eval_env, state.packed)
                tables = (Table_info(f.terminated, rewards_sum, f._last_bid, f._last_bidder, f._call_x, f._call_xx),)
            counts = torch.empty(_capi.EVAL_COUNTS, dtype=torch.int64, device=dev)
            pa = tables[0]._ptrs()
            pb = tables[1]._ptrs() if dup else None
            check(_capi.lib().brl_eval_reduce(eval_env._h, n, C.byref(pa), C.byref(pb) if pb is not None else None,
                                              ptr(stats.bid_count), ptr(state.packed), ptr(counts), _stream()))
            c = sh.allsum(counts).to(torch.float64)   # (exact integer counts: the histograms of all ranks' boards)
            fn = float(sh.n_global)
            steps = stats.step_count.to(torch.float32)
            if sh.active:   # per-board ratios are averaged over all boards: all-reduce their sums
                ratio = sh.allsum(torch.cat([(stats.illegal_prob_sum / steps).to(torch.float64).sum(dim=0),
                                             (stats.pass_count.to(torch.float32) / steps).to(torch.float64).sum(dim=0)]))
                illegal, passes = (ratio[:2] / fn).to(torch.float32), (ratio[2:] / fn).to(torch.float32)
                x = cum_return.to(torch.float64)
                cr = sh.allsum(torch.stack([x.sum(), (x * x).sum()]))
                cr_mean = cr[0] / fn
                cr_se = (((cr[1] - fn * cr_mean * cr_mean) / (fn - 1.0)).clamp_min(0.0).sqrt() / (fn ** 0.5)).to(torch.float32)
                cr_mean = cr_mean.to(torch.float32)
            else:
                illegal = (stats.illegal_prob_sum / steps).mean(dim=0)          # :857-862 (x / y per board, then mean)
                passes = (stats.pass_count.to(torch.float32) / steps).mean(dim=0)  # :1029-1030
                cr_mean = cum_return.mean()
                cr_se = cum_return.std(unbiased=True) / (fn ** 0.5) if dup else None   # :984
            ntab = 2.0 if dup else 1.0

            def both(i):  # (table A + table B) / 2 of a per-table ratio count / n  (:985-1028)
                return ((c[i] + (c[80 + i] if dup else 0.0)) / fn / ntab).to(torch.float32)

            def both_vec(i):
                return ((c[i:i + 35] + (c[80 + i:80 + i + 35] if dup else 0.0)) / fn / ntab).to(torch.float32)

            bid_div = 2.0 if dup else 1.0                                    # :992-993 actor_bid.mean(axis=0) / 2
            actor_bid = (c[160:195] / fn / bid_div).to(torch.float32)
            opp_bid = (c[195:230] / fn / bid_div).to(torch.float32)
            step_count_mean = (c[230] / fn).to(torch.float32)                # state._step_count.mean()
            common = (illegal[0], illegal[1], step_count_mean, actor_bid, opp_bid, both_vec(10), both_vec(45),
                      (c[10:45].sum() + (c[90:125].sum() if dup else 0.0)).div(fn * ntab).to(torch.float32),  # declarer ratios
                      (c[45:80].sum() + (c[125:160].sum() if dup else 0.0)).div(fn * ntab).to(torch.float32),
                      both(1), both(2), both(3), both(4), both(5), both(6), both(7), both(8), both(0))
            if dup:
                score = ((c[9] / fn + c[89] / fn) / 2).to(torch.float32)      # :988
                log_info = (cr_mean, cr_se, score) + common + (passes[0], passes[1])
                return log_info, tables[0], tables[1]
            final = State(eval_env, state.packed).replace(rewards=rewards_sum)  # :582
            return final, (cr_mean,) + common

    def duplicate_evaluate(actor_params, rng_key):
        return run(actor_params, rng_key, True)

    def evaluate(actor_params, rng_key):
        return run(actor_params, rng_key, False)

    return duplicate_evaluate if duplicate else evaluate


def make_evaluate_log(log_info):
    """``make_evaluate_log`` (src/evaluation.py:1035-1115): the 23-entry ``log_info`` of the duplicate evaluator as
    the flat dict ppo.py merges into its wandb log (python floats; the reference keeps jnp scalars)."""
    (imp_mean, std_error, score_mean, actor_illegal, opp_illegal, step_count_mean, actor_bid, opp_bid, actor_contract,
     opp_contract, actor_declarer, opp_declarer, actor_x, actor_xx, opp_x, opp_xx, actor_make, opp_make, actor_down,
     opp_down, pass_out, actor_pass, opp_pass) = log_info
    f = float
    log = {
        "eval/IMP_reward": f(imp_mean), "eval/IMP_SE": f(std_error), "eval/score_reward": f(score_mean),
        "eval/actor_illegal_action_probs": f(actor_illegal), "eval/opp_illegal_action_probs": f(opp_illegal),
        "eval/step count": f(step_count_mean), "eval/actor_declarer_ratio": f(actor_declarer),
        "eval/opp_declarer_ratio": f(opp_declarer), "eval/actor_doubled_ratio": f(actor_x),
        "eval/actor_redoubled_ratio": f(actor_xx), "eval/opp_doubled_ratio": f(opp_x),
        "eval/opp_redoubled_ratio": f(opp_xx), "eval/actor_make_contract_ratio": f(actor_make),
        "eval/opp_make_contract_ratio": f(opp_make), "eval/actor_down_contract_ratio": f(actor_down),
        "eval/opp_down_contract_ratio": f(opp_down), "eval/pass_out_ratio": f(pass_out),
        "eval/actor_pass_ratio": f(actor_pass), "eval/opp_pass_ratio": f(opp_pass),
    }
    ab, ob, ac, oc = (x.detach().cpu().tolist() for x in (actor_bid, opp_bid, actor_contract, opp_contract))
    groups = ({}, {}, {}, {})
    index = 0
    for number in range(1, 8):                 # :1092-1107: bid index = (level - 1) * 5 + strain, strains C D H S NT
        for suit in ("C", "D", "H", "S", "NT"):
            groups[0][f"eval/actor_bid_probs/{number}{suit}"] = ab[index]
            groups[1][f"eval/actor_contract_probs/{number}{suit}"] = ac[index]
            groups[2][f"eval/opp_bid_probs/{number}{suit}"] = ob[index]
            groups[3][f"eval/opp_contract_probs/{number}{suit}"] = oc[index]
            index += 1
    for g in groups:
        log.update(g)
    return log


def make_simple_evaluate(eval_env: BridgeBidding, team1_activation, team1_model_type, team2_activation,
                         team2_model_type, team2_params, num_eval_envs, sync_every: int = 8, shard=None,
                         record_actions=None, record_calls=None):
    """src/evaluation.py:11-66: actor (greedy) vs a fixed opponent (greedy) on single tables, no
    auto-reset; returns the mean total reward of the acting player.  ``team2_params`` replaces the
    reference's pickle path (model files are torch modules here).  ``record_actions``: optional list that receives,
    per loop iteration, the four calls of the macro-step as [4, n] (tests replay them through the oracle; the mirror of the
    reference's macro-step loop runs).  ``record_calls``: optional list that receives one [n] tensor per CALL of the by-turn loop
    (the default loop, forwarding every board: the oracle replay of that loop)."""
    actor_forward_pass = make_forward_pass(team1_activation, team1_model_type)
    opp_forward_pass = make_forward_pass(team2_activation, team2_model_type)

    def by_turn(actor_params, rng):
        """the same evaluation one CALL per iteration (``_eval_loop(by_turn=True)``): the forwards run on the boards still playing
        only (bias + activation in the GEMM epilogue, the last boards through brl_mlp_forward_rows), the loop stops two calls
        after the last board, not up to five.  A board's return is what its opening player collected: rewards are zero except on
        the call that ends the auction, so the sum over macro-steps of src/evaluation.py:60 is that one entry."""
        sh = _Shard(num_eval_envs, shard)
        with torch.no_grad():
            state = sh.init(eval_env, rng)
            opener = state.current_player.to(torch.int64)
            rsum = torch.zeros((sh.n, 4), dtype=torch.float32, device=eval_env.device)
            _eval_loop(eval_env, state, _Forward(actor_forward_pass, actor_params), _Forward(opp_forward_pass, team2_params),
                       None, None, 0, None, rsum, sync_every, record_actions=record_calls, by_turn=True)
            R = rsum.gather(1, opener[:, None])[:, 0]
        if sh.active:   # (scores are integers: the float64 sum over the ranks is exact)
            return (sh.allsum(R.to(torch.float64).sum().reshape(1))[0] / float(num_eval_envs)).to(torch.float32)
        return R.mean()

    def simple_evaluate(actor_params, rng):
        if record_actions is None and os.environ.get("BRL_SIMPLE_EVAL_BY_TURN", "1") != "0":
            return by_turn(actor_params, rng)
        step_fn = single_play_step_two_policy_commpetitive_deterministic(
            step_fn=eval_env.step, actor_forward_pass=actor_forward_pass, actor_params=actor_params,
            opp_forward_pass=opp_forward_pass, opp_params=team2_params)
        subs = [] if record_actions is not None else None
        step_fn.record_actions = subs
        sh = _Shard(num_eval_envs, shard)
        with torch.no_grad():
            state = sh.init(eval_env, rng)
            R = torch.zeros(sh.n, dtype=torch.float32, device=eval_env.device)
            it = 0
            watch = _DoneWatch.take(eval_env, sh.n, False)
            while not watch.finished(it):
                actor = state.current_player.to(torch.int64)
                logits, _ = actor_forward_pass.apply(actor_params, state.observation.to(torch.float32))
                action = masked_mode(logits, state.legal_action_mask)
                state = step_fn(state, action, it * 4)
                if subs is not None:
                    record_actions.append(torch.stack([action.to(torch.int32)] + subs))
                    subs.clear()
                R += state.rewards.gather(1, actor[:, None])[:, 0]   # src/evaluation.py:60
                watch.post(it, state.terminated)
                it += 1
            watch.release()
        if sh.active:   # (scores are integers: the float64 sum over the ranks is exact)
            return (sh.allsum(R.to(torch.float64).sum().reshape(1))[0] / float(num_eval_envs)).to(torch.float32)
        return R.mean()

    return simple_evaluate
