"""``FusedMinibatch`` — the PPO minibatch step of ``src/update.py:74-242`` as a hipGraph-captured PROGRAM of HIP launches, library
GEMMs and (under a process group) RCCL collectives.  ``brl_amd.update.make_update_step`` drives it; DESIGN.md §4.3 / §7 describe
the step and its multi-rank forms."""
from __future__ import annotations

import torch
import torch.distributed as dist

from ._capture import quiet_gc
from .roll_out import Transition


_CAPTURE_GROUP = {}


def _capture_group():
    """The process group whose collectives are issued ONLY while a stream captures (one per process, created on first use — a
    collective call: every rank builds its first fused step at the same point of update_step).  Why a group of its own:
    ProcessGroupNCCL's watchdog thread polls the end events of EAGER collectives it has not yet seen complete (every 100 ms), and HIP
    refuses the query of an event whose stream is capturing ("operation not permitted on an event last recorded in a capturing
    stream": the watchdog throws, the process aborts — scripts/rccl_eager_then_capture_probe.py case F, profiles/r05/).  Round 5
    fenced that with synchronize + 0.3 s of sleep: a probability.  A communicator that never carries an eager collective has
    nothing in its watchdog's list, and the default group's stream — barriers, broadcasts, the evaluators' sums, the optimizer-state
    gather — never captures.  The communicator is connected eagerly (no collective, no work item); RCCL's per-size set-up happens
    inside a throw-away capture (FusedStep.__init__)."""
    key = dist.get_world_size(), dist.get_rank()
    if key not in _CAPTURE_GROUP:
        g = dist.new_group(backend="nccl")
        try:
            dev = torch.device("cuda", torch.cuda.current_device())
            g._get_backend(dev).eager_connect_single_device(dev)
        except Exception:   # (an older torch: the communicator is then created by the first captured collective)
            pass
        _CAPTURE_GROUP[key] = g
    return _CAPTURE_GROUP[key]


class _DistCollectives:
    """The three collectives of the multi-rank step on torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; gloo in the
    CPU / one-GPU rehearsals).  ``capturable``: RCCL's collectives record into a hipGraph (probed: scripts/rccl_capture_probe.py),
    gloo's copy through the host and cannot.  While a stream captures the collectives go through ``_capture_group()``; issued
    eagerly (``gather_optimizer_state``, the per-kernel-group strategy) through the default group.  ``bench.py`` passes an object of
    the same shape whose collectives are no-ops with a collective's stream edges (its one-GPU rehearsal of the step)."""

    def __init__(self):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.capturable = dist.get_backend() == "nccl"
        self.capture_group = _capture_group() if self.capturable else None

    def _group(self):
        return self.capture_group if (self.capturable and torch.cuda.is_current_stream_capturing()) else None

    def all_reduce(self, t, async_op):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._group(), async_op=async_op)

    def reduce_scatter(self, out, inp, async_op):      # out = this rank's slice OF inp (in place)
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self._group(), async_op=async_op)

    def all_gather(self, out, inp, async_op):          # inp = this rank's slice OF out (in place)
        return dist.all_gather_into_tensor(out, inp, group=self._group(), async_op=async_op)

    def agree(self, ok: bool, device) -> bool:
        """True iff ``ok`` on EVERY rank (one eager 4-byte MIN all-reduce on the default group)"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if self.capturable else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))


class FusedStep:
    """What every fused PPO minibatch step shares: the FLAT parameter / gradient / moment buffers the module's parameters and the
    optimizer's state become views of, the device-resident minibatch gather, the step as a PROGRAM — a list of kernel groups
    ("k"), collectives ("c") and waits ("w") built once by the subclass — and its two execution strategies: captured whole,
    collectives included, ``update_graph_steps`` times per hipGraph (single rank, and RCCL, whose collectives record into a graph);
    or, for a backend that cannot be captured (gloo), one graph per kernel group with the collectives issued eagerly between the
    replays.  Subclasses: ``FusedMinibatch`` (the DeepMind MLPs), ``FusedFair`` (the FAIR residual net)."""

    def __init__(self, config, params, opt, mbs: int, device, world: int = 1, log_capacity: int = 0, collective=None):
        from . import _capi
        self.cfg, self.params, self.opt, self.mbs, self.dev = config, params, opt, int(mbs), device
        self.lib, self.capi = _capi.lib(), _capi
        self.world = int(world)
        # force_collectives / BRL_FORCE_DIST=1 under a process group: the multi-rank program at world 1 (tests, bench, rehearsals)
        from .dist import distributed
        multi = self.world > 1 or bool(config.get("force_collectives", False)) or (collective is None and distributed())
        self.multi = multi
        self.allreduce_mode = str(config.get("grad_allreduce", "flat")) if multi else "none"
        if self.allreduce_mode not in ("none", "flat", "sharded"):
            raise ValueError("config['grad_allreduce'] must be 'sharded' or 'flat'")
        self.coll = collective if collective is not None else (_DistCollectives() if multi else None)
        self.rank = int(getattr(self.coll, "rank", 0)) if multi else 0
        # Under a process group the construction is PHASED, with an agreement (MIN over ranks: _agree) after every phase that can
        # fail on one rank alone — a rank that raised would otherwise leave for update_step's own 4-byte agreement while its peers
        # sit in this constructor's 14.7 MB collectives: mismatched collectives, a hang until the RCCL timeout.  Every rank runs
        # the same three agreements or raises at the same one.
        #   phase 1   allocations, the flat buffers, the program, first launches of every kernel group — no collective   -> agree
        #   phase 2   RCCL in the graph: a THROW-AWAY capture of the whole program (RCCL's lazy set-up happens while it records; no
        #             eager collective ever touches the capture group)                                                  -> agree
        #             then its one replay (every rank holds a valid graph: the collectives pair up);
        #             other backends: the program once, eagerly
        #   phase 3   the step's graphs                                                                                 -> agree
        err = None
        self._saved = None
        try:
            self._setup(log_capacity)
        except Exception as e:
            err = e
            self._restore()
        self._agree(err, "allocation / first launches")
        self._capture_all()

    def _restore(self):
        """the warm-up and the captures run real steps on the dummy batch: parameters, moments and counters as they were"""
        if self._saved is not None:
            with torch.no_grad():
                for t, q in zip((self.P, self.M, self.V, self.step, self.mb_index), self._saved):
                    t.copy_(q)
            self._saved = None

    def _setup(self, log_capacity):
        config, params, opt, device, multi = self.cfg, self.params, self.opt, self.dev, self.multi
        _capi = self.capi
        if config.get("tuned_gemm", True):   # committed TunableOp solutions for the step's GEMM shapes (brl_amd/tuned): lookups only
            from . import tuned
            tuned.enable()
        f = self._f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)  # noqa: E731
        B = self.mbs
        self.act = 0 if params.act is torch.relu else 1
        self.ill_coef = float(config.get("illegal_action_l2norm_coef", 0.0) or 0.0)
        # ---- the subclass orders the parameters in the flat buffers and (multi-rank) cuts them into collective buckets
        plist, sizes, offs, lens, n = self._layout()
        assert len(plist) == len(list(params.parameters()))
        self.n = n
        if multi:
            g = _capi.ShardGeom()
            g.nbuckets, g.world = len(offs), self.world
            g.nsub = max(1, 1024 // (self.world * len(offs)))
            for b, (o, ln) in enumerate(zip(offs, lens)):
                g.off[b], g.len[b] = o, ln
            self.geom, self.bucket_off, self.bucket_len = g, offs, lens
            self.norm_partials = f(self.world * g.nbuckets * g.nsub)
        self.P, self.G, self.M, self.V = f(n), f(n), f(n), f(n)
        self.step = torch.zeros((), dtype=torch.float32, device=device)
        self.plist = plist
        off = 0
        views = {}
        with torch.no_grad():
            for q, k in zip(plist, sizes):
                sl = slice(off, off + k)
                self.P[sl].copy_(q.detach().reshape(-1))
                st = opt.state.get(q, {})
                if "exp_avg" in st:  # built after eager steps / from a loaded optimizer: continue from that state
                    self.M[sl].copy_(st["exp_avg"].reshape(-1))
                    self.V[sl].copy_(st["exp_avg_sq"].reshape(-1))
                q.data = self.P[sl].view(q.shape)
                q.grad = self.G[sl].view(q.shape)
                st_step = st.get("step")
                opt.state[q] = {"step": st_step.to(device=device, dtype=torch.float32).reshape(()) if torch.is_tensor(st_step)
                                else torch.zeros((), dtype=torch.float32, device=device),
                                "exp_avg": self.M[sl].view(q.shape), "exp_avg_sq": self.V[sl].view(q.shape)}
                views[q] = sl
                off += k
        self.views = views
        self.x0 = f(B, 480)
        self.mask = torch.zeros((B, 38), dtype=torch.uint8, device=device)
        self.mask[:, 0] = 1  # a valid dummy batch for the warm-up iterations
        self.action = torch.zeros(B, dtype=torch.int32, device=device)
        self.old_v, self.old_lp, self.adv, self.tgt = f(B), f(B), f(B), f(B)
        self.lgroups = (B + 3) // 4                    # 4-sample groups of the loss launch (statistics / Gram partials)
        self.partials = f(self.lgroups, 8)
        self.out = f(8)
        # The logged statistics (src/update.py:136-167) are NOT formed step by step: every step leaves its sums — 8 floats and
        # the 38 x 38 Gram matrix of the illegal-action probabilities, reduced by spare workgroups of the head-backward
        # launch — in row mb_index of these buffers, and ONE launch at the end of the update turns all rows into log rows
        # (rows = minibatch steps of one update_step call; update_step rebuilds this object when an update needs more)
        self._log_cap = max(int(config.get("update_log_capacity", 4096)), int(log_capacity))
        self.log = f(self._log_cap, 8)
        self.stat_sums = f(self._log_cap, 8)
        self.gram_sums = f(self._log_cap, 38 * 38)
        self.norm = f(1)
        self.mb_index = torch.zeros(1, dtype=torch.int32, device=device)  # minibatch step within the current update
        self._alloc()                                  # the subclass's activations, scratch and segment tables
        self.perm = None  # static int64 [epochs * T*N]: every epoch's permutation, filled by begin_update
        # the step's own gather reads ITS arguments from device memory (brl_mb_gather_bind, once per update): the captured
        # step needs no eager launch in front of it.  Until the first update: a dummy trajectory of mbs valid rows.
        self.gargs = torch.zeros(256, dtype=torch.uint8, device=device)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)  # noqa: E731
        dmask = z((B, 38), torch.bool)
        dmask[:, 0] = True
        self._dummy = (Transition(z((B,), torch.bool), z((B,), torch.int32), f(B), f(B), f(B), z((B, 480), torch.bool), dmask),
                       f(B), f(B), (torch.arange(8 * B, device=device) % B).to(torch.int64))
        self._bind_gather(*self._dummy)
        d0 = opt.defaults
        self.lr, (self.b1, self.b2), self.eps = float(d0["lr"]), d0["betas"], float(d0["eps"])
        self.lr_dev = torch.full((1,), float(opt.param_groups[0]["lr"]), dtype=torch.float32, device=device)
        self.max_norm = float(config["max_grad_norm"]) if config.get("global_gradient_clipping", True) else 0.0
        self.program = self._build_program()
        # collectives inside the graph where the backend records into one (RCCL), else eager between per-group graphs
        want = config.get("collective_in_graph")
        self.in_graph = (not multi) or (bool(getattr(self.coll, "capturable", False)) if want is None else bool(want))
        self._works = {}
        self.moments_partial = False    # "sharded": True once a step has run and gather_optimizer_state has not
        self.graph = self.graph_multi = self.segs = None
        # warm-up and capture run real steps on the dummy batch: parameters, moments and counters are put back afterwards
        self._saved = [t.clone() for t in (self.P, self.M, self.V, self.step, self.mb_index)]
        self._side = torch.cuda.Stream()
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side), torch.no_grad():
            for _ in range(2):      # kernel groups only (allocator, library heuristics)
                for item in self.program:
                    if item[0] == "k":
                        item[1]()
        torch.cuda.current_stream().wait_stream(self._side)
        torch.cuda.synchronize()

    def _capture_all(self):
        from ._capture import graph_kwargs
        config, multi, side = self.cfg, self.multi, self._side
        gkw = graph_kwargs()
        try:
            err = None
            scratch = None
            try:
                if multi and self.in_graph:
                    with quiet_gc():
                        scratch = self._capture_steps(1, gkw)
                else:
                    with torch.cuda.stream(side), torch.no_grad():
                        self._run_program(True)     # (single rank: a third warm-up; gloo: the collectives pair up eagerly)
                        self._drain()
                    torch.cuda.current_stream().wait_stream(side)
            except Exception as e:
                err = e
            self._agree(err, "first pass of the program's collectives")
            if scratch is not None:
                scratch.replay()
                torch.cuda.synchronize()
                del scratch
            try:
                with quiet_gc():   # (_capture.py: no collector run while a stream captures)
                    if self.in_graph:
                        self.graph = self._capture_steps(1, gkw)
                        # ... and the same step K times in ONE graph: a replay boundary costs ~5 us (graph launch behind the last
                        # kernel), the step ~0.23 ms; mb_index lives in device memory, so the K copies walk K minibatches
                        self.graph_steps = int(config.get("update_graph_steps", 8))
                        if self.graph_steps > 1:
                            self.graph_multi = self._capture_steps(self.graph_steps, gkw)
                    else:
                        pool = torch.cuda.graph_pool_handle()
                        self.segs = []          # the program with every run of kernel groups replaced by its graph
                        run = []

                        def close():
                            if run:
                                fns = list(run)
                                g = torch.cuda.CUDAGraph()
                                with torch.cuda.graph(g, pool=pool, **gkw), torch.no_grad():
                                    for fn in fns:
                                        fn()
                                self.segs.append(("g", g))
                                run.clear()
                        for item in self.program:
                            if item[0] == "k":
                                run.append(item[1])
                            elif item[0] == "c":
                                close()
                                self.segs.append(item)
                        close()
            except Exception as e:
                err = e
            self._agree(err, "capture of the step")
        finally:
            self._restore()

    def _agree(self, err, what):
        """raises on EVERY rank if the phase failed on any (single rank: re-raises its own error)"""
        agree = getattr(self.coll, "agree", None) if self.multi else None
        ok = err is None
        if agree is not None:
            ok = agree(ok, self.dev)
        if err is not None:
            raise err
        if not ok:
            raise RuntimeError(f"FusedStep: {what} failed on another rank: no rank builds the fused step")

    def _capture_steps(self, k, gkw):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **gkw), torch.no_grad():
            for _ in range(k):
                self._run_program(True)
            self._drain()           # every forked stream joins before the capture ends
        return g

    def _run_program(self, async_op):
        for item in self.program:
            if item[0] == "k":
                item[1]()
            elif item[0] == "c":
                w = item[2](async_op)
                if w is not None:
                    self._works[item[1]] = w
            else:
                w = self._works.pop(item[1], None)
                if w is not None:
                    w.wait()

    def _drain(self):
        for key in list(self._works):
            self._works.pop(key).wait()

    def _bind_gather(self, fl: Transition, adv, tgt, perm, first=True):
        """binds the step's gather to a trajectory / permutation; ``first``: also gathers minibatch *mb_index now (every later
        minibatch is gathered by the Adam launch of the step before it)"""
        import ctypes as C
        tp = self.capi.TransitionPtrs()
        for name in self.capi.TransitionPtrs._names:
            t = getattr(fl, name)
            setattr(tp, name, (t.view(torch.uint8) if t.dtype == torch.bool else t).data_ptr())
        self.capi.check(self.lib.brl_mb_gather_bind(self._di(), C.byref(tp), adv.data_ptr(), tgt.data_ptr(), perm.data_ptr(),
                                                    self.mb_index.data_ptr(), self.mbs, self.x0.data_ptr(), self.mask.data_ptr(),
                                                    self.action.data_ptr(), self.old_v.data_ptr(), self.old_lp.data_ptr(),
                                                    self.adv.data_ptr(), self.tgt.data_ptr(), perm.numel() // self.mbs,
                                                    self.gargs.data_ptr(), torch.cuda.current_stream().cuda_stream))
        if first:
            self.capi.check(self.lib.brl_mb_gather_dev(self._di(), self.gargs.data_ptr(), self.mbs, torch.cuda.current_stream().cuda_stream))

    def _di(self):
        return self.dev.index if self.dev.index is not None else torch.cuda.current_device()

    # ---- clip + Adam on rank slices (multi-rank forms) -----------------------------------------------------------------
    def _shard_norm(self, lo, hi):
        import ctypes as C
        self.capi.check(self.lib.brl_adam_shard_norm(self._di(), self.G.data_ptr(), C.byref(self.geom), lo, hi, 1.0 / self.world,
                                                     self.norm_partials.data_ptr(), self.step.data_ptr(), self.mb_index.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream))

    def _shard_apply(self, lo, hi):
        import ctypes as C
        self.capi.check(self.lib.brl_adam_shard_apply(
            self._di(), self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), C.byref(self.geom), lo, hi,
            self.norm_partials.data_ptr(), self.step.data_ptr(), self.lr, self.lr_dev.data_ptr(), float(self.b1), float(self.b2),
            self.eps, self.max_norm, 1.0 / self.world, self.norm.data_ptr(), self.gargs.data_ptr(), self.mbs,
            torch.cuda.current_stream().cuda_stream))

    # ---- one update_step call -----------------------------------------------------------------------------------
    def begin_update(self, flat: Transition, adv_f, tgt_f, perms):
        """flat: the [T*N, ...] views of the trajectory; adv_f / tgt_f: [T*N]; perms: one permutation of T*N per epoch.
        Binds the step's gather to them (device-resident arguments) and resets the minibatch counter."""
        steps = sum(p.numel() for p in perms) // self.mbs
        if steps > self._log_cap:   # (checked before anything is touched; update_step never gets here: it rebuilds first)
            raise RuntimeError("FusedStep: more minibatch steps per update than its log holds (log_capacity)")
        self._keep = (Transition(*[x.contiguous() for x in flat]), adv_f.contiguous(), tgt_f.contiguous())
        fl, adv_c, tgt_c = self._keep
        self._steps = steps
        with torch.no_grad():
            self._readopt()
            allp = torch.cat(perms)
            if self.perm is None or self.perm.numel() != allp.numel():
                self.perm = torch.empty_like(allp)
            self.perm.copy_(allp)
            self.mb_index.zero_()
            self.step.copy_(self.opt.state[self.plist[0]]["step"])  # the optimizer may have been stepped eagerly / loaded
            self.lr_dev.fill_(float(self.opt.param_groups[0]["lr"]))  # constant within an update (ppo.py:186-192)
            self._bind_gather(fl, adv_c, tgt_c, self.perm)

    def _readopt(self):
        """`opt.load_state_dict` (resume) or a foreign `p.data = ...` replaces tensors that were views of the flat buffers:
        copy their contents in and point them back at the buffers (addresses are baked into the graph)."""
        for q in self.plist:
            sl = self.views[q]
            if q.data.data_ptr() != self.P[sl].data_ptr():
                self.P[sl].copy_(q.data.reshape(-1))
                q.data = self.P[sl].view(q.shape)
            st = self.opt.state.get(q)
            if st is None or "exp_avg" not in st:
                self.M[sl].zero_(); self.V[sl].zero_()
                self.opt.state[q] = {"step": torch.zeros((), dtype=torch.float32, device=self.dev),
                                     "exp_avg": self.M[sl].view(q.shape), "exp_avg_sq": self.V[sl].view(q.shape)}
                continue
            if st["exp_avg"].data_ptr() != self.M[sl].data_ptr():
                self.M[sl].copy_(st["exp_avg"].reshape(-1))
                self.V[sl].copy_(st["exp_avg_sq"].reshape(-1))
                st["exp_avg"], st["exp_avg_sq"] = self.M[sl].view(q.shape), self.V[sl].view(q.shape)
            if not torch.is_tensor(st["step"]) or st["step"].device != self.P.device:
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=self.dev).reshape(())

    def run_steps(self, n: int):
        """the next n minibatch steps of the bound update"""
        if self.allreduce_mode == "sharded" and n > 0:
            self.moments_partial = True
        if self.in_graph:
            k = self.graph_steps if self.graph_multi is not None else 0
            while k and n >= k:
                self.graph_multi.replay()
                n -= k
            for _ in range(n):
                self.graph.replay()
            return
        for _ in range(n):          # a backend that cannot be captured: graphs of kernel groups, eager blocking collectives between
            for item in self.segs:
                if item[0] == "g":
                    item[1].replay()
                else:
                    w = item[2](False)
                    if w is not None:
                        w.wait()

    def gather_optimizer_state(self):
        """"sharded": Adam's moments are current on this rank's slices only — before the optimizer state is SAVED, every rank
        calls this (a collective: two all-gathers per bucket) and ends with the complete moments, as in the replicated forms."""
        if self.allreduce_mode != "sharded":
            return
        self.moments_partial = False
        r = self.rank
        for t in (self.M, self.V):
            for o, ln in zip(self.bucket_off, self.bucket_len):
                w = self.coll.all_gather(t[o:o + self.world * ln], t[o + r * ln:o + (r + 1) * ln], False)
                if w is not None:
                    w.wait()

    def end_update(self):
        """-> the [steps, 8] log of the update (total, value_loss, loss_actor, entropy, approx_kl, clipfrac, illegal-action
        norm / 2, 0): ONE launch over the sums the steps left behind"""
        with torch.no_grad():  # every parameter's step counter (torch keeps one per parameter)
            for q in self.plist:
                self.opt.state[q]["step"].copy_(self.step)
            self.capi.check(self.lib.brl_ppo_stats_rows(self._di(), self.stat_sums.data_ptr(), self.gram_sums.data_ptr(), self._steps,
                                                        self.mbs, float(self.cfg["vf_coef"]), float(self.cfg["ent_coef"]),
                                                        self.ill_coef, self.log.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream))
            self._bind_gather(*self._dummy, first=False)   # (the trajectory may be freed by the caller now)
        self._keep = None
        return self.log[:self._steps]


class FusedMinibatch(FusedStep):
    """One PPO minibatch step of a "DeepMind" MLP (src/update.py:74-242) with the forward layers and the batched weight gradient
    left to the library and everything else hand-written HIP: the minibatch gather (``brl_mb_gather_dev``:
    device-resident arguments), the 39-column head + ``_loss_fn`` + its gradients (``brl_ppo_heads_loss_split``), the head's
    backward (``brl_ppo_heads_bwd``), dz_{l-1} = (dz_l W_l) act'(h_{l-1}) with the bias gradient's tile sums in the epilogue
    (``brl_mlp_gemm``), global-norm clipping + Adam on flat parameter / gradient / moment buffers; the backward pass is written
    out (no autograd); EIGHT steps are one hipGraph; the logged statistics are formed once per update from per-step sums
    (``brl_ppo_stats_rows``).

    Under a process group (``world > 1``, ``config["grad_allreduce"]``) the program (see ``FusedStep``) carries collectives:

      "flat" (default)  the single-rank launches, ONE all-reduce of the flat gradient behind them, clip + Adam replicated on every
          rank.  Nothing overlaps the collective.
      "sharded"  reduce-scatter of the gradient bucket by bucket (one bucket per hidden layer, issued as soon as its weight
          gradient exists, beside the rest of the backward pass; the last bucket = layer 0 + head + biases), the norm's partial
          sums of the rank's own slices + a 4 KB all-gather, clip + Adam on the rank's 1/world slices only, then the all-gather
          of the updated parameters bucket by bucket in forward order — the next step's layer l waits for ITS bucket only.  Same
          ring bytes as an all-reduce; the Adam sweep shrinks by 1/world; the moments are valid on the rank's slices only
          (``gather_optimizer_state`` completes them for a checkpoint).  Opt-in: every kernel on a second branch of the graph
          costs ~10 us per cross-stream edge (scripts/graph_overlap_micro2.py), nine collectives per step pay back what they
          hide, and the emulation (scripts/overlap_probe.py) puts the two forms level.

    Given the same launches (config["per_layer_dw"]) both forms give bit-identical parameters (the norm's partials have one
    layout: csrc/ppo_update.hpp ShardGeom).

    The module's parameters and the optimizer's moments become VIEWS of the flat buffers, so ``params``, ``state_dict``
    checkpoints and the eager path keep working on the same memory.  Mirrors torch.optim.Adam's arithmetic and
    ``clip_grad_norm_``; checked against the float64 numpy restatement and the eager path (tests/test_gpu_parity.py)."""

    @staticmethod
    def supports(config, params) -> bool:
        # the DeepMind MLPs (4 / 6 / 8 x 1024) with either activation (src/models.py:16), reward_scaling and a non-zero
        # illegal_action_l2norm_coef included; the FAIR net takes the autograd path
        return (bool(config.get("fused_update", True)) and str(getattr(params, "model", "")).startswith("DeepMind")
                and getattr(params, "act", None) in (torch.relu, torch.tanh)
                and params.body[0].weight.shape[0] % 256 == 0
                and next(params.parameters()).is_cuda and next(params.parameters()).dtype == torch.float32)

    def _layout(self):
        """-> (parameters in flat order, their sizes, bucket offsets, slice lengths, n)"""
        params, multi = self.params, self.multi
        body = list(params.body)
        nl = len(body)
        H = body[0].weight.shape[0]
        K = params.actor.weight.shape[0] + 1
        self.H, self.K, self.nl = H, K, nl
        # ---- flat layout: W_1 .. W_{nl-1} (consecutive [H, H] blocks: one batched weight-gradient product, one collective bucket
        # each), then the TAIL = W_0 | actor.weight | critic.weight (one [39, H] matrix) | the biases (hidden layers, actor |
        # critic): the tail is what the backward pass finishes last — one bucket —, and its last part (head + biases) is what the
        # sums of partials produce (brl_adam_clip_fin_gather wants those at the END of the buffer)
        plist = [lin.weight for lin in body[1:]] + [body[0].weight, params.actor.weight, params.critic.weight] \
            + [lin.bias for lin in body] + [params.actor.bias, params.critic.bias]
        sizes = [q.numel() for q in plist]
        hid = (nl - 1) * H * H
        tail = sum(sizes) - hid
        # buckets of the multi-rank step: every hidden layer + the tail, each cut into `world` slices of whole float4s
        self.sharded_geom = multi and (H * H) % (4 * self.world) == 0
        if self.allreduce_mode == "sharded" and not self.sharded_geom:
            raise ValueError(f"grad_allreduce='sharded' needs hidden^2 divisible by 4 * world (hidden {H}, world {self.world}): use 'flat'")
        quantum = 4 * self.world if multi else 4
        tail_pad = (tail + quantum - 1) // quantum * quantum
        n = hid + tail_pad                      # zero padding at the end
        offs = lens = None
        if multi:
            if self.sharded_geom:
                offs = [l * H * H for l in range(nl - 1)] + [hid]
                lens = [H * H // self.world] * (nl - 1) + [tail_pad // self.world]
            else:                                # one bucket: the whole buffer
                n = (sum(sizes) + quantum - 1) // quantum * quantum
                offs, lens = [0], [n // self.world]
        self._hid = hid
        return plist, sizes, offs, lens, n

    def _alloc(self):
        import ctypes as C
        params, views, f, B, config = self.params, self.views, self._f, self.mbs, self.cfg
        body = list(params.body)
        nl, H, K, hid = self.nl, self.H, self.K, self._hid
        self.W = [lin.weight for lin in body]                      # [out, in] views of P
        self.b = [lin.bias for lin in body]
        self.GW = [self.G[views[lin.weight]].view(lin.weight.shape) for lin in body]
        self.Gb = [self.G[views[lin.bias]] for lin in body]
        wa = views[params.actor.weight]
        self.Wh = self.P[wa.start:wa.start + K * H].view(K, H)     # actor rows, then the critic row
        self.GWh = self.G[wa.start:wa.start + K * H].view(K, H)
        ba = views[params.actor.bias]
        self.bh = self.P[ba.start:ba.start + K]
        self.Gbh = self.G[ba.start:ba.start + K]
        # Activations and the gradients w.r.t. the pre-activations live in STACKED static buffers (out= costs the fused bias +
        # ReLU GEMM nothing: scripts/fwd_probe.py): the hidden layers' weight gradients are ONE batched product
        self.hs = f(nl, B, H)                       # h_l = act(h_{l-1} W_l^T + b_l)
        self.dzs = f(nl, B, H)                      # d(loss) / d(pre-activation of layer l)
        self.h = [self.hs[l] for l in range(nl)]
        self.GW_hidden = self.G[:hid].view(nl - 1, H, H) if nl > 1 else None   # the gradients of W_1 .. W_{nl-1} as one tensor
        self.dheads = f(B, K)
        self.heads = f(B, K) if self.ill_coef else None      # the gradient of the illegal-action norm re-reads the logits
        self.head_ksplit = max(1, min(4, H // 256))          # K ranges of the heads product (brl_ppo_heads_loss_split)
        self.head_parts = f(self.head_ksplit, B, K)
        self.vec = f(40) if self.ill_coef else None          # v1 [38], sigma_1 of the step's illegal-action matrix
        groups = (B + 15) // 16                        # 16-row tiles of the bias-gradient column sums
        self.gram_partials = f(self.lgroups, 38 * 38)
        self.scratch = f(8192)   # single rank: norm partials (1024 blocks + the finalize blocks that ride in the norm launch)
        self.nsplit = (B + 63) // 64                   # batch splits of the head's weight / bias gradient (brl_ppo_heads_bwd)
        self.dwh_partials = f(self.nsplit, K * H)
        self.dbh_partials = f(self.nsplit, K)
        # the step's 1024^3-class products on this library's own fp32 MFMA kernel (brl_mlp_gemm, csrc/mlp_gemm.hpp) where its
        # fused epilogue removes a launch: dh = dz W with the activation derivative and the bias-gradient tile sums inside
        # (replaces torch.mm + brl_act_bwd_colsum), and — sharded form — each layer's own weight gradient.  The forward layers
        # (config["own_gemm_fwd"]: opt-in, 19.8-20.1 us per layer in the step against the tuned library kernel's 19.0-19.6) and
        # the single-rank batched weight gradient stay with the library.
        self.own_gemm = bool(config.get("own_gemm", True)) and B % 4 == 0 and H % 4 == 0 and nl > 1
        self.own_fwd = self.own_gemm and bool(config.get("own_gemm_fwd", False))
        # config["dw_gemm"] = "bf16x3" (the default; "library": torch.bmm + torch.mm): the single-rank / flat step's weight gradients
        # dz_l^T h_{l-1} (every layer, layer 0 included) as ONE launch of brl_mlp_gemm_x3_group (224 tiles of 128 x 128 at the default
        # shape: one per CU) instead of the library's batched product + a launch for layer 0: 41 us instead of 47 + 11
        # (profiles/r06/r06aa_step_ab.txt: the step 0.229 -> 0.211 ms); the products carry the bf16x3 error (below the exact fp32
        # kernel's rounding: tests/test_gpu_parity.py::test_mlp_gemm_x3_beats_the_exact_kernels_error)
        self.dw_x3 = (config.get("dw_gemm") or "bf16x3") == "bf16x3" and B % 32 == 0 and H % 4 == 0 and self.x0.shape[1] % 4 == 0 and nl <= 8
        if self.dw_x3:
            a_ = [self.dzs[l] for l in range(nl)]
            b_ = [self.x0] + [self.hs[l] for l in range(nl - 1)]
            c_ = [self.GW[l] for l in range(nl)]
            i64, vp = C.c_int64 * nl, C.c_void_p * nl
            self._gdw = (vp(*[t_.data_ptr() for t_ in a_]), i64(*[t_.stride(0) for t_ in a_]), vp(*[t_.data_ptr() for t_ in b_]),
                         i64(*[t_.stride(0) for t_ in b_]), vp(*[t_.data_ptr() for t_ in c_]), i64(*[t_.stride(0) for t_ in c_]),
                         i64(*[t_.shape[0] for t_ in c_]), i64(*[t_.shape[1] for t_ in c_]), i64(*([B] * nl)))
        # the head's weight-gradient role (not on the backward chain) rides with the first launch of the dz chain below the top
        self.dw_deferred = nl > 1
        groups64 = (B + 63) // 64                      # 64-row tiles of brl_mlp_gemm's column sums
        self.tile_rows = [64 if (self.own_gemm and l < nl - 1) else 16 for l in range(nl)]
        self.tile_sums = [f(groups * H) for _ in body]   # per-layer partial column sums (bias gradients)
        # one launch finishes every sum of partials: the head's weight gradient, the hidden layers' bias gradients, the head's
        # bias gradient — in the order they sit at the end of the flat buffer
        nseg = nl + 2
        self._seg_scratch = (C.c_void_p * nseg)(*([self.dwh_partials.data_ptr()] + [t.data_ptr() for t in self.tile_sums]
                                                  + [self.dbh_partials.data_ptr()]))
        self._seg_cols = (C.c_int64 * nseg)(*([K * H] + [H] * nl + [K]))
        self._seg_tiles = (C.c_int64 * nseg)(*([self.nsplit] + [groups64 if r == 64 else groups for r in self.tile_rows] + [self.nsplit]))
        self._seg_db = (C.c_void_p * nseg)(*([self.GWh.data_ptr()] + [g.data_ptr() for g in self.Gb] + [self.Gbh.data_ptr()]))
        self._nseg = nseg

    # ---- the step as a program ---------------------------------------------------------------------------------------
    def _build_program(self):
        """-> [("k", fn) | ("c", key, fn(async_op) -> work) | ("w", key)]: one minibatch step on the current stream"""
        nl = self.nl
        if self.allreduce_mode == "none":
            return [("k", self._step_single)]
        co, G, P = self.coll, self.G, self.P
        if self.allreduce_mode == "flat":
            return [("k", self._grads_chain),
                    ("c", "ar", lambda a: co.all_reduce(G, a)), ("w", "ar"),          # SUM: the sweep scales by 1 / world
                    ("k", lambda: (self._shard_norm(0, self.world), self._shard_apply(0, self.world)))]
        # "sharded"
        r = self.rank
        bucket = lambda t, b: t[self.bucket_off[b]:self.bucket_off[b] + self.world * self.bucket_len[b]]   # noqa: E731
        mine = lambda t, b: t[self.bucket_off[b] + r * self.bucket_len[b]:self.bucket_off[b] + (r + 1) * self.bucket_len[b]]  # noqa: E731
        tail = nl - 1                                  # bucket index of W_0 | head | biases; bucket l - 1 = W_l
        np_ = self.norm_partials
        per = np_.numel() // self.world
        prog = [("w", "ag%d" % tail), ("k", lambda: self._forward(0))]
        for l in range(1, nl):                         # layer l multiplies with W_l: its all-gather (of the step before) must be in
            prog += [("w", "ag%d" % (l - 1)), ("k", lambda l=l: self._forward(l))]
        prog += [("k", self._heads)]
        for l in range(nl - 1, 0, -1):                 # dW_l as soon as dz_l exists -> its bucket leaves; then dz_{l-1}
            prog += [("k", lambda l=l: self._dw(l)),
                     ("c", "rs%d" % (l - 1), lambda a, l=l: co.reduce_scatter(mine(G, l - 1), bucket(G, l - 1), a)),
                     ("k", lambda l=l: self._dz(l))]
        prog += [("k", lambda: (self._dw(0), self._seg_fin())),
                 ("c", "rs%d" % tail, lambda a: co.reduce_scatter(mine(G, tail), bucket(G, tail), a))]
        prog += [("w", "rs%d" % b) for b in range(nl)]
        prog += [("k", lambda: self._shard_norm(r, r + 1)),
                 ("c", "agn", lambda a: co.all_gather(np_, np_[r * per:(r + 1) * per], a)), ("w", "agn"),
                 ("k", lambda: self._shard_apply(r, r + 1))]
        for b in [tail] + list(range(nl - 1)):         # parameters back, in the order the next forward pass needs them
            prog += [("c", "ag%d" % b, lambda a, b=b: co.all_gather(bucket(P, b), mine(P, b), a))]
        return prog

    # ---- kernel groups (each on the current stream) --------------------------------------------------------------------
    def _step_single(self):
        """single rank: forward, heads, the dz chain, the batched weight gradients, sums of partials inside the norm launch, clip +
        Adam (+ the next gather)"""
        for l in range(self.nl):
            self._forward(l)
        self._heads()
        self._backward_chain()
        self._fin_opt()

    def _grads_chain(self):
        """multi-rank "flat" form: everything that produces this rank's gradient — the single-rank chain (batched weight gradients),
        the sums of partials as a launch of their own (they must exist before the all-reduce).  config["per_layer_dw"]: the SAME
        launches as the sharded form instead (a weight gradient per layer): the two forms then reduce bit-identical gradients —
        what the tests compare."""
        for l in range(self.nl):
            self._forward(l)
        self._heads()
        if self.cfg.get("per_layer_dw", False):
            for l in range(self.nl - 1, 0, -1):
                self._dw(l)
                self._dz(l)
            self._dw(0)
        else:
            self._backward_chain()
        self._seg_fin()

    def _forward(self, l):
        """h_l = act(h_{l-1} W_l^T + b_l); layer 0 reads the minibatch the previous step's Adam launch (or the bind) gathered"""
        x = self.x0 if l == 0 else self.h[l - 1]
        W, b = self.W[l], self.b[l]
        if self.own_fwd and l > 0:                            # own kernel: bias + activation in its epilogue
            self.capi.check(self.lib.brl_mlp_gemm(self._di(), 0, 1, x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0),
                                                  self.h[l].data_ptr(), self.h[l].stride(0), self.mbs, W.shape[0], W.shape[1], self.act,
                                                  b.data_ptr(), None, 0, None, None, torch.cuda.current_stream().cuda_stream))
        elif self.act == 0:                                   # ReLU in the GEMM epilogue
            torch._addmm_activation(b, x, W.t(), use_gelu=False, out=self.h[l])
        else:
            torch.addmm(b, x, W.t(), out=self.h[l]).tanh_()

    def _heads(self):
        """heads + `_loss_fn` + output gradients (two launches), backward of the merged head down to the top hidden layer's
        pre-activation (one launch)"""
        L, chk, B = self.lib, self.capi.check, self.mbs
        s = torch.cuda.current_stream().cuda_stream
        di, cfg = self._di(), self.cfg
        x = self.h[self.nl - 1]
        chk(L.brl_ppo_heads_loss_split(di, x.data_ptr(), x.stride(0), self.Wh.data_ptr(), self.bh.data_ptr(), self.H,
                                       self.mask.data_ptr(), self.action.data_ptr(), self.old_v.data_ptr(), self.old_lp.data_ptr(),
                                       self.adv.data_ptr(), self.tgt.data_ptr(), B, float(cfg["clip_eps"]), float(cfg["vf_coef"]),
                                       float(cfg["ent_coef"]), int(bool(cfg.get("actor_illegal_action_mask", True))),
                                       int(bool(cfg.get("value_clipping", True))), int(bool(cfg.get("reward_scaling", False))),
                                       self.heads.data_ptr() if self.ill_coef else None,
                                       self.dheads.data_ptr(), self.partials.data_ptr(), self.gram_partials.data_ptr(),
                                       self.head_parts.data_ptr(), self.head_ksplit, s))
        if self.ill_coef:
            # src/update.py:146-152: + coef * sigma_1(P) / 2 — its gradient needs the step's top singular pair NOW (the logged
            # statistics otherwise wait for the end of the update): one stats launch on the critical path of this configuration
            chk(L.brl_ppo_stats_gram(di, self.partials.data_ptr(), self.lgroups, B, self.gram_partials.data_ptr(), self.lgroups,
                                     float(cfg["vf_coef"]), float(cfg["ent_coef"]), self.ill_coef, self.out.data_ptr(), None,
                                     self.vec.data_ptr(), s))
            chk(L.brl_ppo_illegal_grad(di, self.heads.data_ptr(), self.mask.data_ptr(), self.vec.data_ptr(), self.ill_coef, B,
                                       self.dheads.data_ptr(), s))
        # backward of the head, written out: dz of the top hidden layer (activation derivative applied) and its bias-gradient tile
        # sums; with one hidden layer only also dW_h / db_h partials per batch split (else they ride with the first dz launch)
        top = self.nl - 1
        chk(L.brl_ppo_heads_bwd(di, self.dheads.data_ptr(), x.data_ptr(), x.stride(0), self.Wh.data_ptr(), B, self.H, self.act,
                                self.nsplit, None if self.dw_deferred else self.dwh_partials.data_ptr(),
                                None if self.dw_deferred else self.dbh_partials.data_ptr(),
                                self.dzs[top].data_ptr(), self.tile_sums[top].data_ptr(), self.partials.data_ptr(),
                                self.gram_partials.data_ptr(), self.lgroups, self.mb_index.data_ptr(), self.stat_sums.data_ptr(),
                                self.gram_sums.data_ptr(), s))

    def _dz(self, l):
        """dz_{l-1} = (dz_l W_l) * act'(h_{l-1}) + the tile sums of db_{l-1}; the first of the chain (l = nl - 1) also hosts the head's
        dW_h / db_h partials and the step's statistics sums as extra workgroups"""
        L, chk, B, H = self.lib, self.capi.check, self.mbs, self.H
        s = torch.cuda.current_stream().cuda_stream
        di = self._di()
        first = l == self.nl - 1 and self.dw_deferred
        top = self.h[self.nl - 1]
        dz, out, W, gate, ts = self.dzs[l], self.dzs[l - 1], self.W[l], self.h[l - 1], self.tile_sums[l - 1]
        heads_dw = (self.dheads.data_ptr(), top.data_ptr(), top.stride(0), B, H, self.nsplit, self.dwh_partials.data_ptr(),
                    self.dbh_partials.data_ptr(), self.partials.data_ptr(), self.gram_partials.data_ptr(), self.lgroups,
                    self.mb_index.data_ptr(), self.stat_sums.data_ptr(), self.gram_sums.data_ptr(), s)
        if self.own_gemm:
            if first:
                chk(L.brl_mlp_gemm_dh_heads_dw(di, dz.data_ptr(), H, W.data_ptr(), W.stride(0), out.data_ptr(), H, B, W.shape[1], W.shape[0],
                                               self.act, gate.data_ptr(), H, ts.data_ptr(), *heads_dw))
            else:
                chk(L.brl_mlp_gemm(di, 1, 2, dz.data_ptr(), H, W.data_ptr(), W.stride(0), out.data_ptr(), H, B, W.shape[1], W.shape[0],
                                   self.act, None, gate.data_ptr(), H, ts.data_ptr(), None, s))
            return
        torch.mm(dz, W, out=out)
        if first:
            chk(L.brl_act_bwd_colsum_heads_dw(di, out.data_ptr(), gate.data_ptr(), B, H, H, self.act, ts.data_ptr(), *heads_dw))
        else:
            chk(L.brl_act_bwd_colsum(di, out.data_ptr(), gate.data_ptr(), B, H, H, self.act, ts.data_ptr(), s))

    def _dw(self, l):
        """dW_l = dz_l^T h_{l-1} as a product of its own (sharded form: the layer's bucket leaves as soon as it exists)"""
        src = self.x0 if l == 0 else self.h[l - 1]
        if self.own_gemm and l > 0:
            dz, GW = self.dzs[l], self.GW[l]
            self.capi.check(self.lib.brl_mlp_gemm(self._di(), 2, 0, dz.data_ptr(), dz.stride(0), src.data_ptr(), src.stride(0),
                                                  GW.data_ptr(), GW.stride(0), GW.shape[0], GW.shape[1], self.mbs, self.act, None, None, 0,
                                                  None, None, torch.cuda.current_stream().cuda_stream))
        else:
            torch.mm(self.dzs[l].t(), src, out=self.GW[l])

    def _backward_chain(self):
        """the dz chain first (dz_l for every layer), then the weight gradients — layers 1.. as ONE batched product (three 1024^3
        products: 46-48 us instead of 3 x 19-21, scripts/bmm_probe.py), layer 0 (K = 480 columns) beside it"""
        nl = self.nl
        for l in range(nl - 1, 0, -1):
            self._dz(l)
        if self.dw_x3:
            self.capi.check(self.lib.brl_mlp_gemm_x3_group(self._di(), 2, nl, *self._gdw, torch.cuda.current_stream().cuda_stream))
            return
        if nl > 1:
            torch.bmm(self.dzs[1:].transpose(1, 2), self.hs[:nl - 1], out=self.GW_hidden)
        torch.mm(self.dzs[0].t(), self.x0, out=self.GW[0])

    def _seg_fin(self):
        self.capi.check(self.lib.brl_bias_finalize_ex(self._di(), self._nseg, self._seg_scratch, self._seg_cols, self._seg_tiles,
                                                      self._seg_db, torch.cuda.current_stream().cuda_stream))

    def _fin_opt(self):
        """single rank: every sum of partials is finished by extra workgroups of the norm launch (brl_adam_clip_fin_gather)"""
        self.capi.check(self.lib.brl_adam_clip_fin_gather(
            self._di(), self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.n, self.step.data_ptr(), self.lr,
            self.lr_dev.data_ptr(), float(self.b1), float(self.b2), self.eps, self.max_norm, self.scratch.data_ptr(),
            self.scratch.numel(), self.mb_index.data_ptr(), self.norm.data_ptr(), self.gargs.data_ptr(), self.mbs, self._nseg,
            self._seg_scratch, self._seg_cols, self._seg_tiles, self._seg_db, torch.cuda.current_stream().cuda_stream))



class FusedFair(FusedStep):
    """The same for the "FAIR" network (src/models.py:34-69: eleven 200-wide ``hk.Linear``s in four residual blocks, the observation
    concatenated back in front of the seventh, the two heads on the last block's output): forward and backward written out (no
    autograd) on the flat buffers.  Its 35 products are 80 MFLOP each, so the step is shaped by launch count.  FIVE launches:
    ``brl_fair_chain`` (csrc/fair_chain.hpp: forward, ``_loss_fn`` and the whole backward chain of 16 samples per workgroup — rows
    are independent up to the weight gradients; activations in LDS, weights streamed from L2 a job ahead of the fp32 MFMA tiles that
    use them; 108 us), ``brl_mlp_gemm_group`` (the twelve weight gradients dz^T x as one launch, 21 us), ``brl_bias_finalize_rows``
    (every bias gradient + this step's row of the statistics / Gram sums), clip + Adam through the two shard launches (one bucket =
    the whole buffer); the log rows are formed once per update (``brl_ppo_stats_rows``).  Eight steps per hipGraph: 0.147 ms per
    step at configs[3]'s sizes.  ``fair_chain=False`` (or a minibatch that is not a multiple of 16): the same step launch by launch
    (library products, ``brl_mlp_gemm`` GATE_COLSUM for the chain, ``brl_act_bwd_colsum``; ~70 launches x ~4.7 us = 0.35 ms); the
    autograd step both replace (GraphedMinibatch), which re-gathers the whole trajectory per epoch and copies every minibatch in:
    0.70 ms (profiles/r05/r05t_fair_step_probe.txt).
    Multi-rank: one bucket — "flat" = all-reduce + replicated sweep, "sharded" = reduce-scatter, sweep of the rank's slice,
    all-gather of the parameters; bit-identical.  A non-zero illegal_action_l2norm_coef (src/update.py:146-152) takes the launch-by-launch
    form with two more launches between the loss and the backward pass (the step's top singular pair, its gradient)."""

    SQUARE = (1, 2, 3, 4, 5, 7, 8, 9, 10)      # the 200 x 200 layers, in the order of their stacked buffers
    # input of layer l (name in self.t) — the stacked input buffer holds them in SQUARE's order
    INPUT = {1: "h0", 2: "h1", 3: "g1", 4: "h3", 5: "x2", 7: "h6", 8: "h7", 9: "g3", 10: "h9"}

    @staticmethod
    def supports(config, params) -> bool:
        # (a non-zero illegal_action_l2norm_coef included: that configuration runs launch by launch — the term's gradient needs the
        #  top singular pair of the WHOLE minibatch's illegal-probability matrix between the loss and the backward pass, which one
        #  launch of independent 16-row workgroups cannot have)
        return (bool(config.get("fused_update", True)) and getattr(params, "model", "") == "FAIR"
                and getattr(params, "act", None) in (torch.relu, torch.tanh)
                and next(params.parameters()).is_cuda and next(params.parameters()).dtype == torch.float32)

    def _layout(self):
        p = self.params
        L = list(p.l)
        # the square layers first and consecutive (one batched weight-gradient product), then W_0, W_6, the heads, the biases
        plist = [L[l].weight for l in self.SQUARE] + [L[0].weight, L[6].weight, p.actor.weight, p.critic.weight] \
            + [lin.bias for lin in L] + [p.actor.bias, p.critic.bias]
        sizes = [q.numel() for q in plist]
        quantum = 4 * self.world if self.multi else 4
        n = (sum(sizes) + quantum - 1) // quantum * quantum
        return plist, sizes, ([0] if self.multi else None), ([n // self.world] if self.multi else None), n

    def _alloc(self):
        import ctypes as C
        p, f, B, views = self.params, self._f, self.mbs, self.views
        L = list(p.l)
        H = self.H = L[0].weight.shape[0]
        nsq = len(self.SQUARE)
        self.W, self.b = [lin.weight for lin in L], [lin.bias for lin in L]
        self.GW = [self.G[views[lin.weight]].view(lin.weight.shape) for lin in L]
        self.Gb = [self.G[views[lin.bias]] for lin in L]
        gv = lambda q: self.G[views[q]].view(q.shape)   # noqa: E731
        self.Wa, self.wc, self.ba, self.bc = p.actor.weight, p.critic.weight, p.actor.bias, p.critic.bias
        self.GWa, self.Gwc, self.Gba, self.Gbc = gv(p.actor.weight), gv(p.critic.weight), gv(p.actor.bias), gv(p.critic.bias)
        self.GW_square = self.G[:nsq * H * H].view(nsq, H, H)          # the square layers' weight gradients as one tensor
        # L[6] multiplies cat([z5, obs]): its two column blocks as (strided) views — no concatenation is ever formed
        self.W6a, self.W6b = L[6].weight[:, :H], L[6].weight[:, H:]
        self.GW6 = self.GW[6]
        self.cat6 = f(B, H + 480)                                      # [z5 | obs]: only dW_6 = dz_6^T cat6 reads it whole
        self.inp = f(nsq, B, H)                                        # the square layers' inputs, stacked
        self.dzs = f(nsq, B, H)                                        # ... and their pre-activation gradients
        self.t = {name: self.inp[i] for i, name in enumerate(self.INPUT[l] for l in self.SQUARE)}
        for name in "z0 h2 x1 h4 z6 h8 x3 h10 x4 dc dx dz0 dz6".split():
            self.t[name] = f(B, H)
        self.t["z5"] = self.cat6[:, :H]
        self.dz = {l: self.dzs[i] for i, l in enumerate(self.SQUARE)}
        self.dz[0], self.dz[6] = self.t["dz0"], self.t["dz6"]
        self.logits, self.value = f(B, 38), f(B)
        self.dlogits, self.dvalue, self.illp, self.gram = f(B, 38), f(B), f(B, 38), f(1, 38 * 38)
        self.ones_row = torch.ones((1, B), dtype=torch.float32, device=self.dev)
        # a gate under which act' == 1 (column sums of a gradient that has no activation in front of it)
        self.unit_gate = torch.ones((B, H), dtype=torch.float32, device=self.dev) if self.act == 0 else f(B, H)
        self.own_gemm = bool(self.cfg.get("own_gemm", True)) and B % 4 == 0 and H % 4 == 0
        # bias gradients: every layer's dz leaves column sums per row tile (16 rows: brl_act_bwd_colsum; 64 rows: brl_mlp_gemm's
        # epilogue), ONE launch finishes all eleven
        self.from_gemm = (1, 3, 7, 9) if self.own_gemm else ()
        t16, t64 = (B + 15) // 16, (B + 63) // 64
        self.tiles = [f((t64 if l in self.from_gemm else t16), H) for l in range(11)]
        self._seg_scratch = (C.c_void_p * 11)(*[t.data_ptr() for t in self.tiles])
        self._seg_cols = (C.c_int64 * 11)(*([H] * 11))
        self._seg_tiles = (C.c_int64 * 11)(*[t.shape[0] for t in self.tiles])
        self._seg_db = (C.c_void_p * 11)(*[g.data_ptr() for g in self.Gb])
        if not self.multi:      # single rank: the sweep through the shard launches with a one-bucket, world-1 geometry
            g = self.capi.ShardGeom()
            g.nbuckets, g.world, g.nsub = 1, 1, 1024
            g.off[0], g.len[0] = 0, self.n
            self.geom, self.norm_partials = g, f(1024)
        # forward + loss + backward chain as ONE launch (brl_fair_chain: 16 samples per workgroup, activations in LDS)
        self.chain = bool(self.cfg.get("fair_chain", True)) and B % 16 == 0 and H == 200 and self.x0.shape[1] == 480 and not self.ill_coef
        if self.ill_coef:      # src/update.py:146-152 on the launch-by-launch path: the [B,39] images brl_ppo_illegal_grad works on
            self.heads39, self.dheads39, self.vec = f(B, 39), f(B, 39), f(40)
        if self.chain:
            nwg = B // 16
            assert p.critic.weight.data_ptr() == p.actor.weight.data_ptr() + 38 * H * 4 \
                and p.critic.bias.data_ptr() == p.actor.bias.data_ptr() + 38 * 4      # the heads as one [39, H] / [39] in the flat buffer
            sw, sb = views[p.actor.weight], views[p.critic.bias]
            self.GWh = self.G[sw.start:sw.start + 39 * H].view(39, H)
            self.gates, self.dheads = f(4, B, H), f(B, 40)
            self.ctiles = f(11 * nwg * H + nwg * 39)
            self.cpartials, self.cgram = f(nwg, 8), f(nwg, 38 * 38)
            net, wk = self.capi.FairNet(), self.capi.FairWork()
            for l, lin in enumerate(L):
                net.w[l], net.b[l] = lin.weight.data_ptr(), lin.bias.data_ptr()
            net.head_w, net.head_b = p.actor.weight.data_ptr(), p.actor.bias.data_ptr()
            for name, t_ in (("inp", self.inp), ("dzs", self.dzs), ("gates", self.gates), ("cat6", self.cat6), ("x4", self.t["x4"]),
                             ("dz0", self.dz[0]), ("dz6", self.dz[6]), ("dheads", self.dheads), ("tiles", self.ctiles),
                             ("partials", self.cpartials), ("gram_partials", self.cgram)):
                setattr(wk, name, t_.data_ptr())
            self._net, self._work = net, wk
            # one launch finishes the eleven bias gradients, the heads' and this step's row of the statistics / Gram sums
            parts = [self.ctiles.data_ptr() + 4 * l * nwg * H for l in range(12)] + [self.cpartials.data_ptr(), self.cgram.data_ptr()]
            outs = [g_.data_ptr() for g_ in self.Gb] + [self.G[views[p.actor.bias].start:].data_ptr(), self.stat_sums.data_ptr(),
                                                         self.gram_sums.data_ptr()]
            cols = [H] * 11 + [39, 8, 38 * 38]
            self._cseg = ((C.c_void_p * 14)(*parts), (C.c_int64 * 14)(*cols), (C.c_int64 * 14)(*([nwg] * 14)), (C.c_void_p * 14)(*outs))
            # the twelve weight gradients (eleven layers + the heads) dW = dz^T x as ONE launch (brl_mlp_gemm_group; else: library products)
            self.group_dw = self.own_gemm and bool(self.cfg.get("fair_group_dw", True))
            a_ = [self.dzs[i] for i in range(nsq)] + [self.dz[6], self.dz[0], self.dheads]
            b_ = [self.inp[i] for i in range(nsq)] + [self.cat6, self.x0, self.t["x4"]]
            c_ = [self.GW_square[i] for i in range(nsq)] + [self.GW6, self.GW[0], self.GWh]
            i64, vp = C.c_int64 * 12, C.c_void_p * 12
            self._gdw = (vp(*[t_.data_ptr() for t_ in a_]), i64(*[t_.stride(0) for t_ in a_]), vp(*[t_.data_ptr() for t_ in b_]),
                         i64(*[t_.stride(0) for t_ in b_]), vp(*[t_.data_ptr() for t_ in c_]), i64(*[t_.stride(0) for t_ in c_]),
                         i64(*[t_.shape[0] for t_ in c_]), i64(*[t_.shape[1] for t_ in c_]), i64(*([B] * 12)))

    # ---- the step ------------------------------------------------------------------------------------------------------
    def _build_program(self):
        if self.allreduce_mode == "none":
            return [("k", lambda: (self._grads(), self._shard_norm(0, 1), self._shard_apply(0, 1)))]
        co, G, P, r, W = self.coll, self.G, self.P, self.rank, self.world
        if self.allreduce_mode == "flat":
            return [("k", self._grads), ("c", "ar", lambda a: co.all_reduce(G, a)), ("w", "ar"),
                    ("k", lambda: (self._shard_norm(0, W), self._shard_apply(0, W)))]
        ln = self.bucket_len[0]
        np_ = self.norm_partials
        per = np_.numel() // W
        return [("w", "ag"), ("k", self._grads),
                ("c", "rs", lambda a: co.reduce_scatter(G[r * ln:(r + 1) * ln], G, a)), ("w", "rs"),
                ("k", lambda: self._shard_norm(r, r + 1)),
                ("c", "agn", lambda a: co.all_gather(np_, np_[r * per:(r + 1) * per], a)), ("w", "agn"),
                ("k", lambda: self._shard_apply(r, r + 1)),
                ("c", "ag", lambda a: co.all_gather(P, P[r * ln:(r + 1) * ln], a))]

    def _a(self, x, out):
        return torch.clamp_min(x, 0.0, out=out) if self.act == 0 else torch.tanh(x, out=out)

    def _lin(self, x, l, out, act):
        """out = [act](x W_l^T + b_l)"""
        if act and self.act == 0:
            return torch._addmm_activation(self.b[l], x, self.W[l].t(), use_gelu=False, out=out)
        torch.addmm(self.b[l], x, self.W[l].t(), out=out)
        return out.tanh_() if act else out

    def _colsum(self, l, gate):
        """dz_l *= act'(gate) in place (gate = the activation's OUTPUT; self.unit_gate: no activation) + its column sums per
        16-row tile = the partials of db_l"""
        dz = self.dz[l]
        self.capi.check(self.lib.brl_act_bwd_colsum(self._di(), dz.data_ptr(), gate.data_ptr(), self.mbs, self.H, dz.stride(0), self.act,
                                                    self.tiles[l].data_ptr(), torch.cuda.current_stream().cuda_stream))

    def _dx(self, l, gate, out, bias_of=None):
        """out = (dz_l W_l) * act'(gate); bias_of = k: `out` IS dz_k and the launch leaves db_k's partials (brl_mlp_gemm, GATE_COLSUM)"""
        dz, W = self.dz[l], self.W[l]
        if self.own_gemm:
            cs = self.tiles[bias_of] if bias_of is not None else None
            self.capi.check(self.lib.brl_mlp_gemm(self._di(), 1, 2, dz.data_ptr(), dz.stride(0), W.data_ptr(), W.stride(0), out.data_ptr(),
                                                  out.stride(0), self.mbs, W.shape[1], W.shape[0], self.act, None, gate.data_ptr(),
                                                  gate.stride(0), cs.data_ptr() if cs is not None else None, None,
                                                  torch.cuda.current_stream().cuda_stream))
            return
        torch.mm(dz, W, out=out)
        if bias_of is not None:
            self._colsum(bias_of, gate)        # (out is self.dz[bias_of])
        elif self.act == 0:
            torch.mul(out, (gate > 0), out=out)
        else:
            torch.addcmul(out, out * gate, gate, value=-1.0, out=out)

    def _grads_chain(self):
        """the same through brl_fair_chain: one launch for forward, loss and the backward chain, the weight gradients as four
        products, one launch for every bias gradient and the step's statistics row"""
        cfg, B = self.cfg, self.mbs
        s = torch.cuda.current_stream().cuda_stream
        chk, L, di = self.capi.check, self.lib, self._di()
        chk(L.brl_fair_chain(di, self._net, self.x0.data_ptr(), self.mask.data_ptr(), self.action.data_ptr(), self.old_v.data_ptr(),
                             self.old_lp.data_ptr(), self.adv.data_ptr(), self.tgt.data_ptr(), B, float(cfg["clip_eps"]),
                             float(cfg["vf_coef"]), float(cfg["ent_coef"]), int(bool(cfg.get("actor_illegal_action_mask", True))),
                             int(bool(cfg.get("value_clipping", True))), int(bool(cfg.get("reward_scaling", False))), self.act,
                             self._work, s))
        if self.group_dw:
            chk(L.brl_mlp_gemm_group(di, 2, 12, *self._gdw, s))
        else:
            torch.bmm(self.dzs.transpose(1, 2), self.inp, out=self.GW_square)
            torch.mm(self.dz[6].t(), self.cat6, out=self.GW6)
            torch.mm(self.dz[0].t(), self.x0, out=self.GW[0])
            torch.mm(self.dheads[:, :39].t(), self.t["x4"], out=self.GWh)
        chk(L.brl_bias_finalize_rows(di, 14, *self._cseg, 12, self.mb_index.data_ptr(), s))

    def _grads(self):
        """forward, `_loss_fn`, the step's statistics sums, backward: every gradient into the flat buffer"""
        if self.chain:
            return self._grads_chain()
        t, cfg, B, dz = self.t, self.cfg, self.mbs, self.dz
        a = self._a
        x0 = self.x0
        # ---- forward (src/models.py:34-69)
        self._lin(x0, 0, t["z0"], False)                       # shortcut_1 = the PRE-activation
        a(t["z0"], t["h0"])
        self._lin(t["h0"], 1, t["h1"], True)
        self._lin(t["h1"], 2, t["h2"], True)
        torch.add(t["h2"], t["z0"], out=t["x1"])               # shortcut_2
        a(t["x1"], t["g1"])
        self._lin(t["g1"], 3, t["h3"], True)
        self._lin(t["h3"], 4, t["h4"], True)
        torch.add(t["h4"], t["x1"], out=t["x2"])
        self._lin(t["x2"], 5, t["z5"], False)                  # (into the left block of cat6)
        self.cat6[:, self.H:].copy_(x0)                        # jnp.concatenate([x, input])
        self._lin(self.cat6, 6, t["z6"], False)                # shortcut_3
        a(t["z6"], t["h6"])
        self._lin(t["h6"], 7, t["h7"], True)
        self._lin(t["h7"], 8, t["h8"], True)
        torch.add(t["h8"], t["z6"], out=t["x3"])               # shortcut_4
        a(t["x3"], t["g3"])
        self._lin(t["g3"], 9, t["h9"], True)
        self._lin(t["h9"], 10, t["h10"], True)
        torch.add(t["h10"], t["x3"], out=t["x4"])
        torch.addmm(self.ba, t["x4"], self.Wa.t(), out=self.logits)
        torch.addmv(self.bc, t["x4"], self.wc[0], out=self.value)
        # ---- `_loss_fn` and its gradient w.r.t. (logits, value): one launch; this step's sums for the log (row *mb_index)
        adv = self.adv
        if cfg.get("reward_scaling", False):                   # src/update.py:31-44 (jnp std: ddof = 0)
            adv = (adv - adv.mean()) / (adv.std(unbiased=False) + 1e-8)
        s = torch.cuda.current_stream().cuda_stream
        chk, L, di = self.capi.check, self.lib, self._di()
        chk(L.brl_ppo_loss(di, self.logits.data_ptr(), 38, self.value.data_ptr(), self.mask.data_ptr(), self.action.data_ptr(),
                           self.old_v.data_ptr(), self.old_lp.data_ptr(), adv.data_ptr(), self.tgt.data_ptr(), B, float(cfg["clip_eps"]),
                           float(cfg["vf_coef"]), float(cfg["ent_coef"]), int(bool(cfg.get("actor_illegal_action_mask", True))),
                           int(bool(cfg.get("value_clipping", True))), self.dlogits.data_ptr(), self.dvalue.data_ptr(),
                           self.partials.data_ptr(), self.illp.data_ptr(), s))
        row = self.mb_index.to(torch.int64)
        self.stat_sums.index_copy_(0, row, torch.mm(self.ones_row[:, :self.lgroups], self.partials))
        gram = torch.mm(self.illp.t(), self.illp)
        self.gram_sums.index_copy_(0, row, gram.view(1, -1))
        if self.ill_coef:
            # + coef * sigma_1(P) / 2 (src/update.py:138-152): the step's top singular pair from the Gram matrix (no SVD:
            # brl_ppo_stats_gram), then d sigma_1 / d logits added to the loss gradient (brl_ppo_illegal_grad), as FusedMinibatch does
            chk(L.brl_ppo_stats_gram(di, self.partials.data_ptr(), self.lgroups, B, gram.data_ptr(), 1, float(cfg["vf_coef"]),
                                     float(cfg["ent_coef"]), self.ill_coef, self.out.data_ptr(), None, self.vec.data_ptr(), s))
            self.heads39[:, :38].copy_(self.logits)
            self.heads39[:, 38].copy_(self.value)
            self.dheads39[:, :38].copy_(self.dlogits)
            chk(L.brl_ppo_illegal_grad(di, self.heads39.data_ptr(), self.mask.data_ptr(), self.vec.data_ptr(), self.ill_coef, B,
                                       self.dheads39.data_ptr(), s))
            self.dlogits.copy_(self.dheads39[:, :38])
        # ---- backward of the heads
        dx = t["dx"]
        torch.mm(self.dlogits, self.Wa, out=dx)
        dx.addr_(self.dvalue, self.wc[0])
        torch.mm(self.dlogits.t(), t["x4"], out=self.GWa)
        torch.mm(self.dvalue[None, :], t["x4"], out=self.Gwc)
        torch.mm(self.ones_row, self.dlogits, out=self.Gba.view(1, -1))
        torch.sum(self.dvalue, 0, keepdim=True, out=self.Gbc)
        # ---- a residual block  x_out = act(L_b(act(L_a(g)))) + x_in  with g = act(x_in) (or act of a pre-activation):
        # dx (the gradient w.r.t. x_out) -> dz_b, dz_a (kept: the batched weight-gradient product reads them) and dx += (dz_a W_a) act'(g)

        def block(lb, la, hb, ha, g):
            dz[lb].copy_(dx)
            self._colsum(lb, hb)                                    # dz_b = dx * act'(h_b), db_b partials
            self._dx(lb, ha, dz[la], bias_of=la)                    # dz_a = (dz_b W_b) act'(h_a), db_a partials
            self._dx(la, g, t["dc"])
            dx.add_(t["dc"])

        block(10, 9, t["h10"], t["h9"], t["g3"])                    # x4 = h10 + x3, g3 = act(x3): dx is now d/dx3
        block(8, 7, t["h8"], t["h7"], t["h6"])                      # x3 = h8 + z6, h6 = act(z6): dx is now d/dz6
        dz[6].copy_(dx)
        self._colsum(6, self.unit_gate)                             # db_6 partials
        torch.mm(dz[6], self.W6a, out=dz[5])                        # d/dz5 (L5 has no activation; the obs block needs no gradient)
        self._colsum(5, self.unit_gate)
        torch.mm(dz[5], self.W[5], out=dx)                          # d/dx2
        block(4, 3, t["h4"], t["h3"], t["g1"])                      # x2 = h4 + x1, g1 = act(x1): d/dx1
        block(2, 1, t["h2"], t["h1"], t["h0"])                      # x1 = h2 + z0, h0 = act(z0): d/dz0
        dz[0].copy_(dx)
        self._colsum(0, self.unit_gate)
        # ---- weight gradients: the nine square layers as ONE batched product, W_6 on [z5 | obs], W_0 on obs; every bias gradient
        torch.bmm(self.dzs.transpose(1, 2), self.inp, out=self.GW_square)
        torch.mm(dz[6].t(), self.cat6, out=self.GW6)
        torch.mm(dz[0].t(), x0, out=self.GW[0])
        chk(L.brl_bias_finalize_ex(di, 11, self._seg_scratch, self._seg_cols, self._seg_tiles, self._seg_db, s))
