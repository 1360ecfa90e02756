"""``src/update.py`` — the PPO-clip update.

The data it consumes is the time-major ``Transition`` buffer the HIP rollout kernels wrote; flattening is ``reshape(T*N, ...)`` ->
row ``t*N+n`` (G7, src/update.py:193-206).  Default paths (``brl_amd/fused_update.py``): ``FusedMinibatch`` for the DeepMind MLPs — the
step's big GEMMs stay with the library (hipBLASLt / rocBLAS through torch, committed tuned solutions) or ``brl_mlp_gemm``, everything
else is hand-written HIP (``csrc/ppo_heads.hpp``, ``csrc/ppo_update.hpp``) — and ``FusedFair`` for the FAIR network, eight steps per
hipGraph.  CPU tensors, a non-fp32 network and FAIR with an illegal-action norm term take the torch autograd path (``ppo_loss``).
Under ``torch.distributed`` the fp32 gradient (3 681 319 elements = 14.7 MB for the DeepMind MLP) crosses the ranks once per
minibatch — RCCL over xGMI on MI355X (backend "nccl", the collective a node of the step's hipGraph), gloo in the CPU tests: ONE flat
all-reduce and the replicated clip + Adam sweep by default; ``config["grad_allreduce"] = "sharded"``: reduce-scattered per layer
behind the backward pass, Adam on the rank's slices, parameters all-gathered under the next forward pass (bit-identical given
``per_layer_dw``; DESIGN section 7 for why it is not the default).  Each rank permutes its own shard (statistically equivalent to the
reference's global permutation, not bit-equal — SURVEY §8e caveat: set ``minibatch_size`` to 1024 / world for the reference's global
1024) and uses
``minibatch_size`` PER RANK.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from ._capture import quiet_gc
from .fused_update import FusedFair, FusedMinibatch, FusedStep   # noqa: F401  (the default paths of update_step; re-exported)
from .roll_out import Transition

_NEG = torch.finfo(torch.float32).min


def make_optimizer(config, params: torch.nn.Module):
    """ppo.py:186-211: Adam(eps=1e-5), optional linear anneal; global-norm clipping is applied inside
    ``update_step`` (optax.chain(clip_by_global_norm, adam))."""
    on_gpu = next(params.parameters()).is_cuda
    # fused: one kernel per step; capturable: the step can live inside a hipGraph (see GraphedMinibatch)
    opt = torch.optim.Adam(params.parameters(), lr=config["lr"], eps=1e-5, fused=True if on_gpu else None,
                           capturable=bool(on_gpu))
    sched = None
    if config.get("anneal_lr", False):
        per_update = config["num_minibatches"] * config["update_epochs"]
        sched = torch.optim.lr_scheduler.LambdaLR(
            opt, lambda count: 1.0 - (count // per_update) / config["num_updates"])  # ppo.py:186-192
    return {"opt": opt, "sched": sched}


def masked_log_softmax(logits, mask):
    return torch.log_softmax(torch.where(mask, logits, torch.full_like(logits, _NEG)), dim=-1)


def spectral_norm_nonneg(a: torch.Tensor, squarings: int = 8) -> torch.Tensor:
    """Largest singular value of a non-negative [B, A] matrix without an SVD (which costs more than the whole
    minibatch step and does not capture into a hipGraph): sqrt of the top eigenvalue of the A x A Gram matrix, found
    by repeated squaring (G^(2^k) collapses onto the Perron eigenvector; relative error ~ (l2/l1)^(2^(k+1))).
    Used for the LOGGED illegal-action norm when its coefficient is 0 (src/update.py:136-141); with a non-zero
    coefficient the differentiable ``torch.linalg.matrix_norm(ord=2)`` is used."""
    g = a.t() @ a
    tiny = torch.finfo(g.dtype).tiny
    m = g / g.diagonal().sum().clamp_min(tiny)
    for _ in range(squarings):
        m = m @ m
        m = m / m.diagonal().sum().clamp_min(tiny)
    v = m.sum(dim=1)
    lam = (v @ (g @ v)) / (v @ v).clamp_min(tiny)
    return lam.clamp_min(0).sqrt()


def ppo_loss(config, logits, value, batch: Transition, gae, targets):
    """``_loss_fn`` (src/update.py:90-167) on one minibatch; returns (total_loss, aux tuple)."""
    mask = batch.legal_action_mask
    if config.get("actor_illegal_action_mask", True):
        logp_all = masked_log_softmax(logits, mask)          # src/update.py:12-16
    else:
        logp_all = torch.log_softmax(logits, dim=-1)
    log_prob = logp_all.gather(1, batch.action.long()[:, None])[:, 0]
    if config.get("value_clipping", True):                   # src/update.py:48-60
        v_clipped = batch.value + (value - batch.value).clamp(-config["clip_eps"], config["clip_eps"])
        value_loss = 0.5 * torch.maximum((value - targets) ** 2, (v_clipped - targets) ** 2).mean()
    else:
        value_loss = 0.5 * ((value - targets) ** 2).mean()
    logratio = log_prob - batch.log_prob
    ratio = torch.exp(logratio)
    if config.get("reward_scaling", False):                  # src/update.py:31-44 (jnp std: ddof=0)
        gae = (gae - gae.mean()) / (gae.std(unbiased=False) + 1e-8)
    loss_actor = -torch.minimum(ratio * gae, ratio.clamp(1.0 - config["clip_eps"], 1.0 + config["clip_eps"]) * gae).mean()
    mlp = masked_log_softmax(logits, mask)
    p = mlp.exp()
    entropy = -(torch.where(p > 0, p * mlp, torch.zeros_like(p))).sum(-1).mean()   # distrax: 0 log 0 = 0
    # src/update.py:136-141: the L2 (spectral) norm of the illegal-action probabilities is always computed and
    # logged, whatever its coefficient; it joins the gradient only when the coefficient is non-zero
    coef = config.get("illegal_action_l2norm_coef", 0.0)
    if coef:
        illegal = torch.softmax(logits, dim=-1) * (~mask)
        illegal_loss = torch.linalg.matrix_norm(illegal, ord=2) / 2               # jnp.linalg.norm(2-D, ord=2)
    else:
        with torch.no_grad():
            illegal_loss = spectral_norm_nonneg(torch.softmax(logits, dim=-1) * (~mask)) / 2
    total = loss_actor + config["vf_coef"] * value_loss - config["ent_coef"] * entropy
    if coef:
        total = total + coef * illegal_loss
    with torch.no_grad():
        approx_kl = ((ratio - 1) - logratio).mean()
        clipfrac = ((ratio - 1.0).abs() > config["clip_eps"]).float().mean()
    return total, (value_loss.detach(), loss_actor.detach(), entropy.detach(), approx_kl, clipfrac, illegal_loss.detach())


def fused_loss_ok(config, logits) -> bool:
    """The one-launch HIP loss (``brl_ppo_loss``) covers everything except a non-zero illegal-action coefficient (its
    spectral norm then needs a gradient); CPU tensors take the torch path."""
    return bool(config.get("fused_loss", True)) and logits.is_cuda and not config.get("illegal_action_l2norm_coef", 0.0)


def ppo_loss_fused(config, logits, value, batch: Transition, gae, targets):
    """``_loss_fn`` (src/update.py:90-167) as ONE HIP launch that also returns d(total)/d(logits, value): torch then
    differentiates only the GEMMs (``torch.autograd.backward([logits, value], grads)``) instead of ~120 elementwise
    forward + backward launches per minibatch.  Returns (total, aux, (dlogits, dvalue)); total / aux are detached."""
    from . import _capi
    B = logits.shape[0]
    dev = logits.device
    logits_c = logits.detach()
    if logits_c.stride(1) != 1:
        logits_c = logits_c.contiguous()
    value_c = value.detach().contiguous()
    if config.get("reward_scaling", False):                  # src/update.py:31-44 (jnp std: ddof=0)
        gae = (gae - gae.mean()) / (gae.std(unbiased=False) + 1e-8)
    f32 = lambda x: x.to(torch.float32).contiguous()  # noqa: E731
    mask = batch.legal_action_mask.contiguous()
    action = batch.action.to(torch.int32).contiguous()
    dlogits = torch.empty((B, 38), dtype=torch.float32, device=dev)
    dvalue = torch.empty(B, dtype=torch.float32, device=dev)
    partials = torch.empty(((B + 3) // 4, 8), dtype=torch.float32, device=dev)
    illp = torch.empty((B, 38), dtype=torch.float32, device=dev)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    L = _capi.lib()
    old_v, old_lp, gae_c, tgt_c = f32(batch.value), f32(batch.log_prob), f32(gae), f32(targets)
    _capi.check(L.brl_ppo_loss(dev.index, logits_c.data_ptr(), logits_c.stride(0), value_c.data_ptr(), mask.data_ptr(),
                               action.data_ptr(), old_v.data_ptr(), old_lp.data_ptr(), gae_c.data_ptr(), tgt_c.data_ptr(), B,
                               float(config["clip_eps"]), float(config["vf_coef"]), float(config["ent_coef"]),
                               int(bool(config.get("actor_illegal_action_mask", True))),
                               int(bool(config.get("value_clipping", True))), dlogits.data_ptr(), dvalue.data_ptr(),
                               partials.data_ptr(), illp.data_ptr(), stream))
    gram = illp.t() @ illp                                   # 38 x 38; its top eigenvalue = (largest singular value)^2
    _capi.check(L.brl_ppo_stats(dev.index, partials.data_ptr(), B, gram.data_ptr(), float(config["vf_coef"]),
                                float(config["ent_coef"]), out.data_ptr(), stream))
    # out: total, value_loss, loss_actor, entropy, approx_kl, clipfrac, illegal-action norm / 2
    return out[0], tuple(out[1 + k] for k in range(6)), (dlogits, dvalue)


def loss_and_backward(config, logits, value, batch, gae, targets):
    """forward loss + gradients into ``.grad`` of whatever produced (logits, value); returns (total, aux) detached."""
    if fused_loss_ok(config, logits):
        total, aux, grads = ppo_loss_fused(config, logits, value, batch, gae, targets)
        torch.autograd.backward([logits, value], list(grads))
        return total, aux
    total, aux = ppo_loss(config, logits, value, batch, gae, targets)
    total.backward()
    return total.detach(), aux


def allreduce_gradients(params: torch.nn.Module):
    """One flat all-reduce (mean) of every gradient — 14.7 MB for the DeepMind MLP."""
    from .dist import distributed
    if not distributed():
        return
    grads = [p.grad for p in params.parameters() if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)


class GraphedMinibatch:
    """One PPO minibatch step (forward, loss, backward, global-norm clip, Adam) captured ONCE into a
    hipGraph and replayed per minibatch: the step is ~100 small kernels, host-launch-bound in eager mode
    (2.0 ms per 1024-sample minibatch on MI355X; the GEMMs themselves are ~0.25 ms).  Inputs are copied
    into static buffers before each replay.  Single-process only: with a process group the gradient
    all-reduce stays eager (see update_step)."""

    def __init__(self, config, actor_forward_pass, params, opt, mbs: int, device):
        self.cfg, self.fp, self.params, self.opt = config, actor_forward_pass, params, opt
        self.mbs = int(mbs)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)  # noqa: E731
        self.mb = Transition(z((mbs,), torch.bool), z((mbs,), torch.int32), z((mbs,), torch.float32),
                             z((mbs,), torch.float32), z((mbs,), torch.float32), z((mbs, 480), torch.bool),
                             z((mbs, 38), torch.bool))
        self.mb.legal_action_mask[:, 0] = True  # a valid dummy batch for the warm-up iterations
        self.gae, self.tgt = z((mbs,), torch.float32), z((mbs,), torch.float32)
        # Warm-up and capture run the REAL optimizer on a dummy batch: snapshot the parameters and the optimizer state
        # (moments, step counts — the graph may be built after eager steps or from a loaded optimizer) and put them
        # back IN PLACE afterwards, also when capture fails (the captured graph holds these tensors' addresses).
        saved_p = [p.detach().clone() for p in params.parameters()]
        saved_s = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                   for p, st in opt.state.items()}
        self.graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with quiet_gc(), torch.cuda.graph(graph):   # (_capture.py: no collector run while a stream captures)
                self.out = self._step()
            self.graph = graph
        finally:
            with torch.no_grad():
                for p, q in zip(params.parameters(), saved_p):
                    p.copy_(q)
                for p, st in opt.state.items():
                    old = saved_s.get(p)
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            if old is not None and k in old:
                                v.copy_(old[k])
                            else:
                                v.zero_()  # state created by the warm-up: zero moments / step 0 == a fresh Adam state
                        elif old is not None and k in old:
                            st[k] = old[k]

    def _step(self):
        logits, value = self.fp.apply(self.params, self.mb.obs.to(torch.float32))
        self.opt.zero_grad(set_to_none=False)
        total, aux = loss_and_backward(self.cfg, logits, value, self.mb, self.gae, self.tgt)
        if self.cfg.get("global_gradient_clipping", True):
            torch.nn.utils.clip_grad_norm_(self.params.parameters(), self.cfg["max_grad_norm"])
        self.opt.step()
        return total, torch.stack(aux)

    def run(self, mb: Transition, gae, tgt):
        for name in ("action", "value", "log_prob", "obs", "legal_action_mask"):  # what _loss_fn reads (not done / reward)
            getattr(self.mb, name).copy_(getattr(mb, name))
        self.gae.copy_(gae)
        self.tgt.copy_(tgt)
        self.graph.replay()
        return self.out[0].clone(), self.out[1].clone()


def make_update_step(config, actor_forward_pass, optimizer=None):
    """``make_update_step(config, actor_forward_pass, optimizer)`` (src/update.py:9); returns
    ``update_step(runner_state, traj_batch, advantages, targets) -> (runner_state, loss_info)`` (:74,242).
    ``runner_state[1]`` (opt_state) is the dict from ``make_optimizer``."""

    def update_step(runner_state, traj_batch: Transition, advantages, targets):
        params, opt_state, env_state, last_obs, terminated_count, rng = runner_state
        if opt_state is None:
            opt_state = optimizer if optimizer is not None else make_optimizer(config, params)
        opt, sched = opt_state["opt"], opt_state["sched"]
        T, N = traj_batch.action.shape
        batch_size = T * N
        mbs = int(config["minibatch_size"])
        if batch_size % mbs:
            raise ValueError("batch size must be a multiple of minibatch_size")  # src/update.py:190-192
        num_mb = batch_size // mbs
        flat = Transition(*[x.reshape((batch_size,) + x.shape[2:]) for x in traj_batch])  # G7
        adv_f, tgt_f = advantages.reshape(batch_size), targets.reshape(batch_size)
        gen = torch.Generator(device=adv_f.device)
        gen.manual_seed(int(rng) & 0xFFFFFFFF)   # (the same mod-2^32 convention as the action-draw counter)
        totals, auxes = [], []
        from .dist import distributed
        multi = distributed()      # (under BRL_FORCE_DIST=1 also at world 1: FusedStep then runs the multi-rank program with one peer)
        graphed = None
        fused = None
        fused_cls = next((c for c in (FusedMinibatch, FusedFair) if adv_f.is_cuda and c.supports(config, params)), None)
        want_fused = fused_cls is not None
        need_log = int(config["update_epochs"]) * num_mb
        if config.get("graph_update", True) and adv_f.is_cuda and (want_fused or (sched is None and not multi)):
            graphed = opt_state.get("graphed")
            world = dist.get_world_size() if multi else 1
            # (False = an earlier capture failed, on this or — under a process group — any rank: stay eager, do not retry)
            if graphed is None or (graphed is not False and (
                    graphed.params is not params or graphed.mbs != mbs or isinstance(graphed, FusedStep) != want_fused
                    or getattr(graphed, "world", 1) != world
                    or (isinstance(graphed, FusedStep) and graphed._log_cap < need_log))):   # e.g. minibatch 512: 5120 steps
                prev = graphed
                try:
                    graphed = fused_cls(config, params, opt, mbs, adv_f.device, world, log_capacity=need_log) if want_fused \
                        else GraphedMinibatch(config, actor_forward_pass, params, opt, mbs, adv_f.device)
                except Exception as e:  # capture is an optimisation, never a requirement — but never a SILENT ~1.7x cliff
                    graphed = False
                    opt_state["graph_error"] = repr(e)
                    import warnings
                    warnings.warn(f"brl_amd.update: hipGraph capture of the minibatch step failed ({e!r}); this update runs "
                                  f"on the eager path (~1.7x slower).  opt_state['graph_error'] holds the error.", RuntimeWarning)
                if multi:
                    # every rank must issue the SAME collective sequence: a rank whose capture failed would run the eager path's
                    # one all-reduce per minibatch while the others run FusedMinibatch's collectives, and RCCL would hang
                    # until its timeout.  FusedStep's constructor already ends the same way on every rank (its phases agree,
                    # fused_update.py); this MIN over ranks covers what can differ before it is entered.  All fused, or all eager.
                    ok = torch.tensor([1 if isinstance(graphed, FusedStep) else 0], dtype=torch.int32, device=adv_f.device)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0 and isinstance(graphed, FusedStep):
                        opt_state["graph_error"] = "another rank failed to capture the minibatch step: all ranks run eager"
                        import warnings
                        warnings.warn("brl_amd.update: " + opt_state["graph_error"], RuntimeWarning)
                        graphed = False
                    if graphed is False and isinstance(prev, FusedStep) and prev.moments_partial:
                        # the step being replaced ran "sharded": Adam's moments are current on each rank's own slices only, and the
                        # eager path's replicated Adam would diverge across ranks.  Every rank is here (the agreement above):
                        prev.gather_optimizer_state()
                opt_state["graphed"] = graphed
            if isinstance(graphed, FusedStep):
                fused = graphed
        if fused is not None:
            # minibatches are gathered straight from the un-shuffled buffer by index: no per-epoch take(), no copies
            perms = [torch.randperm(batch_size, device=adv_f.device, generator=gen)            # src/update.py:193
                     for _ in range(int(config["update_epochs"]))]
            fused.begin_update(flat, adv_f, tgt_f, perms)
            fused.run_steps(int(config["update_epochs"]) * num_mb)
            log = fused.end_update()
            if sched is not None:   # ppo.py:186-192: the rate changes between updates only (count // per_update)
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")   # "lr_scheduler.step() before optimizer.step()": Adam ran in HIP
                    for _ in range(int(config["update_epochs"]) * num_mb):
                        sched.step()
            log = log.clone().reshape(int(config["update_epochs"]), num_mb, 8)
            loss_info = (log[..., 0], tuple(log[..., 1 + i] for i in range(6)))
            return (params, opt_state, env_state, last_obs, terminated_count, int(rng) + 1), loss_info
        for _ in range(int(config["update_epochs"])):
            perm = torch.randperm(batch_size, device=adv_f.device, generator=gen)   # src/update.py:193
            # shuffled_batch = take(x, permutation) once per epoch, minibatches are then contiguous views
            # (src/update.py:198-206) — one gather of the whole buffer instead of 7 per minibatch
            shuf = Transition(*[x.index_select(0, perm) for x in flat])
            adv_s, tgt_s = adv_f.index_select(0, perm), tgt_f.index_select(0, perm)
            row_t, row_a = [], []
            for m in range(num_mb):
                sl = slice(m * mbs, (m + 1) * mbs)
                mb = Transition(*[x[sl] for x in shuf])
                if graphed:
                    total_d, aux_d = graphed.run(mb, adv_s[sl], tgt_s[sl])
                    row_t.append(total_d)
                    row_a.append(aux_d)
                    continue
                logits, value = actor_forward_pass.apply(params, mb.obs.to(torch.float32))   # G5
                opt.zero_grad(set_to_none=True)
                total, aux = loss_and_backward(config, logits, value, mb, adv_s[sl], tgt_s[sl])
                allreduce_gradients(params)
                if config.get("global_gradient_clipping", True):
                    torch.nn.utils.clip_grad_norm_(params.parameters(), config["max_grad_norm"])
                opt.step()
                if sched is not None:
                    sched.step()
                row_t.append(total)
                row_a.append(torch.stack(aux))
            totals.append(torch.stack(row_t))
            auxes.append(torch.stack(row_a))
        loss_info = (torch.stack(totals), tuple(torch.stack(auxes)[..., i] for i in range(6)))
        return (params, opt_state, env_state, last_obs, terminated_count, int(rng) + 1), loss_info

    return update_step
