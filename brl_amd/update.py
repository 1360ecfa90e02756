"""``src/update.py`` — the PPO-clip update.

The data it consumes is the time-major ``Transition`` buffer the HIP rollout kernels wrote; flattening is ``reshape(T*N, ...)`` ->
row ``t*N+n`` (G7, src/update.py:193-206).  Default path for the DeepMind MLPs: ``FusedMinibatch`` — the step's big GEMMs stay
with the library (hipBLASLt / rocBLAS through torch, committed tuned solutions), everything else is hand-written HIP
(``csrc/ppo_heads.hpp``, ``csrc/ppo_update.hpp``), eight steps per hipGraph.  Other architectures (FAIR) and CPU tensors take the
torch autograd path (``ppo_loss``).  Under ``torch.distributed`` the fp32 gradient (3 681 319 elements = 14.7 MB for the DeepMind
MLP) is all-reduced per minibatch — RCCL over xGMI on MI355X (backend "nccl"), gloo in the CPU tests: bucketed per layer behind
graph segments in the fused path, one flat all-reduce otherwise.  Each rank permutes its own shard (statistically equivalent to the
reference's global permutation, not bit-equal — SURVEY §8e caveat) and uses ``minibatch_size`` PER RANK.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from ._capture import quiet_gc
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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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


class FusedMinibatch:
    """One PPO minibatch step of a "DeepMind" MLP with NOTHING but its big GEMMs left to torch: the minibatch
    gather (``brl_mb_gather_dev``: device-resident arguments, first node of the captured step), the 39-column head + ``_loss_fn``
    + its gradients (``brl_ppo_heads_loss``), the head's backward (``brl_ppo_heads_bwd``), the activation derivative + bias tile
    sums of the layers below (``brl_act_bwd_colsum``), every sum of partials (``brl_bias_finalize_ex``), global-norm clipping +
    Adam on flat parameter / gradient / moment buffers (``brl_adam_clip``) are single HIP launches; the backward pass is written
    out (no autograd: dz chain, then the hidden layers' weight gradients as ONE batched product); EIGHT steps are one hipGraph;
    the logged statistics are formed once per update from per-step sums (``brl_ppo_stats_rows``).

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

    def __init__(self, config, params, opt, mbs: int, device, world: int = 1, log_capacity: int = 0, collective=None):
        from . import _capi
        self.cfg, self.params, self.opt, self.mbs, self.dev = config, params, opt, int(mbs), device
        # world > 1, config["grad_allreduce"]:
        #   "flat" (default): the step is TWO graphs — forward + the single-rank backward chain + the sums of partials | clip +
        #       Adam — with ONE all-reduce of the flat 14.7 MB gradient between them (ppo.py's pmean); the collective is exposed
        #       (nothing overlaps it), the compute side keeps every single-rank fusion but the sums-inside-the-norm launch;
        #   "bucketed": one graph per all-reduce bucket (see the capture below): collectives of <= 4.2 MB issued asynchronously
        #       behind their segment, overlapping the rest of the backward pass (RCCL over xGMI on MI355X).  Opt-in until it has
        #       run over RCCL on >= 2 GPUs (tests/test_gpu_parity.py::*_rccl; every box of this build so far had one GPU).
        # `collective`: None = torch.distributed.all_reduce; a callable (tensor, async_op) -> work-or-None replaces it (bench.py's
        # one-GPU rehearsal of the multi-rank step: a no-op with the same stream ordering).
        self.world = int(world)
        self.allreduce_mode = str(config.get("grad_allreduce", "flat")) if self.world > 1 else "none"
        if self.allreduce_mode not in ("none", "flat", "bucketed"):
            raise ValueError("config['grad_allreduce'] must be 'flat' or 'bucketed'")
        self.single_chain = self.allreduce_mode != "bucketed"   # the dz chain first, then the batched weight gradients
        self._collective = collective
        if config.get("tuned_gemm", True):   # committed TunableOp solutions for the step's GEMM shapes (brl_amd/tuned): lookups only
            from . import tuned
            tuned.enable()
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=device)  # noqa: E731
        body = list(params.body)
        # flat layout: every hidden layer's W, then actor.weight | critic.weight (one [39,1024] matrix), then the biases
        # (hidden layers, actor | critic): a layer's weight gradient is one contiguous slice (all-reduce buckets), all bias
        # gradients — finished by one launch at the end of the backward pass — another
        plist = [lin.weight for lin in body] + [params.actor.weight, params.critic.weight] \
            + [lin.bias for lin in body] + [params.actor.bias, params.critic.bias]
        assert len(plist) == len(list(params.parameters()))
        sizes = [q.numel() for q in plist]
        self.n = n = (sum(sizes) + 3) // 4 * 4  # brl_adam_clip works on float4s: zero padding at the end
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
        H = body[0].weight.shape[0]
        K = params.actor.weight.shape[0] + 1
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
        self.G_bias = self.G[views[body[0].bias].start:ba.start + K]   # every bias gradient, contiguous
        B = self.mbs
        # Activations and the gradients w.r.t. the pre-activations live in STACKED static buffers (out= costs the fused
        # bias + ReLU GEMM nothing: scripts/fwd_probe.py): segments of a multi-rank step hand them to each other, and the
        # single-rank step forms the hidden layers' weight gradients as ONE batched product at the end of the backward chain.
        nl = len(body)
        self.hs = f(nl, B, H)                       # h_l = act(h_{l-1} W_l^T + b_l)
        self.dzs = f(nl, B, H)                      # d(loss) / d(pre-activation of layer l)
        self.h = [self.hs[l] for l in range(nl)]
        self.dhb = [self.dzs[l] for l in range(nl)]   # (indexed by layer)
        # W_1 .. W_{nl-1} are consecutive [H, H] blocks of the flat buffer: their gradients as one [nl - 1, H, H] tensor
        w1 = views[body[1].weight] if nl > 1 else None
        self.GW_hidden = self.G[w1.start:w1.start + (nl - 1) * H * H].view(nl - 1, H, H) if nl > 1 else None
        self.x0 = f(B, 480)
        self.mask = torch.zeros((B, 38), dtype=torch.uint8, device=device)
        self.mask[:, 0] = 1  # a valid dummy batch for the warm-up iterations
        self.action = torch.zeros(B, dtype=torch.int32, device=device)
        self.old_v, self.old_lp, self.adv, self.tgt = f(B), f(B), f(B), f(B)
        self.dheads = f(B, K)
        self.ill_coef = float(config.get("illegal_action_l2norm_coef", 0.0) or 0.0)
        self.heads = f(B, K) if self.ill_coef else None      # the gradient of the illegal-action norm re-reads the logits
        self.head_ksplit = max(1, min(4, H // 256))          # K ranges of the heads product (brl_ppo_heads_loss_split)
        # config["fuse_heads_fwd"]: the LAST hidden layer's forward launch (own kernel) leaves the heads' partial products, one
        # per 64-column tile (brl_mlp_gemm_fwd_heads): no k_heads_product launch on the chain
        # Built, checked, NOT the default: the fused launch takes 23.7-24.4 us in the step where the library's forward launch +
        # k_heads_product take 19.9 + 5.1 (the epilogue — 24 MFMAs per wave, an LDS exchange between the two column halves, the
        # part's stores — is not hidden behind anything): 0.2295-0.2307 vs 0.2300-0.2309 ms per step.
        self.fuse_heads_fwd = (bool(config.get("fuse_heads_fwd", False)) and bool(config.get("own_gemm", True)) and self.single_chain
                               and B % 4 == 0 and H % 4 == 0 and (H + 63) // 64 <= 32 and K == 39 and len(body) > 1)
        # (64 x 32 tiles — two workgroups per CU, the faster form in the step — where that gives <= 32 parts)
        self.head_nparts = ((H + 31) // 32 if (H + 31) // 32 <= 32 else (H + 63) // 64) if self.fuse_heads_fwd else 0
        self.head_parts = f(max(self.head_ksplit, self.head_nparts), B, K)
        self.vec = f(40) if self.ill_coef else None          # v1 [38], sigma_1 of the step's illegal-action matrix
        self.H, self.K = H, K
        self.act = 0 if params.act is torch.relu else 1
        groups = (B + 15) // 16                        # 16-row tiles of the bias-gradient column sums
        self.groups = groups
        self.lgroups = (B + 3) // 4                    # 4-sample groups of brl_ppo_heads_loss (statistics / Gram partials)
        self.partials = f(self.lgroups, 8)
        self.gram_partials = f(self.lgroups, 38 * 38)
        self.out = f(8)
        self.scratch = f(8192)   # norm partials: 1024 blocks + the finalize blocks that ride in the norm launch (single rank)
        self.nsplit = (B + 63) // 64                   # batch splits of the head's weight / bias gradient (brl_ppo_heads_bwd)
        self.dwh_partials = f(self.nsplit, K * H)
        self.dbh_partials = f(self.nsplit, K)
        # the step's 1024^3-class products on this library's own fp32 MFMA kernel (brl_mlp_gemm, csrc/mlp_gemm.hpp) where its
        # fused epilogue removes a launch: hidden layers' forward (bias + activation inside), dh = dz W with the activation
        # derivative and the bias-gradient tile sums inside (replaces torch.mm + brl_act_bwd_colsum).  Layer 0 (K = 480) and the
        # weight gradients (one batched library product) stay with the library.  config["own_gemm"]: True / False.
        self.own_gemm = bool(config.get("own_gemm", True)) and self.single_chain and B % 4 == 0 and H % 4 == 0 and nl > 1
        # (the forward layers on the own kernel: opt-in — in the step it takes 19.8-20.1 us per layer where the tuned library
        #  kernel with the same epilogue takes 19.0-19.6; profiles/r04/r04_experiments.txt)
        self.own_fwd = self.own_gemm and bool(config.get("own_gemm_fwd", False))
        # The Adam sweep off the dependency chain (config["adam_ride"]): the step's own clip + Adam launch updates only what the
        # next forward pass needs at once — layers 0 and 1 and every bias; the rest is owed (self.pending) and paid by extra
        # workgroups of the NEXT step's forward launches: layer l's launch (l = 1 .. nl - 2) updates layer l + 1's weights, the
        # last of them also the head's (HBM-bound beside MFMA-bound); `_flush_adam` pays what the last step of a run owes.
        # Built, bit-compatible, measured and NOT the default: the riders slow their host launches by more than the chain's launch
        # shrinks (+ 4.5 and + 6.3 us on two forward launches for - 7.5 us of the Adam launch: 0.2359 vs 0.2337 ms per step).
        self.adam_ride = self.own_fwd and self.world == 1 and nl >= 3 and bool(config.get("adam_ride", False))
        self.pending = torch.zeros(1, dtype=torch.int32, device=device)
        if self.adam_ride:
            bias0 = views[body[0].bias].start
            self.defer = (views[body[2].weight].start, bias0)
            self.ride = {l: (views[body[l + 1].weight].start,
                             views[body[l + 2].weight].start if l + 2 < nl else bias0) for l in range(1, nl - 1)}
        # both backward products of a hidden layer in ONE launch (brl_mlp_gemm_bwd_pair): config["bwd_pair"].  Built, checked,
        # NOT the default: the pair takes 37.7-38.2 us in the step, no less than its two launches (own dh 21.3 + a third of the
        # batched library product 15.8): 0.2312 vs 0.2295 ms per step (profiles/r04/r04_experiments.txt §6)
        self.bwd_pair = self.own_gemm and bool(config.get("bwd_pair", False))
        groups64 = (B + 63) // 64                      # 64-row tiles of brl_mlp_gemm's column sums
        self.tile_rows = [64 if (self.own_gemm and l < nl - 1) else 16 for l in range(nl)]
        self.tile_sums = [f(groups * H) for _ in body]   # per-layer partial column sums (bias gradients)
        import ctypes as C
        # one launch finishes every sum of partials: the hidden layers' bias gradients, the head's bias and weight gradients
        nseg = len(body) + 2
        self._seg_scratch = (C.c_void_p * nseg)(*([t.data_ptr() for t in self.tile_sums]
                                                  + [self.dbh_partials.data_ptr(), self.dwh_partials.data_ptr()]))
        self._seg_cols = (C.c_int64 * nseg)(*([H] * len(body) + [K, K * H]))
        self._seg_tiles = (C.c_int64 * nseg)(*([groups64 if r == 64 else groups for r in self.tile_rows] + [self.nsplit, self.nsplit]))
        self._seg_db = (C.c_void_p * nseg)(*([g.data_ptr() for g in self.Gb] + [self.Gbh.data_ptr(), self.GWh.data_ptr()]))
        self._nseg = nseg
        self.npartials = 1024 + sum((int(c) + 63) // 64 for c in self._seg_cols)   # k_adam_norm_fin's partial sums
        # The logged statistics (src/update.py:136-167) are NOT formed step by step: every step leaves its sums — 8 floats and
        # the 38 x 38 Gram matrix of the illegal-action probabilities, reduced by spare workgroups of the head-backward
        # launch — in row mb_index of these buffers, and ONE launch at the end of the update turns all rows into log rows
        # (a per-step statistics kernel on a parallel graph branch cost ~20 us of fork / join per step, 7 % of the update).
        # (rows = minibatch steps of one update_step call; update_step rebuilds this object when an update needs more)
        self._log_cap = max(int(config.get("update_log_capacity", 4096)), int(log_capacity))
        self.log = f(self._log_cap, 8)
        self.stat_sums = f(self._log_cap, 8)
        self.gram_sums = f(self._log_cap, 38 * 38)
        self.norm = f(1)
        self.mb_index = torch.zeros(1, dtype=torch.int32, device=device)  # minibatch step within the current update
        self.perm = None  # static int64 [epochs * T*N]: every epoch's permutation, filled by begin_update
        self.lib, self.capi = _capi.lib(), _capi
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
        # warm-up and capture run real steps on the dummy batch: put parameters, moments and counters back afterwards
        saved = [t.clone() for t in (self.P, self.M, self.V, self.step, self.mb_index)]
        self.graph = self.segs = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(3):
                    self._step()
            torch.cuda.current_stream().wait_stream(side)
            nl = len(self.W)
            # under a process group the NCCL watchdog thread queries events while this thread captures: "global" capture mode
            # would turn that into a capture error
            gkw = {"capture_error_mode": "thread_local"} if (dist.is_available() and dist.is_initialized()) else {}
            with quiet_gc():   # (_capture.py: no collector run while a stream captures)
                if self.world == 1:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, **gkw), torch.no_grad():
                        self._step()
                    self.graph = graph
                    # ... and the same step K times in ONE graph: a replay boundary costs ~5 us (graph launch behind the last
                    # kernel), the step ~0.3 ms; mb_index lives in device memory, so the K copies walk K minibatches
                    self.multi = int(config.get("update_graph_steps", 8))
                    self.graph_multi = None
                    if self.multi > 1:
                        gm = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gm, **gkw), torch.no_grad():
                            for _ in range(self.multi):
                                self._step()
                        self.graph_multi = gm
                elif self.allreduce_mode == "flat":
                    pool = torch.cuda.graph_pool_handle()
                    self.segs = []
                    # three graphs: [gradients] | [clip + Adam of step i, then the gradients of step i + 1] | [clip + Adam]: a run of n
                    # steps is first, (all-reduce, middle) x (n - 1), all-reduce, last — ONE replay and one collective per step
                    # (a replay costs ~20 us of host -> device latency; as gradient graph + Adam graph the step took 0.282 ms, see
                    # bench.py's config4_rehearsal)
                    for seg in (self._grads, lambda: (self._opt(), self._grads()), self._opt):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=pool, **gkw), torch.no_grad():
                            seg()
                        self.segs.append(g)
                    self.buckets = [self.G]
                    self.graph = self.segs[0]
                else:
                    # one graph per all-reduce bucket: forward + loss + head backward | each hidden layer's backward | bias
                    # gradients | clip + Adam; the bucket's all-reduce is issued (async) behind its graph and overlaps with
                    # the graphs that follow (RCCL over xGMI: five collectives of <= 4.2 MB beside ~0.2 ms of backward GEMMs)
                    pool = torch.cuda.graph_pool_handle()
                    self.segs = []
                    for seg in [lambda: self._seg_head()] + [(lambda l=l: self._seg_layer(l)) for l in range(nl - 1, -1, -1)] \
                            + [lambda: self._seg_fin(), lambda: self._opt()]:
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=pool, **gkw), torch.no_grad():
                            seg()
                        self.segs.append(g)
                    # (the head's weight gradient is finished by _seg_fin, beside the bias gradients: contiguous in the flat buffer)
                    self.buckets = [None] + [self.GW[l] for l in range(nl - 1, -1, -1)] + [self.G[wa.start:ba.start + K]]
                    self.graph = self.segs[0]
        finally:
            with torch.no_grad():
                for t, q in zip((self.P, self.M, self.V, self.step, self.mb_index), saved):
                    t.copy_(q)
                self.pending.zero_()   # (nothing is owed: the warm-up's deferred sweeps were discarded with its parameters)

    def _bind_gather(self, fl: Transition, adv, tgt, perm, first=True):
        """binds the step's gather to a trajectory / permutation; ``first``: also gathers minibatch *mb_index now (every later
        minibatch is gathered by the Adam launch of the step before it)"""
        import ctypes as C
        tp = self.capi.TransitionPtrs()
        for name in self.capi.TransitionPtrs._names:
            t = getattr(fl, name)
            setattr(tp, name, (t.view(torch.uint8) if t.dtype == torch.bool else t).data_ptr())
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.capi.check(self.lib.brl_mb_gather_bind(di, C.byref(tp), adv.data_ptr(), tgt.data_ptr(), perm.data_ptr(),
                                                    self.mb_index.data_ptr(), self.mbs, self.x0.data_ptr(), self.mask.data_ptr(),
                                                    self.action.data_ptr(), self.old_v.data_ptr(), self.old_lp.data_ptr(),
                                                    self.adv.data_ptr(), self.tgt.data_ptr(), perm.numel() // self.mbs,
                                                    self.gargs.data_ptr(), torch.cuda.current_stream().cuda_stream))
        if first:
            self.capi.check(self.lib.brl_mb_gather_dev(di, self.gargs.data_ptr(), self.mbs, torch.cuda.current_stream().cuda_stream))

    def _step(self):
        """one minibatch step on the current stream (what the graphs capture)"""
        if self.world == 1:
            self._seg_head()
            self._backward_chain()
            self._fin_opt()       # sums of partials inside the norm launch, clip + Adam (+ the next gather)
        elif self.allreduce_mode == "flat":
            self._grads()
            self._opt()
        else:
            self._seg_head()
            for l in range(len(self.W) - 1, -1, -1):
                self._seg_layer(l)
            self._seg_fin()
            self._opt()

    def _grads(self):
        """multi-rank "flat" form, first graph: everything that produces this rank's gradient — the single-rank chain with the
        sums of partials as a launch of their own (they must exist before the all-reduce)"""
        self._seg_head()
        self._backward_chain()
        self._seg_fin()

    def _seg_head(self):
        """forward, heads + loss + output gradients (one launch), the logged statistics (parallel branch), backward of the
        merged head down to the top hidden layer's pre-activation (one launch)"""
        L, chk, B = self.lib, self.capi.check, self.mbs
        s = torch.cuda.current_stream().cuda_stream
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        cfg = self.cfg
        x = self.x0   # minibatch *mb_index of the bound trajectory: gathered by the previous step's Adam launch (or by the bind)
        for l, (W, b) in enumerate(zip(self.W, self.b)):          # forward: bias + activation
            if self.fuse_heads_fwd and l == len(self.W) - 1:      # last hidden layer + the heads' partial products
                chk(L.brl_mlp_gemm_fwd_heads(di, x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0), self.h[l].data_ptr(),
                                             self.h[l].stride(0), B, W.shape[0], W.shape[1], self.act, b.data_ptr(),
                                             self.Wh.data_ptr(), self.Wh.stride(0), self.head_parts.data_ptr(),
                                             self.head_nparts, s))
                x = self.h[l]
                continue
            if self.adam_ride and l in self.ride:                 # + the owed Adam sweep of the next layer's weights
                lo, hi = self.ride[l]
                chk(L.brl_mlp_gemm_adam(di, x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0), self.h[l].data_ptr(),
                                        self.h[l].stride(0), B, W.shape[0], W.shape[1], self.act, b.data_ptr(), self.P.data_ptr(),
                                        self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), lo, hi, self.scratch.data_ptr(),
                                        self.npartials, self.step.data_ptr(), self.lr, self.lr_dev.data_ptr(), float(self.b1),
                                        float(self.b2), self.eps, self.max_norm, 1.0, self.pending.data_ptr(), s))
                x = self.h[l]
                continue
            if self.own_fwd and l > 0:                            # own kernel: bias + activation in its epilogue
                chk(L.brl_mlp_gemm(di, 0, 1, x.data_ptr(), x.stride(0), W.data_ptr(), W.stride(0), self.h[l].data_ptr(),
                                   self.h[l].stride(0), B, W.shape[0], W.shape[1], self.act, b.data_ptr(), None, 0, None, None, s))
                x = self.h[l]
                continue
            if self.act == 0:                                     # ReLU in the GEMM epilogue
                x = torch._addmm_activation(b, x, W.t(), use_gelu=False, out=self.h[l])
            else:
                x = torch.addmm(b, x, W.t(), out=self.h[l]).tanh_()
        if self.fuse_heads_fwd:   # (one launch: the loss on bias + the parts the last layer's launch left)
            chk(L.brl_ppo_heads_loss_parts(di, self.bh.data_ptr(), self.head_parts.data_ptr(), self.head_nparts,
                                           self.mask.data_ptr(), self.action.data_ptr(), self.old_v.data_ptr(), self.old_lp.data_ptr(),
                                           self.adv.data_ptr(), self.tgt.data_ptr(), B, float(cfg["clip_eps"]), float(cfg["vf_coef"]),
                                           float(cfg["ent_coef"]), int(bool(cfg.get("actor_illegal_action_mask", True))),
                                           int(bool(cfg.get("value_clipping", True))), int(bool(cfg.get("reward_scaling", False))),
                                           self.heads.data_ptr() if self.ill_coef else None, self.dheads.data_ptr(),
                                           self.partials.data_ptr(), self.gram_partials.data_ptr(), s))
        else:
          # (two launches: the heads product split over K across workgroups, then the loss on bias + its parts)
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
        # backward of the head, written out: dW_h / db_h partials per batch split, dz of the top hidden layer (activation
        # derivative applied) and its bias-gradient tile sums
        nl = len(self.W)
        top = nl - 1
        # single rank, more than one hidden layer: the head's weight-gradient role (not on the backward chain) rides with the first
        # activation-derivative launch of _backward_chain (brl_act_bwd_colsum_heads_dw); here only the activation-gradient role
        self.dw_deferred = self.single_chain and nl > 1   # (it rides with the first launch of the dz chain below the top layer)
        chk(L.brl_ppo_heads_bwd(di, self.dheads.data_ptr(), x.data_ptr(), x.stride(0), self.Wh.data_ptr(), B, self.H, self.act,
                                self.nsplit, None if self.dw_deferred else self.dwh_partials.data_ptr(),
                                None if self.dw_deferred else self.dbh_partials.data_ptr(),
                                self.dhb[top].data_ptr(), self.tile_sums[nl - 1].data_ptr(), self.partials.data_ptr(),
                                self.gram_partials.data_ptr(), self.lgroups, self.mb_index.data_ptr(), self.stat_sums.data_ptr(),
                                self.gram_sums.data_ptr(), s))

    def _seg_layer(self, l):
        """backward of hidden layer l (multi-rank form: one segment per all-reduce bucket): dz_l (below the top layer: dh -> dz
        in place + tile sums), dW_l, dh of the layer below"""
        L, chk, B = self.lib, self.capi.check, self.mbs
        s = torch.cuda.current_stream().cuda_stream
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        dz = self.dzs[l]
        if l != len(self.W) - 1:   # (the top layer's activation derivative and tile sums came with brl_ppo_heads_bwd)
            chk(L.brl_act_bwd_colsum(di, dz.data_ptr(), self.h[l].data_ptr(), B, self.H, self.H, self.act,
                                     self.tile_sums[l].data_ptr(), s))
        torch.mm(dz.t(), self.h[l - 1] if l > 0 else self.x0, out=self.GW[l])
        if l > 0:
            torch.mm(dz, self.W[l], out=self.dzs[l - 1])

    def _backward_chain(self):
        """single-rank form: the dh chain first (dz_l for every layer), then the weight gradients — layers 1.. as ONE batched
        product (three 1024^3 products: 58 us instead of 66, scripts/bmm_probe.py), layer 0 (K = 480 columns) beside it"""
        L, chk, B = self.lib, self.capi.check, self.mbs
        s = torch.cuda.current_stream().cuda_stream
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        nl = len(self.W)
        if self.bwd_pair:
            # per hidden layer ONE launch: dz_{l-1} (activation derivative + bias tile sums inside) AND dW_l = dz_l^T h_{l-1}; the
            # first also hosts the head's dW role.  Then dW_0 (K = 480 columns) with the library.
            for l in range(nl - 1, 0, -1):
                first = l == nl - 1 and self.dw_deferred
                top = self.h[nl - 1]
                chk(L.brl_mlp_gemm_bwd_pair(
                    di, self.dzs[l].data_ptr(), self.H, self.W[l].data_ptr(), self.W[l].stride(0), self.h[l - 1].data_ptr(), self.H,
                    self.dzs[l - 1].data_ptr(), self.H, self.GW[l].data_ptr(), self.GW[l].stride(0), B, self.W[l].shape[0],
                    self.W[l].shape[1], self.act, self.tile_sums[l - 1].data_ptr(), None,
                    self.dheads.data_ptr() if first else None, top.data_ptr(), top.stride(0), self.H, self.nsplit,
                    self.dwh_partials.data_ptr(), self.dbh_partials.data_ptr(), self.partials.data_ptr(), self.gram_partials.data_ptr(),
                    self.lgroups, self.mb_index.data_ptr(), self.stat_sums.data_ptr(), self.gram_sums.data_ptr(), s))
            torch.mm(self.dzs[0].t(), self.x0, out=self.GW[0])
            return
        for l in range(nl - 1, 0, -1):
            if self.own_gemm and l == nl - 1 and self.dw_deferred:   # + the head's dW_h / db_h partials and the step's statistics sums
                top = self.h[nl - 1]
                chk(L.brl_mlp_gemm_dh_heads_dw(di, self.dzs[l].data_ptr(), self.H, self.W[l].data_ptr(), self.W[l].stride(0),
                                               self.dzs[l - 1].data_ptr(), self.H, B, self.W[l].shape[1], self.W[l].shape[0], self.act,
                                               self.h[l - 1].data_ptr(), self.H, self.tile_sums[l - 1].data_ptr(),
                                               self.dheads.data_ptr(), top.data_ptr(), top.stride(0), B, self.H, self.nsplit,
                                               self.dwh_partials.data_ptr(), self.dbh_partials.data_ptr(), self.partials.data_ptr(),
                                               self.gram_partials.data_ptr(), self.lgroups, self.mb_index.data_ptr(),
                                               self.stat_sums.data_ptr(), self.gram_sums.data_ptr(), s))
                continue
            if self.own_gemm:   # dz_{l-1} = (dz_l W_l) * act'(h_{l-1}) + the 64-row tile sums of db_{l-1}: ONE launch
                chk(L.brl_mlp_gemm(di, 1, 2, self.dzs[l].data_ptr(), self.H, self.W[l].data_ptr(), self.W[l].stride(0),
                                   self.dzs[l - 1].data_ptr(), self.H, B, self.W[l].shape[1], self.W[l].shape[0], self.act, None,
                                   self.h[l - 1].data_ptr(), self.H, self.tile_sums[l - 1].data_ptr(), None, s))
                continue
            torch.mm(self.dzs[l], self.W[l], out=self.dzs[l - 1])
            if l == nl - 1 and self.dw_deferred:   # + the head's dW_h / db_h partials and the step's statistics sums
                top = self.h[nl - 1]
                chk(L.brl_act_bwd_colsum_heads_dw(di, self.dzs[l - 1].data_ptr(), self.h[l - 1].data_ptr(), B, self.H, self.H, self.act,
                                                  self.tile_sums[l - 1].data_ptr(), self.dheads.data_ptr(), top.data_ptr(),
                                                  top.stride(0), B, self.H, self.nsplit, self.dwh_partials.data_ptr(),
                                                  self.dbh_partials.data_ptr(), self.partials.data_ptr(),
                                                  self.gram_partials.data_ptr(), self.lgroups, self.mb_index.data_ptr(),
                                                  self.stat_sums.data_ptr(), self.gram_sums.data_ptr(), s))
                continue
            chk(L.brl_act_bwd_colsum(di, self.dzs[l - 1].data_ptr(), self.h[l - 1].data_ptr(), B, self.H, self.H, self.act,
                                     self.tile_sums[l - 1].data_ptr(), s))
        if nl > 1:
            torch.bmm(self.dzs[1:].transpose(1, 2), self.hs[:nl - 1], out=self.GW_hidden)
        torch.mm(self.dzs[0].t(), self.x0, out=self.GW[0])

    def _seg_fin(self):
        s = torch.cuda.current_stream().cuda_stream
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.capi.check(self.lib.brl_bias_finalize_ex(di, self._nseg, self._seg_scratch, self._seg_cols, self._seg_tiles,
                                                      self._seg_db, s))

    def _fin_opt(self):
        """single rank: every sum of partials is finished by extra workgroups of the norm launch (brl_adam_clip_fin_gather)"""
        s = torch.cuda.current_stream().cuda_stream
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        if self.adam_ride:
            self.capi.check(self.lib.brl_adam_clip_fin_gather_defer(
                di, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.n, self.step.data_ptr(), self.lr,
                self.lr_dev.data_ptr(), float(self.b1), float(self.b2), self.eps, self.max_norm, self.scratch.data_ptr(),
                self.scratch.numel(), self.mb_index.data_ptr(), self.norm.data_ptr(), self.gargs.data_ptr(), self.mbs, self._nseg,
                self._seg_scratch, self._seg_cols, self._seg_tiles, self._seg_db, self.defer[0], self.defer[1],
                self.pending.data_ptr(), s))
            return
        self.capi.check(self.lib.brl_adam_clip_fin_gather(di, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                          self.n, self.step.data_ptr(), self.lr, self.lr_dev.data_ptr(), float(self.b1),
                                                          float(self.b2), self.eps, self.max_norm, self.scratch.data_ptr(),
                                                          self.scratch.numel(), self.mb_index.data_ptr(), self.norm.data_ptr(),
                                                          self.gargs.data_ptr(), self.mbs, self._nseg, self._seg_scratch, self._seg_cols,
                                                          self._seg_tiles, self._seg_db, s))

    def _flush_adam(self):
        """pays the part of the last step's Adam sweep that no forward pass followed (adam_ride); clears the flag"""
        if not self.adam_ride:
            return
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.capi.check(self.lib.brl_adam_apply_range(
            di, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), self.defer[0], self.defer[1],
            self.scratch.data_ptr(), self.npartials, self.step.data_ptr(), self.lr, self.lr_dev.data_ptr(), float(self.b1),
            float(self.b2), self.eps, self.max_norm, 1.0, self.pending.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))

    def _opt(self):
        s = torch.cuda.current_stream().cuda_stream
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.capi.check(self.lib.brl_adam_clip_gather(di, self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                      self.n, self.step.data_ptr(), self.lr, self.lr_dev.data_ptr(), float(self.b1),
                                                      float(self.b2), self.eps,
                                                      self.max_norm, 1.0 / self.world, self.scratch.data_ptr(),
                                                      self.mb_index.data_ptr(), self.norm.data_ptr(), self.gargs.data_ptr(),
                                                      self.mbs, s))

    # ---- one update_step call -----------------------------------------------------------------------------------
    def begin_update(self, flat: Transition, adv_f, tgt_f, perms):
        """flat: the [T*N, ...] views of the trajectory; adv_f / tgt_f: [T*N]; perms: one permutation of T*N per epoch.
        Binds the step's gather to them (device-resident arguments) and resets the minibatch counter."""
        steps = sum(p.numel() for p in perms) // self.mbs
        if steps > self._log_cap:   # (checked before anything is touched; update_step never gets here: it rebuilds first)
            raise RuntimeError("FusedMinibatch: more minibatch steps per update than its log holds (log_capacity)")
        self._keep = (Transition(*[x.contiguous() for x in flat]), adv_f.contiguous(), tgt_f.contiguous())
        fl, adv_c, tgt_c = self._keep
        self._steps = steps
        with torch.no_grad():
            self._readopt()
            assert self._steps <= self._log_cap, "update_step sizes the log (log_capacity) before it binds an update"
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
        if self.world == 1:
            k = self.multi if self.graph_multi is not None else 0
            while k and n >= k:
                self.graph_multi.replay()
                n -= k
            for _ in range(n):
                self.graph.replay()
            self._flush_adam()
            return
        ar = self._collective if self._collective is not None else \
            (lambda t, async_op: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op))
        if self.allreduce_mode == "flat":
            if n <= 0:
                return
            self.segs[0].replay()
            for _ in range(n - 1):
                ar(self.G, False)                                     # brl_adam_clip divides by world (grad_scale)
                self.segs[1].replay()
            ar(self.G, False)
            self.segs[2].replay()
            return
        for _ in range(n):
            works = []
            for g, bucket in zip(self.segs[:-1], self.buckets):      # brl_adam_clip divides by world (grad_scale)
                g.replay()
                if bucket is not None:
                    works.append(ar(bucket, True))
            for w in works:
                if w is not None:
                    w.wait()
            self.segs[-1].replay()

    def end_update(self):
        """-> the [steps, 8] log of the update (total, value_loss, loss_actor, entropy, approx_kl, clipfrac, illegal-action
        norm / 2, 0): ONE launch over the sums the steps left behind"""
        di = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        with torch.no_grad():  # every parameter's step counter (torch keeps one per parameter)
            for q in self.plist:
                self.opt.state[q]["step"].copy_(self.step)
            self.capi.check(self.lib.brl_ppo_stats_rows(di, self.stat_sums.data_ptr(), self.gram_sums.data_ptr(), self._steps,
                                                        self.mbs, float(self.cfg["vf_coef"]), float(self.cfg["ent_coef"]),
                                                        self.ill_coef, self.log.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream))
            self._bind_gather(*self._dummy, first=False)   # (the trajectory may be freed by the caller now)
        self._keep = None
        return self.log[:self._steps]


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
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        graphed = None
        fused = None
        want_fused = adv_f.is_cuda and FusedMinibatch.supports(config, params)
        need_log = int(config["update_epochs"]) * num_mb
        if (adv_f.is_cuda and not want_fused and config.get("fused_update", True) and not opt_state.get("warned_unfused")
                and not str(getattr(params, "model", "")).startswith("DeepMind")):
            # a supported configuration on the slower path because of the network type: say so once (src/models.py:34-69)
            import warnings
            opt_state["warned_unfused"] = True
            warnings.warn(f"brl_amd.update: FusedMinibatch covers the DeepMind MLPs only; model_type "
                          f"{getattr(params, 'model', type(params).__name__)!r} takes the hipGraph-captured autograd step "
                          f"(GraphedMinibatch: ~1.6x slower per minibatch at minibatch 1024).", RuntimeWarning)
        if config.get("graph_update", True) and adv_f.is_cuda and (want_fused or (sched is None and not multi)):
            graphed = opt_state.get("graphed")
            world = dist.get_world_size() if multi else 1
            # (False = an earlier capture failed, on this or — under a process group — any rank: stay eager, do not retry)
            if graphed is None or (graphed is not False and (
                    graphed.params is not params or graphed.mbs != mbs or isinstance(graphed, FusedMinibatch) != want_fused
                    or getattr(graphed, "world", 1) != world
                    or (isinstance(graphed, FusedMinibatch) and graphed._log_cap < need_log))):   # e.g. minibatch 512: 5120 steps
                try:
                    graphed = FusedMinibatch(config, params, opt, mbs, adv_f.device, world, log_capacity=need_log) if want_fused \
                        else GraphedMinibatch(config, actor_forward_pass, params, opt, mbs, adv_f.device)
                except Exception as e:  # capture is an optimisation, never a requirement — but never a SILENT ~1.7x cliff
                    graphed = False
                    opt_state["graph_error"] = repr(e)
                    import warnings
                    warnings.warn(f"brl_amd.update: hipGraph capture of the minibatch step failed ({e!r}); this update runs "
                                  f"on the eager path (~1.7x slower).  opt_state['graph_error'] holds the error.", RuntimeWarning)
                if multi:
                    # every rank must issue the SAME collective sequence: a rank whose capture failed would run ONE flat
                    # all-reduce per minibatch while the others run FusedMinibatch's bucketed ones, and RCCL would hang
                    # until its timeout.  Agree (MIN over ranks): all fused, or all eager.
                    ok = torch.tensor([1 if isinstance(graphed, FusedMinibatch) else 0], dtype=torch.int32, device=adv_f.device)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    if int(ok.item()) == 0 and isinstance(graphed, FusedMinibatch):
                        opt_state["graph_error"] = "another rank failed to capture the minibatch step: all ranks run eager"
                        import warnings
                        warnings.warn("brl_amd.update: " + opt_state["graph_error"], RuntimeWarning)
                        graphed = False
                opt_state["graphed"] = graphed
            if isinstance(graphed, FusedMinibatch):
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
