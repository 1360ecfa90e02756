"""The policy/value network of the reference (src/models.py) as a small torch module.

north_star assigns the MLP to PyTorch-ROCm (dense GEMMs -> rocBLAS/hipBLASLt MFMA kernels);
only the architecture is mirrored here: "DeepMind" = 480 -> 4 x 1024 (ReLU) -> 38 logits + 1
value (src/models.py:23-33; "DeepMind_6" / "DeepMind_8" = the deeper variants of wb5/models.py:34-88),
"FAIR" = the 200-wide residual net (src/models.py:34-69).
``make_forward_pass`` keeps the Haiku-style ``init / apply(params, x)`` surface
(src/models.py:73-83); ``params`` is the torch module itself.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn


def _haiku_linear(fan_in: int, fan_out: int) -> nn.Linear:
    """hk.Linear default init: truncated normal(stddev = 1/sqrt(fan_in)), zero bias."""
    lin = nn.Linear(fan_in, fan_out)
    std = 1.0 / math.sqrt(fan_in)
    nn.init.trunc_normal_(lin.weight, mean=0.0, std=std, a=-2 * std, b=2 * std)
    nn.init.zeros_(lin.bias)
    return lin


class ActorCritic(nn.Module):
    def __init__(self, action_dim: int = 38, activation: str = "relu", model: str = "DeepMind", obs_dim: int = 480):
        super().__init__()
        self.model = model
        self.act = torch.relu if activation == "relu" else torch.tanh
        if model.startswith("DeepMind"):  # "DeepMind" = 4 hidden layers; "DeepMind_6" / "DeepMind_8": wb5/models.py:34-88
            depth = int(model.split("_")[1]) if "_" in model else 4
            self.body = nn.ModuleList([_haiku_linear(obs_dim, 1024)] + [_haiku_linear(1024, 1024) for _ in range(depth - 1)])
            self.actor = _haiku_linear(1024, action_dim)
            self.critic = _haiku_linear(1024, 1)
        elif model == "FAIR":
            self.l = nn.ModuleList(
                [_haiku_linear(obs_dim, 200)] + [_haiku_linear(200, 200) for _ in range(5)]
                + [_haiku_linear(200 + obs_dim, 200)] + [_haiku_linear(200, 200) for _ in range(4)]
            )
            self.actor = _haiku_linear(200, action_dim)
            self.critic = _haiku_linear(200, 1)
        else:
            raise ValueError(model)

    def _fair_forward(self, x):
        """``actor(x), critic(x)`` of the FAIR network as ONE launch (brl_fair_forward: csrc/fair_chain.hpp's forward half, 16 rows per
        workgroup) instead of ~25 — inference only (rollouts, evaluators: no autograd), fp32 on the GPU.  None = not applicable
        (any shape, layout, device or alignment the entry point would refuse — the caller then runs the layers one by one)."""
        if torch.is_grad_enabled() or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2) \
                or self.act not in (torch.relu, torch.tanh) or os.environ.get("BRL_FAIR_FORWARD", "1") == "0":
            return None
        # the kernel's fixed geometry (src/models.py:34-69): 480 observation bits, eleven 200-wide layers (the seventh reads 200 + 480
        # inputs), 38 + 1 head rows — all parameters fp32, contiguous, 16-byte aligned, on x's device
        dev = x.device
        ins = [480] + [200] * 5 + [680] + [200] * 4
        if x.shape[1] != 480 or len(self.l) != 11 or self.actor.weight.shape != (38, 200) or self.critic.weight.shape != (1, 200):
            return None
        tensors = [self.actor.weight, self.actor.bias, self.critic.weight, self.critic.bias]
        for lin, k in zip(self.l, ins):
            if lin.weight.shape != (200, k) or lin.bias.shape != (200,):
                return None
            tensors += [lin.weight, lin.bias]
        if any(t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16 for t in tensors
               if t is not self.critic.bias and t is not self.critic.weight):
            return None
        from . import _capi
        x = x.contiguous()
        if x.data_ptr() % 16:
            return None
        n = x.shape[0]
        net = _capi.FairNet()
        for l, lin in enumerate(self.l):
            net.w[l], net.b[l] = lin.weight.data_ptr(), lin.bias.data_ptr()
        aw, cw, ab, cb = self.actor.weight, self.critic.weight, self.actor.bias, self.critic.bias
        if cw.device == dev and cb.device == dev and cw.data_ptr() == aw.data_ptr() + 38 * 200 * 4 and cb.data_ptr() == ab.data_ptr() + 38 * 4:
            net.head_w, net.head_b = aw.data_ptr(), ab.data_ptr()       # (FusedFair's flat buffer: the heads are one [39, 200] already)
        else:   # the two heads as one matrix: a buffer of this module, rewritten when a head parameter has changed (its _version)
            if cw.device != dev or cb.device != dev or cw.dtype != torch.float32 or cb.dtype != torch.float32:
                return None
            key = (dev, aw._version, cw._version, ab._version, cb._version, aw.data_ptr(), cw.data_ptr(), ab.data_ptr(), cb.data_ptr())
            hw = getattr(self, "_fair_head_w", None)
            capturing = torch.cuda.is_current_stream_capturing()
            if hw is None or hw.device != dev:
                hw = self._fair_head_w = torch.empty((39, 200), dtype=torch.float32, device=dev)
                self._fair_head_b = torch.empty(39, dtype=torch.float32, device=dev)
                self._fair_head_key = None
            hb = self._fair_head_b
            if capturing or getattr(self, "_fair_head_key", None) != key:   # (inside a captured graph: re-read per replay)
                hw[:38].copy_(aw.detach()); hw[38:].copy_(cw.detach()); hb[:38].copy_(ab.detach()); hb[38:].copy_(cb.detach())
                self._fair_head_key = None if capturing else key
            net.head_w, net.head_b = hw.data_ptr(), hb.data_ptr()
        logits = torch.empty((n, 38), dtype=torch.float32, device=dev)
        value = torch.empty(n, dtype=torch.float32, device=dev)
        di = dev.index if dev.index is not None else torch.cuda.current_device()
        rc = _capi.lib().brl_fair_forward(di, net, x.data_ptr(), n, 0 if self.act is torch.relu else 1, logits.data_ptr(),
                                          value.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:       # refused (the documented fallback: the layers one by one), never an exception from an optimisation
            return None
        return logits, value

    def forward(self, x):
        a = self.act
        if self.model.startswith("DeepMind"):
            for lin in self.body:
                x = a(lin(x))
        else:
            fast = self._fair_forward(x)
            if fast is not None:
                return fast
            inp = x
            L = self.l
            x = L[0](x); s1 = x
            x = a(x); x = a(L[1](x)); x = a(L[2](x)); x = x + s1; s2 = x
            x = a(x); x = a(L[3](x)); x = a(L[4](x)); x = x + s2
            x = L[5](x)
            x = torch.cat([x, inp], dim=-1)
            x = L[6](x); s3 = x
            x = a(x); x = a(L[7](x)); x = a(L[8](x)); x = x + s3; s4 = x
            x = a(x); x = a(L[9](x)); x = a(L[10](x)); x = x + s4
        return self.actor(x), self.critic(x).squeeze(-1)


class InferenceSnapshot:
    """Inference-only copy of a "DeepMind" ReLU ActorCritic (no autograd): weights transposed (and cast to `dtype`
    when given) ONCE, bias + ReLU in the GEMM epilogue (``torch._addmm_activation`` -> hipBLASLt), the actor and
    critic heads as one 39-row GEMM.  With a 16-bit `dtype` and a library handle (`env`) the hidden layers run on the
    library's own kernel instead (``brl_linear_act``) and ``head_parts`` lets the last layer's launch compute the heads'
    share (``brl_linear_act_heads``).  Build one per rollout / evaluation call — it does not follow later weight
    updates (``refresh`` re-reads them in place).  ``make`` returns None for architectures it does not cover (callers fall
    back to ``module(x)``)."""

    def __init__(self, module: "ActorCritic", dtype=None, env=None, own_cast=True, views=False, gemm=None):
        # env: a BridgeBidding whose library runs the 16-bit hidden layers (brl_linear_act) and — own_cast — converts the 0/1
        # observation bytes to `dtype` (brl_obs_cast: 3 us instead of 15 us of GPU time per forward, but a slower launch than
        # torch's .to(): only worth it in hipGraph replays)
        self.env = env
        self.own_cast = bool(own_cast)
        self.dtype = dtype or torch.float32
        dt = self.dtype
        # gemm (config["inference_gemm"], else BRL_INFERENCE_GEMM, else "bf16x3"): "bf16x3" = fp32 forwards of >= 4096 rows run their hidden
        # layers on brl_mlp_gemm_x3 — the fp32 product as six bf16 MFMA products of three EXACT bf16 pieces per operand, fp32 accumulators
        # by magnitude class: 0.07-0.44 x the exact fp32 kernel's error against float64 and 1.4 x its rate on such batches
        # (csrc/mlp_gemm_x3.hpp) — on nn.Linear's own [out, in] weights; "library" = torch's fp32 GEMM for every batch.
        # Either way an fp32 forward (src/models.py:23-33); the two are not bit-identical to each other.
        self.gemm_x3 = (gemm or os.environ.get("BRL_INFERENCE_GEMM", "") or "bf16x3") == "bf16x3" and dt == torch.float32 \
            and all(lin.weight.shape[1] % 32 == 0 and lin.weight.shape[0] % 4 == 0 for lin in module.body)   # (whole 32-deep K chunks)
        # nn.Linear's own [out, in] layout: the module's parameters themselves under `views`, else copies `refresh` re-reads (their
        # addresses are baked into captured graphs; the update re-points the module's parameters at its flat buffers)
        self.lin = None
        self.wp = None
        if self.gemm_x3:
            self.lin = [(lin.weight.detach(), lin.bias.detach()) if views else (lin.weight.detach().clone(), lin.bias.detach().clone())
                        for lin in module.body]
            # ... and, where the shapes allow (out % 128, in % 32), on brl_linear_x3p (csrc/mlp_linear_x3p.hpp): the SAME products on
            # operands already split into bf16 planes — the weights here, once per refresh (they are constant over a rollout's 128
            # forwards), an activation by the launch that produces it — so that the K loop has no vector work.
            # BRL_INFERENCE_PLANES=0: brl_mlp_gemm_x3 (the split in registers) for every layer.
            if os.environ.get("BRL_INFERENCE_PLANES", "1") != "0" and all(
                    w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.shape[0] % 128 == 0 and w.shape[1] % 32 == 0
                    and w.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 for w, b in self.lin):
                self.wp = [torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device) for w, _ in self.lin]
                self._split_weights()
        # views (fp32 only): the hidden layers multiply with the module's own weights through transposed VIEWS — nothing is
        # copied (an evaluator builds its snapshots per call: nine launches per network otherwise) and nothing needs a refresh
        self.views = bool(views) and dt == torch.float32 and all(lin.weight.dtype == torch.float32 for lin in module.body)
        if self.views:
            self.body = [(lin.weight.detach().t(), lin.bias.detach()) for lin in module.body]
        else:
            self.body = [(lin.weight.detach().to(dt).t().contiguous(), lin.bias.detach().to(dt)) for lin in module.body]
        self.n_actions = module.actor.weight.shape[0]
        self.head_b = torch.cat([module.actor.bias, module.critic.bias], 0).detach().to(dt)
        # head_wt — for the step kernel that forms the heads itself (brl_macro_ext.head_h): [39, hidden] row-major in `dtype`;
        # head_bf: the bias as float holding the SAME (rounded) values the GEMM path adds
        if self.views:   # (two launches: the merged rows ARE head_wt, head_w is their transposed view)
            self.head_wt = torch.cat([module.actor.weight, module.critic.weight], 0).detach()
            self.head_w = self.head_wt.t()
        else:
            self.head_w = torch.cat([module.actor.weight, module.critic.weight], 0).detach().to(dt).t().contiguous()
            self.head_wt = self.head_w.t().contiguous()
        self.head_bf = self.head_b.float().contiguous()
        # 16-bit inference with a library handle: the hidden layers run on the library's own kernel (brl_linear_act,
        # csrc/mlp_infer.hpp) — nn.Linear's [out, in] layout in `dtype`, the bias as float holding the rounded values
        self.body_nk = None
        if env is not None and dt in (torch.bfloat16, torch.float16) and os.environ.get("BRL_LINEAR16", "1") != "0" \
                and all(lin.weight.shape[0] % 128 == 0 and lin.weight.shape[1] % 8 == 0 for lin in module.body):
            self.body_nk = [(lin.weight.detach().to(dt).contiguous(), lin.bias.detach().to(dt).float().contiguous())
                            for lin in module.body]

    def refresh(self, module: "ActorCritic"):
        """Re-read the weights of `module` INTO the existing tensors (their addresses are baked into captured graphs)."""
        if self.body_nk is not None:
            return self._refresh16(module)
        if self.lin is not None:
            if self.views:
                self.lin = [(lin.weight.detach(), lin.bias.detach()) for lin in module.body]
            else:
                for (w, b), lin in zip(self.lin, module.body):
                    w.copy_(lin.weight.detach())
                    b.copy_(lin.bias.detach())
            if self.wp is not None:
                self._split_weights()
        if self.views:   # (the hidden layers alias the module's parameters: only the merged heads are copies)
            self.body = [(lin.weight.detach().t(), lin.bias.detach()) for lin in module.body]
        else:
            for (w, b), lin in zip(self.body, module.body):
                w.copy_(lin.weight.detach().t())
                b.copy_(lin.bias.detach())
        k = self.n_actions
        self.head_w[:, :k].copy_(module.actor.weight.detach().t())
        self.head_w[:, k:].copy_(module.critic.weight.detach().t())
        self.head_b[:k].copy_(module.actor.bias.detach())
        self.head_b[k:].copy_(module.critic.bias.detach())
        if not self.views:
            self.head_wt.copy_(self.head_w.t())
        self.head_bf.copy_(self.head_b)

    def _split_weights(self):
        """the hidden layers' weights -> their three bf16 planes (brl_split_planes: w == hi + mid + lo exactly)"""
        from . import _capi
        L = _capi.lib()
        for (w, _), wp in zip(self.lin, self.wp):
            di = w.device.index if w.device.index is not None else torch.cuda.current_device()
            _capi.check(L.brl_split_planes(di, w.data_ptr(), w.numel(), wp.data_ptr(), w.numel(),
                                           torch.cuda.current_stream(w.device).cuda_stream))

    def planes_for(self, n):
        """True where a forward of `n` rows runs on brl_linear_x3p: its input may then come as bf16 (the 0/1 observation is exact in it)"""
        return self.wp is not None and n >= 4096

    def _body_planes(self, x):
        """the hidden layers on brl_linear_x3p: x bf16 [n, in] — the 0/1 observation, exact as given: ONE plane (an fp32 input stays on
        brl_mlp_gemm_x3: splitting it first costs what the first layer then saves); every layer writes the planes of its output, the last
        one fp32 (what the heads' product reads)"""
        from . import _capi
        L, st = _capi.lib(), torch.cuda.current_stream(x.device).cuda_stream
        di = x.device.index if x.device.index is not None else torch.cuda.current_device()
        n, k = x.shape
        xp, npx, sx = x, 1, 0
        y = None
        for li, ((w, b), wp) in enumerate(zip(self.lin, self.wp)):
            out, last = w.shape[0], li + 1 == len(self.lin)
            y = torch.empty((n, out), dtype=torch.float32, device=x.device) if last else None
            yp = None if last else torch.empty((3, n, out), dtype=torch.bfloat16, device=x.device)
            _capi.check(L.brl_linear_x3p(di, xp.data_ptr(), npx, k, sx, wp.data_ptr(), k, out * k, b.data_ptr(), 1,
                                         y.data_ptr() if last else None, out, None if last else yp.data_ptr(), out, n * out, n, out, k, st))
            xp, npx, sx, k = yp, 3, n * out, out
        return y

    def _refresh16(self, module):
        """`refresh` of the 16-bit layout: a handful of multi-tensor copies instead of ~30 cast launches per network and
        rollout (contiguous -> contiguous; the transposed [in, out] copies of the library-GEMM path are rebuilt only if a
        caller falls back to it)."""
        k = self.n_actions

        def copy_all(dst, src):
            if hasattr(torch, "_foreach_copy_"):
                torch._foreach_copy_(dst, src)
            else:
                for d, s_ in zip(dst, src):
                    d.copy_(s_)

        with torch.no_grad():
            ws = [w for w, _ in self.body_nk] + [self.head_wt[:k], self.head_wt[k:]]
            src = [lin.weight.detach() for lin in module.body] + [module.actor.weight.detach(), module.critic.weight.detach()]
            copy_all(ws, src)
            # biases: rounded to `dtype` first — the float copies hold the SAME values a 16-bit GEMM epilogue would add
            b16 = [b for _, b in self.body] + [self.head_b[:k], self.head_b[k:]]
            copy_all(b16, [lin.bias.detach() for lin in module.body] + [module.actor.bias.detach(), module.critic.bias.detach()])
            copy_all([b for _, b in self.body_nk] + [self.head_bf], [b for _, b in self.body] + [self.head_b])
            self.head_w.copy_(self.head_wt.t())
        self._body_stale = True   # self.body's weights ([in, out]) no longer match: rebuilt on demand (_body)

    @staticmethod
    def covers(module):
        """the architectures a snapshot exists for ("DeepMind" bodies with ReLU)"""
        return str(getattr(module, "model", "")).startswith("DeepMind") and getattr(module, "act", None) is torch.relu

    @staticmethod
    def make(module, dtype=None, env=None, own_cast=True, views=False, gemm=None):
        if not InferenceSnapshot.covers(module):
            return None
        return InferenceSnapshot(module, dtype, env, own_cast, views, gemm)

    _FMT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

    def _input(self, obs):
        if self.env is not None and self.own_cast and obs.dtype in (torch.bool, torch.uint8) and obs.is_cuda and obs.is_contiguous() \
                and self.dtype in self._FMT:
            from . import _capi
            from .bridge_bidding import _stream
            dt = torch.bfloat16 if (self.dtype == torch.float32 and obs.dim() == 2 and self.planes_for(obs.shape[0])) else self.dtype
            x = torch.empty(obs.shape, dtype=dt, device=obs.device)
            _capi.check(_capi.lib().brl_obs_cast(self.env._h, obs.data_ptr(), obs.numel() // 480, x.data_ptr(), self._FMT[dt], _stream()))
            return x
        if self.dtype == torch.float32 and obs.dim() == 2 and self.planes_for(obs.shape[0]):
            if obs.dtype in (torch.bool, torch.uint8):
                return obs.to(torch.bfloat16)      # 0 / 1: exact — one plane for brl_linear_x3p
            if obs.dtype == torch.bfloat16:
                return obs                         # (already cast by the caller: brl_obs_cast_rows with format 1)
        return obs.to(self.dtype)

    def _body(self, x):
        """the hidden layers: x [n, 480] in ``self.dtype`` -> [n, hidden]"""
        if self.body_nk is not None and x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.shape[0] > 0:
            from . import _capi
            from .bridge_bidding import _stream
            L, fmt, st = _capi.lib(), self._FMT[self.dtype], _stream()
            for w, b in self.body_nk:
                y = torch.empty((x.shape[0], w.shape[0]), dtype=self.dtype, device=x.device)
                _capi.check(L.brl_linear_act(self.env._h, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), b.data_ptr(),
                                           y.data_ptr(), y.stride(0), x.shape[0], w.shape[0], w.shape[1], 1, fmt, st))
                x = y
            return x
        if x.dim() == 2 and self.planes_for(x.shape[0]) and x.is_cuda and x.is_contiguous() and x.data_ptr() % 16 == 0 \
                and x.dtype == torch.bfloat16 and self.dtype == torch.float32:
            return self._body_planes(x)
        if x.dtype != self.dtype:
            raise RuntimeError(f"InferenceSnapshot: input in {x.dtype} where the layers run in {self.dtype} (bf16 input is taken by the "
                               "planes path only: >= 4096 contiguous rows)")
        if self.gemm_x3 and x.is_cuda and x.dim() == 2 and x.shape[0] >= 4096 and x.is_contiguous() and x.data_ptr() % 16 == 0:
            from . import _capi
            L, st = _capi.lib(), torch.cuda.current_stream(x.device).cuda_stream
            di = x.device.index if x.device.index is not None else torch.cuda.current_device()
            for w, b in self.lin:      # nn.Linear's own layout: y = relu(x W^T + b), W [out, in]
                y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
                _capi.check(L.brl_mlp_gemm_x3(di, 0, 1, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), y.data_ptr(), y.stride(0),
                                              x.shape[0], w.shape[0], w.shape[1], 0, b.data_ptr(), None, 0, None, None, 0, st))
                x = y
            return x
        if getattr(self, "_body_stale", False):   # (only after a 16-bit refresh, and only if this path is taken at all)
            for (w, _), (wn, _) in zip(self.body, self.body_nk):
                w.copy_(wn.t())
            self._body_stale = False
        fused = hasattr(torch, "_addmm_activation")
        for w, b in self.body:
            x = torch._addmm_activation(b, x, w, use_gelu=False) if fused else torch.addmm(b, x, w).relu_()
        return x

    HEAD_PART_LD = 40

    def head_parts(self, obs, x=None):
        """obs -> the heads as PARTIAL products f32 [hidden / 128, n, 40] (brl_linear_act_heads: the last hidden layer multiplies
        each of its 128-column tiles with the head weights while it holds it, and is itself never written to memory);
        ``brl_policy_step_ex`` adds the parts and the bias (brl_macro_ext.head_part).  None when the library's own layer kernel
        does not apply (fp32, no handle, other widths): callers use ``hidden`` / ``heads``."""
        if self.body_nk is None or os.environ.get("BRL_HEAD_PARTS", "1") == "0":
            return None
        if x is None:
            x = self._input(obs)
        if not (x.is_cuda and x.dim() == 2 and x.is_contiguous() and x.shape[0] > 0):
            return None
        from . import _capi
        from .bridge_bidding import _stream
        L, fmt, st, n = _capi.lib(), self._FMT[self.dtype], _stream(), x.shape[0]
        for w, b in self.body_nk[:-1]:
            y = torch.empty((n, w.shape[0]), dtype=self.dtype, device=x.device)
            _capi.check(L.brl_linear_act(self.env._h, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), b.data_ptr(),
                                         y.data_ptr(), y.stride(0), n, w.shape[0], w.shape[1], 1, fmt, st))
            x = y
        w, b = self.body_nk[-1]
        parts = torch.empty((w.shape[0] // 128, n, self.HEAD_PART_LD), dtype=torch.float32, device=x.device)
        _capi.check(L.brl_linear_act_heads(self.env._h, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), b.data_ptr(), None, 0,
                                           n, w.shape[0], w.shape[1], 1, fmt, self.head_wt.data_ptr(), self.head_wt.stride(0),
                                           self.head_wt.shape[0], parts.data_ptr(), parts.stride(1), parts.stride(0), st))
        return parts

    def hidden(self, obs, x=None):
        """obs -> the last hidden layer's output [n, hidden] in ``self.dtype`` (what the heads are applied to)"""
        if x is None:
            x = self._input(obs)
        return self._body(x)

    def heads(self, obs, x=None, raw=False):
        """obs: [n, 480] bool / float -> f32 [n, 39]: the 38 logits and the value as ONE matrix (row stride 39; the
        kernels take the logits as a strided slice of it).  ``x``: the observation already in ``self.dtype`` (written
        by the step kernel that produced it, brl_macro_ext.obs_cast) — then ``obs`` is not read.  ``raw``: the matrix in
        ``self.dtype`` as the GEMM wrote it (brl_policy_step_ex converts while reading: brl_macro_ext.in_fmt)."""
        if x is None:
            x = self._input(obs)
        x = self._body(x)
        out = torch.addmm(self.head_b, x, self.head_w)
        return out if raw else out.float()

    def __call__(self, obs):
        """obs: [n, 480] bool / float -> (logits f32 [n, 38], value f32 [n])"""
        out = self.heads(obs)
        return out[:, :self.n_actions], out[:, self.n_actions]


class ForwardPass:
    """``hk.without_apply_rng(hk.transform(forward_fn))`` look-alike (src/models.py:73-83)."""

    def __init__(self, activation: str, model_type: str):
        self.activation = activation
        self.model_type = model_type

    def init(self, rng, x=None, device=None):
        seed = int(rng) if not torch.is_tensor(rng) else int(rng.reshape(-1)[0].item())
        devs = [] if device is None or torch.device(device).type == "cpu" else [device]
        with torch.random.fork_rng(devices=devs):
            torch.manual_seed(seed)
            net = ActorCritic(38, self.activation, self.model_type)
        return net.to(device) if device is not None else net

    def apply(self, params: nn.Module, x):
        return params(x)


def make_forward_pass(activation: str, model_type: str) -> ForwardPass:
    return ForwardPass(activation, model_type)
