"""float64 numpy restatement of ONE PPO minibatch step of the reference (src/update.py:86-178 + the optax chain of
ppo.py:195-211) for the "DeepMind" ReLU MLP (src/models.py:23-33) and the "FAIR" residual net (src/models.py:34-69): forward,
`_loss_fn`, hand-derived backward, clip_by_global_norm, Adam(eps=1e-5).  TEST INFRASTRUCTURE — the checker for brl_amd/update.py
and brl_amd/fused_update.py on CPU and GPU."""
import numpy as np


def params_of(net):
    """[(W [out,in], b [out])] for body layers, actor head, critic head — float64 copies."""
    lins = list(net.body) + [net.actor, net.critic]
    return [(l.weight.detach().cpu().double().numpy().copy(), l.bias.detach().cpu().double().numpy().copy()) for l in lins]


def forward(params, x):
    hs, zs = [x], []
    for W, b in params[:-2]:
        z = hs[-1] @ W.T + b
        zs.append(z)
        hs.append(np.maximum(z, 0.0))
    logits = hs[-1] @ params[-2][0].T + params[-2][1]
    value = (hs[-1] @ params[-1][0].T + params[-1][1])[:, 0]
    return logits, value, hs, zs


def head_loss(cfg, logits, value, mask, action, old_value, old_log_prob, gae, tgt):
    """`_loss_fn` on given network outputs (float64) -> (total, (value_loss, loss_actor, entropy, approx_kl, clipfrac),
    dlogits, dvalue): the derivative w.r.t. the outputs, hand-derived.  cfg["actor_illegal_action_mask"] False = the
    unmasked policy for the log-prob (src/update.py:18-21); the entropy is always the masked policy's (:132-135)."""
    B = logits.shape[0]
    eps = cfg["clip_eps"]
    mask = mask.astype(bool)
    masked = cfg.get("actor_illegal_action_mask", True)
    ml = np.where(mask, logits, -np.inf)
    ml = ml - ml.max(1, keepdims=True)
    lsm = ml - np.log(np.exp(ml).sum(1, keepdims=True))
    p = np.exp(lsm)
    ul = logits - logits.max(1, keepdims=True)
    lsm_u = ul - np.log(np.exp(ul).sum(1, keepdims=True))
    idx = np.arange(B)
    lsel, psel = (lsm, p) if masked else (lsm_u, np.exp(lsm_u))
    lp = lsel[idx, action]
    logratio = lp - old_log_prob
    ratio = np.exp(logratio)
    if cfg.get("value_clipping", True):
        vc = old_value + np.clip(value - old_value, -eps, eps)
        l1, l2 = (value - tgt) ** 2, (vc - tgt) ** 2
        value_loss = 0.5 * np.maximum(l1, l2).mean()
        dv = np.where(l1 >= l2, value - tgt, (vc - tgt) * (np.abs(value - old_value) < eps))
    else:
        value_loss = 0.5 * ((value - tgt) ** 2).mean()
        dv = value - tgt
    dv = dv / B * cfg["vf_coef"]
    a1, a2 = ratio * gae, np.clip(ratio, 1 - eps, 1 + eps) * gae
    loss_actor = -np.minimum(a1, a2).mean()
    inside_r = (ratio > 1 - eps) & (ratio < 1 + eps)
    dlp = -np.where((a1 < a2) | inside_r, gae, 0.0) / B * ratio
    plogp = np.where(mask, p * np.where(mask, lsm, 0.0), 0.0)
    ent_i = -plogp.sum(1)
    entropy = ent_i.mean()
    total = loss_actor + cfg["vf_coef"] * value_loss - cfg["ent_coef"] * entropy
    onehot = np.zeros_like(p)
    onehot[idx, action] = 1.0
    dlogits = dlp[:, None] * (onehot - psel)
    if masked:
        dlogits = np.where(mask, dlogits, 0.0)
    dH = -np.where(mask, p * (np.where(mask, lsm, 0.0) + ent_i[:, None]), 0.0)
    dlogits = dlogits - cfg["ent_coef"] * dH / B
    approx_kl = ((ratio - 1) - logratio).mean()
    clipfrac = (np.abs(ratio - 1.0) > eps).mean()
    return total, (value_loss, loss_actor, entropy, approx_kl, clipfrac), dlogits, dv


def loss_and_grads(cfg, params, obs, mask, action, old_value, old_log_prob, gae, tgt):
    """-> (total, (value_loss, loss_actor, entropy, approx_kl, clipfrac), grads like params)"""
    B = obs.shape[0]
    eps = cfg["clip_eps"]
    mask = mask.astype(bool)
    logits, value, hs, zs = forward(params, obs.astype(np.float64))
    ml = np.where(mask, logits, -np.inf)
    ml = ml - ml.max(1, keepdims=True)
    lsm = ml - np.log(np.exp(ml).sum(1, keepdims=True))          # masked log-softmax (src/update.py:12-16)
    p = np.exp(lsm)
    idx = np.arange(B)
    lp = lsm[idx, action]
    logratio = lp - old_log_prob
    ratio = np.exp(logratio)
    # value loss (src/update.py:48-60)
    vc = old_value + np.clip(value - old_value, -eps, eps)
    l1, l2 = (value - tgt) ** 2, (vc - tgt) ** 2
    value_loss = 0.5 * np.maximum(l1, l2).mean()
    inside_v = np.abs(value - old_value) < eps
    dv = np.where(l1 >= l2, value - tgt, (vc - tgt) * inside_v) / B * cfg["vf_coef"]
    # actor loss (src/update.py:116-130)
    a1, a2 = ratio * gae, np.clip(ratio, 1 - eps, 1 + eps) * gae
    loss_actor = -np.minimum(a1, a2).mean()
    inside_r = (ratio > 1 - eps) & (ratio < 1 + eps)
    dratio = -np.where((a1 < a2) | inside_r, gae, 0.0) / B
    dlp = dratio * ratio
    # entropy of the masked policy (src/update.py:132-138), 0 log 0 = 0
    plogp = np.where(mask, p * np.where(mask, lsm, 0.0), 0.0)
    ent_i = -plogp.sum(1)
    entropy = ent_i.mean()
    total = loss_actor + cfg["vf_coef"] * value_loss - cfg["ent_coef"] * entropy
    onehot = np.zeros_like(p)
    onehot[idx, action] = 1.0
    dlogits = dlp[:, None] * (onehot - p)
    dH = -np.where(mask, p * (np.where(mask, lsm, 0.0) + ent_i[:, None]), 0.0)   # dH_i / dz_a
    dlogits += -cfg["ent_coef"] * dH / B
    dlogits = np.where(mask, dlogits, 0.0)
    # backward through the MLP
    grads = [None] * len(params)
    h = hs[-1]
    grads[-2] = (dlogits.T @ h, dlogits.sum(0))
    grads[-1] = (dv[None, :] @ h, np.array([dv.sum()]))
    dh = dlogits @ params[-2][0] + dv[:, None] * params[-1][0]
    for k in range(len(params) - 3, -1, -1):
        dz = dh * (zs[k] > 0)
        grads[k] = (dz.T @ hs[k], dz.sum(0))
        dh = dz @ params[k][0]
    approx_kl = ((ratio - 1) - logratio).mean()
    clipfrac = (np.abs(ratio - 1.0) > eps).mean()
    return total, (value_loss, loss_actor, entropy, approx_kl, clipfrac), grads


def adam_first_step(cfg, params, grads):
    """optax.chain(clip_by_global_norm(max_grad_norm), adam(lr, eps=1e-5)) from a fresh state (ppo.py:195-211)."""
    gn = np.sqrt(sum((gw ** 2).sum() + (gb ** 2).sum() for gw, gb in grads))
    scale = min(1.0, cfg["max_grad_norm"] / gn) if cfg.get("global_gradient_clipping", True) else 1.0
    out = []
    for (W, b), (gw, gb) in zip(params, grads):
        new = []
        for x, g in ((W, gw), (b, gb)):
            g = g * scale
            m_hat, v_hat = g, g * g            # (1-b1) g / (1-b1) ; (1-b2) g^2 / (1-b2)
            new.append(x - cfg["lr"] * m_hat / (np.sqrt(v_hat) + 1e-5))
        out.append(tuple(new))
    return out, gn


# ---------------------------------------------------------------------------------------------------------------
# the FAIR network (src/models.py:34-69), written as the reference writes it: x = L(x); shortcut = x; x = act(x); ...
# ---------------------------------------------------------------------------------------------------------------
def fair_params_of(net):
    lins = list(net.l) + [net.actor, net.critic]
    return [(l.weight.detach().cpu().double().numpy().copy(), l.bias.detach().cpu().double().numpy().copy()) for l in lins]


def fair_loss_and_grads(cfg, params, obs, mask, action, old_value, old_log_prob, gae, tgt, activation="relu"):
    """-> (total, aux, grads like params): the loss of `head_loss` on the FAIR forward pass and its gradient by reverse-mode
    differentiation of the reference's own statement order (a tape of (kind, ...) entries; no shared code with the build)."""
    act = (lambda z: np.maximum(z, 0.0)) if activation == "relu" else np.tanh
    dact = (lambda out: (out > 0).astype(np.float64)) if activation == "relu" else (lambda out: 1.0 - out * out)
    x = obs.astype(np.float64)
    inp = x
    tape = []

    def lin(k, v):
        tape.append(("lin", k, v))
        return v @ params[k][0].T + params[k][1]

    def activate(v):
        out = act(v)
        tape.append(("act", out))
        return out

    x = lin(0, x); s1 = x
    x = activate(x); x = lin(1, x); x = activate(x); x = lin(2, x); x = activate(x)
    x = x + s1; tape.append(("add", "s1")); s2 = x
    x = activate(x); x = lin(3, x); x = activate(x); x = lin(4, x); x = activate(x)
    x = x + s2; tape.append(("add", "s2"))
    x = lin(5, x)
    x = np.concatenate([x, inp], axis=-1); tape.append(("cat", 200))
    x = lin(6, x); s3 = x
    x = activate(x); x = lin(7, x); x = activate(x); x = lin(8, x); x = activate(x)
    x = x + s3; tape.append(("add", "s3")); s4 = x
    x = activate(x); x = lin(9, x); x = activate(x); x = lin(10, x); x = activate(x)
    x = x + s4; tape.append(("add", "s4"))
    logits = x @ params[11][0].T + params[11][1]
    value = (x @ params[12][0].T + params[12][1])[:, 0]
    total, aux, dlogits, dv = head_loss(cfg, logits, value, mask, action, old_value, old_log_prob, gae, tgt)
    grads = [None] * len(params)
    grads[11] = (dlogits.T @ x, dlogits.sum(0))
    grads[12] = (dv[None, :] @ x, np.array([dv.sum()]))
    d = dlogits @ params[11][0] + dv[:, None] * params[12][0]
    # the shortcuts were taken right AFTER the linear layers 0 and 6 (s1, s3) and after the first / third block's sum (s2, s4):
    # a pending shortcut gradient is added where its value was defined
    pending = {}
    defined_after = {"s4": ("add", "s3"), "s2": ("add", "s1")}     # s4 = the value after `add s3`, s2 = the value after `add s1`
    for entry in reversed(tape):
        if entry[0] == "add":
            name = entry[1]
            pending[name] = d.copy()                                 # x = x + s: the gradient flows to both
            for later, where in defined_after.items():
                if where == entry and later in pending:              # this sum's OUTPUT was also taken as shortcut `later`
                    d = d + pending.pop(later)
                    pending[name] = d.copy()
        elif entry[0] == "act":
            d = d * dact(entry[1])
        elif entry[0] == "cat":
            d = d[:, :entry[1]]
        else:
            _, k, v = entry
            if k == 6 and "s3" in pending:
                d = d + pending.pop("s3")
            if k == 0 and "s1" in pending:
                d = d + pending.pop("s1")
            grads[k] = (d.T @ v, d.sum(0))
            d = d @ params[k][0]
    assert not pending
    return total, aux, grads
