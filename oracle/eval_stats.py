"""numpy restatement of the statistics of ``make_evaluate`` — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/src/evaluation.py line by line, in float32 where the reference's jnp arrays are float32:
the per-step log (:299-378 single table, :649-748 duplicate), ``make_terminated_log`` / ``make_contract_log``
(:463-563, :841-984) and the two ``log_info`` tuples (:583-605, :985-1031)."""
from __future__ import annotations

import numpy as np

f32 = np.float32


class StepLog:
    def __init__(self, n):
        self.total_illegal = np.zeros((n, 2), f32)   # [:,0] actor (players 0,1)  [:,1] opp (players 2,3)
        self.step_count = np.zeros((n, 2), f32)
        self.bid = np.zeros((n, 2, 35), f32)
        self.pass_count = np.zeros((n, 2), f32)

    def update(self, terminated, current_player, mask, logits, action, bid_set):
        """make_step_log / update_log_info: boards whose state is terminated log nothing (:736-748 / :380-388).
        logits: the UNMASKED logits of the network that acted; illegal mass = dot(softmax(logits), ~mask) (:664-665)."""
        n = len(action)
        lg = logits.astype(f32)
        e = np.exp(lg - lg.max(1, keepdims=True))
        probs = e / e.sum(1, keepdims=True)
        illegal = (probs * (mask == 0)).sum(1).astype(f32)
        team = (current_player >= 2).astype(np.int64)
        live = terminated == 0
        idx = np.arange(n)[live]
        t = team[live]
        self.total_illegal[idx, t] += illegal[live]
        self.step_count[idx, t] += 1
        a = action[live]
        isbid = a >= 3
        if bid_set:   # single table: .at[action - 3].set(1)  (:349-358)
            self.bid[idx[isbid], t[isbid], a[isbid] - 3] = 1
        else:         # duplicate: one-hot + actor_bid  (:704-713)
            np.add.at(self.bid, (idx[isbid], t[isbid], a[isbid] - 3), 1)
        ispass = a == 0
        np.add.at(self.pass_count, (idx[ispass], t[ispass]), 1)


def terminated_log(last_bid, last_bidder, call_x, call_xx, r0):
    """make_terminated_log + make_contract_log for one table's arrays (:841-984; :463-563 with r0 = cum_return)."""
    n = len(last_bid)
    pass_out = (last_bidder == -1) & (last_bid == -1)
    actor_side = (last_bidder < 2) & ~pass_out
    opp_side = (last_bidder >= 2) & ~pass_out
    actor_contract = np.zeros((n, 35), f32)
    opp_contract = np.zeros((n, 35), f32)
    actor_contract[np.nonzero(actor_side)[0], last_bid[actor_side]] = 1
    opp_contract[np.nonzero(opp_side)[0], last_bid[opp_side]] = 1
    nonneg = r0 >= 0
    return {
        "pass_out": pass_out, "actor_contract": actor_contract, "opp_contract": opp_contract,
        "actor_doubled": actor_side & (call_x != 0), "actor_redoubled": actor_side & (call_xx != 0),
        "opp_doubled": opp_side & (call_x != 0), "opp_redoubled": opp_side & (call_xx != 0),
        # the reference's labels (:951-984): rewards[0] >= 0 x declaring team
        "actor_make": actor_side & nonneg, "opp_make": opp_side & nonneg,
        "actor_down": actor_side & ~nonneg, "opp_down": opp_side & ~nonneg,
    }


def duplicate_log_info(cum_return, log: StepLog, step_count, A, B):
    """log_info of duplicate_evaluate (:985-1031).  A / B: oracle TABLE_INFO arrays."""
    n = f32(len(cum_return))
    ta = terminated_log(A["last_bid"], A["last_bidder"], A["call_x"], A["call_xx"], A["rewards"][:, 0])
    tb = terminated_log(B["last_bid"], B["last_bidder"], B["call_x"], B["call_xx"], B["rewards"][:, 0])
    m = lambda k: (ta[k].astype(f32).mean(axis=0) + tb[k].astype(f32).mean(axis=0)) / 2  # noqa: E731
    ratio = lambda k: (ta[k].sum() / n + tb[k].sum() / n) / 2  # noqa: E731
    cr = cum_return.astype(f32)
    return (
        cr.mean(), cr.std(ddof=1) / np.sqrt(n),
        (A["rewards"][:, 0].mean() + B["rewards"][:, 0].mean()) / 2,
        (log.total_illegal[:, 0] / log.step_count[:, 0]).mean(), (log.total_illegal[:, 1] / log.step_count[:, 1]).mean(),
        step_count.astype(f32).mean(),
        log.bid[:, 0].mean(axis=0) / 2, log.bid[:, 1].mean(axis=0) / 2,
        m("actor_contract"), m("opp_contract"), ratio("actor_contract"), ratio("opp_contract"),
        m("actor_doubled"), m("actor_redoubled"), m("opp_doubled"), m("opp_redoubled"),
        m("actor_make"), m("opp_make"), m("actor_down"), m("opp_down"), ratio("pass_out"),
        (log.pass_count[:, 0] / log.step_count[:, 0]).mean(), (log.pass_count[:, 1] / log.step_count[:, 1]).mean(),
    )


def single_log_info(cum_return, log: StepLog, state):
    """log_info of evaluate (:583-605).  state: oracle STATE array of the finished boards."""
    n = f32(len(cum_return))
    t = terminated_log(state["last_bid"], state["last_bidder"], state["call_x"], state["call_xx"], cum_return)
    m = lambda k: t[k].astype(f32).mean(axis=0)  # noqa: E731
    return (
        cum_return.astype(f32).mean(),
        (log.total_illegal[:, 0] / log.step_count[:, 0]).mean(), (log.total_illegal[:, 1] / log.step_count[:, 1]).mean(),
        state["step_count"].astype(f32).mean(), log.bid[:, 0].mean(axis=0), log.bid[:, 1].mean(axis=0),
        m("actor_contract"), m("opp_contract"), t["actor_contract"].sum() / n, t["opp_contract"].sum() / n,
        m("actor_doubled"), m("actor_redoubled"), m("opp_doubled"), m("opp_redoubled"),
        m("actor_make"), m("opp_make"), m("actor_down"), m("opp_down"), t["pass_out"].sum() / n,
    )


EVAL_LOG_KEYS = [  # make_evaluate_log (:1062-1082), in order
    "eval/IMP_reward", "eval/IMP_SE", "eval/score_reward", "eval/actor_illegal_action_probs",
    "eval/opp_illegal_action_probs", "eval/step count", "eval/actor_declarer_ratio", "eval/opp_declarer_ratio",
    "eval/actor_doubled_ratio", "eval/actor_redoubled_ratio", "eval/opp_doubled_ratio", "eval/opp_redoubled_ratio",
    "eval/actor_make_contract_ratio", "eval/opp_make_contract_ratio", "eval/actor_down_contract_ratio",
    "eval/opp_down_contract_ratio", "eval/pass_out_ratio", "eval/actor_pass_ratio", "eval/opp_pass_ratio"]
