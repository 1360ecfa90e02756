"""The PPO self-play loop of ``ppo.py:224-572`` on the MI355X-native path — BASELINE configs 4/5.

    python -m brl_amd.train num_envs=8192 num_steps=32 total_timesteps=2621440 [dds_results_dir=dds_results]
    python -m torch.distributed.run --nproc-per-node 8 -m brl_amd.train ...   # env shards per GPU, RCCL gradient all-reduce

Same phases, config keys and defaults as the reference (``PPOConfig``, ppo.py:40-180): periodic checkpoints
(:351-362, torch state_dicts instead of pickles), the three evaluations (:366-381), the opponent pool — latest / FSP
uniform / PFSP ``softmax(-IMP / prior_t)`` behind the ``threshold_model_zoo`` gate (:376-460) —, ``imp_opp_before/after``,
roll_out -> calc_gae -> update_step (:467-479), the log dict (:501-519), rotation of the double-dummy hash tables after
``hash_size`` finished boards (:525-549, G14) and the final params + opt_state files (:550-570).  wandb is replaced by
one JSON line per iteration.  Differences, all host-side: the training tables live in ONE handle whose LUT is swapped in
place (``brl_set_lut``) instead of one env + jitted roll_out per file; evaluation has its own handle; under
torch.distributed rank 0 draws the opponent and broadcasts its index.  Without ``dds_results_dir`` synthetic tables are
used (``synthetic_lut_files`` of them, so that rotation still happens).
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

DEFAULTS = dict(  # ppo.py:122-180 (PPOConfig), same names and defaults
    seed=0, lr=0.000001, num_envs=8192, num_steps=32, total_timesteps=2_621_440_000, update_epochs=10,
    minibatch_size=1024,
    dds_results_dir=None, hash_size=100_000,                                                       # dataset
    num_eval_envs=10000, eval_opp_activation="relu", eval_opp_model_type="DeepMind", eval_opp_model_path=None,
    num_eval_step=10,                                                                               # eval
    save_model=True, save_model_interval=1, log_path="rl_log", exp_name="exp_0000", save_model_path="rl_params",
    track=False,                                                                                    # log (no wandb here)
    load_initial_model=False, initial_model_path=None, actor_activation="relu", actor_model_type="DeepMind",
    game_mode="competitive", self_play=True, opp_activation="relu", opp_model_type="DeepMind", opp_model_path=None,
    ratio_model_zoo=0.0, num_model_zoo=100_000, threshold_model_zoo=-24.0, prioritized_fictitious=False, prior_t=0.1,
    num_prioritized_envs=100,                                                                       # opponent pool
    gamma=1.0, gae_lambda=0.95, clip_eps=0.2, ent_coef=0.001, vf_coef=0.5,
    value_clipping=True, global_gradient_clipping=True, anneal_lr=False, reward_scaling=False, max_grad_norm=0.5,
    reward_scale=7600.0,
    actor_illegal_action_mask=True, actor_illegal_action_penalty=False, illegal_action_penalty=-1.0,
    illegal_action_l2norm_coef=0.0,
    # build-side knobs (not in the reference)
    lut_len=100_000, synthetic_lut_files=3, inference_dtype=None, inference_gemm=None, dw_gemm=None, graph_rollout=False, evaluate=True,
    tunable_gemm=False,  # torch TunableOp: time every rocBLAS / hipBLASLt solution once per GEMM shape (update: -5 %)
    memoize_eval=True, memoize_eval_check_every=0,   # the per-iteration duplicate evaluations: play each distinct pair once (train())
    grad_allreduce="flat",      # the gradient step under a process group: "flat" | "sharded" (brl_amd/fused_update.py)
    check_rank_sync=True,       # under a process group: a parameter checksum compared across the ranks after every update
)


def parse_cli(argv):
    """``key=value`` arguments like the reference's OmegaConf CLI (ppo.py:183)."""
    cfg = dict(DEFAULTS)
    for a in argv:
        k, v = a.split("=", 1)
        if k not in cfg:
            raise SystemExit(f"unknown option {k}")
        d = DEFAULTS[k]
        if isinstance(d, bool):
            cfg[k] = v.lower() in ("1", "true")
        elif d is None:
            cfg[k] = None if v.lower() in ("none", "null") else v
        else:
            cfg[k] = type(d)(v)
    return cfg


# ---------------------------------------------------------------------------------------------------------------
# pieces of the loop that are pure host logic (unit-tested on CPU)
# ---------------------------------------------------------------------------------------------------------------
def pfsp_probabilities(imp_list, prior_t: float):
    """PFSP sampling weights ``softmax(-imp / prior_t)`` (ppo.py:424-435: the maximum is subtracted first)."""
    x = -np.asarray(imp_list, np.float64)
    e = np.exp((x - np.max(x, axis=-1, keepdims=True)) / prior_t)
    return e / np.sum(e, axis=-1, keepdims=True)


PFSP = -3   # opponent_draw: "draw from the league with PFSP weights" — the league evaluation comes first


def opponent_draw(config, imp_opp: float, n_pool: int, rng: np.random.RandomState):
    """First half of the decision tree of ppo.py:381-460, the part that needs no evaluation: None = keep the current
    opponent (below the threshold), -1 = "latest", an index = uniform FSP draw, PFSP = a prioritised draw follows."""
    if imp_opp < config["threshold_model_zoo"]:
        return None
    if n_pool != 0 and rng.binomial(size=1, n=1, p=config["ratio_model_zoo"])[0]:
        if config["prioritized_fictitious"]:
            return PFSP
        return int(rng.choice(n_pool))
    return -1


def pfsp_draw(config, imps, rng: np.random.RandomState) -> int:
    """Second half (ppo.py:399-435): checkpoint k with probability softmax(-imp_k / prior_t)."""
    probabilities = pfsp_probabilities(imps, config["prior_t"])
    return int(rng.choice(len(probabilities), p=probabilities))


def choose_opponent(config, imp_opp: float, params_list, rng: np.random.RandomState, league_imps=None):
    """The decision tree of ppo.py:381-460.  Returns the index into ``params_list`` of the checkpoint to play against,
    -1 for "latest" (the current weights), or None for "keep the current opponent" (below the threshold).
    ``league_imps``: callable returning the IMP of the learner against every checkpoint (PFSP only)."""
    d = opponent_draw(config, imp_opp, len(params_list), rng)
    return pfsp_draw(config, league_imps(), rng) if d == PFSP else d


class LutRotation:
    """ppo.py:297-301,525-549: one hash table at a time; after ``hash_size`` finished boards move to the next file,
    reshuffling the order once all have been used."""

    def __init__(self, num_files: int, hash_size: int, rng: np.random.RandomState):
        self.order = np.arange(num_files)
        self.hash_index = 0
        self.board_count = 0
        self.hash_size = hash_size
        self.rng = rng

    @property
    def current(self) -> int:
        return int(self.order[self.hash_index])

    def advance(self, terminated_count: int) -> bool:
        """True if the table must be switched now (then ``current`` is the new file)."""
        if (terminated_count - self.board_count) // self.hash_size < 1:
            return False
        self.hash_index += 1
        self.board_count = terminated_count
        if self.hash_index == len(self.order):
            self.hash_index = 0
            self.rng.shuffle(self.order)
        return True


def linear_schedule(config, count):
    """ppo.py:186-192"""
    frac = 1.0 - (count // (config["num_minibatches"] * config["update_epochs"])) / config["num_updates"]
    return config["lr"] * frac


# ---------------------------------------------------------------------------------------------------------------
def _weight_delta(a: torch.nn.Module, b) -> float:
    """max |actor - opponent| over the parameters (a log field of this build): NaN when the two trees differ in shape (e.g. a FAIR
    opponent of a DeepMind actor) or are the same object; ONE reduction and one host read, whatever the number of tensors"""
    if b is a or not isinstance(b, torch.nn.Module):
        return float("nan")
    pa, pb = list(a.parameters()), list(b.parameters())
    if len(pa) != len(pb) or any(x.shape != y.shape for x, y in zip(pa, pb)):
        return float("nan")
    with torch.no_grad():
        return float(torch.stack([(x - y).abs().max() for x, y in zip(pa, pb)]).max())


def train(config, log=print, on_rollout=None):
    """``on_rollout(i, runner_state, traj_batch, roll_out)``: optional observer called right after iteration i's roll_out (tests
    replay a rank's shard through the oracle from it; the buffers are the loop's own — copy what you keep)."""
    import torch.distributed as dist

    import brl_amd
    from brl_amd import checkpoint as ckpt
    from brl_amd.bridge_bidding import load_dds_table
    from brl_amd.dist import broadcast_int, broadcast_parameters, distributed, rank_world, shard_offset, sum_over_ranks
    from brl_amd.evaluation import (make_evaluate, make_evaluate_log, make_simple_duplicate_evaluate,
                                    make_simple_evaluate)
    from brl_amd.models import make_forward_pass
    from brl_amd.update import make_optimizer, make_update_step
    if config.get("tunable_gemm") and torch.cuda.is_available():
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(30)

    rank, world = rank_world()
    if (world > 1 or os.environ.get("BRL_FORCE_DIST") == "1") and not dist.is_initialized():
        # BRL_DIST_BACKEND=gloo: rehearsal on a box with fewer GPUs than ranks (ranks share the devices round-robin)
        backend = os.environ.get("BRL_DIST_BACKEND", "nccl")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        if backend == "nccl":   # RCCL: one GPU per rank, bound to the communicator from the start (no lazy device guess)
            dist.init_process_group(backend, device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", torch.cuda.current_device())
    config = dict(DEFAULTS, **config)
    # ppo.py:225-227; under a process group num_envs is PER RANK (each rank owns its env shard: weak scaling), so an update
    # consumes world * num_envs * num_steps of the total_timesteps
    config["num_updates"] = config["total_timesteps"] // config["num_steps"] // (config["num_envs"] * world)
    config["num_minibatches"] = config["num_envs"] * config["num_steps"] // config["minibatch_size"]  # ppo.py:228-230
    host_rng = np.random.RandomState(config["seed"])  # the reference uses numpy's global RNG for the pool / shuffles

    # ---- double-dummy tables (ppo.py:297-308); kept in host memory, ONE device copy at a time
    if config["dds_results_dir"]:
        d = config["dds_results_dir"]
        train_files = sorted(p for p in os.listdir(d) if "train" in p)
        luts = [load_dds_table(os.path.join(d, f)) for f in train_files]
        test_path = os.path.join(d, "test_000.npy")
        eval_lut = load_dds_table(test_path) if os.path.exists(test_path) else luts[0]      # ppo.py:252
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import synthetic_lut
        train_files = [f"synthetic_{i:03}" for i in range(config["synthetic_lut_files"])]
        luts = [synthetic_lut(config["lut_len"], i) for i in range(len(train_files))]
        eval_lut = synthetic_lut(config["lut_len"], 10_000)
    rotation = LutRotation(len(luts), config["hash_size"], host_rng)
    env = brl_amd.BridgeBidding(lut=luts[rotation.current], device=dev, env_offset=shard_offset(rank, config["num_envs"]))
    # evaluation has its OWN handle: env.init(seed) re-keys a handle's RNG, which must not leak into the training tables.
    # Under a process group every evaluation is SHARDED: rank r plays num_eval_envs / world of the boards (the same global
    # boards a single process would deal) and the sums are all-reduced, so every rank holds the same statistics and the pool
    # decisions agree (ppo.py:366-381,461-484 runs 3-4 evaluations of 10 000 boards per iteration)
    eval_env = brl_amd.BridgeBidding(lut=eval_lut, device=dev)
    sharded = distributed()

    actor_fp = make_forward_pass(config["actor_activation"], config["actor_model_type"])
    opp_fp = make_forward_pass(config["opp_activation"], config["opp_model_type"])
    params = actor_fp.init(config["seed"], device=dev)               # same weights on every rank
    if config["load_initial_model"]:                                  # ppo.py:246-248
        params = ckpt.load_params(config["initial_model_path"], config["actor_activation"], config["actor_model_type"], dev)
    broadcast_parameters(params)                                      # one broadcast at start
    opt_state = make_optimizer(config, params)

    def load_opponent(path, activation, model_type):
        return ckpt.load_params(path, activation, model_type, dev)

    latest_snapshot = {}

    def snapshot_latest(params):
        """The "latest" opponent = the learner's weights AS THEY ARE NOW (ppo.py:455-460: an immutable pytree that stays at
        the pre-update weights).  `params` is a live module that update_step changes in place, so the opponent is a copy
        in a persistent module of its own (14.7 MB, once per iteration) — never an alias."""
        net = latest_snapshot.get("net")
        if net is None:
            net = latest_snapshot["net"] = actor_fp.init(0, device=dev)
            for q in net.parameters():
                q.requires_grad_(False)
        with torch.no_grad():
            for q, src in zip(net.parameters(), params.parameters()):
                q.copy_(src.detach())
        return net

    eval_opp = (load_opponent(config["eval_opp_model_path"], config["eval_opp_activation"], config["eval_opp_model_type"])
                if config["eval_opp_model_path"] else
                make_forward_pass(config["eval_opp_activation"], config["eval_opp_model_type"]).init(config["seed"] + 1, device=dev))
    do_eval = bool(config["evaluate"])
    simple_evaluate = make_simple_evaluate(eval_env, config["actor_activation"], config["actor_model_type"],
                                           config["eval_opp_activation"], config["eval_opp_model_type"], eval_opp,
                                           config["num_eval_envs"], shard=sharded)                    # ppo.py:253-261
    simple_duplicate_evaluate = make_simple_duplicate_evaluate(
        eval_env, config["actor_activation"], config["actor_model_type"], config["actor_activation"],
        config["actor_model_type"], config["num_prioritized_envs"], shard=sharded)                  # ppo.py:262-269
    duplicate_evaluate = make_evaluate(eval_env, config["actor_activation"], config["actor_model_type"],
                                       config["eval_opp_activation"], config["eval_opp_model_type"], eval_opp,
                                       config["num_eval_envs"], game_mode=config["game_mode"], duplicate=True,
                                       shard=sharded)                                                # :270-280
    eval_rng = config["seed"] + 12345

    roll_out = brl_amd.make_roll_out(config, env, actor_fp, opp_fp)
    calc_gae = brl_amd.make_calc_gae(config, actor_fp)
    update_step = make_update_step(config, actor_fp)
    env_state = env.init(config["seed"], num_envs=config["num_envs"])                                # ppo.py:314-318
    runner_state = (params, opt_state, env_state, env_state.observation, 0, 0)

    if not config["self_play"]:                                                                      # ppo.py:335-338
        opp_params = load_opponent(config["opp_model_path"] or config["eval_opp_model_path"],
                                   config["opp_activation"], config["opp_model_type"]) \
            if (config["opp_model_path"] or config["eval_opp_model_path"]) else eval_opp
    else:
        opp_params = snapshot_latest(params)
    pool_dir = os.path.join(config["log_path"], config["exp_name"], config["save_model_path"])
    if config["save_model"] and rank == 0:
        os.makedirs(pool_dir, exist_ok=True)                                                         # ppo.py:339-346
    steps = 0
    history = []
    i = -1
    # ppo.py plays jit_simple_duplicate_evaluate(params, opp_params, eval_rng) three times per iteration (:376 imp_opp, :461
    # imp_opp_before, :480 imp_opp_after) with ONE eval_rng for the whole run (:251).  The evaluation is a deterministic function
    # of (params, opp_params, eval_rng) — arg-max play on the same boards, integer IMPs — so two of those calls repeat an earlier
    # one whenever neither network changed in between: imp_opp of iteration i + 1 is imp_opp_after of iteration i, and
    # imp_opp_before is imp_opp where the pool left the opponent alone.  Each distinct (parameter version, opponent) pair is
    # played once (~12 ms per evaluation at num_eval_envs = 10000; the logged numbers are the same).
    # (`opp_version` counts the loop's assignments to opp_params: the "latest" opponent is a persistent module rewritten in place)
    memo = {"key": None, "imp": None}
    opp_version = 0

    # config["memoize_eval"] (default on): off = the reference's three evaluations per iteration, played each time;
    # config["memoize_eval_check_every"] = N > 0: every N-th memo hit is re-played and must give the remembered IMP (a guard
    # against a future in-place change of `params` / `opp_params` outside the counted sites: the key is bookkeeping, not a checksum)
    memoize = bool(config.get("memoize_eval", True))
    check_every = int(config.get("memoize_eval_check_every", 0) or 0)
    hits = [0]

    def duplicate_imp(params, opp, version):
        key = (version, opp_version)
        if memoize and memo["key"] == key:
            hits[0] += 1
            if check_every and hits[0] % check_every == 0:
                again = float(simple_duplicate_evaluate(params, opp, eval_rng)[0][0])
                if again != memo["imp"]:
                    raise RuntimeError(f"brl_amd.train: a memoised evaluation is stale (key {key}: {memo['imp']} remembered, {again} "
                                       f"re-played): parameters changed outside the loop's counted sites — set memoize_eval=False")
            return memo["imp"]
        imp = float(simple_duplicate_evaluate(params, opp, eval_rng)[0][0])
        memo.update(key=key, imp=imp)
        return imp

    for i in range(config["num_updates"]):
        params = runner_state[0]
        if i != 0 and i % config["save_model_interval"] == 0 and config["save_model"] and rank == 0:  # ppo.py:351-362
            ckpt.save_params(params, os.path.join(pool_dir, f"params-{i:08}.pt"))
        if sharded:
            dist.barrier()  # the pool listing below must see rank 0's file
        rec = {}
        t_eval = time.perf_counter()
        if do_eval:
            rec["train/score"] = float(simple_evaluate(params, eval_rng))                            # ppo.py:366
            if i % config["num_eval_step"] == 0:                                                     # ppo.py:369-374
                log_info, _, _ = duplicate_evaluate(params, eval_rng)
                rec.update(make_evaluate_log(log_info))
        opp_name = "latest"
        if config["self_play"]:                                                                      # ppo.py:376-460
            imp_opp = duplicate_imp(params, opp_params, i)
            params_list = ckpt.list_checkpoints(pool_dir)[-int(config["num_model_zoo"]):]

            def league_imps():
                imps = np.zeros(len(params_list))
                for k, name in enumerate(params_list):                                               # ppo.py:399-421
                    other = load_opponent(os.path.join(pool_dir, name), config["actor_activation"], config["actor_model_type"])
                    imps[k] = float(simple_duplicate_evaluate(params, other, eval_rng)[0][0])
                return imps

            # rank 0 draws (its host RNG is the run's), every rank follows; a PFSP draw needs the league evaluation first,
            # which all ranks run together (sharded boards, one all-reduce per checkpoint)
            draw = opponent_draw(config, float(imp_opp), len(params_list), host_rng) if rank == 0 else None
            code = broadcast_int({None: -2}.get(draw, draw) if rank == 0 else 0, dev)
            if code == PFSP:
                imps = league_imps()
                code = broadcast_int(pfsp_draw(config, imps, host_rng) if rank == 0 else 0, dev)
            if code >= 0:
                opp_name = params_list[code]
                opp_params = load_opponent(os.path.join(pool_dir, opp_name), config["actor_activation"], config["actor_model_type"])
                opp_version += 1
            elif code == -1:
                opp_params = snapshot_latest(params)
                opp_version += 1
            else:
                opp_name = "unchanged"
        imp_before = duplicate_imp(params, opp_params, i) if do_eval else float("nan")                # ppo.py:461
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner_state, traj = roll_out(runner_state, opp_params)                                       # ppo.py:467
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if on_rollout is not None:
            on_rollout(i, runner_state, traj, roll_out)
        adv, tgt = calc_gae(runner_state, traj)                                                       # ppo.py:471
        torch.cuda.synchronize(); t2 = time.perf_counter()
        runner_state, loss_info = update_step(runner_state, traj, adv, tgt)                           # ppo.py:473
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if sharded and config.get("check_rank_sync", True):
            # every rank must hold the SAME parameters after an update (all-reduced / reduce-scattered gradients, all-gathered
            # slices): one scalar per rank, MIN and MAX over the ranks — a desynchronised run fails here, not silently later
            with torch.no_grad():
                ck = torch.stack([q.detach().double().sum() for q in runner_state[0].parameters()]).sum().reshape(1)
                lo, hi = ck.clone(), ck.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if float(lo) != float(hi):
                raise RuntimeError(f"brl_amd.train: the ranks' parameters differ after update {i} (checksum {float(lo)!r} .. {float(hi)!r})")
        imp_after = duplicate_imp(runner_state[0], opp_params, i + 1) if do_eval else float("nan")    # ppo.py:480
        steps += config["num_envs"] * config["num_steps"] * world                                     # ppo.py:489
        total_loss, (value_loss, loss_actor, entropy, approx_kl, clipfracs, illegal_action_loss) = loss_info
        board_num = int(sum_over_ranks(float(runner_state[4].item()), dev))
        rec.update({                                                                                  # ppo.py:501-519
            "train/total_loss": float(total_loss[-1][-1]), "train/value_loss": float(value_loss[-1][-1]),
            "train/loss_actor": float(loss_actor[-1][-1]), "train/illegal_action_loss": float(illegal_action_loss[-1][-1]),
            "train/policy_entropy": float(entropy[-1][-1]), "train/clipflacs": float(clipfracs[-1][-1]),
            "train/approx_kl": float(approx_kl[-1][-1]),
            "train/lr": float(linear_schedule(config, (i + 1) * config["update_epochs"] * config["num_minibatches"])),
            "train/imp_opp_before": imp_before, "train/imp_opp_after": imp_after, "board_num": board_num, "steps": steps,
            # build-side extras
            "update": i, "opponent": opp_name,
            "opp_weight_delta": _weight_delta(runner_state[0], opp_params), "hash_table": train_files[rotation.current],
            "eval_s": t0 - t_eval, "rollout_s": t1 - t0, "gae_s": t2 - t1, "update_s": t3 - t2,
            "macro_steps_per_s": config["num_envs"] * config["num_steps"] * world / (t3 - t0)})
        if rotation.advance(board_num):                                                               # ppo.py:525-549 (G14)
            env.set_lut(luts[rotation.current])
            env_state = env.init(config["seed"] + 1 + i, num_envs=config["num_envs"])   # every table re-dealt, new key
            runner_state = (runner_state[0], runner_state[1], env_state, env_state.observation, runner_state[4],
                            runner_state[5])
            rec["hash_table_next"] = train_files[rotation.current]
        history.append(rec)
        if rank == 0:
            log(json.dumps(rec))
    fused = runner_state[1].get("graphed") if isinstance(runner_state[1], dict) else None
    if hasattr(fused, "gather_optimizer_state"):
        # sharded Adam: every rank holds the moments of its own slices only — the runner_state this returns (and any checkpoint of
        # it) carries complete moments; a collective, so unconditional: every rank is here
        fused.gather_optimizer_state()
    if config["save_model"] and rank == 0:                                                            # ppo.py:550-570
        ckpt.save_params(runner_state[0], os.path.join(pool_dir, f"params-{i + 1:08}.pt"))
        ckpt.save_opt_state(runner_state[1], os.path.join(pool_dir, f"opt_state-{i + 1:08}.pt"))
    return runner_state, history


if __name__ == "__main__":
    cfg = parse_cli(sys.argv[1:])
    if cfg["save_model"] and int(os.environ.get("RANK", "0")) == 0:                                   # ppo.py:627-645
        os.makedirs(os.path.join(cfg["log_path"], cfg["exp_name"]), exist_ok=True)
        with open(os.path.join(cfg["log_path"], cfg["exp_name"], "config.json"), "w") as f:
            json.dump(cfg, f, indent=2, ensure_ascii=False)
    t_sta = time.time()
    train(cfg)
    print("training: time", time.time() - t_sta)
