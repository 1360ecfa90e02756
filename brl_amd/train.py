"""Minimal PPO self-play loop with the reference's phases (ppo.py:348-549) — BASELINE configs 4/5.

    python -m brl_amd.train num_envs=8192 num_steps=32 total_timesteps=2621440 [dds_results_dir=...]
    torchrun --nproc-per-node 8 -m brl_amd.train ...        # env shards per GPU, RCCL gradient all-reduce

NOT the experiment driver of ppo.py (wandb, pickles, FSP/PFSP pool, LUT rotation are out of scope,
SURVEY §2 row 9): just roll_out -> calc_gae -> update_step with the same config keys and defaults
(ppo.py:40-180), a duplicate evaluation against the initial weights every ``eval_interval`` updates, and
one JSON line per update.  Without ``dds_results_dir`` a synthetic LUT is used.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

DEFAULTS = dict(  # ppo.py:40-180 / README.md:68-72
    seed=0, lr=1e-5, num_envs=8192, num_steps=32, total_timesteps=8192 * 32 * 4, update_epochs=10,
    minibatch_size=1024, gamma=1.0, gae_lambda=0.95, clip_eps=0.2, ent_coef=0.001, vf_coef=0.5,
    value_clipping=True, global_gradient_clipping=True, anneal_lr=False, reward_scaling=False, max_grad_norm=0.5,
    reward_scale=7600.0, actor_illegal_action_mask=True, illegal_action_l2norm_coef=0.0,
    actor_activation="relu", actor_model_type="DeepMind", opp_activation="relu", opp_model_type="DeepMind",
    game_mode="competitive", self_play=True, num_eval_envs=1024, eval_interval=0, dds_results_dir=None,
    lut_len=100_000, inference_dtype=None,
)


def parse_cli(argv):
    cfg = dict(DEFAULTS)
    for a in argv:
        k, v = a.split("=", 1)
        if k not in cfg:
            raise SystemExit(f"unknown option {k}")
        d = DEFAULTS[k]
        cfg[k] = (v.lower() in ("1", "true")) if isinstance(d, bool) else (type(d)(v) if d is not None else v)
    return cfg


def train(config, log=print):
    import torch.distributed as dist

    import brl_amd
    from brl_amd.dist import rank_world, shard_offset, sum_over_ranks
    from brl_amd.evaluation import make_simple_duplicate_evaluate
    from brl_amd.models import make_forward_pass
    from brl_amd.update import make_optimizer, make_update_step

    rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    config = dict(config)
    config["num_updates"] = config["total_timesteps"] // config["num_steps"] // config["num_envs"]     # ppo.py:225-227
    config["num_minibatches"] = config["num_envs"] * config["num_steps"] // config["minibatch_size"]  # ppo.py:228-230

    if config["dds_results_dir"]:
        files = sorted(f for f in os.listdir(config["dds_results_dir"]) if "train" in f)   # ppo.py:297-300
        lut = brl_amd.bridge_bidding.load_dds_table(os.path.join(config["dds_results_dir"], files[0]))
    else:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import synthetic_lut
        lut = synthetic_lut(config["lut_len"], 0)
    env = brl_amd.BridgeBidding(lut=lut, device=dev, env_offset=shard_offset(rank, config["num_envs"]))
    fp = make_forward_pass(config["actor_activation"], config["actor_model_type"])
    params = fp.init(config["seed"], device=dev)               # same weights on every rank
    initial = fp.init(config["seed"], device=dev)
    opt_state = make_optimizer(config, params)
    roll_out = brl_amd.make_roll_out(config, env, fp, fp)
    calc_gae = brl_amd.make_calc_gae(config, fp)
    update_step = make_update_step(config, fp)
    # evaluation has its OWN handle: env.init(seed) re-keys a handle's RNG, which must not leak into the training tables
    eval_env = brl_amd.BridgeBidding(lut=lut, device=dev, env_offset=shard_offset(rank, config["num_eval_envs"]))
    evaluate = make_simple_duplicate_evaluate(eval_env, config["actor_activation"], config["actor_model_type"],
                                              config["opp_activation"], config["opp_model_type"], config["num_eval_envs"])
    env_state = env.init(config["seed"], num_envs=config["num_envs"])
    runner_state = (params, opt_state, env_state, env_state.observation, 0, 0)
    steps = 0
    history = []
    for i in range(config["num_updates"]):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opp_params = params if config["self_play"] else initial                     # ppo.py:344-347
        runner_state, traj = roll_out(runner_state, opp_params)                     # ppo.py:467
        torch.cuda.synchronize(); t1 = time.perf_counter()
        adv, tgt = calc_gae(runner_state, traj)                                     # ppo.py:471
        torch.cuda.synchronize(); t2 = time.perf_counter()
        runner_state, loss_info = update_step(runner_state, traj, adv, tgt)         # ppo.py:473
        torch.cuda.synchronize(); t3 = time.perf_counter()
        steps += config["num_envs"] * config["num_steps"] * world                   # ppo.py:489
        rec = {"update": i, "steps": steps, "rollout_s": t1 - t0, "gae_s": t2 - t1, "update_s": t3 - t2,
               "macro_steps_per_s": config["num_envs"] * config["num_steps"] * world / (t3 - t0),
               "total_loss": float(loss_info[0].mean()), "value_loss": float(loss_info[1][0].mean()),
               "entropy": float(loss_info[1][2].mean()), "approx_kl": float(loss_info[1][3].mean()),
               "terminated_count": sum_over_ranks(float(runner_state[4].item()), dev)}
        if config["eval_interval"] and (i + 1) % config["eval_interval"] == 0:
            (imp, se, win), _, _ = evaluate(runner_state[0], initial, 10_000 + i)
            rec.update(imp_vs_initial=float(imp), imp_se=float(se), win_rate=float(win))
        history.append(rec)
        if rank == 0:
            log(json.dumps(rec))
    return runner_state, history


if __name__ == "__main__":
    train(parse_cli(sys.argv[1:]))
