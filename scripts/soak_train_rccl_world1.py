"""brl_amd.train with EVERY collective of the multi-rank loop really issued — over RCCL with one peer (BRL_FORCE_DIST=1, world 1) — for N
iterations, once per form of the gradient step: parameter broadcast, per-iteration barrier, the sharded evaluators' all-reduces, the
opponent-index broadcast, rollout capture, the update's graph with RCCL nodes inside, rank-sync checksums, the final optimizer-state
gather.  What it exercises is the ORDER of eager collectives, captures and replays under ProcessGroupNCCL's watchdog thread on the one
GPU of a box (what crosses xGMI needs a node).  usage: python scripts/soak_train_rccl_world1.py [iterations] [num_envs]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode, iters, n_envs, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", BRL_FORCE_DIST="1",
                      BRL_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from brl_amd.train import train
    cfg = dict(num_envs=n_envs, num_steps=32, minibatch_size=1024, update_epochs=2, total_timesteps=n_envs * 32 * iters, graph_rollout=True,
               evaluate=True, num_eval_envs=1000, num_prioritized_envs=200, num_eval_step=3, save_model=True, save_model_interval=2,
               ratio_model_zoo=0.5, lut_len=20000, hash_size=30000, log_path=tempfile.mkdtemp(), exp_name="soak_" + mode,
               grad_allreduce=mode, check_rank_sync=True)
    t0 = time.perf_counter()
    rs, hist = train(cfg, log=lambda line: None)
    torch.cuda.synchronize()
    fm = rs[1].get("graphed")
    print(f"{mode:8s}: {len(hist)} iterations under a world-1 RCCL process group in {time.perf_counter() - t0:.1f} s; backend {dist.get_backend()}; step "
          f"object {type(fm).__name__} (mode {getattr(fm, 'allreduce_mode', None)}, collectives inside the graph: {getattr(fm, 'in_graph', None)}); "
          f"evaluations sharded + all-reduced, {sum('hash_table_next' in h for h in hist)} LUT rotations, opponents {sorted({h['opponent'].split('-')[0] for h in hist})}; "
          f"last loss {hist[-1]['train/total_loss']:.5f}; peak memory {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    if len(sys.argv) > 3:            # child: one form per process (a process group is created once per process)
        run(sys.argv[3], iters, n_envs, int(sys.argv[4]))
    else:
        import subprocess
        for k, mode in enumerate(("flat", "sharded")):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(iters), str(n_envs), mode, str(29570 + k)])
            if r.returncode:
                print(f"{mode}: exit code {r.returncode}", flush=True)
                sys.exit(r.returncode)
