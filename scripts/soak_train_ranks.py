"""brl_amd.train on TWO gloo ranks sharing this box's GPU for N iterations, once per form of the multi-rank gradient step: the
per-update cross-rank parameter checksum (check_rank_sync) must hold all the way, device memory must stay flat.
usage: python scripts/soak_train_ranks.py [iterations] [num_envs]"""
import os
import socket
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, iters, n_envs, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      BRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from brl_amd.train import train
    cfg = dict(num_envs=n_envs, num_steps=32, minibatch_size=1024, update_epochs=2, total_timesteps=world * n_envs * 32 * iters,
               graph_rollout=True, evaluate=True, num_eval_envs=1000, num_prioritized_envs=200, num_eval_step=5, save_model=True,
               save_model_interval=5, ratio_model_zoo=0.5, lut_len=20000, hash_size=50000, log_path=out, exp_name="soak_" + mode,
               grad_allreduce=mode, check_rank_sync=True)
    mem = []
    t0 = time.perf_counter()
    rs, hist = train(cfg, log=lambda line: mem.append(torch.cuda.memory_allocated() / 2**20))
    torch.cuda.synchronize()
    if rank == 0:
        fm = rs[1].get("graphed")
        print(f"{mode:8s}: {len(hist)} iterations on {world} gloo ranks x {n_envs} tables in {time.perf_counter() - t0:.1f} s; step object "
              f"{type(fm).__name__} (mode {getattr(fm, 'allreduce_mode', None)}, collectives in graph: {getattr(fm, 'in_graph', None)}); "
              f"rank checksums agreed after every update; allocated {mem[0]:.0f} -> {mem[-1]:.0f} MiB (peak "
              f"{torch.cuda.max_memory_allocated() / 2**20:.0f}); last loss {hist[-1]['train/total_loss']:.5f}, opponents used: "
              f"{sorted({h['opponent'].split('-')[0] for h in hist})}", flush=True)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    for mode in ("flat", "sharded"):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        mp.start_processes(worker, args=(2, port, iters, n_envs, mode, tempfile.mkdtemp()), nprocs=2, join=True, start_method="spawn")
