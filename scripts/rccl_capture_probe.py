"""Does an RCCL collective capture into a hipGraph and replay under torch 2.10 / ROCm 7?  

One process, one GPU, `init_process_group("nccl", world_size=1)`: enough to learn whether ProcessGroupNCCL lets all_reduce /
reduce_scatter_tensor / all_gather_into_tensor be recorded by `torch.cuda.graph` (capture_error_mode "global" and
"thread_local"), whether replays give the eager values, and what a replayed collective costs beside an eager one.  The process
group is created BEFORE anything touches the GPU.  Output: one line per experiment (gpurun_out/rccl_capture_probe.txt).

    python scripts/rccl_capture_probe.py [out_file]
"""
import os
import sys
import time
import traceback

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist

N = 3_681_320   # the DeepMind MLP's flat gradient (floats)


def main(out_path):
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.cuda.set_device(dev)
    say(f"torch {torch.__version__} hip {torch.version.hip} nccl {torch.cuda.nccl.version()} world {dist.get_world_size()}")
    g = torch.ones(N, device=dev)
    shard = torch.empty(N, device=dev)
    dist.all_reduce(g)
    dist.reduce_scatter_tensor(shard, g)
    dist.all_gather_into_tensor(g, shard)
    torch.cuda.synchronize()

    def body(kind):
        g.mul_(2.0)
        if kind == "all_reduce":
            dist.all_reduce(g)
        elif kind == "rs_ag":
            dist.reduce_scatter_tensor(shard, g)
            shard.add_(1.0)
            dist.all_gather_into_tensor(g, shard)
        g.add_(1.0)

    for kind in ("all_reduce", "rs_ag"):
        for mode in ("global", "thread_local", "relaxed"):
            try:
                g.fill_(1.0)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    body(kind)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                eager = float(g[0])
                graph = torch.cuda.CUDAGraph()
                g.fill_(1.0)
                torch.cuda.synchronize()
                with torch.cuda.graph(graph, capture_error_mode=mode):
                    for _ in range(8):
                        body(kind)
                g.fill_(1.0)
                graph.replay()
                torch.cuda.synchronize()
                want = 1.0
                for _ in range(8):
                    want = want * 2.0 + (1.0 if kind == "rs_ag" else 0.0) + 1.0
                got = float(g[0])
                ok = got == want and bool((g == got).all())
                # timing: 50 replays of 8 bodies vs 400 eager bodies
                for _ in range(3):
                    graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    g.fill_(1.0)
                    graph.replay()
                torch.cuda.synchronize()
                t_graph = (time.perf_counter() - t0) / 400
                t0 = time.perf_counter()
                for _ in range(50):
                    g.fill_(1.0)
                    for _ in range(8):
                        body(kind)
                torch.cuda.synchronize()
                t_eager = (time.perf_counter() - t0) / 400
                say(f"{kind:10s} capture_error_mode={mode:12s}: captured, replay {'== eager' if ok else 'WRONG'} "
                    f"(one body eager {eager}, 8 bodies {got} want {want}); per body: graph {t_graph * 1e6:.1f} us, eager {t_eager * 1e6:.1f} us")
            except Exception as e:   # noqa: BLE001
                say(f"{kind:10s} capture_error_mode={mode:12s}: REFUSED: {type(e).__name__}: {str(e).splitlines()[0][:300]}")
                traceback.print_exc()
                torch.cuda.synchronize()
    dist.destroy_process_group()
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        with open(out_path, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rccl_capture_probe.txt")
