"""An eager RCCL collective shortly BEFORE a hipGraph capture on the same process group: does ProcessGroupNCCL's watchdog thread survive?
(Round 5: a FusedStep warm-up that ran its collectives eagerly right before the capture aborted the process — "operation not permitted on
an event last recorded in a capturing stream", raised in the watchdog's WorkNCCL::isCompleted.)  Each case runs in its own process
(world-1 process group, one GPU):
    A  eager all_reduce, then capture a graph WITHOUT collectives
    B  eager all_reduce, then capture a graph WITH an all_reduce inside
    C  as B, with torch.cuda.synchronize() + 0.5 s between the eager collective and the capture (the watchdog polls every 100 ms)
    D  as B under TORCH_NCCL_CUDA_EVENT_CACHE=0
    E  capture WITH a collective first, eager all_reduce afterwards (the order every earlier run of this build had)
    F  as B, but the capture is LONG (0.6 s of host time with the collective at its start: the watchdog wakes up inside it)
    G  as F, with torch.cuda.synchronize() + 0.3 s in front of the capture
    H  as F, the long capture WITHOUT a collective inside (RCCL's stream is not part of the capture)
    I  as F, but the captured collective goes through a process group OF ITS OWN (dist.new_group + eager_connect_single_device:
       round 6's fence, brl_amd/fused_update.py::_capture_group) that never carries an eager collective — no synchronize, no sleep;
       REPEAT (default 20) times in one process: eager all_reduce on the default group, at once a 0.6 s capture with a collective
    J  as I, with a throw-away capture + replay of the collective on that group first (FusedStep's phase 2)
    python scripts/rccl_eager_then_capture_probe.py [out_file]
"""
import os
import subprocess
import sys
import time

BODY = r'''
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", os.environ["PORT"])
import torch, torch.distributed as dist
case = os.environ["CASE"]
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.cuda.set_device(dev)
g = torch.ones(1 << 20, device=dev)
mode = {"capture_error_mode": "thread_local"}
cap_group = None
if case in "IJ":
    cap_group = dist.new_group(backend="nccl")
    try:
        cap_group._get_backend(dev).eager_connect_single_device(dev)
    except Exception as e:
        print("eager_connect_single_device:", repr(e), flush=True)
def capture(with_collective, long=False):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, **mode):
        g.mul_(1.0)
        if with_collective:
            w = dist.all_reduce(g, group=cap_group, async_op=True); w.wait()
        g.add_(0.0)
        if long:
            for _ in range(30):           # 0.6 s of host time inside the capture, 30 more nodes
                g.add_(0.0)
                time.sleep(0.02)
    return graph
if case in "IJ":
    free0 = torch.cuda.mem_get_info()[0]
    if case == "J":
        scratch = capture(True); scratch.replay(); torch.cuda.synchronize(); del scratch
    for rep in range(int(os.environ.get("REPEAT", "20"))):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            w = dist.all_reduce(g, async_op=True); w.wait()      # eager, default group
        torch.cuda.current_stream().wait_stream(side)
        graph = capture(True, long=True)                          # at once: 0.6 s of capture with a collective on the other group
        graph.replay()
    torch.cuda.synchronize()
    print("capture group: device memory taken since its creation", (free0 - torch.cuda.mem_get_info()[0]) >> 20, "MiB", flush=True)
elif case in "ABCDFGH":
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        w = dist.all_reduce(g, async_op=True); w.wait()
    torch.cuda.current_stream().wait_stream(side)
    if case == "C":
        torch.cuda.synchronize(); time.sleep(0.5)
    if case == "G":
        torch.cuda.synchronize(); time.sleep(0.3)
    graph = capture(case not in "AH", long=case in "FGH")
else:
    graph = capture(True)
    dist.all_reduce(g)
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
time.sleep(1.0)          # several watchdog periods
dist.all_reduce(g)
torch.cuda.synchronize()
print("CASE", case, "survived", flush=True)
dist.destroy_process_group()
'''


def main(out_path):
    lines = []
    only = os.environ.get("CASES", "ABCDEFGHIJ")
    for i, case in enumerate("ABCDEFGHIJ"):
        if case not in only:
            continue
        env = dict(os.environ, CASE=case, PORT=str(29560 + i), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if case == "D":
            env["TORCH_NCCL_CUDA_EVENT_CACHE"] = "0"
        r = subprocess.run([sys.executable, "-c", BODY], env=env, capture_output=True, text=True, timeout=300)
        ok = f"CASE {case} survived" in r.stdout
        why = "" if ok else next((l.strip()[:200] for l in r.stderr.splitlines() if "HIP error" in l or "Error" in l), f"rc {r.returncode}")
        extra = " ".join(l.strip() for l in r.stdout.splitlines() if l.startswith("capture group") or l.startswith("eager_connect"))
        lines.append(f"case {case}: {'survived' if ok else 'ABORTED: ' + why}" + (f"  [{extra}]" if extra else ""))
        print(lines[-1], flush=True)
    if out_path:
        os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
        open(out_path, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rccl_eager_then_capture_probe.txt")
