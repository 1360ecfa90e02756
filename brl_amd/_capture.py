"""hipGraph capture with Python's cyclic garbage collector out of the way.

A dead reference cycle that still owns a ``torch.cuda.CUDAGraph`` (an earlier rollout / update object) is freed whenever the
collector happens to run; the graph's destructor synchronises the device (torch does that on ROCm: hipGraphExecDestroy frees
lazily), and a device synchronisation while a stream of this process is capturing aborts the process.  ``torch.cuda.graph`` no
longer collects before it captures (torch.compiler.config.force_cudagraph_gc), so every capture of this package goes through
``quiet_gc``: collect once up front — cycles die BEFORE the capture — and keep the collector off until the captures are done.
"""
import contextlib
import gc


@contextlib.contextmanager
def quiet_gc():
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def graph_kwargs() -> dict:
    """``torch.cuda.graph`` keyword arguments for a capture of this package: ``capture_error_mode="thread_local"`` ONLY while an
    NCCL / RCCL process group lives — its watchdog thread queries events beside the capturing thread, which the default
    ("global") mode reports as a capture error.  Every other capture (no process group, gloo) keeps the default, where an unsafe
    call from ANY thread fails the capture instead of going unnoticed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return {"capture_error_mode": "thread_local"}
    return {}
