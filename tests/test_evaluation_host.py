"""Host-side logic of the evaluators' loop (brl_amd/evaluation.py) that needs no GPU: the ladder of batch sizes the forwards on
the boards still playing step down through (`_ActiveRows.update`)."""
import torch

from brl_amd.evaluation import _ActiveRows


def ladder(n):
    """every batch size the loop can take for n boards, by feeding every possible number of finished boards in order"""
    rows, seen = _ActiveRows(n), [n]
    idx = torch.arange(n)
    for finished in range(0, n + 1):
        rows.update((finished, idx))
        if rows.m != seen[-1]:
            seen.append(rows.m)
        live = n - finished
        assert rows.m >= min(live, n) or live == 0            # never fewer rows than boards still playing
        assert rows.m >= 256 or rows.m == n                   # (small evaluations are never cut below one tile)
        assert rows.idx is None or rows.idx.numel() == rows.m
    return seen


def test_batch_sizes_step_down_a_short_ladder():
    seen = ladder(8192)
    assert seen == [8192, 6144, 4096, 3072, 2048, 1536, 1024, 768, 512, 256]     # ten GEMM shapes per layer, not thirty-two
    seen = ladder(10000)
    assert seen[0] == 10000 and seen[-1] == 256 and len(seen) <= 12
    assert all(a > b for a, b in zip(seen, seen[1:])) and all(m % 256 == 0 for m in seen[1:])
    assert ladder(300) == [300, 256]
    assert ladder(100) == [100]                                                    # nothing below one tile: full batches


def test_batch_size_never_grows_and_ignores_missing_lists():
    rows = _ActiveRows(4096)
    idx = torch.arange(4096)
    rows.update(None)
    rows.update((4000, None))          # a post without an index list (compaction off)
    assert rows.m == 4096 and rows.idx is None
    rows.update((3000, idx))           # 1096 playing -> 1536
    assert rows.m == 1536
    rows.update((100, idx))            # an older, larger count arriving late must not grow the batch again
    assert rows.m == 1536


def test_done_watch_tags_identify_loop_and_iteration():
    """The finished-board count reaches the host as (tag << 32) | count in pinned memory (brl_live_index); a tag names the loop
    (epoch) and the iteration, fits the kernel's 31 bits, and no two iterations the host could confuse share one: the ring has 4
    slots, a slot is rewritten every 4 iterations, and a new loop on the same watch takes a new epoch."""
    from brl_amd.evaluation import _DoneWatch
    w = object.__new__(_DoneWatch)
    seen = set()
    for epoch in (1, 2, 0x7FFF):
        w.epoch = epoch
        tags = [w._tag(i) for i in range(0, 5000)]
        assert all(0 < t < (1 << 31) for t in tags)
        assert len(set(tags)) == len(tags)                      # within a loop: distinct for 65535 iterations
        assert not (seen & set(tags))                           # across loops: disjoint
        seen |= set(tags)
    assert _DoneWatch.DEPTH < _DoneWatch.RING                   # the slot polled for iteration i - DEPTH is not the one being written
