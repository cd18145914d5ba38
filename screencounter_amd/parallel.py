"""Read-sharded multi-GPU counting: one process per GPU, torch.distributed over RCCL.

Reads (or read pairs) are independent units, so the stream is cut into contiguous shards, one
per rank, the library is replicated on every GPU, and the only exchange step is one
``all_reduce(SUM)`` of the int32 count vector plus the read totals -- the device counterpart of the
reference's serial ``handler.reduce()`` over per-thread states
(kaori/handlers/SingleBarcodeSingleEnd.hpp:119-125, kaori/process_data.hpp:115-124) and of the
``cbind`` / ``combineComboCounts`` that follows BiocParallel in ``matrixOf*``.

The count vector is 400 KB for 100 k barcodes: latency-bound over xGMI, so it is reduced in one
collective, never bucketed.  Integer addition is order-independent, hence the N-GPU result equals
the 1-GPU result exactly.

Everything here also runs on CPU tensors with the gloo backend (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` of `world` (sizes differ by <= 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def assign_files(n_files: int, rank: int, world: int) -> Sequence[int]:
    """Round-robin file indices for `rank` (the matrixOf* case: one FASTQ per worker at a time)."""
    return list(range(rank, n_files, world))


def all_reduce_counts(counts, total: int, group=None):
    """Sum the per-rank count vector (torch int32 tensor, CUDA -> RCCL, CPU -> gloo) and totals.

    Returns (counts tensor reduced in place, global total).  The total travels as int64 next to
    the counts so that one extra tiny collective suffices."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return counts, int(total)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    t = torch.tensor([int(total)], dtype=torch.int64, device=counts.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return counts, int(t.item())


class ShardedPlan:
    """A Plan whose counters live in a torch tensor that is all-reduced across ranks on finish()."""

    def __init__(self, plan, device):
        import torch
        self.plan = plan
        self.device = torch.device(device)
        self.counters = torch.zeros(max(plan.num_counters, 1), dtype=torch.int32, device=self.device)
        plan.bind_counters(self.counters)
        plan.reset()

    def count(self, *args, **kwargs) -> None:
        self.plan.count(*args, **kwargs)

    def count_paired(self, *args, **kwargs) -> None:
        self.plan.count_paired(*args, **kwargs)

    def finish(self, group=None):
        """Synchronise, reduce over ranks, return (counts int32 ndarray [num_counters], global total)."""
        import torch
        _, local_total = self.plan.read()        # synchronises the launch stream
        counts, total = all_reduce_counts(self.counters, local_total, group)
        torch.cuda.synchronize(self.device)
        return counts[:self.plan.num_counters].cpu().numpy().astype(np.int32), total


def gather_columns(local_cols: dict, n_files: int, group=None):
    """matrixOf* across ranks: every rank counted the files of assign_files(); returns the list of
    per-file results in file order on every rank (all_gather_object: results are small)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [local_cols[i] for i in range(n_files)]
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, local_cols, group=group)
    merged = {}
    for d in gathered:
        merged.update(d)
    return [merged[i] for i in range(n_files)]
