"""Replica-level helpers for multi-GPU runs.

The decode path does not shard one sequence across devices (north_star): N GPUs run N independent
replicas, one process per GPU.  `torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is used ONLY to line the
replicas up for a measurement and to agree on the slowest one -- there is no data-path collective.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Optional

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def timed_region(run: Callable[[], None], dist=None, device: Optional[torch.device] = None, reduce_on_cpu: bool = False) -> float:
    """barrier + device synchronize on both sides of `run`, then MAX of the elapsed time over all ranks."""
    def sync():
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)

    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if (device is None or reduce_on_cpu) else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_throughput(units_per_rank_per_step: int, steps: int, world: int, seconds: float) -> float:
    """Whole-job rate: units of ALL replicas divided by the slowest replica's time (weak scaling)."""
    return world * units_per_rank_per_step * steps / seconds
