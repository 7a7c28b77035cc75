"""CPU, world_size 2, gloo: the N>1 measurement path of bench.py (replicas, barrier, max-over-ranks)."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mustafar_amd.replicas import aggregate_throughput, env_rank_world, timed_region
    assert env_rank_world() == (rank, world, rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 1 is the slow replica: every rank must report ITS time (max over ranks)
        dt = timed_region(lambda: time.sleep(0.05 + 0.25 * rank), dist=dist, device=None)
        value = aggregate_throughput(8, 10, world, dt)
        # replicas hold different data (seeded by rank) and never exchange it: no data-path collective
        torch.manual_seed(42 + rank)
        x = torch.randn(4)
        q.put((rank, dt, value, x.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_replicas_report_the_slowest_time():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, v0, x0), (r1, dt1, v1, x1) = res
    assert abs(dt0 - dt1) < 1e-9 and dt0 >= 0.29          # both ranks agree on the slow replica's time
    assert v0 == v1 == pytest.approx(2 * 8 * 10 / dt0)     # whole-job units / max time
    assert x0 != x1                                        # independent replicas
