"""The N > 1 path on CPU: world_size-2 gloo processes run the same weight-broadcast / row-sharding /
id-gather plumbing bench.py uses with RCCL (mgea/dist.py).  No GPU, no compute kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_rows, q):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "music-generation-emotion-adaptive_amd"))
    from mgea import dist as mdist
    from mgea import synth
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # one broadcast moves the whole weight arena; only rank 0 holds real weights
    n = 4096
    arena = torch.from_numpy(synth.uniform(9, "arena", (n,))) if rank == 0 else torch.zeros(n)
    mdist.broadcast_arena(arena, 0)
    ok_w = bool(np.array_equal(arena.numpy(), synth.uniform(9, "arena", (n,))))
    # prompts are sharded by contiguous rows; each rank "generates" ids that encode (row, step)
    rows = mdist.shard_rows(n_rows, rank, world)
    local = torch.tensor([[100 * b + t for t in range(5)] for b in rows], dtype=torch.int32).reshape(len(rows), 5)
    allids = mdist.gather_ids(local, n_rows)
    if rank == 0:
        want = torch.tensor([[100 * b + t for t in range(5)] for b in range(n_rows)], dtype=torch.int32)
        q.put((ok_w, bool(torch.equal(allids, want))))
    else:
        q.put((ok_w, allids is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [8, 5])
def test_broadcast_shard_gather_world2(n_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(a and b for a, b in res), res
