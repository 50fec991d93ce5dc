"""The N>1 path on CPU: two gloo ranks run the gradient-sync plan (early/late slices of the
flat arena) and the rank-strided batch sharding; result = SUM over ranks, averaged by the
optimizer's grad_scale = 1/world (refid_clip_adamw)."""
import os
import socket

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


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from refid_amd.dist import GradSync, get_dist_info, shard_batch
    from refid_amd.engine import ParamArena, param_shapes
    assert get_dist_info() == (rank, world)
    A = ParamArena(param_shapes(6, base=8), torch.device("cpu"))
    g = torch.Generator().manual_seed(rank)
    A.flat_g.copy_(torch.rand(A.total, generator=g))
    mine = A.flat_g.clone()
    sync = GradSync(A.flat_g, A.offsets)
    sync("early")
    sync("late")
    other = torch.rand(A.total, generator=torch.Generator().manual_seed(1 - rank))
    ok = torch.allclose(A.flat_g, mine + other)
    # parameter broadcast as in the model wrapper (DDP's rank-0 broadcast)
    A.flat_p.fill_(float(rank + 1))
    dist.broadcast(A.flat_p, src=0)
    ok = ok and float(A.flat_p.max()) == 1.0
    ok = ok and shard_batch(8, rank, world) == list(range(rank, 8, world))
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_grad_sync_two_ranks_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_rank_core_pinning_splits_the_allowed_cores():
    """bench.py pins each rank to its share of the cores next to its GPU (here: unknown topology -> even split of the
    cores this process may use); the shares of all ranks are disjoint and restore cleanly."""
    import os
    from refid_amd.dist import _cpulist, pin_to_local_cores
    assert _cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    before = os.sched_getaffinity(0)
    try:
        shares = []
        for r in range(4):
            os.sched_setaffinity(0, before)
            shares.append(pin_to_local_cores(r, 4, numa_nodes=[-1] * 4))
            assert os.sched_getaffinity(0) == set(shares[-1]) and shares[-1]
        if len(before) >= 4:
            assert sum(len(s) for s in shares) == len(set().union(*map(set, shares)))      # disjoint
    finally:
        os.sched_setaffinity(0, before)
