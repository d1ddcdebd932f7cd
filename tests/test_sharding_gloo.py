"""N>1 path on CPU: world_size-2 gloo processes shard a seeded batch, solve their shard (the oracle stands in for
the GPU solve -- tests may use it) and all-gather the costs; the result must equal the single-process batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(total):
    from tests.helpers import golden, oracle_system

    case = golden()["cases"]["POS_ORN_SYS"]
    rng = np.random.default_rng(7)
    q0s = np.asarray(case["problem"]["q0"])[None, :] + rng.uniform(-0.2, 0.2, (total, 7))
    return case, q0s


def _solve(case, q0):
    from tests.helpers import oracle_system, orc, u0_of

    pr = dict(case["problem"])
    pr["q0"] = list(q0)
    s = oracle_system(pr)
    return orc.solve_recursive(s, u0_of(pr), 3, True, False)["cost"]


def _worker(rank, world, port, total, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ilqr_planner_amd.sharding import gather_costs, shard_range

    case, q0s = _problem(total)
    lo, hi = shard_range(total, rank, world)
    local = torch.tensor([_solve(case, q0s[i]) for i in range(lo, hi)], dtype=torch.float64)
    full = gather_costs(local, total)
    if rank == 0:
        ret.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _bench_worker(rank, world, port, total, ret):
    """bench.py's own multi-rank plumbing (fence, timed_region, gather_step, max_over_ranks) with a stubbed solve: the stub lives here,
    in the test -- the product has no CPU path."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time

    import bench
    from ilqr_planner_amd.sharding import shard_range

    lo, hi = shard_range(total, rank, world)
    T, nx, nu = 5, 3, 2
    calls = {"n": 0}
    out = {}

    def step():  # stand-in for "solve the shard, read the costs back": instance i costs i + 0.5, X[i] = i everywhere; rank 1 is slower
        calls["n"] += 1
        time.sleep(0.02 * (1 + rank))
        cost = torch.arange(lo, hi, dtype=torch.float64) + 0.5
        X = torch.arange(lo, hi, dtype=torch.float64)[:, None, None].expand(hi - lo, T, nx).contiguous()
        U = -torch.arange(lo, hi, dtype=torch.float64)[:, None, None].expand(hi - lo, T - 1, nu).contiguous()
        out.update(bench.gather_step(cost, total, X, U))

    mine = bench.timed_region(step, 3, lambda: None, dist)
    slowest = bench.max_over_ranks(mine, dist, "cpu")
    assert calls["n"] == 3
    assert slowest >= mine - 1e-9 and slowest >= 3 * 0.02 * world - 1e-3  # the MAX over ranks is rank (world-1)'s time
    np.testing.assert_array_equal(out["cost"].numpy(), np.arange(total) + 0.5)  # every rank holds all costs, in instance order
    if rank == 0:
        assert out["X"].shape == (total, T, nx) and out["U"].shape == (total, T - 1, nu)
        np.testing.assert_array_equal(out["X"][:, 0, 0].numpy(), np.arange(total))
        np.testing.assert_array_equal(out["U"][:, -1, -1].numpy(), -np.arange(total, dtype=float))
        ret.put((mine, slowest))
    else:
        assert out["X"] is None and out["U"] is None
    dist.barrier()
    dist.destroy_process_group()


def test_bench_multirank_plumbing_under_gloo():
    """bench.py's N>1 code path on CPU: world-size-2 gloo, ragged shards (4 + 3), fence + gather of costs and trajectories + MAX of
    the per-rank wall times -- the functions bench.py's main() calls, not copies of them."""
    total, world = 7, 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    mine, slowest = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slowest >= mine


def test_bench_helpers_single_process():
    """Without a process group the helpers degrade to the single-GPU case."""
    import bench

    n = {"k": 0}
    t = bench.timed_region(lambda: n.__setitem__("k", n["k"] + 1), 4, lambda: None, None)
    assert n["k"] == 4 and t >= 0
    assert bench.max_over_ranks(1.25, None) == 1.25


def test_shard_range_partitions():
    from ilqr_planner_amd.sharding import shard_range

    for total in (0, 1, 7, 4096, 32768, 33):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_two_rank_gloo_gather_matches_single_process():
    total, world = 7, 2  # ragged: shards of 4 and 3
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    full = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case, q0s = _problem(total)
    ref = np.array([_solve(case, q0s[i]) for i in range(total)])
    np.testing.assert_array_equal(full, ref)
