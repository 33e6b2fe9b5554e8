"""N > 1 path on CPU: two gloo ranks own disjoint problem slices, solve them independently (here with the oracle, the
GPU is not available in this tier) and all-gather the trajectories with the same helper bench.py's RCCL path uses."""
import os
import socket

import numpy as np
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


def _worker(rank, world, port, total, n_intervals, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bipedal_control_amd import distributed as bd, scenarios
    from tests import oracle_bridge as ob
    itf = scenarios.h1_interface()
    lo, hi = bd.shard_range(total, world, rank)
    prob = scenarios.trot_problem(itf, batch=hi - lo, n_intervals=n_intervals, offset=lo)
    xs, us = [], []
    for b in range(hi - lo):
        xo, uo, _, _ = ob.oracle_solve_like(prob, b)
        xs.append(xo); us.append(uo)
    x_all, u_all = bd.gather_trajectories(torch.from_numpy(np.stack(xs)), torch.from_numpy(np.stack(us)))
    stats = bd.reduce_stats([float(hi - lo), float(np.stack(xs).sum())])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x_all.numpy(), u=u_all.numpy(), stats=stats.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range():
    from bipedal_control_amd.distributed import shard_range
    assert [shard_range(4096, 8, r) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    cover = [shard_range(10, 4, r) for r in range(4)]
    assert cover == [(0, 3), (3, 6), (6, 8), (8, 10)]


@pytest.mark.timeout(600)
def test_two_rank_gather(tmp_path):
    total, n_int = 4, 6
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, n_int, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["x"], r1["x"]) and np.array_equal(r0["u"], r1["u"])
    # equals the single-process solve of all four problems in order
    from bipedal_control_amd import scenarios
    from tests import oracle_bridge as ob
    itf = scenarios.h1_interface()
    prob = scenarios.trot_problem(itf, batch=total, n_intervals=n_int)
    for b in range(total):
        xo, uo, _, _ = ob.oracle_solve_like(prob, b)
        assert np.array_equal(r0["x"][b], xo) and np.array_equal(r0["u"][b], uo)
    assert r0["stats"][0] == total
