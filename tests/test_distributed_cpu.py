"""World-size-2 gloo tests of the multi-GPU plumbing on CPU: the shard of series each rank
owns, the one scalar all-reduce per step, and the sample gather.  The arithmetic under it (HIP) is
covered by the -m gpu tests; here the per-rank "loss" is the oracle's MLL so the reduced value can
be checked against a single-process run."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import volt_oracle as vo
        from volt_amd import distributed as vd
        from volt_amd.synthetic import sde_batch
        lo, hi = vd.shard_range(total)
        x, F, vol = sde_batch(hi - lo, n, seed=2019, first=lo)      # this rank's series only
        K = vo.volatility_kernel(np.repeat(x[None], hi - lo, 0)[..., None], vol[..., None])
        y = np.log(F[:, 1:])
        o = vo.mll_and_grads(K, y, np.full_like(y, 2.3), 1e-5)
        local = torch.tensor([-o["mll"].sum(), -o["d_raw"].sum(), float(hi - lo)], dtype=torch.float64)
        red = vd.all_reduce_scalars(local)
        assert local[2] == hi - lo                                   # input untouched
        g = torch.tensor([1.0 + rank])
        vd.all_reduce_(g)
        samples = torch.full((hi - lo, 3, 2), float(rank))
        gathered = vd.gather_samples(samples)
        q.put((rank, lo, hi, red.tolist(), float(g), [tuple(t.shape) for t in gathered],
               [float(t.mean()) if t.numel() else -1.0 for t in gathered]))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_scalar_allreduce():
    world, total, n = 2, 5, 48
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, red0, g0, shapes0, means0), (r1, lo1, hi1, red1, g1, shapes1, means1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 3, 3, 5)                      # contiguous, remainder to low ranks
    assert red0 == red1 and red0[2] == total and g0 == g1 == 3.0
    assert shapes0 == [(3, 3, 2), (2, 3, 2)] and means0 == [0.0, 1.0]
    # single-process reference over all series
    from oracle import volt_oracle as vo
    from volt_amd.synthetic import sde_batch
    x, F, vol = sde_batch(total, n, seed=2019)
    K = vo.volatility_kernel(np.repeat(x[None], total, 0)[..., None], vol[..., None])
    y = np.log(F[:, 1:])
    o = vo.mll_and_grads(K, y, np.full_like(y, 2.3), 1e-5)
    np.testing.assert_allclose(red0[0], -o["mll"].sum(), rtol=1e-12)
    np.testing.assert_allclose(red0[1], -o["d_raw"].sum(), rtol=1e-12)


def test_single_process_helpers_are_identity():
    from volt_amd import distributed as vd
    t = torch.tensor([1.0, 2.0])
    assert vd.all_reduce_scalars(t) is t
    assert vd.shard_range(10) == (0, 10)
    assert len(vd.gather_samples(torch.zeros(2, 3, 4))) == 1
