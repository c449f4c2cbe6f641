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


def _replay_worker(rank, world, port, agree_collectively, q):
    """The loop driver of train_utils on CPU tensors: `iteration` all-reduces once (as TrainVoltMagpieBatch's reduce does)
    and, on rank 1 only, reports ONE failed factorisation in the middle of a deferred stretch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    try:
        from volt_amd import gp, train_utils
        p = torch.zeros(1, requires_grad=True)
        opt = torch.optim.Adam([p], lr=0.1)
        calls = {"n": 0, "collectives": 0, "trace": []}

        def iteration():
            calls["n"] += 1
            chk = gp.deferred_checks._active
            bad = 1 if (rank == 1 and calls["n"] == 17 and not chk.immediate) else 0      # fails once, while deferred
            chk.note(torch.tensor([bad], dtype=torch.int32))
            loss = (p * p).sum() + 1.0
            loss.backward()
            t = torch.tensor([float(calls["n"])])
            dist.all_reduce(t)                                   # the step's collective: pairs up with the SAME iteration?
            calls["collectives"] += 1
            calls["trace"].append(float(t))
            return loss

        def agree(bad):
            flag = torch.tensor([1.0 if bad else 0.0])
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            return bool(flag.item() > 0)

        train_utils._run_iterations(iteration, opt, 60, False, defer=True, agree=agree if agree_collectively else None)
        q.put((rank, calls["collectives"], calls["trace"]))
    except Exception as e:                                       # noqa: BLE001 -- a hang shows up as a gloo time-out
        q.put((rank, -1, repr(e)[:200]))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:                                        # noqa: BLE001
            pass


def test_deferred_replay_decision_is_collective():
    """with the info check deferred, a failed factorisation on ONE rank must make EVERY rank replay the stretch
    (each replayed iteration issues the step's all-reduce again); decided rank-locally the ranks issue different numbers
    of collectives and iterations pair up with the wrong partner.  With `agree` both ranks issue the same number of
    collectives and every all-reduce sums the SAME iteration number on both (trace entries are 2 x the local count)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replay_worker, args=(r, world, port, True, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, c0, t0), (r1, c1, t1) = res
    assert c0 == c1 > 60, (c0, c1, t0 if c0 < 0 else "", t1 if c1 < 0 else "")       # both replayed the failed stretch
    assert t0 == t1 and all(abs(v - 2 * (i + 1)) < 1e-6 for i, v in enumerate(t0))    # same iteration on both ranks, always
