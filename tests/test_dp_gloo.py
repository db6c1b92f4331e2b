"""CPU, world size 2, gloo: the flat-bucket data-parallel path (rdst_amd/dp.py) — N>1 coverage without GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3, padding=1), torch.nn.GELU(), torch.nn.Conv2d(4, 1, 3, padding=1))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rdst_amd import dp
    torch.set_num_threads(1)
    net = _tiny()
    if rank == 1:  # start from different weights: broadcast_parameters must repair it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(net)
    bucket = dp.FlatGradBucket(net.parameters())
    assert bucket.check_views()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(4, 1, 8, 8, generator=g)
    t = torch.rand(4, 1, 8, 8, generator=g)
    xs, ts = x[rank * 2:(rank + 1) * 2], t[rank * 2:(rank + 1) * 2]     # this rank's shard of the global batch
    opt = torch.optim.Adam(bucket.params, lr=1e-2)
    for it in range(2):
        if it == 0:                       # both filling protocols give the same bucket
            bucket.zero()
            F.l1_loss(net(xs), ts).backward()
        else:
            bucket.detach_grads()
            F.l1_loss(net(xs), ts).backward()
            bucket.gather()
        bucket.all_reduce_mean()
        assert bucket.check_views()
        opt.step()
    # plain numpy (pickled by value): torch tensors would travel as shared-memory handles of a dying process
    q.put((rank, bucket.flat.numpy().copy(), [p.detach().numpy().copy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single-process reference: same global batch, mean loss over 4 = mean of the two shard means
    net = _tiny()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(4, 1, 8, 8, generator=g)
    t = torch.rand(4, 1, 8, 8, generator=g)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(2):
        opt.zero_grad()
        F.l1_loss(net(x), t).backward()
        opt.step()
    flat_ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    for rank, flat, params in res:
        assert torch.allclose(torch.from_numpy(flat), flat_ref, atol=1e-6), rank   # averaged grads == full-batch grads
        for a, b in zip(params, net.parameters()):
            assert torch.allclose(torch.from_numpy(a), b.detach(), atol=1e-6)
    assert (res[0][1] == res[1][1]).all()                                 # both ranks hold identical buckets


def test_bucket_single_process_is_noop_and_zeroes():
    from rdst_amd import dp
    net = _tiny()
    b = dp.FlatGradBucket(net.parameters())
    F.l1_loss(net(torch.rand(1, 1, 8, 8)), torch.rand(1, 1, 8, 8)).backward()
    assert b.flat.abs().sum() > 0 and b.check_views()
    b.all_reduce_mean()      # no process group: no-op
    b.zero()
    assert b.flat.abs().sum() == 0 and all(p.grad.abs().sum() == 0 for p in net.parameters())
    assert b.nbytes == 4 * sum(p.numel() for p in net.parameters())


def _guard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rdst_amd.trainer import DPTrainStep
    torch.set_num_threads(1)
    net = _tiny()
    tr = DPTrainStep(net, loss_threshold=0.5)
    before = [p.detach().clone() for p in net.parameters()]
    # rank 0's local loss is below the threshold, rank 1's is not: the decision must be the same on both ranks
    local = torch.tensor(0.1 if rank == 0 else 0.9)
    keep_mixed = tr._keep_step(local)
    keep_all = tr._keep_step(torch.tensor(0.1))
    keep_nan = tr._keep_step(torch.tensor(float("nan") if rank == 0 else 0.1))
    # a whole step on the skip path (targets far away on rank 1 only): neither rank enters the gradient all-reduce,
    # nothing hangs, the iteration counter and the cost record advance as trans_sr_trainer.py:134,176-178 do
    x = torch.rand(2, 1, 8, 8)
    t = torch.rand(2, 1, 8, 8) + (100.0 if rank == 1 else 0.0)
    tr.loss_threshold = 50.0
    tr.step(x, t)
    same = all(torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))
    q.put((rank, keep_mixed, keep_all, keep_nan, tr.current_epoch, len(tr.training_epoch_costs), same))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_loss_threshold_guard_is_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, keep_mixed, keep_all, keep_nan, epoch, ncost, same in res:
        assert keep_mixed is False and keep_all is True and keep_nan is False, rank
        assert epoch == 1 and ncost == 1 and same, rank
