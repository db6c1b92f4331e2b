"""The real data-parallel path on a GPU with TWO ranks (both on cuda:0, gloo backend with device tensors): RDSTSR on the
HIP kernels + FlatGradBucket in its detach_grads()/gather() protocol (the kernels write parameter gradients straight
into the bucket views) + one all-reduce + FlatAdam, against a single process that sees the concatenated batch.
(An 8-GPU RCCL run is the driver's; this covers everything but the transport.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O

pytestmark = pytest.mark.gpu

CFG = O.make_cfg(img_size=16, in_chans=1, sr_scale=4, embed_dim=60, dense_layer_depths=[2], num_heads=[6], window_size=[8],
                 rdb_depths=[2], mlp_ratio=2.0, growth_rate=30, pre_norm=True, feature_last_operation=True)
STEPS, LR = 2, 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    g = torch.Generator().manual_seed(99)
    return torch.rand(4, 1, 16, 16, generator=g), torch.rand(4, 1, 64, 64, generator=g)


def _run(rank, world, dtype):
    """STEPS training steps; returns (flat gradient of the last step, parameters) as numpy."""
    from rdst_amd import dp, optim
    from util import build_net
    dev = torch.device("cuda:0")
    net = build_net(CFG)
    sd = O.make_weights(CFG, 5)
    if rank == 1:   # a rank that starts elsewhere: broadcast_parameters must repair it
        sd = {k: (v + 0.25 if v.dtype.is_floating_point and not k.startswith(("sub_mean", "add_mean")) else v) for k, v in sd.items()}
    net.load_state_dict(sd, strict=True)
    net.to(dev).train().set_compute_dtype(dtype)
    dp.broadcast_parameters(net)
    bucket = dp.FlatGradBucket(net.parameters())
    opt = optim.FlatAdam(bucket.params, lr=LR, betas=(0.9, 0.99), eps=1e-8, bucket=bucket)
    x, t = _data()
    n = x.shape[0] // world
    xs, ts = x[rank * n:(rank + 1) * n].to(dev), t[rank * n:(rank + 1) * n].to(dev)
    for _ in range(STEPS):
        bucket.detach_grads()                 # the HIP ops take the bucket views as gradient destinations
        F.l1_loss(net(xs), ts).backward()
        bucket.gather()
        assert bucket.check_views()
        bucket.all_reduce_mean()
        opt.step()
    torch.cuda.synchronize()
    return bucket.flat.cpu().numpy().copy(), np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in bucket.params])


def _worker(rank, world, port, dtype_name, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    try:
        g, p = _run(rank, world, getattr(torch, dtype_name))
        q.put((rank, g, p))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_two_ranks_match_single_process(dtype_name):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g_ref, p_ref = _run(0, 1, getattr(torch, dtype_name))       # this process: the whole batch, no process group
    (_, g0, p0), (_, g1, p1) = res
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)    # both ranks hold identical buckets and parameters
    # the mean of the two shard gradients is the full-batch gradient (same kernels, other batch partition: fp32 order /
    # bf16 rounding differences only)
    tol = 2e-4 if dtype_name == "float32" else 3e-2
    rel = np.linalg.norm(g0 - g_ref) / np.linalg.norm(g_ref)
    assert rel <= tol, rel
    # Adam's step is lr * sign-like for tiny |g|: compare where it is well conditioned, bound the rest by 2 lr per step
    dpar = np.abs(p0 - p_ref)
    assert dpar.max() <= 2 * LR * STEPS + 1e-6
    assert np.median(dpar) <= (1e-5 if dtype_name == "float32" else 5e-4)
