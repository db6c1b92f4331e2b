"""Gradient ACCUMULATION through the HIP backward (round-3 advisor finding): the reduction batch a dense join opens spans
several autograd nodes, so parameter gradients are written after the node that returns them has returned.  That is only
sound when nobody reads them before the batch ends.  These tests run the flows in which autograd DOES read them at once
(``p.grad`` already defined: ``p.grad += g`` right after each node) and compare with the oracle's gradients (the fixture
``net_tiny_b4`` stores every gradient of the reference elementwise)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from util import NET_CASES, build_net, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tiny():
    cfg, seed = NET_CASES["net_tiny_b4"]
    g = load_golden("net_tiny_b4")
    net = build_net(cfg)
    net.load_state_dict(O.make_weights(cfg, seed), strict=True)
    net.to(DEV).train()
    return net, g, torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["target"]).to(DEV)


def _check(net, g, factor, tol=1e-3):
    params = dict(net.named_parameters())
    n = 0
    for k in g:
        if k.startswith("grad::"):
            ref = factor * g[k]
            got = params[k[6:]].grad.cpu().numpy()
            assert np.linalg.norm(got - ref) <= tol * max(np.linalg.norm(ref), 1e-12), k
            n += 1
    assert n > 100


def test_backward_twice_accumulates_vs_oracle():
    net, g, x, tgt = _tiny()
    for _ in range(2):
        F.l1_loss(net(x), tgt).backward()          # second pass: p.grad is defined, autograd accumulates
    torch.cuda.synchronize()
    _check(net, g, 2.0)


def test_zero_grad_keep_tensors_then_backward_vs_oracle():
    net, g, x, tgt = _tiny()
    F.l1_loss(net(x), tgt).backward()
    net.zero_grad(set_to_none=False)
    F.l1_loss(net(x), tgt).backward()
    torch.cuda.synchronize()
    _check(net, g, 1.0)


def test_bucket_zero_then_backward_vs_oracle():
    """INTEGRATION.md's documented flow: FlatGradBucket.zero() + backward accumulates into the bucket views."""
    from rdst_amd.dp import FlatGradBucket
    net, g, x, tgt = _tiny()
    bucket = FlatGradBucket(net.parameters())
    for rep in range(2):                            # twice: the second pass must not see the first one's values
        bucket.zero()
        F.l1_loss(net(x), tgt).backward()
        torch.cuda.synchronize()
        assert bucket.check_views()
        _check(net, g, 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_accumulate_equals_assign_e1_dims(dtype):
    """RDST-E1 widths (C = 60 / 90 / 120 inside every RDSTB; the one-pass bf16 kernels with their G4 slabs): gradients
    accumulated into zeroed tensors are the bits the assign-to-None flow produces (same kernels, same order)."""
    cfg = O.make_cfg(**{**O.CFG_E1, "img_size": 16, "dense_layer_depths": [2, 2], "num_heads": [6, 6],
                        "window_size": [8, 8], "rdb_depths": [3, 3]})
    net = build_net(cfg)
    net.load_state_dict(O.make_weights(cfg, 5), strict=True)
    net.to(DEV).train()
    net.set_compute_dtype(dtype)
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(2, 1, 16, 16, generator=gen).to(DEV)
    tgt = torch.rand(2, 1, 64, 64, generator=gen).to(DEV)
    F.l1_loss(net(x), tgt).backward()
    torch.cuda.synchronize()
    want = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=False)
    # dirty the allocator between the passes so that a gradient read before its deferred write would see garbage
    junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]
    del junk
    F.l1_loss(net(x), tgt).backward()
    torch.cuda.synchronize()
    for k, p in net.named_parameters():
        if k in want:
            if dtype == torch.bfloat16:      # every bf16 kernel sums in a fixed order
                assert torch.equal(p.grad, want[k]), k
            else:                            # the fp32 K2 adds its d(table) partials with LDS float atomics (order varies)
                assert (p.grad - want[k]).norm().item() <= 1e-5 * max(want[k].norm().item(), 1e-9), k


def test_dense_layer_with_stochastic_depth_in_a_later_block():
    """DenseSTLayer(drop_path=[0, 0.1]) in train(): block 0 fuses (GradSink, the join opens the layer's reduction batch),
    block 1 runs as its op-level chain (_LnLinear / _WindowAttention nodes INSIDE that open batch — the d(table) slabs
    of rdst_wattn_bwd must outlive their node).  Against the cat composition of the same layers with the same masks."""
    from rdst_amd.networks.rdst_variations import RDSTB, _apply_res_connection
    torch.manual_seed(5)
    B, H, W, C = 2, 16, 16, 60
    blk = RDSTB(C, (H, W), layer_depth=2, num_heads=6, window_size=8, mlp_ratio=2.0, img_size=H, patch_size=1,
                growth_rate=30, num_blocks=3, drop_path=[0.0, 0.1], pre_norm=True).to(DEV).train()
    keep = 1.0 / 0.9
    x0 = torch.randn(B, H * W, C, device=DEV)
    gy = torch.randn(B, H * W, C, device=DEV)

    def run(dense):
        for layer in blk.body:
            b1 = layer.body.blocks[1]
            assert not b1.fuses_input_gradient() and layer.body.blocks[0].fuses_input_gradient()
            masks = iter([torch.tensor([keep, 0.0]).view(B, 1, 1), torch.tensor([keep, keep]).view(B, 1, 1)])
            b1.drop_path.mask = (lambda it: (lambda t: next(it).to(t.device, t.dtype)))(masks)
        for p in blk.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if dense:
            y = blk(x, (H, W))
        else:
            t = x
            for m in blk.body:
                t = m(t, (H, W))
            y = _apply_res_connection(blk.conv, t.view(B, H, W, t.shape[-1]), residual=x.view(B, H, W, C),
                                      out_scale=blk.residual_scale).view(B, H * W, C)
        junk = [torch.full((1 << 20,), float("nan"), device=DEV) for _ in range(8)]
        del junk
        y.backward(gy)
        torch.cuda.synchronize()
        return [("y", y.detach()), ("dx", x.grad)] + [(k, p.grad.clone()) for k, p in blk.named_parameters()]

    a, b = run(True), run(False)
    for (k, u), (_, v) in zip(a, b):
        assert torch.isfinite(u).all(), k
        assert (u - v).norm().item() <= 1e-5 * max(v.norm().item(), 1e-6), k
    # and the accumulate flow on the same module (p.grad defined)
    first = {k: v.clone() for k, v in a[2:]}
    x = x0.clone().requires_grad_(True)
    for layer in blk.body:
        masks = iter([torch.tensor([keep, 0.0]).view(B, 1, 1), torch.tensor([keep, keep]).view(B, 1, 1)])
        layer.body.blocks[1].drop_path.mask = (lambda it: (lambda t: next(it).to(t.device, t.dtype)))(masks)
    blk(x, (H, W)).backward(gy)
    torch.cuda.synchronize()
    for k, p in blk.named_parameters():
        assert (p.grad - 2 * first[k]).norm().item() <= 1e-5 * max(first[k].norm().item(), 1e-6), k


def test_two_threads_backward_at_once_vs_oracle():
    """Two Python threads call backward() on the same device at the same time (two networks, forwards done one after the
    other): autograd's device thread serves both passes and may interleave their nodes, so a node of one pass can run
    inside the reduction batch the other pass's dense join opened.  ``_ReduceBatch`` keeps its state per thread (as the
    C side's queues are) and flushes at every node while two passes are mixed: both gradients equal the oracle's."""
    import threading
    nets, losses = [], []
    for _ in range(2):
        net, g, x, tgt = _tiny()
        nets.append(net)
        losses.append(F.l1_loss(net(x), tgt))
    torch.cuda.synchronize()
    errs = []

    def run(l):
        try:
            l.backward()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    for rep in range(3):
        if rep:
            for net in nets:
                net.zero_grad(set_to_none=True)
            losses = [F.l1_loss(net(x), tgt) for net in nets]
        ts = [threading.Thread(target=run, args=(l,)) for l in losses]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        assert not errs, errs
        for net in nets:
            _check(net, g, 1.0)
    # the counters live on autograd's DEVICE thread: read them there, from a hook on the last node of one more (solo) pass.
    # A batch the mixed passes left open would make this pass nest inside it and end one level up.
    from rdst_amd import ops
    seen = []
    net = nets[0]
    net.zero_grad(set_to_none=True)
    first = next(p for p in net.parameters() if p.requires_grad)
    h = first.register_hook(lambda g_: seen.append((ops._ReduceBatch.depth, ops._ReduceBatch.mixed)))
    F.l1_loss(net(x), tgt).backward()
    torch.cuda.synchronize()
    h.remove()
    assert seen and seen[-1] == (0, False), seen
    _check(net, g, 1.0)


def test_backward_that_raises_then_clean_pass_vs_oracle():
    """A hook raises in the middle of a backward: the dense join's reduction batch of that pass stays open on autograd's
    device thread.  The next pass must not run those queued jobs as its own and must not lose its own: without a reset it
    nests inside the dead batch and flushes at every node; ``ops.reset_backward_state()`` (what the trainer calls after a
    failed capture) makes the device thread drop the dead batch, and the counters return to zero."""
    from rdst_amd import ops
    net, g, x, tgt = _tiny()
    mid = [p for n, p in net.named_parameters() if n.endswith("attn.proj.weight")][0]   # a block inside the first layer

    def boom(_g):
        raise RuntimeError("boom")
    h = mid.register_hook(boom)
    try:
        F.l1_loss(net(x), tgt).backward()
        raised = False
    except RuntimeError:
        raised = True
    h.remove()
    assert raised
    torch.cuda.synchronize()
    for reset in (False, True):
        if reset:
            ops.reset_backward_state()
        net.zero_grad(set_to_none=True)
        seen = []
        first = next(p for p in net.parameters() if p.requires_grad)
        hk = first.register_hook(lambda g_: seen.append(ops._ReduceBatch.depth))
        F.l1_loss(net(x), tgt).backward()
        torch.cuda.synchronize()
        hk.remove()
        _check(net, g, 1.0)
        if reset:
            assert seen[-1] == 0, seen
