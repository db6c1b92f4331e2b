"""FlatAdam (one HIP launch over flat buffers) against torch.optim.Adam, the optimizer the reference
steps (utils/optim.py:30-53, models/trans_sr_trainer.py:170-173)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _nets():
    torch.manual_seed(3)
    a = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.GELU(), torch.nn.Linear(13, 5, bias=False), torch.nn.LayerNorm(5)).cuda()
    return a, copy.deepcopy(a)


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_matches_torch_adam(wd):
    from rdst_amd.optim import FlatAdam
    a, b = _nets()
    assert sum(p.numel() for p in a.parameters()) % 4 != 0          # exercises the scalar tail of the kernel
    oa = FlatAdam(a.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, milestones=[3], gamma=0.5)    # utils/optim.py:56-75
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, milestones=[3], gamma=0.5)
    g = torch.Generator(device="cuda").manual_seed(0)
    for it in range(6):
        # the SAME gradient values go to both optimizers (two backward passes could differ in the last
        # bits through GEMM kernel selection, which is not what this test is about)
        oa.zero_grad()
        ob.zero_grad()
        for pa, pb in zip(a.parameters(), b.parameters()):
            gr = torch.randn(pa.shape, device="cuda", generator=g) * 10.0 ** float(it - 3)
            pa.grad.copy_(gr)
            pb.grad = gr.clone()
        oa.step()
        ob.step()
        sa.step()
        sb.step()
        for pa, pb in zip(a.parameters(), b.parameters()):
            # fp32, same operation order as torch: a few ulp of the parameter magnitude
            torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7)
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    for k in sd_b["state"]:
        torch.testing.assert_close(sd_a["state"][k]["exp_avg"], sd_b["state"][k]["exp_avg"], rtol=1e-5, atol=1e-9)
        torch.testing.assert_close(sd_a["state"][k]["exp_avg_sq"], sd_b["state"][k]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
        assert float(sd_a["state"][k]["step"]) == float(sd_b["state"][k]["step"]) == 6.0


def test_flat_adam_resumes_from_torch_adam_state():
    """The optimizer half of a reference checkpoint.tar (models/basic_trainer.py:187-208) loads."""
    from rdst_amd.optim import FlatAdam
    a, b = _nets()
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, betas=(0.9, 0.99))
    x = torch.randn(4, 7, device="cuda")
    for _ in range(2):
        ob.zero_grad()
        b(x).square().mean().backward()
        ob.step()
    a.load_state_dict(b.state_dict())
    oa = FlatAdam(a.parameters(), lr=1e-3, betas=(0.9, 0.99))
    oa.load_state_dict(ob.state_dict())
    for o, n in ((oa, a), (ob, b)):
        o.zero_grad()
        n(x).square().mean().backward()
        o.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7)


def test_dp_train_step_matches_reference_loop_and_resumes(tmp_path):
    """DPTrainStep (flat bucket + FlatAdam + MultiStepLR) against the reference's inner loop
    (models/trans_sr_trainer.py:141-173) with torch.optim.Adam, then a checkpoint round trip."""
    from rdst_amd.trainer import DPTrainStep
    a, b = _nets()
    tr = DPTrainStep(a, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, milestones=[2], gamma=0.5)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, betas=(0.9, 0.99), eps=1e-8)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, milestones=[2], gamma=0.5)
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = [torch.randn(6, 7, device="cuda", generator=g) for _ in range(5)]
    ys = [torch.randn(6, 5, device="cuda", generator=g) for _ in range(5)]
    for i in range(3):
        la = tr.step(xs[i], ys[i])
        ob.zero_grad()
        lb = torch.nn.functional.l1_loss(b(xs[i]), ys[i])
        lb.backward()
        ob.step()
        sb.step()
        torch.testing.assert_close(la, lb.detach(), rtol=1e-5, atol=1e-6)
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-4, atol=1e-6)
    path = str(tmp_path / "checkpoint.tar")
    tr.save_checkpoint(path)
    a2, _ = _nets()
    tr2 = DPTrainStep(a2, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, milestones=[2], gamma=0.5)
    tr2.load_checkpoint(path)
    for i in range(3, 5):
        l1 = tr.step(xs[i], ys[i])
        l2 = tr2.step(xs[i], ys[i])
        torch.testing.assert_close(l1, l2, rtol=0, atol=0)       # a resumed run continues bit-identically
    for p1, p2 in zip(a.parameters(), a2.parameters()):
        assert torch.equal(p1, p2)
    # loss-threshold guard (trans_sr_trainer.py:162): nothing moves when the loss is not below the threshold
    tr3 = DPTrainStep(_nets()[0], lr=1e-3, loss_threshold=1e-12)
    w0 = [p.detach().clone() for p in tr3.net.parameters()]
    tr3.step(xs[0], ys[0])
    for p, w in zip(tr3.net.parameters(), w0):
        assert torch.equal(p, w)
