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
