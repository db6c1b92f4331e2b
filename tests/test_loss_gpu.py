"""N2: the stem of the seg-UNet perceptual loss (rdst_amd.loss.SegUNet_F on stem_conv.h + segunet.hip) against the same function
in plain torch on the CPU: conv 7x7/2 -> BatchNorm2d (training mode) -> ReLU -> MSE / L1 between SR and HR features
(loss/seg_unet.py:80-107).  Parity with the reference's own UNet is unpinned (no smp, no weights in the image)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(conv, bn, sr, hr, layers, use_mse):
    lossf = F.mse_loss if use_mse else F.l1_loss
    feats = lambda x: [x, F.relu(bn(conv(x)))]          # noqa: E731
    fs = feats(sr)
    with torch.no_grad():
        fh = feats(hr)
    loss = 0
    for l in layers:
        loss = loss + lossf(fs[l], fh[l])
        loss = loss / len(layers)
    return loss


@pytest.mark.parametrize("mode,layers,cin,shape", [("encoder-L1", [1], 1, (3, 1, 40, 32)), ("encoder-L2", [1], 1, (2, 1, 33, 31)),
                                                   ("encoder-L1", [0, 1], 4, (2, 4, 24, 24))])
def test_stem_loss_vs_torch(mode, layers, cin, shape):
    from rdst_amd.loss import SegUNet_F
    torch.manual_seed(3)
    data = "BraTS" if cin == 4 else "OASIS"
    mod = SegUNet_F({mode: layers}, data, unet_path="/nonexistent", allow_random_init=True)
    with torch.no_grad():
        mod.encoder.bn1.weight.copy_(1 + 0.2 * torch.randn(64))
        mod.encoder.bn1.bias.copy_(0.1 * torch.randn(64))
    conv = nn.Conv2d(cin, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64)
    conv.load_state_dict(mod.encoder.conv1.state_dict())
    bn.load_state_dict(mod.encoder.bn1.state_dict())
    conv.train(); bn.train()
    mod.to(DEV)
    sr = torch.rand(shape)
    hr = (sr + 0.1 * torch.randn(shape)).clamp(0, 1)
    srr = sr.clone().requires_grad_(True)
    want = _reference(conv, bn, srr, hr, layers, "L1" in mode)
    (3.0 * want).backward()
    srg = sr.to(DEV).requires_grad_(True)
    got, rep = mod(srg, hr.to(DEV))
    (3.0 * got).backward()
    torch.cuda.synchronize()
    assert abs(got.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item())) + 1e-7
    assert list(rep) == ["SegUNet({})".format(mode)] and abs(rep["SegUNet({})".format(mode)] - got.item()) < 1e-9
    rel = (srg.grad.cpu() - srr.grad).norm().item() / srr.grad.norm().item()
    assert rel <= 2e-4, rel
    assert torch.allclose(mod.encoder.bn1.running_mean.cpu(), bn.running_mean, atol=1e-6)
    assert torch.allclose(mod.encoder.bn1.running_var.cpu(), bn.running_var, rtol=1e-5, atol=1e-7)
    assert int(mod.encoder.bn1.num_batches_tracked) == int(bn.num_batches_tracked) == 2


def test_bad_modes_and_missing_gpu_say_so():
    from rdst_amd.loss import SegUNet_F
    with pytest.raises(ValueError):
        SegUNet_F({"labels": []}, "OASIS", allow_random_init=True)                 # loss/seg_unet.py:124-125
    with pytest.raises(ValueError):
        SegUNet_F({"encoder-L1": [6]}, "OASIS", allow_random_init=True)            # six feature maps: 0..5
    with pytest.raises(ValueError, match="Pre-trained UNet not exist"):
        SegUNet_F({"encoder-L1": [1]}, "OASIS", unet_path="/nonexistent")           # :47-48, unless allow_random_init
    mod = SegUNet_F({"encoder-L1": [1]}, "OASIS", allow_random_init=True)
    with pytest.raises(RuntimeError):
        mod(torch.rand(1, 1, 16, 16), torch.rand(1, 1, 16, 16))   # CPU tensors: no fallback
