"""BASELINE.json configs[4] end to end: RDSTSR with the seg-UNet loss IN ITS BACKWARD PATH, as the reference trainer runs its
'UNet-F' training state (config_files/RDST_E1_OASIS_example_SRx4.ini:34,46; loss/sr_loss.py:35-51):
    loss = 0.1 * L1(sr, hr) + 1 * SegUNet_F(sr, hr)
for the shipped layer set ({'encoder-L1': [1]}, RDST-E1) and for 'label-hr' (RDST-HRL: resnet34-UNet + multiclass Dice).
Checked against the composition of the two CPU oracles (rdst_oracle: pinned to the reference; segunet_oracle: parity
unpinned, see its header) — every one of the 750 parameter gradients — and through the trainer-step shell with the step
captured into a HIP graph (no host sync inside)."""
import types

import pytest
import torch
import torch.nn.functional as F

from oracle import rdst_oracle as O
from oracle import segunet_oracle as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _paras(mode, layers):
    return types.SimpleNamespace(gpu_id=0, precision=False, training_losses=["L1", "UNet-F"],
                                 loss_scalars={"WarmUP": {"L1": 1}, "UNet-F": {"L1": 0.1, "UNet-F": 1}},
                                 training_states=["WarmUP", "UNet-F"], unet_loss_layers={mode: layers}, unet_loss_mode="OASIS",
                                 unet_path="/nonexistent", unet_allow_random=True)


def _loss(mode, layers, dtype, seed=2):
    from rdst_amd.loss import SRLoss
    sl = SRLoss(_paras(mode, layers))
    usd = S.make_unet_weights(1, 4, seed)
    sl.loss_functions["UNet-F"].load_state_dict(usd, strict=True)
    sl.loss_functions["UNet-F"].set_compute_dtype(dtype)
    sl.set_training_state("UNet-F")
    return sl, usd


@pytest.mark.parametrize("mode,layers,dtype", [("encoder-L1", [1], torch.float32), ("encoder-L1", [1], torch.bfloat16),
                                               ("label-hr", [], torch.float32), ("label-hr", [], torch.bfloat16)])
def test_config5_train_step_all_gradients_vs_oracle(mode, layers, dtype):
    from util import build_net
    cfg = O.CFG_E1
    B = 2
    sd = O.make_weights(cfg, 21)
    net = build_net(cfg)
    net.load_state_dict(sd, strict=True)
    net.to(DEV).train().set_compute_dtype(dtype)
    sl, usd = _loss(mode, layers, dtype)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(B, 1, 64, 64, generator=g)
    tgt = torch.rand(B, 1, 256, 256, generator=g)
    y = net(x.to(DEV))
    loss, rep = sl(y, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert set(rep.keys()) == {"Rec_L1", "SegUNet({})".format(mode)}

    params = dict(net.named_parameters())
    osd = {k: (v.clone().requires_grad_(True) if (k in params and params[k].requires_grad) else v) for k, v in sd.items()}
    oy = O.rdstsr_forward(x, osd, cfg)
    l1 = F.l1_loss(oy, tgt)
    lu = S.segunet_loss(oy, tgt, usd, mode, layers)
    oloss = 0.1 * l1 + 1 * lu
    oloss.backward()
    assert abs(rep["Rec_L1"] - l1.item()) <= (1e-5 if dtype == torch.float32 else 2e-3)
    lu_tol = {(torch.float32, "encoder-L1"): 1e-4, (torch.float32, "label-hr"): 1e-3,
              (torch.bfloat16, "encoder-L1"): 2e-2, (torch.bfloat16, "label-hr"): 3e-2}[(dtype, mode)]
    assert abs(rep["SegUNet({})".format(mode)] - lu.item()) <= lu_tol * max(1.0, abs(lu.item())), (rep["SegUNet({})".format(mode)], lu.item())

    tot_d = tot_r = 0.0
    worst, n, dot, ng, nr = (0.0, None), 0, 0.0, 0.0, 0.0
    for k, p in params.items():
        if not p.requires_grad:
            continue
        ref = osd[k].grad
        got = p.grad.float().cpu()
        d, rn = (got - ref).norm().item(), ref.norm().item()
        tot_d += d * d; tot_r += rn * rn; n += 1
        dot += (got * ref).sum().item(); ng += got.norm().item() ** 2; nr += rn * rn
        if d / max(rn, 1e-12) > worst[0]:
            worst = (d / max(rn, 1e-12), k)
    total = (tot_d / tot_r) ** 0.5
    cos = dot / (ng * nr) ** 0.5
    print(f"\nconfig 5 {mode} {dtype}: loss {loss.item():.6f} vs {oloss.item():.6f}; {n} gradients: total rel L2 {total:.2e}, "
          f"cosine {cos:.5f}, worst {worst[0]:.2e} ({worst[1]})")
    assert n == 750
    if dtype == torch.float32:
        assert total <= (1e-3 if mode == "encoder-L1" else 1e-2), total
        assert worst[0] <= (5e-3 if mode == "encoder-L1" else 5e-2), worst
    else:
        # bf16 throughput mode.  'label-hr': the HR labels are an argmax of bf16 logits, so a small share of boundary pixels
        # carries another label than in fp32 — the Dice gradient is compared by direction and size, not element by element
        assert total <= (3e-2 if mode == "encoder-L1" else 0.35), total
        assert cos >= (0.999 if mode == "encoder-L1" else 0.94), cos


def test_unetf_state_through_graph_captured_trainer_step_equals_eager():
    """DPTrainStep(loss_fn=SRLoss, graph=True): the 'UNet-F' step replayed from a HIP graph lands on the SAME parameters,
    BatchNorm statistics and loss as the eager step (bit for bit: every reduction has a fixed order), and the per-component
    report stays lazy (device scalars) until it is read."""
    from rdst_amd.loss import LazyScalars
    from rdst_amd.trainer import DPTrainStep
    from util import build_net
    cfg = O.make_cfg(img_size=16, in_chans=1, sr_scale=4, embed_dim=60, dense_layer_depths=[2], num_heads=[6], window_size=[8],
                     rdb_depths=[2], mlp_ratio=2.0, growth_rate=30, pre_norm=True, feature_last_operation=True)
    g = torch.Generator().manual_seed(5)
    data = [(torch.rand(2, 1, 16, 16, generator=g).to(DEV), torch.rand(2, 1, 64, 64, generator=g).to(DEV)) for _ in range(5)]
    res = {}
    for use_graph in (False, True):
        net = build_net(cfg)
        net.load_state_dict(O.make_weights(cfg, 9), strict=True)
        net.to(DEV).train().set_compute_dtype(torch.bfloat16)
        sl, _ = _loss("label-hr", [], torch.bfloat16)
        tr = DPTrainStep(net, lr=1e-3, loss_fn=sl, graph=use_graph, graph_warmup=2)
        losses = []
        for x, t in data:
            losses.append(tr.step(x, t).clone())
        torch.cuda.synchronize()
        assert (tr.graph is not None) == use_graph
        assert isinstance(tr.last_report, LazyScalars) and isinstance(tr.last_report.raw("Rec_L1"), torch.Tensor)
        unet = sl.loss_functions["UNet-F"]
        res[use_graph] = ([l.item() for l in losses], tr.optimizer.flat_param.clone(),
                          unet.decoder.blocks[4].conv2[1].running_var.clone(), int(unet.encoder.bn1.num_batches_tracked))
        ck = tr.checkpoint()
        assert "UNet-F" in ck["loss"] and "tail.0.weight" in ck["loss"]["UNet-F"]      # basic_loss.py:77-88 layout
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])
    assert res[False][3] == res[True][3] == 10
