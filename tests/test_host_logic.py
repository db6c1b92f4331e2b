"""CPU: host-side mirror of the reference's nn.Module API (no GPU compute)."""
import ctypes
import json
import os
import re
import types

import pytest
import torch
import torch.nn as nn

from oracle import rdst_oracle as O
from util import GOLDEN, NET_CASES, build_net, load_golden

ROOT = os.path.dirname(GOLDEN.rstrip("/").rsplit("/", 1)[0])


def test_library_exports_every_declared_symbol():
    from rdst_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rdst_hip.h")).read()
    declared = set(re.findall(r"\b(rdst_\w+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().rdst_abi_version() == _lib.ABI_VERSION == 11


@pytest.mark.parametrize("name", sorted(NET_CASES))
def test_state_dict_matches_reference_layout(name):
    cfg, seed = NET_CASES[name]
    g = load_golden(name)
    mean = g["mean"].tolist() if "mean" in g else None
    std = g["std"].tolist() if "std" in g else None
    net = build_net(cfg, mean, std)
    ref = json.load(open(os.path.join(GOLDEN, f"state_dict_{name}.json")))
    mine = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    assert list(mine) == list(ref["entries"])          # same keys, same order
    assert mine == ref["entries"]                      # same shapes and dtypes
    assert sum(p.numel() for p in net.parameters()) == ref["n_params"]
    assert sum(p.numel() for p in net.parameters() if p.requires_grad) == ref["n_trainable"]
    assert [k for k, p in net.named_parameters() if p.requires_grad] == ref["trainable"]
    # strict load of reference-layout weights; buffers we build ourselves equal the oracle's restatement
    sd = O.make_weights(cfg, seed, mean, std)
    fresh = net.state_dict()
    for k, v in sd.items():
        if k.endswith("relative_position_index") or k.endswith("attn_mask") or "_mean." in k:
            assert torch.equal(fresh[k], v), k
    net.load_state_dict(sd, strict=True)


def test_init_matches_reference_rules():
    net = build_net(O.CFG_TINY)
    for m in net.modules():
        if isinstance(m, nn.Linear):
            # trunc_normal_(std=.02) with timm's absolute cut at +-2 (rdst_variations.py:1311) ~ N(0, .02).  timm's
            # recipe (uniform_(-1, 1) -> erfinv -> clamp) maps the closed lower end of the uniform to -inf and the clamp
            # turns it into a weight of exactly -2: about one construction in 150 holds such an element (measured), and
            # ONE of them lifts the std of a 72 x 144 matrix from 0.020 to 0.028 — so the spread is checked on the bulk.
            w = m.weight.detach()
            assert w.abs().max().item() <= 2.0
            bulk = w[w.abs() < 0.5]
            assert bulk.numel() >= w.numel() - 2 and 0.015 < bulk.std().item() < 0.025 and torch.all(m.bias == 0)
        if isinstance(m, nn.LayerNorm):
            assert torch.all(m.weight == 1) and torch.all(m.bias == 0)
    assert all(not p.requires_grad for p in list(net.sub_mean.parameters()) + list(net.add_mean.parameters()))


def test_constructor_validation():
    from rdst_amd.networks.rdst_variations import RDSTSR, DenseSTLayer
    with pytest.raises(ValueError):
        RDSTSR(act_in_conv="swish")
    with pytest.raises(ValueError):
        RDSTSR(in_chans=1, mean=[0., 0.], std=[1., 1.])
    with pytest.raises(AssertionError):
        RDSTSR(rdb_depths=[3, 3], window_size=[4, 4, 4, 4])
    with pytest.raises(AssertionError):
        DenseSTLayer(input_dim=50, input_resolution=(8, 8), num_heads=6, growth_rate=30)
    with pytest.raises(NotImplementedError):
        RDSTSR(scale_free=True)
    net = RDSTSR(drop_path_rate=0.1, use_checkpoint=True)   # accepted and inert, as in the reference
    assert net.drop_path_rate == 0.1


def test_make_rdstsr_from_paras_namespace():
    from networks.swinIR_variations import make_RDSTSR        # the path the reference's trainer imports
    from networks.rdst_variations import make_RDSTSR as mk2
    assert make_RDSTSR is mk2
    p = types.SimpleNamespace(
        patch_size=24, input_channel=1, sr_scale=4.0, swin_patch_size=1, rdst_pre_norm=True,
        rdst_global_bottleneck=False, rdst_global_bottleneck_ratio=1., rdst_feature_last_operation=True,
        swin_hidden_ratio=2., swin_qkv_bias=True, swin_qk_scale=None, swin_drop_rate=0., swin_attn_drop_rate=0.,
        swin_drop_path_rate=0.1, rdst_embed_dim=60, rdst_dense_layer_depths=[2] * 8, rdst_num_heads=[6] * 8,
        rdst_window_size=[8] * 8, rdst_rdb_depths=[3] * 8, rdst_layer_norm=True, rdst_ape=False,
        rdst_patch_norm=True, rdst_use_checkpoint=False, rdst_res_connection='1conv', rdst_growth_rate=30,
        rdst_dense_scale=1., rdst_dim_modify_mode='tail', rdst_rdb_residual_scale=1., rdst_global_res_scale=1.,
        rdst_act_in_conv='leaky_relu', rdst_bn_in_conv=None, scale_free=False)
    net = make_RDSTSR(p, mean=[0.], std=[1.])
    assert sum(q.numel() for q in net.parameters()) == 4464965 and len(net.state_dict()) == 826
    assert net.sr_scale == 4 and isinstance(net.sr_scale, int)


def test_cpu_forward_fails_loudly():
    net = build_net(O.CFG_TINY)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 16, 16))


def test_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rdst_amd")
    for dp, _dn, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dp, fn)


@pytest.mark.parametrize("name", ["swinir_ps_x4", "swinir_psd_x2_rgb", "swinir_denoise", "swinir_nearest_x4", "rdstsr_n_mlp", "rdstsr_n_conv",
                                  "rdstsr_head_pre", "rdstsr_head_post", "rdstsr_identity_norm", "rdstsr_ape", "rdstsr_qk_scale"])
def test_next_row_models_state_dict_layout(name):
    """SwinIR baseline, RDSTSR_N ("next" rows) and the RDSTSR constructor branches ('head' dim modifier, nn.Identity norms, ape,
    qk_scale): same state-dict keys / order / shapes / dtypes as the reference."""
    from util import MODEL_CASES, model_kwargs
    kind, kw, _x, _seed, _train = MODEL_CASES[name]
    if kind == "swinir":
        from networks.swin_transformer_sr import SwinIR as cls
    elif kind == "rdstsr":
        from networks.rdst_variations import RDSTSR as cls
    else:
        from networks.rdst_variations import RDSTSR_N as cls
    net = cls(**model_kwargs(kw))
    ref = json.load(open(os.path.join(GOLDEN, f"state_dict_{name}.json")))["entries"]
    mine = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    assert list(mine) == list(ref) and mine == ref


def test_flat_adam_state_dict_layout_and_cpu_refusal():
    """FlatAdam keeps torch.optim.Adam's state-dict layout (so reference checkpoints load) and, like every
    product path, refuses to run without the GPU library path (no CPU fallback)."""
    import pytest
    from rdst_amd.optim import FlatAdam
    net = torch.nn.Sequential(torch.nn.Linear(4, 5), torch.nn.Linear(5, 3))
    ref = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
    net(torch.randn(2, 4)).sum().backward()
    ref.step()
    w0 = net[0].weight.detach().clone()
    opt = FlatAdam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
    assert torch.equal(net[0].weight, w0)                       # flattening keeps the values
    assert net[0].weight.data_ptr() == opt.flat_param.data_ptr()  # ... and the module now reads the flat buffer
    opt.load_state_dict(ref.state_dict())
    sd, rd = opt.state_dict(), ref.state_dict()
    assert sd["param_groups"][0]["betas"] == rd["param_groups"][0]["betas"]
    for k in rd["state"]:
        assert torch.equal(sd["state"][k]["exp_avg"], rd["state"][k]["exp_avg"])
        assert float(sd["state"][k]["step"]) == 1.0
    assert opt.bucket.check_views()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()


def test_trainer_checkpoint_layout_and_resume_from_reference_style_checkpoint(tmp_path):
    """DPTrainStep writes the reference's checkpoint keys (models/basic_trainer.py:187-208) and resumes from a
    checkpoint assembled the way the reference trainer does (model + torch.optim.Adam + MultiStepLR state dicts)."""
    from rdst_amd.trainer import DPTrainStep
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(4, 6), nn.Linear(6, 2))
    ref_opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.99), eps=1e-8)
    ref_sch = torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones=[2, 4], gamma=0.5)
    for _ in range(3):
        ref_opt.zero_grad()
        net(torch.randn(5, 4)).abs().mean().backward()
        ref_opt.step()
        ref_sch.step()
    ref_ck = {"Time": "x", "model_g": {k: v.clone() for k, v in net.state_dict().items()}, "optimizer_g": ref_opt.state_dict(),
              "scheduler_g": ref_sch.state_dict(), "loss": {}, "training_loss_names": ["L1"],
              "training_loss_records": {"L1": [0.3, 0.2, 0.1]}, "quick_validation_reports": [],
              "current_training_state_id": 0, "current_epoch": 3, "training_epoch_costs": [0.1, 0.1, 0.1]}
    path = str(tmp_path / "checkpoint.tar")
    torch.save(ref_ck, path)

    net2 = nn.Sequential(nn.Linear(4, 6), nn.Linear(6, 2))
    tr = DPTrainStep(net2, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, milestones=[2, 4], gamma=0.5)
    tr.load_checkpoint(path)
    for k, v in ref_ck["model_g"].items():
        assert torch.equal(net2.state_dict()[k], v)
    assert net2[0].weight.data_ptr() == tr.optimizer.flat_param.data_ptr()      # still views of the flat buffer
    assert tr.current_epoch == 3 and tr.training_loss_records["L1"] == [0.3, 0.2, 0.1]
    tr._pending = {"L1": [9.0]}            # a step taken before a (second) load belongs to the history the load replaces
    tr.load_checkpoint(path)
    assert tr.loss_records()["L1"] == [0.3, 0.2, 0.1] and tr._pending == {}
    assert tr.optimizer.param_groups[0]["lr"] == ref_opt.param_groups[0]["lr"] == 1e-4 * 0.5   # past milestone 2
    ck = tr.checkpoint()
    assert set(ref_ck) <= set(ck)                                                # every reference key is written
    for k in ref_opt.state_dict()["state"]:
        assert torch.equal(ck["optimizer_g"]["state"][k]["exp_avg"], ref_opt.state_dict()["state"][k]["exp_avg"])
    tr.save_checkpoint(str(tmp_path / "ck2.tar"))
    back = torch.load(str(tmp_path / "ck2.tar"), weights_only=False)
    assert back["current_epoch"] == 3 and "scheduler_g" in back


def test_metrics_psnr_ssim_definitions():
    """rdst_amd.metrics restates skimage's PSNR / SSIM (metrics/sr_metrics.py:8-14); PSNR must equal the oracle's (which
    the golden fixtures pin), SSIM is checked on closed-form cases (scikit-image is not installed: parity unpinned)."""
    import numpy as np
    from rdst_amd.metrics import SRMetrics, psnr, ssim
    from oracle import rdst_oracle as O
    rng = np.random.default_rng(0)
    gt = rng.random((2, 1, 40, 32)).astype(np.float32)
    pr = np.clip(gt + 0.05 * rng.standard_normal(gt.shape).astype(np.float32), 0, 1)
    rep = SRMetrics("psnr ssim")(torch.from_numpy(gt), torch.from_numpy(pr), margin=4)
    for i in range(2):
        assert abs(rep["psnr"][i] - O.psnr(torch.from_numpy(gt[i:i + 1]), torch.from_numpy(pr[i:i + 1]), 4)) < 1e-9
    assert ssim(gt[0, 0], gt[0, 0]) == pytest.approx(1.0, abs=1e-12)
    # constant images a, b: every local variance is zero -> SSIM = (2ab + C1) / (a^2 + b^2 + C1)
    a, b = 0.3, 0.5
    want = (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)
    assert ssim(np.full((16, 16), a), np.full((16, 16), b)) == pytest.approx(want, rel=1e-12)
    # y = x + c keeps the structure term at 1: SSIM = luminance term averaged over the windows
    x = rng.random((24, 24))
    s = ssim(x, x + 0.1)
    assert 0.0 < s < 1.0 and ssim(x, x + 0.1) > ssim(x, x + 0.3)
    assert psnr(np.zeros((8, 8)), np.full((8, 8), 0.1)) == pytest.approx(20.0, abs=1e-9)
    # numpy (N, H, W, C) inputs and the mean mode, as the reference's docstring uses them
    m = SRMetrics("psnr", "mean")(gt.transpose(0, 2, 3, 1), pr.transpose(0, 2, 3, 1), margin=0)
    assert isinstance(m["psnr"], float)
    with pytest.raises(ValueError):
        SRMetrics("fid")


# ---- round 3: the loss side of the boundary (loss/sr_loss.py, loss/seg_unet.py), host logic only ---------------------------
def _loss_paras(**kw):
    base = dict(gpu_id=-1, precision=False, training_losses=["L1"], loss_scalars={"WarmUP": {"L1": 1}}, training_states=["WarmUP"])
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_srloss_weights_states_and_lazy_report():
    """SRLoss mirrors loss/sr_loss.py:35-51 (weighted sum over the current state's scalars) and loss/basic_loss.py:94-95
    (set_training_state); its report converts to floats only when read; components outside section 8 say so."""
    from rdst_amd.loss import LazyScalars, SRLoss
    sl = SRLoss(_loss_paras(training_losses=["L1", "MSE"], loss_scalars={"A": {"L1": 1}, "B": {"L1": 0.25, "MSE": 2.0}},
                            training_states=["A", "B"]))
    assert sl.loss_components == ["Rec_L1", "Rec_MSE"] and sl.current_training_state == "A"
    g = torch.Generator().manual_seed(0)
    p, t = torch.rand(2, 1, 8, 8, generator=g), torch.rand(2, 1, 8, 8, generator=g)
    la, ra = sl(p, t)
    assert torch.allclose(la, torch.nn.functional.l1_loss(p, t)) and list(ra.keys()) == ["Rec_L1"]
    sl.set_training_state("B")
    lb, rb = sl(p, t)
    want = 0.25 * torch.nn.functional.l1_loss(p, t) + 2.0 * torch.nn.functional.mse_loss(p, t)
    assert torch.allclose(lb, want)
    assert isinstance(rb, LazyScalars) and isinstance(rb.raw("Rec_MSE"), torch.Tensor)       # still a device scalar ...
    assert isinstance(rb["Rec_MSE"], float) and abs(rb["Rec_MSE"] - torch.nn.functional.mse_loss(p, t).item()) < 1e-7
    assert dict(rb.items()).keys() == {"Rec_L1", "Rec_MSE"} and all(isinstance(v, float) for v in rb.values())
    assert sl.state_dict() == {}                       # RecLoss carries no state (basic_loss.py:77-88)
    with pytest.raises(NotImplementedError):
        SRLoss(_loss_paras(training_losses=["VGG54"], loss_scalars={"WarmUP": {"VGG54": 1}}))


def test_segunet_f_state_dict_is_smp_layout_and_loads_smp_checkpoint(tmp_path):
    """SegUNet_F carries smp 0.3.x's resnet34-UNet keys (24.4 M parameters), loads a checkpoint written under smp's own names
    (segmentation_head.* -> tail.*, loss/seg_unet.py:57) strictly, raises on a missing file like loss/seg_unet.py:47-48, and
    freezes the UNet.  (Parity with smp itself is unpinned: smp is not in the image.)"""
    from oracle import segunet_oracle as S
    from rdst_amd.loss import SegUNet_F
    lay = S.unet_layout(1, 4)
    mod = SegUNet_F({"label-hr": []}, "OASIS", unet_path="/nonexistent", allow_random_init=True)
    sd = mod.state_dict()
    assert list(sd) == list(lay) and all(tuple(sd[k].shape) == lay[k] for k in lay)
    assert sum(p.numel() for p in mod.parameters()) == 24430532 - 0 * 1      # smp.Unet('resnet34', in_channels=1, classes=4)
    assert all(not p.requires_grad for p in mod.parameters())
    w = S.make_unet_weights(1, 4, 5)
    path = str(tmp_path / "unet_oasis.pt")
    torch.save({("segmentation_head." + k[5:] if k.startswith("tail.") else k): v for k, v in w.items()}, path)
    m2 = SegUNet_F({"encoder-L1": [1, 2]}, "OASIS", unet_path=path)
    assert torch.equal(m2.tail[0].weight, w["tail.0.weight"]) and torch.equal(m2.encoder.layer3[5].bn2.running_var, w["encoder.layer3.5.bn2.running_var"])
    assert m2.use_mse and m2.loss_names == ["SegUNet(encoder-L1)"]          # 'L1' in the name selects MSELoss (:73-74)
    assert not SegUNet_F({"encoder-L2": [1]}, "OASIS", unet_path=path).use_mse
    assert SegUNet_F({"label-gt": []}, "BraTS tumor_only", allow_random_init=True).dice_classes == [1, 2, 3]      # :40-44
    with pytest.raises(ValueError, match="Pre-trained UNet not exist"):
        SegUNet_F({"label-hr": []}, "OASIS", unet_path=str(tmp_path / "missing.pt"))
    with pytest.raises(ValueError):
        SegUNet_F({"label-hr": []}, "MARS", allow_random_init=True)


def test_dropout_probabilities_are_validated_like_nn_dropout():
    """Dropout p > 0 is accepted by the constructors (reference: swin_transformer_sr.py:21, :102, :105); p outside [0, 1) is
    refused at construction time (1.0 would divide by zero in the attention kernels' 1 / (1 - p))."""
    import pytest
    from rdst_amd.networks.swin_transformer_sr import Mlp, SwinTransformerBlock, WindowAttention
    assert Mlp(60, 120, drop=0.1).drop.p == 0.1
    assert WindowAttention(60, (8, 8), 6, attn_drop=0.2, proj_drop=0.3).active_attn_drop() == 0.2
    blk = SwinTransformerBlock(60, (16, 16), 6, window_size=8, shift_size=4, drop=0.1, attn_drop=0.1)
    assert blk._stochastic() and not blk.fuses_input_gradient()
    blk.eval()
    assert not blk._stochastic() and blk.attn.active_attn_drop() == 0.0
    for bad in (1.0, -0.1, 1.5):
        with pytest.raises(ValueError):
            Mlp(60, 120, drop=bad)
        with pytest.raises(ValueError):
            WindowAttention(60, (8, 8), 6, attn_drop=bad)


def test_compute_dtype_switch_and_dtype_codes():
    """set_compute_dtype: torch.float32 / 'fp32' = exact parity mode, 'fp32x3' = fp32 tensors with the split-bf16 GEMMs (the C ABI's
    RDST_F32X3), torch.bfloat16 / 'bf16' = throughput mode; anything else is refused.  The mode is the MODULE's
    (``compute_code``, applied by ``ops.compute_scope`` around its forward): setting it on one network changes neither another
    network nor the process default ``ops.F32_SPLIT`` (which only governs fp32 ops called outside any network)."""
    from rdst_amd import _lib, ops
    net, other = build_net(O.CFG_TINY), build_net(O.CFG_TINY)
    x32, xbf = torch.zeros(2, 4), torch.zeros(2, 4, dtype=torch.bfloat16)
    try:
        assert net.compute_dtype == torch.float32 and net.compute_code == _lib.F32 and not ops.F32_SPLIT
        assert ops._dtype_code(x32) == _lib.F32 and ops._dtype_code(xbf) == _lib.BF16
        assert net.set_compute_dtype("fp32x3") is net
        assert net.compute_dtype == torch.float32 and net.compute_code == _lib.F32X3 == 2
        assert not ops.F32_SPLIT and other.compute_code == _lib.F32           # nobody else changed
        assert ops._dtype_code(x32) == _lib.F32                                # outside the network: the default
        with ops.compute_scope(net.compute_code):                              # what net.forward opens
            assert ops._dtype_code(x32) == _lib.F32X3 and ops._dtype_code(xbf) == _lib.BF16
            with ops.compute_scope(other.compute_code):                        # a nested forward of the other network
                assert ops._dtype_code(x32) == _lib.F32
            assert ops._dtype_code(x32) == _lib.F32X3
        assert ops._dtype_code(x32) == _lib.F32
        net.set_compute_dtype("bf16")
        assert net.compute_dtype == torch.bfloat16 and net.compute_code == _lib.BF16
        with ops.compute_scope(net.compute_code):
            assert ops._dtype_code(x32) == _lib.F32 and ops._dtype_code(xbf) == _lib.BF16   # fp32 side tensors of a bf16 net: exact
        net.set_compute_dtype(torch.float32)
        assert net.compute_dtype == torch.float32 and net.compute_code == _lib.F32
        ops.set_f32_split(True)                                                # the default, for op-level calls
        assert ops._dtype_code(x32) == _lib.F32X3
        with ops.compute_scope(_lib.F32):
            assert ops._dtype_code(x32) == _lib.F32                            # a module's own mode outranks it
        assert ops.PackPlan.signature(net) != ops.PackPlan.signature(net.set_compute_dtype("fp32x3"))
        with pytest.raises(ValueError):
            net.set_compute_dtype(torch.float16)
        with pytest.raises(TypeError):
            ops._dtype_code(torch.zeros(1, dtype=torch.float16))
    finally:
        ops.set_f32_split(False)


def test_reduce_batch_bookkeeping_nesting_foreign_pass_and_reset(monkeypatch):
    """The host-side bookkeeping of the batched slab reductions (ops._ReduceBatch) against a recording stand-in for the library:
    a DenseSTLayer's outer batch nests the blocks' own (ONE begin / end pair reaches the C side); a dense join that finds the batch of
    ANOTHER live backward pass open nests inside it and makes every node flush (it must not drop the other pass's queued jobs); a join
    that finds a leftover of its OWN pass drops it; ops.reset_backward_state() makes the thread drop a stale batch the next time it
    touches one (how the trainer reaches autograd's device thread after a failed capture)."""
    from rdst_amd import ops
    RB = ops._ReduceBatch
    calls = []

    class Lib:
        def __getattr__(self, name):
            def f(*a):
                calls.append(name)
                return 0
            return f
    lib = Lib()
    task = {"id": 7}
    monkeypatch.setattr(RB, "_task", staticmethod(lambda: task["id"]))
    monkeypatch.setattr(ops, "_stream", lambda: None)
    RB.abandon(lib)
    calls.clear()
    # one pass: join opens the layer's batch, two blocks nest, the first block of the layer closes it
    RB.begin_layer(lib)
    for _ in range(2):
        RB.begin(lib); RB.settle(lib); RB.end(lib)
    RB.end(lib)
    assert calls == ["rdst_reduce_batch_begin", "rdst_reduce_batch_end"] and RB.depth == 0 and not RB.mixed
    # a second pass's join runs while the first pass's batch is open: nest + flush at every node, nothing aborted
    calls.clear()
    RB.begin_layer(lib)              # pass 7
    task["id"] = 8
    RB.begin_layer(lib)              # pass 8 inside it
    assert RB.depth == 2 and RB.mixed and "rdst_reduce_batch_abort" not in calls
    RB.begin(lib); RB.settle(lib); RB.end(lib)      # a node of pass 8: flushed (end + begin) before it returns
    assert calls.count("rdst_reduce_batch_end") == 1 and calls.count("rdst_reduce_batch_begin") == 2
    RB.end(lib)                      # pass 8's layer closes
    task["id"] = 7
    RB.end(lib)                      # pass 7's layer closes: the batch ends
    assert RB.depth == 0 and "rdst_reduce_batch_abort" not in calls
    # a leftover of the SAME pass (a node raised, the pass re-entered): dropped
    calls.clear()
    RB.begin_layer(lib)
    RB.begin_layer(lib)
    assert calls == ["rdst_reduce_batch_begin", "rdst_reduce_batch_abort", "rdst_reduce_batch_begin"] and RB.depth == 1
    # reset_backward_state(): the stale batch is dropped the next time this thread touches one
    calls.clear()
    type(RB).EPOCH += 1
    RB.begin(lib)
    assert calls[:2] == ["rdst_reduce_batch_abort", "rdst_reduce_batch_begin"] and RB.depth == 1
    RB.end(lib)
    assert RB.depth == 0
