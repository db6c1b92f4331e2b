"""CPU oracle for the RDST hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement, in plain functional torch ops on CPU tensors, of the
algorithm of GinZhu/RDST's data-parallel hot path (window-attention Swin blocks,
residual-dense blocks, pixel-shuffle upsampler).  It exists to CHECK the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``rdst_amd/`` imports this file, and the product path
raises if the HIP library is missing rather than falling back to it.

Pinning: ``tests/golden/make_golden.py`` (run in the build container, where
``/root/reference`` is mounted) imports the reference network, loads the same
deterministic weights into it and into this oracle, and asserts equality before it
writes the committed fixtures in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
re-checks the oracle against those fixtures everywhere (no reference needed).
The reference holds no tests or golden vectors of its own, so the
fixtures generated from the reference import are the pin.

Every function cites the reference lines it restates (paths relative to the
reference root).  Weights are passed as a flat mapping ``sd`` using the reference's
state-dict keys, so the same dictionary drives the reference, the oracle and the
HIP-backed modules.

All arithmetic is floating point in the dtype of the inputs (fp32 for parity with
the reference, fp64 for tight gradient checks).
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]


# --------------------------------------------------------------------------------------
# window helpers
# --------------------------------------------------------------------------------------
def window_partition(x: Tensor, ws: int) -> Tensor:
    """(B,H,W,C) -> (B*nW, ws, ws, C).  networks/swin_transformer_sr.py:32-43."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def window_reverse(windows: Tensor, ws: int, H: int, W: int) -> Tensor:
    """(B*nW, ws, ws, C) -> (B,H,W,C).  networks/swin_transformer_sr.py:46-59."""
    nW = (H // ws) * (W // ws)
    B = windows.shape[0] // nW
    x = windows.reshape(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def relative_position_index(ws: int) -> Tensor:
    """(N,N) int64 index into the ((2ws-1)^2, heads) bias table.

    networks/swin_transformer_sr.py:89-98: idx(i,j) = (yi-yj+ws-1)*(2ws-1) + (xi-xj+ws-1).
    """
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    y = ys.reshape(-1)
    x = xs.reshape(-1)
    dy = y[:, None] - y[None, :] + ws - 1
    dx = x[:, None] - x[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


def calculate_mask(H: int, W: int, ws: int, shift: int, dtype=torch.float32) -> Tensor:
    """(nW, N, N) additive mask (0 / -100) of a shifted block.

    networks/swin_transformer_sr.py:211-232: the (rolled) image is cut into 3x3 regions
    by the slices [0,-ws), [-ws,-shift), [-shift,end) on both axes; tokens of different
    regions inside one window do not attend to each other.
    """
    img = torch.zeros(H, W, dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = window_partition(img.reshape(1, H, W, 1), ws).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def effective_window(input_resolution: Sequence[int], ws: int, shift: int) -> Tuple[int, int]:
    """Window clamp done at construction time.  networks/swin_transformer_sr.py:188-191."""
    if min(input_resolution) <= ws:
        return min(input_resolution), 0
    return ws, shift


# --------------------------------------------------------------------------------------
# attention core (what the fused HIP kernel K1/K2 replaces)
# --------------------------------------------------------------------------------------
def window_attention_core(qkv: Tensor, table: Tensor, heads: int, ws: int, shift: int,
                          scale: float, drop_mult: Optional[Tensor] = None) -> Tensor:
    """qkv (B,H,W,3C) token-major, inner order [3][heads][d] -> out (B,H,W,C).

    Restates roll -> window_partition -> (q*scale)@k^T + bias (+mask) -> softmax -> @v ->
    window_reverse -> roll back: networks/swin_transformer_sr.py:244-251 (roll/partition),
    :117-138 (attention without the qkv/proj Linears), :260-267 (reverse / un-roll).
    Rolling qkv instead of the LayerNorm output is equivalent because the qkv Linear acts
    per token.
    """
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    N = ws * ws
    x = qkv
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).reshape(-1, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0], xw[1], xw[2]
    q = q * scale
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(ws).reshape(-1)
    bias = table[idx].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        mask = calculate_mask(H, W, ws, shift, dtype=qkv.dtype)
        nW = mask.shape[0]
        attn = attn.reshape(-1, nW, heads, N, N) + mask[None, :, None]
        attn = attn.reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    if drop_mult is not None:
        # self.attn_drop(attn) (:136) with the mask made explicit: nn.Dropout multiplies by 0 or 1 / (1 - p); the random
        # stream itself is not part of the restatement (the tests export the HIP path's mask and pass it here)
        attn = attn * drop_mult.reshape(attn.shape)
    out = (attn @ v).transpose(1, 2).reshape(-1, ws, ws, C)
    out = window_reverse(out, ws, H, W)
    if shift > 0:
        out = torch.roll(out, shifts=(shift, shift), dims=(1, 2))
    return out


def layer_norm(x: Tensor, w: Optional[Tensor], b: Optional[Tensor], eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm over the last dim (eps 1e-5, biased variance); identity if w is None
    (``norm_layer = nn.Identity`` when ``rdst_layer_norm`` is False, rdst_variations.py:1399)."""
    if w is None:
        return x
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu(x: Tensor) -> Tensor:
    """Exact (erf) GELU = nn.GELU default.  networks/swin_transformer_sr.py:14,19."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _get(sd: SD, key: str) -> Optional[Tensor]:
    return sd[key] if key in sd else None


# --------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------
def swin_block(x: Tensor, x_size: Tuple[int, int], sd: SD, pfx: str, heads: int, ws: int,
               shift: int, qk_scale: Optional[float] = None) -> Tensor:
    """SwinTransformerBlock.forward.  networks/swin_transformer_sr.py:234-274.

    ``ws``/``shift`` are the already-clamped values (see :func:`effective_window`).
    """
    H, W = x_size
    B, L, C = x.shape
    d = C // heads
    scale = qk_scale or d ** -0.5  # :82
    shortcut = x
    h = layer_norm(x, _get(sd, pfx + "norm1.weight"), _get(sd, pfx + "norm1.bias"))  # :240
    qkv = F.linear(h, sd[pfx + "attn.qkv.weight"], _get(sd, pfx + "attn.qkv.bias"))  # :117
    a = window_attention_core(qkv.reshape(B, H, W, 3 * C), sd[pfx + "attn.relative_position_bias_table"],
                              heads, ws, shift, scale).reshape(B, L, C)
    a = F.linear(a, sd[pfx + "attn.proj.weight"], sd[pfx + "attn.proj.bias"])  # :139
    x = shortcut + a  # :271 (drop_path is Identity on this path)
    h = layer_norm(x, _get(sd, pfx + "norm2.weight"), _get(sd, pfx + "norm2.bias"))
    h = F.linear(h, sd[pfx + "mlp.fc1.weight"], sd[pfx + "mlp.fc1.bias"])  # :24
    h = gelu(h)  # :25
    h = F.linear(h, sd[pfx + "mlp.fc2.weight"], sd[pfx + "mlp.fc2.bias"])  # :27
    return x + h  # :272


def basic_layer(x: Tensor, x_size, sd: SD, pfx: str, depth: int, heads: int, ws: int,
                input_resolution, qk_scale=None) -> Tensor:
    """BasicLayer.forward: blocks alternate shift 0 / ws//2.  swin_transformer_sr.py:373-398."""
    for i in range(depth):
        shift = 0 if i % 2 == 0 else ws // 2  # :376
        ews, eshift = effective_window(input_resolution, ws, shift)
        x = swin_block(x, x_size, sd, f"{pfx}blocks.{i}.", heads, ews, eshift, qk_scale)
    return x


def dense_st_layer(x: Tensor, x_size, sd: SD, pfx: str, depth: int, heads: int, ws: int,
                   input_resolution, growth: int, dense_scale: float, pre_norm: bool,
                   qk_scale=None) -> Tensor:
    """DenseSTLayer.forward in 'tail' mode.  networks/rdst_variations.py:305-320, :335-341."""
    C = x.shape[-1]
    short_cut = x
    y = basic_layer(x, x_size, sd, pfx + "body.", depth, heads, ws, input_resolution, qk_scale)
    if C != growth:  # :308
        if pre_norm:  # LN(hidden) -> Linear(hidden, growth)  :309-313
            y = layer_norm(y, _get(sd, pfx + "tail.0.weight"), _get(sd, pfx + "tail.0.bias"))
            y = F.linear(y, sd[pfx + "tail.1.weight"], sd[pfx + "tail.1.bias"])
        else:  # Linear -> LN(growth)  :315-318
            y = F.linear(y, sd[pfx + "tail.0.weight"], sd[pfx + "tail.0.bias"])
            y = layer_norm(y, _get(sd, pfx + "tail.1.weight"), _get(sd, pfx + "tail.1.bias"))
    return torch.cat((short_cut, y * dense_scale), 2)  # :340


def tokens_to_nchw(x: Tensor, x_size) -> Tensor:
    """PatchUnEmbed.forward.  networks/swin_transformer_sr.py:552-555."""
    B, L, C = x.shape
    return x.transpose(1, 2).reshape(B, C, x_size[0], x_size[1])


def nchw_to_tokens(x: Tensor) -> Tensor:
    """PatchEmbed.forward without the norm.  networks/swin_transformer_sr.py:515-516."""
    return x.flatten(2).transpose(1, 2)


def res_conv(x: Tensor, sd: SD, pfx: str, mode: str) -> Tensor:
    """'1conv' / '3conv' residual connection conv.  networks/rdst_variations.py:420-428."""
    if mode == "1conv":
        return F.conv2d(x, sd[pfx + "weight"], sd[pfx + "bias"], padding=1)
    x = F.conv2d(x, sd[pfx + "0.weight"], sd[pfx + "0.bias"], padding=1)
    x = F.leaky_relu(x, 0.2)
    x = F.conv2d(x, sd[pfx + "2.weight"], sd[pfx + "2.bias"])
    x = F.leaky_relu(x, 0.2)
    return F.conv2d(x, sd[pfx + "4.weight"], sd[pfx + "4.bias"], padding=1)


def rdstb(x: Tensor, x_size, sd: SD, pfx: str, cfg: dict, i_block: int, input_resolution) -> Tensor:
    """RDSTB.forward.  networks/rdst_variations.py:438-445."""
    short_cut = x
    for l in range(cfg["rdb_depths"][i_block]):
        x = dense_st_layer(x, x_size, sd, f"{pfx}body.{l}.", cfg["dense_layer_depths"][i_block],
                           cfg["num_heads"][i_block], cfg["window_size"][i_block], input_resolution,
                           cfg["growth_rate"], cfg["dense_scale"], cfg["pre_norm"], cfg.get("qk_scale"))
    y = res_conv(tokens_to_nchw(x, x_size), sd, pfx + "conv.", cfg["resi_connection"])
    return nchw_to_tokens(y) * cfg["rdb_residual_scale"] + short_cut  # :444-445


def upsampler_tail(x: Tensor, sd: SD, sr_scale: int) -> Tensor:
    """tail = UpSampler (conv 3x3 C->4C + PixelShuffle(2), log2(s) times; or C->9C + PS(3))
    then conv3x3 C->in_chans.  networks/common.py:125-148, rdst_variations.py:1299-1304."""
    if sr_scale > 1:
        if sr_scale & (sr_scale - 1) == 0:
            for i in range(int(math.log2(sr_scale))):
                x = F.conv2d(x, sd[f"tail.0.{2 * i}.weight"], sd[f"tail.0.{2 * i}.bias"], padding=1)
                x = F.pixel_shuffle(x, 2)
        elif sr_scale == 3:
            x = F.conv2d(x, sd["tail.0.0.weight"], sd["tail.0.0.bias"], padding=1)
            x = F.pixel_shuffle(x, 3)
        else:
            raise NotImplementedError(f"SR scale {sr_scale} is not valid.")
        last = "tail.1."
    else:
        last = "tail.0."
    return F.conv2d(x, sd[last + "weight"], sd[last + "bias"], padding=1)


DEFAULT_CFG = dict(
    img_size=48, in_chans=1, sr_scale=2, embed_dim=60, dense_layer_depths=[2] * 4,
    num_heads=[6] * 4, window_size=[4] * 4, rdb_depths=[3] * 4, mlp_ratio=4.0, qk_scale=None,
    layer_norm=True, patch_norm=True, resi_connection="1conv", growth_rate=30, dense_scale=1.0,
    rdb_residual_scale=1.0, global_res_scale=1.0, pre_norm=False, feature_last_operation=False,
)


def make_cfg(**kw) -> dict:
    cfg = dict(DEFAULT_CFG)
    cfg.update(kw)
    return cfg


def rdstsr_forward(x: Tensor, sd: SD, cfg: dict) -> Tensor:
    """RDSTSR.forward.  networks/rdst_variations.py:1326-1360."""
    img = cfg["img_size"]
    input_resolution = (img, img) if isinstance(img, int) else tuple(img)
    x = F.conv2d(x, sd["sub_mean.weight"], sd["sub_mean.bias"])  # :1343
    x = F.conv2d(x, sd["head.weight"], sd["head.bias"], padding=1)  # :1344
    x_size = (x.shape[2], x.shape[3])
    t = nchw_to_tokens(x)  # :1329
    if cfg["patch_norm"]:
        t = layer_norm(t, _get(sd, "patch_embed.norm.weight"), _get(sd, "patch_embed.norm.bias"))
    for b in range(len(cfg["rdb_depths"])):  # :1334-1335
        t = rdstb(t, x_size, sd, f"body.{b}.", cfg, b, input_resolution)
    t = layer_norm(t, _get(sd, "norm.weight"), _get(sd, "norm.bias"))  # :1337
    res = tokens_to_nchw(t, x_size) * cfg["global_res_scale"]  # :1338, :1347
    if cfg["feature_last_operation"]:  # :1348-1349
        res = res_conv(res, sd, "conv_after_body.", cfg["resi_connection"])
    res = res + x  # :1350
    y = upsampler_tail(res, sd, cfg["sr_scale"])  # :1356
    return F.conv2d(y, sd["add_mean.weight"], sd["add_mean.bias"])  # :1358


def psnr(gt: Tensor, pred: Tensor, border: int = 0) -> float:
    """10*log10(1/mse) in float64 with data_range=1 (metrics/sr_metrics.py:8-9), after
    cropping ``border`` pixels (= ceil(sr_scale), metrics/sr_metrics.py:108-115)."""
    g = gt.detach().double()
    p = pred.detach().double()
    if border:
        g = g[..., border:-border, border:-border]
        p = p[..., border:-border, border:-border]
    mse = torch.mean((g - p) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(1.0 / mse)


# --------------------------------------------------------------------------------------
# state-dict layout (what the reference's strict loads expect) and deterministic weights
# --------------------------------------------------------------------------------------
def state_dict_layout(cfg: dict) -> Dict[str, Tuple[Tuple[int, ...], str, str]]:
    """key -> (shape, dtype-name, kind) of RDSTSR(**cfg).state_dict(), in the reference's
    registration order; checked against the reference import in tests/golden/make_golden.py
    and pinned in tests/golden/state_dict_*.json.  kind is one of ln_w ln_b lin_w lin_b
    conv_w conv_b table index mask shift_w shift_b."""
    out: Dict[str, Tuple[Tuple[int, ...], str, str]] = {}
    nc = cfg["in_chans"]
    E = cfg["embed_dim"]
    img = cfg["img_size"]
    res = (img, img) if isinstance(img, int) else tuple(img)
    f32, i64 = "float32", "int64"
    ln = cfg["layer_norm"]

    def lin(p, o, i):
        out[p + "weight"] = ((o, i), f32, "lin_w")
        out[p + "bias"] = ((o,), f32, "lin_b")

    def norm(p, c):
        if ln:
            out[p + "weight"] = ((c,), f32, "ln_w")
            out[p + "bias"] = ((c,), f32, "ln_b")

    def conv(p, o, i, k):
        out[p + "weight"] = ((o, i, k, k), f32, "conv_w")
        out[p + "bias"] = ((o,), f32, "conv_b")

    def resconv(p, cin, cout):
        if cfg["resi_connection"] == "1conv":
            conv(p, cout, cin, 3)
        else:
            conv(p + "0.", cin // 4, cin, 3)
            conv(p + "2.", cin // 4, cin // 4, 1)
            conv(p + "4.", cout, cin // 4, 3)

    for m in ("add_mean.", "sub_mean."):
        out[m + "weight"] = ((nc, nc, 1, 1), f32, "shift_w")
        out[m + "bias"] = ((nc,), f32, "shift_b")
    conv("head.", E, nc, 3)
    if cfg["patch_norm"]:
        norm("patch_embed.norm.", E)
    G = cfg["growth_rate"]
    for b in range(len(cfg["rdb_depths"])):
        C = E
        heads = cfg["num_heads"][b]
        for l in range(cfg["rdb_depths"][b]):
            p = f"body.{b}.body.{l}."
            if C != G:
                if cfg["pre_norm"]:
                    norm(p + "tail.0.", C)
                    lin(p + "tail.1.", G, C)
                else:
                    lin(p + "tail.0.", G, C)
                    norm(p + "tail.1.", G)
            for k in range(cfg["dense_layer_depths"][b]):
                q = f"{p}body.blocks.{k}."
                ws, shift = effective_window(res, cfg["window_size"][b],
                                             0 if k % 2 == 0 else cfg["window_size"][b] // 2)
                if shift > 0:
                    nW = (res[0] // ws) * (res[1] // ws)
                    out[q + "attn_mask"] = ((nW, ws * ws, ws * ws), f32, "mask")
                norm(q + "norm1.", C)
                out[q + "attn.relative_position_bias_table"] = (((2 * ws - 1) ** 2, heads), f32, "table")
                out[q + "attn.relative_position_index"] = ((ws * ws, ws * ws), i64, "index")
                lin(q + "attn.qkv.", 3 * C, C)
                lin(q + "attn.proj.", C, C)
                norm(q + "norm2.", C)
                hid = int(C * cfg["mlp_ratio"])
                lin(q + "mlp.fc1.", hid, C)
                lin(q + "mlp.fc2.", C, hid)
            C += G
        resconv(f"body.{b}.conv.", C, E)
    norm("norm.", E)
    resconv("conv_after_body.", E, E)
    s = cfg["sr_scale"]
    if s > 1:
        if s & (s - 1) == 0:
            for i in range(int(math.log2(s))):
                conv(f"tail.0.{2 * i}.", 4 * E, E, 3)
        else:
            conv("tail.0.0.", 9 * E, E, 3)
        conv("tail.1.", nc, E, 3)
    else:
        conv("tail.0.", nc, E, 3)
    return out


def make_weights(cfg: dict, seed: int = 0, mean=None, std=None, dtype=torch.float32) -> Dict[str, Tensor]:
    """Deterministic, torch-RNG-independent weights keyed by state-dict name (numpy PCG64
    seeded by (seed, crc32(key))), so the reference (in the build container), the oracle and
    the HIP modules (on the GPU box) all see identical parameters without shipping them.

    Scales are chosen to exercise every path: LayerNorm affine away from (1,0), a
    relative-position table with O(0.5) entries, non-zero biases.
    """
    import zlib
    import numpy as np

    nc = cfg["in_chans"]
    mean = [0.0] * nc if mean is None else list(mean)
    std = [1.0] * nc if std is None else list(std)
    img = cfg["img_size"]
    res = (img, img) if isinstance(img, int) else tuple(img)
    sd: Dict[str, Tensor] = {}
    for key, (shape, _dt, kind) in state_dict_layout(cfg).items():
        if kind == "index":
            sd[key] = relative_position_index(int(round(math.sqrt(shape[0]))))
            continue
        if kind == "mask":
            ws = int(round(math.sqrt(shape[1])))
            sd[key] = calculate_mask(res[0], res[1], ws, ws // 2, dtype=dtype)
            continue
        if kind in ("shift_w", "shift_b"):  # MeanShift, networks/common.py:151-167
            s = torch.tensor(std, dtype=dtype)
            m = torch.tensor(mean, dtype=dtype)
            eye = torch.eye(nc, dtype=dtype).reshape(nc, nc, 1, 1)
            if key.startswith("sub_mean."):
                sd[key] = eye / s.reshape(nc, 1, 1, 1) if kind == "shift_w" else -m / s
            else:
                sd[key] = eye * s.reshape(nc, 1, 1, 1) if kind == "shift_w" else m
            continue
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
        n = rng.standard_normal(shape)
        if kind == "table":
            a = 0.5 * n
        elif kind == "ln_w":
            a = 1.0 + 0.1 * n
        elif kind == "ln_b":
            a = 0.05 * n
        elif kind in ("lin_b", "conv_b"):
            a = 0.02 * n
        elif kind == "conv_w":
            a = n / math.sqrt(shape[1] * shape[2] * shape[3])
        else:  # lin_w
            a = 0.7 * n / math.sqrt(shape[1])
        sd[key] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return sd


CFG_TINY = make_cfg(  # BASELINE.json configs[0] ("RDST-E tiny"): SURVEY.md §8 cfg1
    img_size=64, in_chans=1, sr_scale=4, embed_dim=48, dense_layer_depths=[2, 2], num_heads=[6, 6],
    window_size=[8, 8], rdb_depths=[3, 3], mlp_ratio=2.0, growth_rate=24, pre_norm=True,
    feature_last_operation=True)

CFG_E1 = make_cfg(  # config_files/RDST_E1_OASIS_example_SRx4.ini:188-240, img_size per BASELINE.json
    img_size=64, in_chans=1, sr_scale=4, embed_dim=60, dense_layer_depths=[2] * 8, num_heads=[6] * 8,
    window_size=[8] * 8, rdb_depths=[3] * 8, mlp_ratio=2.0, growth_rate=30, pre_norm=True,
    feature_last_operation=True)

CFG_WS16 = make_cfg(  # BASELINE.json configs[3]: x2, 3-channel, 128x128, window 16 (2 blocks here)
    img_size=128, in_chans=3, sr_scale=2, embed_dim=60, dense_layer_depths=[2] * 8, num_heads=[6] * 8,
    window_size=[16] * 8, rdb_depths=[3] * 8, mlp_ratio=2.0, growth_rate=30, pre_norm=True,
    feature_last_operation=True)
