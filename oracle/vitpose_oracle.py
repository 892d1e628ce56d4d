"""CPU oracle for the ViTPose hot path (TEST INFRASTRUCTURE, not product code).

A functional fp32 restatement, in plain torch CPU ops, of what the reference
computes on the path  ViT backbone -> TopdownHeatmapSimpleHead -> JointsMSELoss
(+ PCK accuracy, flip_back, layer-decay parameter groups).  Every function
cites the reference file:line it follows.  Parity is pinned: tests/golden/*.npz
hold outputs of the reference's own source files executed in the build
container (tests/golden/make_golden.py), and tests/test_oracle_golden.py checks
this file against them; the loss / PCK / flip_back functions are additionally
checked against the known answers in the reference's own unit tests.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (vitpose_amd/) never does.

Weights are passed as a flat {state_dict key: tensor} mapping with exactly the
reference's key names (backbone keys un-prefixed here, see `split_state`).
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# model hyper-parameters of the four reference configs
# (configs/body/2d_kpt_sview_rgb_img/topdown_heatmap/coco/ViTPose_{small,base,
#  large,huge}_coco_256x192.py, `model=dict(backbone=...)` blocks)
# ----------------------------------------------------------------------------
VIT_CFGS = {
    'S': dict(embed_dim=384, depth=12, num_heads=12, drop_path_rate=0.1),
    'B': dict(embed_dim=768, depth=12, num_heads=12, drop_path_rate=0.3),
    'L': dict(embed_dim=1024, depth=24, num_heads=16, drop_path_rate=0.5),
    'H': dict(embed_dim=1280, depth=32, num_heads=16, drop_path_rate=0.55),
}


def vit_cfg(name: str, **over) -> dict:
    c = dict(img_size=(256, 192), patch_size=16, in_chans=3, mlp_ratio=4, qkv_bias=True,
             ratio=1, last_norm=True)
    c.update(VIT_CFGS[name])
    c.update(over)
    return c


def classic_head_cfg(in_channels: int) -> dict:
    return dict(in_channels=in_channels, out_channels=17, num_deconv_layers=2,
                num_deconv_filters=(256, 256), num_deconv_kernels=(4, 4),
                extra=dict(final_conv_kernel=1), upsample=0)


def simple_head_cfg(in_channels: int) -> dict:
    return dict(in_channels=in_channels, out_channels=17, num_deconv_layers=0,
                num_deconv_filters=[], num_deconv_kernels=[], upsample=4,
                extra=dict(final_conv_kernel=3))


# ----------------------------------------------------------------------------
# backbone
# ----------------------------------------------------------------------------
def patch_grid(cfg: dict) -> Tuple[int, int]:
    """Token grid of the patch-embedding conv (vit.py:146-157: kernel = patch,
    stride = patch // ratio, padding = 4 + 2 * (ratio // 2 - 1))."""
    H, W = cfg['img_size']
    p = cfg['patch_size']
    stride = p // cfg.get('ratio', 1)
    pad = 4 + 2 * (cfg.get('ratio', 1) // 2 - 1)
    return (H + 2 * pad - p) // stride + 1, (W + 2 * pad - p) // stride + 1


def patch_embed(sd: Dict[str, Tensor], img: Tensor, cfg: dict) -> Tuple[Tensor, Tuple[int, int]]:
    """vit.py:157-165: Conv2d(in, E, k=patch, s=patch//ratio, p=4+2*(ratio//2-1)),
    then flatten(2).transpose(1, 2) -> [B, Hp*Wp, E]."""
    p = cfg['patch_size']
    ratio = cfg.get('ratio', 1)
    y = F.conv2d(img, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'],
                 stride=p // ratio, padding=4 + 2 * (ratio // 2 - 1))
    Hp, Wp = y.shape[2], y.shape[3]
    return y.flatten(2).transpose(1, 2), (Hp, Wp)


def attention(sd: Dict[str, Tensor], pre: str, x: Tensor, num_heads: int,
              taps: Optional[dict] = None) -> Tensor:
    """vit.py:99-115: qkv Linear -> [B,N,3,h,d] -> q*d^-0.5 -> softmax(q k^T) v
    -> transpose/reshape to [B,N,E] -> proj Linear."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[pre + 'qkv.weight'], sd.get(pre + 'qkv.bias'))
    qkv = qkv.reshape(B, N, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    d = q.shape[-1]
    s = (q * (d ** -0.5)) @ k.transpose(-2, -1)
    p = s.softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, -1)
    if taps is not None:
        taps['qkv'] = qkv
        taps['attn_o'] = o
    return F.linear(o, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'])


def mlp(sd: Dict[str, Tensor], pre: str, x: Tensor, taps: Optional[dict] = None,
        indices: Optional[Tensor] = None) -> Tensor:
    """vit.py:71-76: fc2(GELU_erf(fc1(x)))  (nn.GELU default = exact erf).
    With expert weights in `sd` (ViTMoE), vit_moe.py:98-115 (MoEMlp.forward): fc2 emits the shared
    channels, every expert is evaluated on every token and masked by `indices == i` (one dataset index per
    image), the masked sum is concatenated behind the shared channels."""
    u = F.linear(x, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias'])
    h = F.gelu(u)
    if taps is not None:
        taps['fc1_pre'] = u
        taps['fc1_act'] = h
    shared = F.linear(h, sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias'])
    if pre + 'experts.0.weight' not in sd:
        return shared
    part = sd[pre + 'experts.0.weight'].shape[0]
    expert_x = torch.zeros_like(x[:, :, -part:])
    idx = indices.view(-1, 1, 1)
    i = 0
    while pre + f'experts.{i}.weight' in sd:
        cur = F.linear(h, sd[pre + f'experts.{i}.weight'], sd[pre + f'experts.{i}.bias']) * (idx == i)
        expert_x = expert_x + cur
        i += 1
    return torch.cat([shared, expert_x], dim=-1)


def drop_path_rates(cfg: dict) -> List[float]:
    """vit.py:233: torch.linspace(0, drop_path_rate, depth)."""
    return [v.item() for v in torch.linspace(0, cfg.get('drop_path_rate', 0.), cfg['depth'])]


def drop_path_scale(mask_u: Tensor, p: float) -> Tensor:
    """timm drop_path (called at vit.py:56): floor(keep + U[0,1)) / keep per sample.
    `mask_u` holds the uniform draws, shape [B]."""
    keep = 1.0 - p
    return torch.floor(keep + mask_u) / keep


def block(sd: Dict[str, Tensor], i: int, x: Tensor, cfg: dict,
          dp: Optional[Tuple[Tensor, Tensor]] = None, taps: Optional[dict] = None,
          indices: Optional[Tensor] = None) -> Tensor:
    """vit.py:137-140: x += DropPath(attn(LN1(x))); x += DropPath(mlp(LN2(x))),
    LayerNorm eps = 1e-6 (vit.py:212).  `dp` = per-sample scale vectors [B] for
    the two DropPath sites (None = eval / rate 0)."""
    pre = f'blocks.{i}.'
    E = x.shape[-1]
    xn = F.layer_norm(x, (E,), sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'], 1e-6)
    a = attention(sd, pre + 'attn.', xn, cfg['num_heads'], taps)
    if dp is not None:
        a = a * dp[0].view(-1, 1, 1)
    x = x + a
    xn2 = F.layer_norm(x, (E,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-6)
    m = mlp(sd, pre + 'mlp.', xn2, taps, indices)
    if dp is not None:
        m = m * dp[1].view(-1, 1, 1)
    if taps is not None:
        taps['ln1'] = xn
        taps['x_mid'] = x
        taps['ln2'] = xn2
    return x + m


def vit_forward(sd: Dict[str, Tensor], img: Tensor, cfg: dict,
                dp_scales: Optional[Sequence[Tuple[Tensor, Tensor]]] = None,
                return_tokens: bool = False, dataset_source: Optional[Tensor] = None) -> Tensor:
    """vit.py:313-332: patch_embed -> + pos_embed[:,1:] + pos_embed[:,:1] -> blocks
    -> last_norm -> permute(0,2,1).reshape(B,E,Hp,Wp).  ViTMoE (vit_moe.py:357-376) is the same
    sequence with `dataset_source` handed to every Block's MoEMlp."""
    B = img.shape[0]
    x, (Hp, Wp) = patch_embed(sd, img, cfg)
    pe = sd['pos_embed']
    x = x + pe[:, 1:] + pe[:, :1]
    for i in range(cfg['depth']):
        x = block(sd, i, x, cfg, None if dp_scales is None else dp_scales[i], indices=dataset_source)
    if cfg.get('last_norm', True):
        x = F.layer_norm(x, (x.shape[-1],), sd['last_norm.weight'], sd['last_norm.bias'], 1e-6)
    if return_tokens:
        return x
    return x.permute(0, 2, 1).reshape(B, -1, Hp, Wp).contiguous()


# ----------------------------------------------------------------------------
# head
# ----------------------------------------------------------------------------
def deconv_cfg(k: int) -> Tuple[int, int, int]:
    """topdown_heatmap_base_head.py:105-120."""
    if k == 4:
        return 4, 1, 0
    if k == 3:
        return 3, 1, 1
    if k == 2:
        return 2, 0, 0
    raise ValueError(f'Not supported num_kernels ({k}).')


def head_resize_concat(inputs: Sequence[Tensor], in_index: Sequence[int], align_corners: bool = False) -> Tensor:
    """topdown_heatmap_simple_head.py:289-298 (input_transform='resize_concat'): the selected maps, each resized
    bilinearly to the first one's size (mmpose/models/utils/ops.py:7-29 -> F.interpolate), joined along the channels."""
    xs = [inputs[i] for i in in_index]
    size = xs[0].shape[2:]
    return torch.cat([F.interpolate(x, size=size, mode='bilinear', align_corners=align_corners) for x in xs], dim=1)


def head_forward(sd: Dict[str, Tensor], feat: Tensor, cfg: dict, training: bool = False,
                 bn_momentum: float = 0.1, taps: Optional[dict] = None) -> Tensor:
    """topdown_heatmap_simple_head.py:197-202 (+ :269-287, :306-337):
    optional ReLU + bilinear x`upsample`; [ConvTranspose2d(k,s=2,bias=False), BN, ReLU]
    x num_deconv_layers; final Conv2d(k = final_conv_kernel).
    In training mode BN uses batch statistics and `sd`'s running stats are
    updated in place (torch BatchNorm2d semantics, momentum 0.1, unbiased var)."""
    x = feat
    if cfg.get('upsample', 0) > 0:
        x = F.interpolate(F.relu(x), scale_factor=cfg['upsample'], mode='bilinear',
                          align_corners=cfg.get('align_corners', False))
    for li in range(cfg.get('num_deconv_layers', 0)):
        k, pad, opad = deconv_cfg(cfg['num_deconv_kernels'][li])
        x = F.conv_transpose2d(x, sd[f'deconv_layers.{3 * li}.weight'], None, stride=2,
                               padding=pad, output_padding=opad)
        if taps is not None:
            taps[f'deconv{li}'] = x
        bn = f'deconv_layers.{3 * li + 1}.'
        x = F.batch_norm(x, sd[bn + 'running_mean'], sd[bn + 'running_var'],
                         sd[bn + 'weight'], sd[bn + 'bias'], training, bn_momentum, 1e-5)
        if taps is not None:
            taps[f'bn{li}'] = x
        x = F.relu(x)
    extra = cfg.get('extra') or {}
    fk = extra.get('final_conv_kernel', 1)
    if fk == 0:
        return x
    # optional [Conv2d(C, C, k, padding (k-1)/2), BN, ReLU] x num_conv_layers in front of the final conv
    # (topdown_heatmap_simple_head.py:112-138); final_layer is then an nn.Sequential and its keys carry indices
    nconv = extra.get('num_conv_layers', 0)
    kernels = extra.get('num_conv_kernels', [1] * nconv)
    for i in range(nconv):
        x = F.conv2d(x, sd[f'final_layer.{3 * i}.weight'], sd[f'final_layer.{3 * i}.bias'], padding=(kernels[i] - 1) // 2)
        bn = f'final_layer.{3 * i + 1}.'
        x = F.batch_norm(x, sd[bn + 'running_mean'], sd[bn + 'running_var'],
                         sd[bn + 'weight'], sd[bn + 'bias'], training, bn_momentum, 1e-5)
        x = F.relu(x)
    fl = f'final_layer.{3 * nconv}.' if nconv else 'final_layer.'
    return F.conv2d(x, sd[fl + 'weight'], sd[fl + 'bias'], padding=1 if fk == 3 else 0)


def joints_mse_loss(output: Tensor, target: Tensor, target_weight: Optional[Tensor],
                    use_target_weight: bool = True, loss_weight: float = 1.0) -> Tensor:
    """mse_loss.py:24-44: sum_k MSE(pred_k * w_k, gt_k * w_k) / K * loss_weight,
    MSE = mean over the B*H*W elements of joint k."""
    B, K = output.shape[:2]
    p = output.reshape(B, K, -1)
    t = target.reshape(B, K, -1)
    if use_target_weight:
        w = target_weight.reshape(B, K, 1)
        p, t = p * w, t * w
    per_joint = ((p - t) ** 2).mean(dim=(0, 2))
    return per_joint.sum() / K * loss_weight


# ----------------------------------------------------------------------------
# accuracy / post-processing (numpy, as in the reference)
# ----------------------------------------------------------------------------
def max_preds(heatmaps: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """core/evaluation/top_down_eval.py:63-95 (_get_max_preds)."""
    N, K, _, W = heatmaps.shape
    flat = heatmaps.reshape(N, K, -1)
    idx = flat.argmax(2)
    maxvals = flat.max(2)[..., None]
    preds = np.stack([idx % W, idx // W], axis=-1).astype(np.float32)
    preds = np.where(maxvals > 0.0, preds, -1)
    return preds, maxvals


def pck_accuracy(output: np.ndarray, target: np.ndarray, mask: np.ndarray,
                 thr: float = 0.05) -> Tuple[np.ndarray, float, int]:
    """core/evaluation/top_down_eval.py:136-176 (pose_pck_accuracy with
    normalize=[H, W]) -> :98-133 (keypoint_pck_accuracy), :10-54 (distances)."""
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    norm = np.tile(np.array([[H, W]], dtype=np.float32), (N, 1))
    pred, _ = max_preds(output)
    gt, _ = max_preds(target)
    dist = np.full((N, K), -1, dtype=np.float32)
    d = np.linalg.norm((pred - gt) / norm[:, None, :], axis=-1)
    dist[mask] = d[mask]
    acc = np.full((K,), -1.0)
    for k in range(K):
        valid = dist[:, k] != -1
        if valid.sum() > 0:
            acc[k] = (dist[valid, k] < thr).sum() / valid.sum()
    ok = acc[acc >= 0]
    cnt = len(ok)
    return acc, (float(ok.mean()) if cnt > 0 else 0), cnt


def flip_back(output_flipped: np.ndarray, flip_pairs) -> np.ndarray:
    """core/post_processing/post_transforms.py:110-147 (GaussianHeatmap case):
    swap left/right channels, reverse W."""
    out = output_flipped.copy()
    for left, right in flip_pairs:
        out[:, left] = output_flipped[:, right]
        out[:, right] = output_flipped[:, left]
    return out[..., ::-1]


# ----------------------------------------------------------------------------
# optimizer parameter groups
# ----------------------------------------------------------------------------
def layer_decay_groups(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], num_layers_cfg: int,
                       layer_decay_rate: float, base_lr: float, base_wd: float):
    """mmcv_custom/layer_decay_optimizer_constructor.py:6-78.  `named_shapes`
    holds (name, shape) of every trainable parameter with the `backbone.` /
    `keypoint_head.` prefixes of the TopDown module.  Returns a list of
    (name, layer_id, lr_scale, lr, weight_decay)."""
    num_layers = num_layers_cfg + 2
    out = []
    for name, shape in named_shapes:
        if len(shape) == 1 or name.endswith('.bias') or 'pos_embed' in name:
            wd = 0.
        else:
            wd = base_wd
        if name in ('backbone.cls_token', 'backbone.mask_token', 'backbone.pos_embed'):
            lid = 0
        elif name.startswith('backbone.patch_embed'):
            lid = 0
        elif name.startswith('backbone.blocks'):
            lid = int(name.split('.')[2]) + 1
        else:
            lid = num_layers - 1
        scale = layer_decay_rate ** (num_layers - lid - 1)
        out.append((name, lid, scale, scale * base_lr, wd))
    return out


# ----------------------------------------------------------------------------
# deterministic synthetic weights / inputs (shared by the golden generator, the
# tests, smoke() and bench.py so that nothing but small outputs is stored)
# ----------------------------------------------------------------------------
def vit_param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    E = cfg['embed_dim']
    Hd = int(E * cfg.get('mlp_ratio', 4))
    p = cfg['patch_size']
    Hp, Wp = patch_grid(cfg)
    # pos_embed is sized from img_size // patch (vit.py:150,231), not from the conv grid
    npos = (cfg['img_size'][0] // p) * (cfg['img_size'][1] // p) * cfg.get('ratio', 1) ** 2 + 1
    out = [('pos_embed', (1, npos, E)),
           ('patch_embed.proj.weight', (E, cfg.get('in_chans', 3), p, p)),
           ('patch_embed.proj.bias', (E,))]
    for i in range(cfg['depth']):
        b = f'blocks.{i}.'
        out += [(b + 'norm1.weight', (E,)), (b + 'norm1.bias', (E,)),
                (b + 'attn.qkv.weight', (3 * E, E)), (b + 'attn.qkv.bias', (3 * E,)),
                (b + 'attn.proj.weight', (E, E)), (b + 'attn.proj.bias', (E,)),
                (b + 'norm2.weight', (E,)), (b + 'norm2.bias', (E,)),
                (b + 'mlp.fc1.weight', (Hd, E)), (b + 'mlp.fc1.bias', (Hd,))]
        part = cfg.get('part_features') or 0          # ViTMoE: vit_moe.py:86,92-96
        out += [(b + 'mlp.fc2.weight', (E - part, Hd)), (b + 'mlp.fc2.bias', (E - part,))]
        if part:
            for j in range(cfg.get('num_expert', 1)):
                out += [(b + f'mlp.experts.{j}.weight', (part, Hd)), (b + f'mlp.experts.{j}.bias', (part,))]
    if cfg.get('last_norm', True):
        out += [('last_norm.weight', (E,)), ('last_norm.bias', (E,))]
    return out


def synth_vit_weights(cfg: dict, seed: int = 0, gain: float = 1.0) -> Dict[str, Tensor]:
    """Non-degenerate 'trained-like' weights: matrices ~ N(0, gain/fan_in) so every
    branch contributes O(1) to the residual stream, LN affine and all biases
    randomised (the reference's default zeros/ones would hide bias/affine bugs)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in vit_param_shapes(cfg):
        if name == 'pos_embed':
            t = 0.5 * torch.randn(shape, generator=g)
        elif name.endswith('norm1.weight') or name.endswith('norm2.weight') or name == 'last_norm.weight':
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith('.bias'):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (gain / math.sqrt(fan_in))
        sd[name] = t
    return sd


def head_param_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    out = []
    cin = cfg['in_channels']
    for li in range(cfg.get('num_deconv_layers', 0)):
        k = cfg['num_deconv_kernels'][li]
        co = cfg['num_deconv_filters'][li]
        out.append((f'deconv_layers.{3 * li}.weight', (cin, co, k, k)))
        bn = f'deconv_layers.{3 * li + 1}.'
        out += [(bn + 'weight', (co,)), (bn + 'bias', (co,)),
                (bn + 'running_mean', (co,)), (bn + 'running_var', (co,))]
        cin = co
    extra = cfg.get('extra') or {}
    fk = extra.get('final_conv_kernel', 1)
    if fk:
        nconv = extra.get('num_conv_layers', 0)
        kernels = extra.get('num_conv_kernels', [1] * nconv)
        for i in range(nconv):
            out += [(f'final_layer.{3 * i}.weight', (cin, cin, kernels[i], kernels[i])), (f'final_layer.{3 * i}.bias', (cin,))]
            bn = f'final_layer.{3 * i + 1}.'
            out += [(bn + 'weight', (cin,)), (bn + 'bias', (cin,)), (bn + 'running_mean', (cin,)), (bn + 'running_var', (cin,))]
        fl = f'final_layer.{3 * nconv}.' if nconv else 'final_layer.'
        out += [(fl + 'weight', (cfg['out_channels'], cin, fk, fk)), (fl + 'bias', (cfg['out_channels'],))]
    return out


def synth_head_weights(cfg: dict, seed: int = 1) -> Dict[str, Tensor]:
    """Head weights scaled so heatmaps come out O(1) (the reference's default
    std=0.001 init yields ~1e-4 heatmaps, which makes an absolute tolerance vacuous)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in head_param_shapes(cfg):
        if name.startswith('deconv_layers') and len(shape) == 4:
            # stride-2 transposed conv: each output pixel sums (k/2)^2 taps x Cin
            taps = max(1, (shape[2] // 2) * (shape[3] // 2))
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / (shape[0] * taps))
        elif name.endswith('running_var'):
            t = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith('running_mean'):
            t = 0.2 * torch.randn(shape, generator=g)
        elif name.endswith('.weight') and len(shape) == 1:
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith('.bias'):
            t = 0.1 * torch.randn(shape, generator=g)
        else:  # final conv
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * math.sqrt(1.0 / fan_in)
        sd[name] = t
    return sd


def synth_batch(B: int, img_size=(256, 192), K: int = 17, seed: int = 0):
    """img ~ N(0,1) (ImageNet-normalised crops), target in [0,1), weights in {0,1}."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, img_size[0], img_size[1], generator=g)
    target = torch.rand(B, K, img_size[0] // 4, img_size[1] // 4, generator=g)
    tw = (torch.rand(B, K, 1, generator=g) > 0.2).float()
    return img, target, tw


def split_state(sd: Dict[str, Tensor]) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """Split a TopDown state dict into (backbone, keypoint_head) un-prefixed dicts."""
    bb = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    hd = {k[len('keypoint_head.'):]: v for k, v in sd.items() if k.startswith('keypoint_head.')}
    return bb, hd


def topdown_moe_losses(bb: Dict[str, Tensor], heads: Sequence[Dict[str, Tensor]], img: Tensor, target: Tensor,
                       tw: Tensor, dataset_source: Tensor, vcfg: dict, hcfgs: Sequence[dict], training: bool = True,
                       dp_scales=None, use_target_weight: bool = True):
    """models/detectors/top_down_moe.py:166-203 (forward_train): the routed backbone once, then every head on
    every image with the targets / target weights of the other datasets' images zeroed.  Returns
    (losses per head, heat maps per head)."""
    feat = vit_forward(bb, img, vcfg, dp_scales, dataset_source=dataset_source)
    losses, heats = [], []
    for i, (hd, hcfg) in enumerate(zip(heads, hcfgs)):
        sel = (dataset_source == i)
        out = head_forward(hd, feat, hcfg, training)
        losses.append(joints_mse_loss(out, target * sel.view(-1, 1, 1, 1), tw * sel.view(-1, 1, 1),
                                      use_target_weight))
        heats.append(out)
    return losses, heats


def topdown_forward(bb: Dict[str, Tensor], hd: Dict[str, Tensor], img: Tensor, vcfg: dict,
                    hcfg: dict, training: bool = False, dp_scales=None) -> Tensor:
    """models/detectors/top_down.py:202-218 (forward_dummy): head(backbone(img))."""
    return head_forward(hd, vit_forward(bb, img, vcfg, dp_scales), hcfg, training)
