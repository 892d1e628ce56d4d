"""The reference's model classes for the ViTPose hot path, re-implemented over the HIP plans:

    ViT                        mmpose/models/backbones/vit.py:200-341
    TopdownHeatmapSimpleHead   mmpose/models/heads/topdown_heatmap_simple_head.py:16-350
    JointsMSELoss              mmpose/models/losses/mse_loss.py:9-44
    TopDown                    mmpose/models/detectors/top_down.py:22-218 (+ detectors/base.py:38-126)

Same registry names, constructor keywords, state-dict keys, exceptions and call contract, so the
reference's `model = dict(type='TopDown', backbone=dict(type='ViT', ...), ...)` configs build
them unchanged.  The torch sub-modules (nn.Linear, nn.LayerNorm, nn.Conv2d, nn.ConvTranspose2d,
nn.BatchNorm2d) are PARAMETER HOLDERS only: they give the reference's parameter names, shapes
and default initialisation, and are never called.  All arithmetic runs in the gfx950 kernels of
libvitpose_hip.so; there is no CPU execution path (forward on a CPU tensor raises).

Storage.  On a HIP device every module keeps its parameters in one flat f32 buffer
(engine.ParamStore: master weights + bf16 mirror + gradient buffer); the nn.Parameters are views
into it, and after a backward pass `param.grad` is a view into the flat gradient buffer.  A
TopDown that holds a native backbone and head fuses both into ONE store and one launch plan
(the token-major backbone output feeds the head directly, the loss is fused into the head).
"""
import warnings
from collections import OrderedDict
from functools import partial
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from .builder import BACKBONES, HEADS, LOSSES, POSENETS, build_backbone, build_head, build_loss, build_neck
from .engine import ParamStore, VitRunner, Workspace
from .head_engine import KPAD, HeadRunner, bilinear_tables
from .plan import Plan


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _need_hip(t: torch.Tensor, what: str):
    if t.device.type != 'cuda':
        raise RuntimeError(f'{what} runs on an MI355X (HIP device) only; there is no CPU path '
                           f'(got a tensor on {t.device})')
    _lib.lib()


def _param_versions(params):
    return sum(p._version for p in params)


class _Holder(nn.Module):
    """A sub-module that only names parameters; the enclosing HIP module computes."""

    def forward(self, *a, **k):
        raise NotImplementedError(f'{type(self).__name__} is evaluated inside the HIP plan of its parent module')


class _FlatBound(nn.Module):
    """Mixin: keeps the module's parameters (and float buffers) as views of one ParamStore while
    they live on a HIP device, and re-creates the store whenever the module is moved."""
    _owned = False      # True while a TopDown has fused this module's parameters into its own store
    _store: Optional[ParamStore] = None
    _prefix = ''        # this module's name inside the owner's store ('backbone.' / 'keypoint_head.')

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if not self._owned:
            self._bind()
        return out

    def _bind(self):
        params = list(self.named_parameters())
        self._store, self._runners, self._plans = None, {}, {}
        if not params or params[0][1].device.type != 'cuda':
            return
        st = ParamStore([(n, tuple(p.shape)) for n, p in params], params[0][1].device)
        for n, p in params:
            st.view(n).copy_(p.data)
            p.data = st.view(n)
            p.grad = None
        self._store = st
        self._seen_version = None

    def _grad_slice(self):
        """This module's part of the flat gradient buffer (its parameters are contiguous there)."""
        st = self._store
        names = [self._prefix + n for n, _ in self.named_parameters()]
        lo = min(st.offsets[n] for n in names)
        hi = max(st.offsets[n] + st.gview(n).numel() for n in names)
        return st.grad[lo:hi]

    def _publish_grads(self, stash):
        """After a backward plan wrote fresh gradients into the flat buffer: expose them as
        `.grad` views (adding what was there before, as autograd would)."""
        st = self._store
        if stash is not None:
            self._grad_slice().add_(stash)
        for n, p in self.named_parameters():
            if p.requires_grad:
                gv = st.gview(self._prefix + n)
                if p.grad is None or p.grad.data_ptr() == gv.data_ptr():
                    p.grad = gv
                else:           # a foreign .grad tensor: accumulate into it
                    p.grad.add_(gv)

    def _stash_live_grads(self):
        """Gradients of an earlier backward still held in the flat buffer (no zero_grad since)."""
        st = self._store
        for n, p in self.named_parameters():
            if p.grad is not None and p.grad.data_ptr() == st.gview(self._prefix + n).data_ptr():
                return self._grad_slice().clone()
        return None


# =====================================================================================
# backbone
# =====================================================================================
class DropPath(_Holder):
    """Stochastic depth per sample (vit.py:48-59); the rate is applied by the HIP epilogues."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def extra_repr(self):
        return 'p={}'.format(self.drop_prob)


class Mlp(_Holder):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)


class Attention(_Holder):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads, self.dim = num_heads, dim
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, head_dim * num_heads * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(head_dim * num_heads, dim)
        self.proj_drop = nn.Dropout(proj_drop)


class Block(_Holder):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), drop=drop)


class PatchEmbed(_Holder):
    """vit.py:143-165: Conv2d(in, E, patch, stride = patch // ratio, padding = 4 + 2*(ratio//2 - 1))."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, ratio=1):
        super().__init__()
        img_size, patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * (ratio ** 2)
        self.patch_shape = (int(img_size[0] // patch_size[0] * ratio), int(img_size[1] // patch_size[1] * ratio))
        self.origin_patch_shape = (int(img_size[0] // patch_size[0]), int(img_size[1] // patch_size[1]))
        self.img_size, self.patch_size = img_size, patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=(patch_size[0] // ratio),
                              padding=4 + 2 * (ratio // 2 - 1))


class _ViTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, img, *params):
        ctx.mod, ctx.B = mod, img.shape[0]
        return mod._run(img, keep=True)

    @staticmethod
    def backward(ctx, dfeat):
        ctx.mod._run_backward(dfeat, ctx.B)
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


@BACKBONES.register_module()
class ViT(_FlatBound):
    """Plain ViT backbone of ViTPose (vit.py:200-341) on the HIP kernels."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=80, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., hybrid_backbone=None, norm_layer=None, use_checkpoint=False,
                 frozen_stages=-1, ratio=1, last_norm=True, patch_padding='pad', freeze_attn=False,
                 freeze_ffn=False):
        super().__init__()
        if hybrid_backbone is not None:
            raise NotImplementedError('hybrid_backbone (CNN patch embedding) is outside the HIP path')
        if norm_layer is not None:
            raise NotImplementedError('the HIP path implements LayerNorm(eps=1e-6), the reference default')
        if drop_rate or attn_drop_rate:
            raise NotImplementedError('drop_rate / attn_drop_rate > 0 (all ViTPose configs use 0)')
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.frozen_stages = frozen_stages
        self.use_checkpoint = use_checkpoint   # accepted; every activation fits HBM, nothing is recomputed
        self.patch_padding = patch_padding
        self.freeze_attn, self.freeze_ffn = freeze_attn, freeze_ffn
        self.depth = depth
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, ratio)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate, dpr[i], norm_layer)
            for i in range(depth)])
        self.last_norm = norm_layer(embed_dim) if last_norm else nn.Identity()
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        self._cfg = dict(img_size=to_2tuple(img_size), patch_size=to_2tuple(patch_size)[0], in_chans=in_chans,
                         embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                         qkv_bias=qkv_bias, qk_scale=qk_scale, ratio=ratio, last_norm=last_norm,
                         drop_path_rate=drop_path_rate)
        self._freeze_stages()

    # ---- reference API ----
    def _freeze_stages(self):
        """vit.py:249-284."""
        def freeze(m):
            m.eval()
            for p in m.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 0:
            freeze(self.patch_embed)
        for i in range(1, self.frozen_stages + 1):
            freeze(self.blocks[i])
        if self.freeze_attn:
            for blk in self.blocks:
                freeze(blk.attn)
                freeze(blk.norm1)
        if self.freeze_ffn:
            self.pos_embed.requires_grad = False
            freeze(self.patch_embed)
            for blk in self.blocks:
                freeze(blk.mlp)
                freeze(blk.norm2)

    def init_weights(self, pretrained=None):
        """vit.py:286-304 / base_backbone.py:17-34: a str loads a checkpoint, None self-initialises."""
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False, patch_padding=self.patch_padding)
        elif pretrained is None:
            def _init(m):
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=.02)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.LayerNorm):
                    nn.init.constant_(m.bias, 0)
                    nn.init.constant_(m.weight, 1.0)
            self.apply(_init)
        else:
            raise TypeError('pretrained must be a str or None.'
                            f' But received {type(pretrained)}.')

    def get_num_layers(self):
        return len(self.blocks)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        return self

    # ---- HIP execution ----
    def _runner(self):
        r = self._runners.get('vit')
        if r is None:
            r = self._runners['vit'] = VitRunner(self._cfg, self._store, self._prefix)
            prep = Plan()
            prep.cast_f32_bf16(self._store.data, self._store.half, self._store.numel)
            self._runners['prep'] = prep
        return r

    def _refresh(self):
        v = _param_versions(self.parameters())
        if v != self._seen_version:
            self._runners['prep'].run()
            self._runners['vit'].refresh_derived()
            self._seen_version = v

    def _plan(self, B, keep):
        key = (B, keep)
        if key not in self._plans:
            vr = self._runner()
            ws = self._runners.setdefault(('ws', keep), Workspace(self._store.device))
            H, W = vr.img_hw
            io = dict(img=ws.f32('img', B, vr.C, H, W), tok=ws.f32('tok_out', B * vr.T, vr.E),
                      out=ws.f32('feat_nchw', B, vr.E, vr.Hp, vr.Wp))
            fwd = Plan()
            dp = None
            if keep:
                dp = io['dp'] = ws.f32('dp', vr.depth, 2, B)
                dp.fill_(1.0)
            feat = vr.record_forward(fwd, ws, io['img'], B, training=keep, dp=dp, tokens_f32=io['tok'])
            # token-major [B][T][E] -> NCHW [B][E][Hp][Wp] (vit.py:330)
            fwd.strided_cast(io['tok'], io['out'], (B, vr.E, vr.T, 1), (vr.T * vr.E, 1, vr.E, 0))
            bwd = None
            if keep:
                io['dfeat'] = ws.f32('dfeat_nchw', B, vr.E, vr.Hp, vr.Wp)
                dcl = ws.bf16('dfeat_cl', B * vr.T, vr.E)
                bwd = Plan()
                bwd.nchw_to_cl(io['dfeat'], dcl, B, vr.E, vr.T, vr.E)
                vr.record_backward(bwd, ws, dcl, B, dp=dp)
            self._plans[key] = (fwd, bwd, io)
        return self._plans[key]

    def _run(self, img, keep):
        B, C, H, W = img.shape
        vr = self._runner()
        if (H, W) != tuple(vr.img_hw) or C != vr.C:
            raise ValueError(f'input {tuple(img.shape)} does not match img_size {vr.img_hw} / in_chans {vr.C}')
        self._refresh()
        fwd, _, io = self._plan(B, keep)
        io['img'].copy_(img)
        if vr.part:
            src = self._dataset_source
            vr.set_routing(self._runners[('ws', keep)], B, src if src is not None else [0] * B)
        if keep:
            if self.training and max(vr.dp_rates) > 0:
                vr.sample_drop_path(io['dp'], active=[blk.training for blk in self.blocks])
            else:
                io['dp'].fill_(1.0)
        fwd.run()
        return io['out'].clone()

    def _run_backward(self, dfeat, B):
        _, bwd, io = self._plan(B, True)
        stash = self._stash_live_grads()
        io['dfeat'].copy_(dfeat)
        bwd.run()
        self._publish_grads(stash)

    def forward_features(self, x):
        _need_hip(x, 'ViT.forward')
        if self._store is None:
            raise RuntimeError('move the module to the HIP device first (model.cuda())')
        params = [p for p in self.parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _ViTFunction.apply(self, x.float(), *params)
        return self._run(x.float(), keep=False)

    _dataset_source = None

    def forward(self, x):
        return self.forward_features(x)


class MoEMlp(_Holder):
    """vit_moe.py:78-115: fc2 emits the shared out - part_features channels, expert `indices[b]` (a
    Linear(hidden, part_features)) the rest of image b."""

    def __init__(self, num_expert=1, in_features=1024, hidden_features=None, out_features=None, drop=0.,
                 part_features=256):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.part_features = part_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features - part_features)
        self.drop = nn.Dropout(drop)
        self.num_expert = num_expert
        self.experts = nn.ModuleList([nn.Linear(hidden_features, part_features) for _ in range(num_expert)])


@BACKBONES.register_module()
class ViTMoE(ViT):
    """ViTPose+ backbone (vit_moe.py:241-385): a ViT whose FFNs route the last `part_features` channels through
    one of `num_expert` expert Linears chosen per image by `dataset_source`.  The reference evaluates every
    expert on every token and masks (vit_moe.py:107-111); here the GEMM reads the image's expert rows directly."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=80, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., hybrid_backbone=None, norm_layer=None, use_checkpoint=False,
                 frozen_stages=-1, ratio=1, last_norm=True, patch_padding='pad', freeze_attn=False,
                 freeze_ffn=False, num_expert=1, part_features=None):
        super().__init__(img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads, mlp_ratio,
                         qkv_bias, qk_scale, drop_rate, attn_drop_rate, drop_path_rate, hybrid_backbone, norm_layer,
                         use_checkpoint, frozen_stages, ratio, last_norm, patch_padding, freeze_attn, freeze_ffn)
        if not part_features:
            raise TypeError('ViTMoE needs part_features (vit_moe.py:86 subtracts it from the fc2 width)')
        self.part_features = part_features
        for blk in self.blocks:
            blk.mlp = MoEMlp(num_expert, embed_dim, int(embed_dim * mlp_ratio), drop=drop_rate,
                             part_features=part_features)
        self._cfg.update(num_expert=num_expert, part_features=part_features)
        self._freeze_stages()

    def init_weights(self, pretrained=None):
        """vit_moe.py:330-348; a checkpoint's fc2 rows are split into the shared part and (copied to every
        expert) the last part_features rows (mmcv_custom/checkpoint.py:396-404)."""
        if isinstance(pretrained, str):
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False, patch_padding=self.patch_padding,
                            part_features=self.part_features)
        else:
            super().init_weights(pretrained)

    def forward_features(self, x, dataset_source=None):
        self._dataset_source = dataset_source
        try:
            return super().forward_features(x)
        finally:
            self._dataset_source = None

    def forward(self, x, dataset_source=None):
        return self.forward_features(x, dataset_source)


# =====================================================================================
# loss
# =====================================================================================
class _MSEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, target, tw, loss_weight):
        B, K = output.shape[:2]
        HW = output[0, 0].numel()
        out = output.contiguous().float()
        loss = torch.zeros(1, dtype=torch.float32, device=out.device)
        dpred = torch.empty_like(out) if output.requires_grad else None
        p = Plan()
        p.mse_loss_nchw(out, target.contiguous().float(), None if tw is None else tw.reshape(B, K).contiguous().float(),
                        loss, dpred, B, K, HW, float(loss_weight))
        p.run()
        ctx.dpred = dpred
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.dpred * g if ctx.dpred is not None else None), None, None, None


@LOSSES.register_module()
class JointsMSELoss(nn.Module):
    """mse_loss.py:9-44: mean over joints of the MSE between (weighted) heat maps."""

    def __init__(self, use_target_weight=False, loss_weight=1.):
        super().__init__()
        self.use_target_weight = use_target_weight
        self.loss_weight = loss_weight

    def forward(self, output, target, target_weight):
        _need_hip(output, 'JointsMSELoss.forward')
        return _MSEFunction.apply(output, target, target_weight if self.use_target_weight else None,
                                  self.loss_weight)


# =====================================================================================
# head
# =====================================================================================
def flip_back(output_flipped, flip_pairs, target_type='GaussianHeatmap'):
    """core/post_processing/post_transforms.py:110-147 on a host array [N, K, H, W]: the maps of mirrored joints trade
    places and the width axis is reversed ('CombinedTarget': three maps per joint, the x-offset map changes sign)."""
    if output_flipped.ndim != 4:
        raise AssertionError('output_flipped should be [batch_size, num_keypoints, height, width]')
    N, C, H, W = output_flipped.shape
    per_joint = 3 if target_type.lower() == 'combinedtarget' else 1
    if per_joint == 3:
        output_flipped[:, 1::3] *= -1          # in place, as the reference does (the caller's array sees it)
    maps = output_flipped.reshape(N, C // per_joint, per_joint, H, W)
    order = np.arange(C // per_joint)
    for left, right in flip_pairs:
        order[left], order[right] = right, left
    return maps[:, order].reshape(N, C, H, W)[..., ::-1]


def transform_preds(coords, center, scale, output_size, use_udp=False):
    """core/post_processing/post_transforms.py:150-194: heat-map pixels -> image pixels of the box (center, scale * 200);
    UDP divides by size - 1.  coords [..., >= 2] (columns beyond x, y are returned as ones, like the reference)."""
    if coords.shape[-1] not in (2, 4, 5) or len(center) != 2 or len(scale) != 2 or len(output_size) != 2:
        raise AssertionError('coords [.., 2|4|5], center [2], scale [2], output_size [2] expected')
    box = np.asarray(scale, dtype=np.float64) * 200.0
    grid = np.asarray(output_size, dtype=np.float64) - (1.0 if use_udp else 0.0)
    out = np.ones_like(coords)
    out[..., :2] = coords[..., :2] * (box / grid) + np.asarray(center, dtype=np.float64) - 0.5 * box
    return out


def gaussian_kernel1d(ksize: int) -> np.ndarray:
    """The 1-D kernel cv2.GaussianBlur(src, (ksize, ksize), 0) builds (the reference blurs with sigma 0,
    top_down_eval.py:366,435,575): OpenCV's fixed tables for ksize <= 7, otherwise
    exp(-x^2 / (2 sigma^2)) with sigma = 0.3 * ((ksize - 1) / 2 - 1) + 0.8, normalised in double, stored as f32."""
    assert ksize % 2 == 1 and ksize > 0
    fixed = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize in fixed:
        return np.asarray(fixed[ksize], dtype=np.float32)
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-0.5 * x * x / (sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def _stream_ptr():
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# vp_decode_heatmaps / vp_decode_blurred modes
_MODE_NONE, _MODE_SHIFT, _MODE_DARK_UDP, _MODE_DARK, _MODE_MEGVII, _MODE_COMBINED = 0, 1, 2, 3, 4, 5


def _post_process_mode(unbiased, post_process, kernel, use_udp, target_type):
    """The argument rules of keypoints_from_heatmaps (top_down_eval.py:517-560) -> (device mode, normalised
    post_process).  Legacy spellings (booleans, unbiased=True) are accepted with a DeprecationWarning."""
    if unbiased and post_process in (False, None, 'megvii'):
        raise AssertionError('unbiased decoding needs post_process "default" / "unbiased"')
    if post_process in ('megvii', 'unbiased') and not kernel > 0:
        raise AssertionError('post_process "megvii" / "unbiased" blur the maps: kernel must be > 0')
    if use_udp and post_process == 'megvii':
        raise AssertionError('UDP decoding has no "megvii" variant')
    legacy = None
    if post_process is False:
        legacy, post_process = 'post_process=False', None
    elif post_process is True:
        legacy = f'post_process=True, unbiased={bool(unbiased)}'
        post_process = 'unbiased' if unbiased is True else 'default'
    elif post_process == 'default' and unbiased is True:
        legacy, post_process = 'unbiased=True', 'unbiased'
    if legacy is not None:
        warnings.warn(f'{legacy} is a deprecated spelling of post_process={post_process!r}', DeprecationWarning)
    if use_udp:
        tt = target_type.lower()
        if tt == 'gaussianheatmap':
            return _MODE_DARK_UDP, post_process
        if tt == 'combinedtarget':
            return _MODE_COMBINED, post_process
        raise ValueError("target_type should be either 'GaussianHeatmap' or 'CombinedTarget'")
    return {None: _MODE_NONE, 'default': _MODE_SHIFT, 'unbiased': _MODE_DARK, 'megvii': _MODE_MEGVII}[post_process], post_process


def device_decode_mode(unbiased=False, post_process='default', use_udp=False, target_type='GaussianHeatmap', kernel=11):
    """Which device mode evaluates these keypoints_from_heatmaps settings (every combination the reference
    accepts has one)."""
    return _post_process_mode(unbiased, post_process, kernel, use_udp, target_type)[0]


def keypoints_from_heatmaps_device(heatmaps: torch.Tensor, center, scale, mode: int, kernel=11, use_udp=False,
                                   valid_radius_factor=0.0546875):
    """The per-map work of keypoints_from_heatmaps (arg-max, blur, log, Newton / Taylor step, offsets) in one HIP
    launch -- heat maps stay on the GPU, [N, K, 3] numbers come back -- then the map to image coordinates."""
    import ctypes
    _need_hip(heatmaps, 'keypoints_from_heatmaps')
    N, C, H, W = heatmaps.shape
    K = C // 3 if mode == _MODE_COMBINED else C
    hm = heatmaps.detach().contiguous().float()
    preds_d = torch.empty(N * K, 2, dtype=torch.float32, device=hm.device)
    max_d = torch.empty(N * K, dtype=torch.float32, device=hm.device)
    L = _lib.lib()

    def carr(a):
        return (ctypes.c_float * len(a))(*np.asarray(a, dtype=np.float32).tolist())
    if mode <= _MODE_DARK_UDP:
        taps = gaussian_kernel1d(kernel) if mode == _MODE_DARK_UDP else np.zeros(1, np.float32)
        rc = L.vp_decode_heatmaps(hm.data_ptr(), preds_d.data_ptr(), max_d.data_ptr(), N * K, H, W, mode,
                                  carr(taps), len(taps), _stream_ptr())
    else:
        taps = gaussian_kernel1d(kernel)
        taps2 = gaussian_kernel1d(2 * kernel + 1) if mode == _MODE_COMBINED else taps
        rc = L.vp_decode_blurred(hm.data_ptr(), preds_d.data_ptr(), max_d.data_ptr(), N * K, H, W, mode,
                                 carr(taps), len(taps), carr(taps2), len(taps2), float(valid_radius_factor * H),
                                 _stream_ptr())
    if rc != 0:
        raise RuntimeError(f'heat-map decode failed with status {rc} (mode {mode}, kernel {kernel}, maps {H}x{W})')
    coords = preds_d.cpu().numpy().reshape(N, K, 2)
    maxvals = max_d.cpu().numpy().reshape(N, K, 1)
    # heat-map pixels -> image pixels, every person at once (transform_preds per person in the reference)
    box = np.asarray(scale, dtype=np.float64).reshape(N, 1, 2) * 200.0
    grid = np.array([W, H], dtype=np.float64) - (1.0 if use_udp else 0.0)
    preds = (coords * (box / grid) + np.asarray(center, dtype=np.float64).reshape(N, 1, 2) - 0.5 * box).astype(np.float32)
    return preds, maxvals


def keypoints_from_heatmaps(heatmaps, center, scale, unbiased=False, post_process='default', kernel=11,
                            valid_radius_factor=0.0546875, use_udp=False, target_type='GaussianHeatmap'):
    """core/evaluation/top_down_eval.py:474-622, every branch (post_process None / 'default' / 'unbiased' /
    'megvii', UDP with 'GaussianHeatmap' or 'CombinedTarget' targets), evaluated on the GPU: `heatmaps` may be
    a device tensor (stays there) or a host array (uploaded to the current HIP device; the array is not
    modified).  Returns (preds [N, K, 2] in image pixels, maxvals [N, K, 1]) as numpy arrays."""
    mode, _ = _post_process_mode(unbiased, post_process, kernel, use_udp, target_type)
    if not torch.is_tensor(heatmaps):
        if not torch.cuda.is_available():
            raise RuntimeError('keypoints_from_heatmaps runs on an MI355X (HIP device) only; there is no CPU path')
        heatmaps = torch.from_numpy(np.ascontiguousarray(heatmaps, dtype=np.float32)).cuda()
    return keypoints_from_heatmaps_device(heatmaps, center, scale, mode, kernel=kernel, use_udp=use_udp,
                                          valid_radius_factor=valid_radius_factor)


def flip_back_avg_device(heat: torch.Tensor, heat_flipped: torch.Tensor, flip_pairs, shift: bool) -> torch.Tensor:
    """(heat + flip_back(heat_flipped)) * 0.5 on the device (post_transforms.py:54-95, top_down.py:181-189)."""
    N, K, H, W = heat.shape
    perm = list(range(K))
    for left, right in flip_pairs:
        perm[left], perm[right] = right, left
    perm_d = torch.tensor(perm, dtype=torch.int32, device=heat.device)
    out = torch.empty_like(heat)
    rc = _lib.lib().vp_flip_back_avg(heat.data_ptr(), heat_flipped.data_ptr(), perm_d.data_ptr(), int(bool(shift)),
                                     out.data_ptr(), N, K, H, W, _stream_ptr())
    if rc != 0:
        raise RuntimeError(f'vp_flip_back_avg failed with status {rc}')
    return out


def pose_pck_accuracy_device(output, target, mask, thr=0.05):
    """pose_pck_accuracy (core/evaluation/top_down_eval.py:136-176) with the two arg-max passes
    (`_get_max_preds`, :63-95) on the device: only [N, K] indices and maxima cross to the host,
    not the 2 x N*K*H*W heat maps the reference copies every iteration.
    Returns (acc [K], avg_acc, cnt) like the reference."""
    N, K, H, W = output.shape
    if K == 0:
        return None, 0, 0
    idx = torch.empty(2, N * K, dtype=torch.int32, device=output.device)
    val = torch.empty(2, N * K, dtype=torch.float32, device=output.device)
    p = Plan()
    p.argmax_rows(output.detach().contiguous().float(), N * K, H * W, idx[0], val[0])
    p.argmax_rows(target.detach().contiguous().float(), N * K, H * W, idx[1], val[1])
    p.run()
    idx_h, val_h = idx.cpu().numpy().astype(np.int64), val.cpu().numpy()
    coords = []
    for i in range(2):
        xy = np.stack([idx_h[i] % W, idx_h[i] // W], -1).astype(np.float32).reshape(N, K, 2)
        coords.append(np.where((val_h[i] > 0.0).reshape(N, K, 1), xy, -1))
    pred, gt = coords
    norm = np.tile(np.array([[H, W]], dtype=np.float32), (N, 1))
    mask = np.asarray(mask).copy()
    # _calc_distances (:10-54): joints whose normaliser is <= 0 are masked out
    mask[np.where((norm <= 0).sum(1))[0], :] = False
    dist = np.full((N, K), -1, dtype=np.float32)
    norm_c = norm.copy()
    norm_c[np.where(norm_c <= 0)] = 1e6
    d = np.linalg.norm((pred - gt) / norm_c[:, None, :], axis=-1)
    dist[mask] = d[mask]
    dist = dist.T
    acc = np.array([((dk[dk != -1] < thr).sum() / (dk != -1).sum()) if (dk != -1).sum() > 0 else -1 for dk in dist])
    valid = acc[acc >= 0]
    cnt = len(valid)
    return acc, (valid.mean() if cnt > 0 else 0), cnt


class _HeadFunction(torch.autograd.Function):
    """nx feature maps (one, or the maps 'resize_concat' joins), then the parameters."""

    @staticmethod
    def forward(ctx, mod, nx, *args):
        xs = args[:nx]
        ctx.mod, ctx.shapes = mod, tuple(tuple(x.shape) for x in xs)
        return mod._run(xs, keep=True)

    @staticmethod
    def backward(ctx, dheat):
        dxs = ctx.mod._run_backward(dheat, ctx.shapes)
        return (None, None) + tuple(dxs) + (None,) * (len(ctx.needs_input_grad) - 2 - len(dxs))


@HEADS.register_module()
class TopdownHeatmapSimpleHead(_FlatBound):
    """topdown_heatmap_simple_head.py:16-350: deconv x n + BN + ReLU + final conv ("classic"), or
    ReLU + bilinear upsample + final conv ("simple decoder"), on the HIP kernels."""

    def __init__(self, in_channels, out_channels, num_deconv_layers=3, num_deconv_filters=(256, 256, 256),
                 num_deconv_kernels=(4, 4, 4), extra=None, in_index=0, input_transform=None, align_corners=False,
                 loss_keypoint=None, train_cfg=None, test_cfg=None, upsample=0):
        super().__init__()
        self.in_channels = in_channels
        self.loss = build_loss(loss_keypoint)
        self.upsample = upsample
        self.train_cfg = {} if train_cfg is None else train_cfg
        self.test_cfg = {} if test_cfg is None else test_cfg
        self.target_type = self.test_cfg.get('target_type', 'GaussianHeatmap')
        self._init_inputs(in_channels, in_index, input_transform)
        self.in_index = in_index
        self.align_corners = align_corners
        if extra is not None and not isinstance(extra, dict):
            raise TypeError('extra should be dict or None.')
        head_in = self.in_channels
        if num_deconv_layers > 0:
            self.deconv_layers = self._make_deconv_layer(num_deconv_layers, num_deconv_filters, num_deconv_kernels)
        elif num_deconv_layers == 0:
            self.deconv_layers = nn.Identity()
        else:
            raise ValueError(f'num_deconv_layers ({num_deconv_layers}) should >= 0.')
        identity_final_layer = False
        if extra is not None and 'final_conv_kernel' in extra:
            assert extra['final_conv_kernel'] in [0, 1, 3]
            if extra['final_conv_kernel'] == 3:
                padding = 1
            elif extra['final_conv_kernel'] == 1:
                padding = 0
            else:
                identity_final_layer = True
            kernel_size = extra['final_conv_kernel']
        else:
            kernel_size, padding = 1, 0
        if identity_final_layer:
            self.final_layer = nn.Identity()
        else:
            conv_channels = num_deconv_filters[-1] if num_deconv_layers > 0 else self.in_channels
            layers = []
            if extra is not None:
                num_conv_layers = extra.get('num_conv_layers', 0)
                num_conv_kernels = extra.get('num_conv_kernels', [1] * num_conv_layers)
                for i in range(num_conv_layers):
                    layers += [nn.Conv2d(conv_channels, conv_channels, num_conv_kernels[i], 1,
                                         (num_conv_kernels[i] - 1) // 2),
                               nn.BatchNorm2d(conv_channels), nn.ReLU(inplace=True)]
            layers.append(nn.Conv2d(conv_channels, out_channels, kernel_size, 1, padding))
            self.final_layer = nn.Sequential(*layers) if len(layers) > 1 else layers[0]
        self._cfg = dict(in_channels=head_in, out_channels=out_channels, num_deconv_layers=num_deconv_layers,
                         num_deconv_filters=tuple(num_deconv_filters), num_deconv_kernels=tuple(num_deconv_kernels),
                         extra=dict(extra or {}), upsample=upsample, align_corners=align_corners)
        self._cfg['extra'].setdefault('final_conv_kernel', 1)

    # ---- reference API ----
    def _init_inputs(self, in_channels, in_index, input_transform):
        """topdown_heatmap_simple_head.py:229-267."""
        if input_transform is not None:
            assert input_transform in ['resize_concat', 'multiple_select']
        self.input_transform = input_transform
        self.in_index = in_index
        if input_transform is not None:
            assert isinstance(in_channels, (list, tuple))
            assert isinstance(in_index, (list, tuple))
            assert len(in_channels) == len(in_index)
            self.in_channels = sum(in_channels) if input_transform == 'resize_concat' else in_channels
        else:
            assert isinstance(in_channels, int)
            assert isinstance(in_index, int)
            self.in_channels = in_channels

    @staticmethod
    def _get_deconv_cfg(deconv_kernel):
        """topdown_heatmap_base_head.py:105-120."""
        if deconv_kernel == 4:
            return 4, 1, 0
        if deconv_kernel == 3:
            return 3, 1, 1
        if deconv_kernel == 2:
            return 2, 0, 0
        raise ValueError(f'Not supported num_kernels ({deconv_kernel}).')

    def _make_deconv_layer(self, num_layers, num_filters, num_kernels):
        if num_layers != len(num_filters):
            raise ValueError(f'num_layers({num_layers}) != length of num_filters({len(num_filters)})')
        if num_layers != len(num_kernels):
            raise ValueError(f'num_layers({num_layers}) != length of num_kernels({len(num_kernels)})')
        layers = []
        for i in range(num_layers):
            kernel, padding, output_padding = self._get_deconv_cfg(num_kernels[i])
            planes = num_filters[i]
            layers += [nn.ConvTranspose2d(self.in_channels, planes, kernel, stride=2, padding=padding,
                                          output_padding=output_padding, bias=False),
                       nn.BatchNorm2d(planes), nn.ReLU(inplace=True)]
            self.in_channels = planes
        return nn.Sequential(*layers)

    def init_weights(self):
        """topdown_heatmap_simple_head.py:339-350."""
        for _, m in self.deconv_layers.named_modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, 0, 0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.final_layer.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, 0, 0.001)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def get_loss(self, output, target, target_weight):
        losses = dict()
        assert not isinstance(self.loss, nn.Sequential)
        assert target.dim() == 4 and target_weight.dim() == 3
        losses['heatmap_loss'] = self.loss(output, target, target_weight)
        return losses

    def get_accuracy(self, output, target, target_weight):
        accuracy = dict()
        if self.target_type == 'GaussianHeatmap':
            _, avg_acc, _ = pose_pck_accuracy_device(
                output, target, target_weight.detach().cpu().numpy().squeeze(-1) > 0)
            accuracy['acc_pose'] = float(avg_acc)
        return accuracy

    def inference_model(self, x, flip_pairs=None):
        output = self.forward(x)
        if flip_pairs is not None:
            output_heatmap = flip_back(output.detach().cpu().numpy(), flip_pairs, target_type=self.target_type)
            if self.test_cfg.get('shift_heatmap', False):
                output_heatmap[:, :, :, 1:] = output_heatmap[:, :, :, :-1]
        else:
            output_heatmap = output.detach().cpu().numpy()
        return output_heatmap

    def decode(self, img_metas, output, **kwargs):
        """topdown_heatmap_base_head.py:40-103: heat maps -> `preds` [N, K, 3] (x, y, score in image pixels),
        `boxes` [N, 6] (center, scale, area, score), `image_paths`, `bbox_ids`.  The heat maps are decoded on
        the GPU whatever test_cfg selects (a host array is uploaded first)."""
        n = len(img_metas)
        center = np.array([np.asarray(m['center'], dtype=np.float32).reshape(2) for m in img_metas], dtype=np.float32)
        scale = np.array([np.asarray(m['scale'], dtype=np.float32).reshape(2) for m in img_metas], dtype=np.float32)
        score = np.array([np.asarray(m['bbox_score']).reshape(-1)[0] if 'bbox_score' in m else 1.0 for m in img_metas],
                         dtype=np.float64)
        cfg = self.test_cfg
        preds, maxvals = keypoints_from_heatmaps(
            output, center, scale, unbiased=cfg.get('unbiased_decoding', False),
            post_process=cfg.get('post_process', 'default'), kernel=cfg.get('modulate_kernel', 11),
            valid_radius_factor=cfg.get('valid_radius_factor', 0.0546875), use_udp=cfg.get('use_udp', False),
            target_type=cfg.get('target_type', 'GaussianHeatmap'))
        boxes = np.concatenate([center, scale, np.prod(scale * 200.0, axis=1, keepdims=True), score[:, None]],
                               axis=1).astype(np.float32)
        return dict(preds=np.concatenate([preds[:, :, :2], maxvals], axis=2).astype(np.float32), boxes=boxes,
                    image_paths=[m['image_file'] for m in img_metas],
                    bbox_ids=[m['bbox_id'] for m in img_metas] if 'bbox_id' in img_metas[0] else None)

    def _transform_inputs(self, inputs):
        """topdown_heatmap_simple_head.py:269-304 (the ReLU + resize of a tensor input happens inside
        the HIP plan; feature lists select `in_index`)."""
        if not isinstance(inputs, list):
            return inputs
        if self.input_transform == 'resize_concat':
            # the maps are resized and joined inside the HIP plan (vp_resize_to_cl_bf16 writes each one into its
            # channel range of the channels-last matrix)
            return [inputs[i] for i in self.in_index]
        if self.input_transform == 'multiple_select':
            # the reference hands the list on to nn.Sequential (topdown_heatmap_simple_head.py:199-201, :299-300),
            # which cannot take one: the option is not executable with this head there either
            raise NotImplementedError("input_transform='multiple_select': this head has a single input")
        return inputs[self.in_index]

    # BatchNorm2d.num_batches_tracked (a LongTensor buffer the reference's modules increment every training
    # forward): counted on the host and added to the buffers when the state dict is read -- one device add per
    # BatchNorm per step is pure launch overhead in the step
    _bn_steps_pending = 0

    def flush_bn_counters(self):
        if self._bn_steps_pending:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None:
                    m.num_batches_tracked += self._bn_steps_pending
            self._bn_steps_pending = 0

    def state_dict(self, *args, **kwargs):
        self.flush_bn_counters()
        return super().state_dict(*args, **kwargs)

    def _save_to_state_dict(self, *args, **kwargs):
        self.flush_bn_counters()
        return super()._save_to_state_dict(*args, **kwargs)

    # ---- HIP execution ----
    def _bn_buffers(self):
        return {self._prefix + n: b for n, b in self.named_buffers() if b.dtype == torch.float32}

    def _runner(self, grid):
        key = ('head', grid)
        r = self._runners.get(key)
        if r is None:
            r = self._runners[key] = HeadRunner(self._cfg, self._store, self._bn_buffers(), grid, self._prefix)
            prep = Plan()
            prep.cast_f32_bf16(self._store.data, self._store.half, self._store.numel)
            r.record_weight_prep(prep)
            self._runners[('prep', grid)] = prep
            self._seen_version = None
        return r

    def _resize_tabs(self, src_hw, dst_hw):
        """Device tables of the bilinear resize src_hw -> dst_hw (ops.py:7-29 with `size=`): (forward, transposed)."""
        key = ('rtab', src_hw, dst_hw)
        if key not in self._runners:
            dev = self._store.device
            ty = bilinear_tables(src_hw[0], dst_hw[0], None, bool(self.align_corners))
            tx = bilinear_tables(src_hw[1], dst_hw[1], None, bool(self.align_corners))
            self._runners[key] = (tuple(t.reshape(-1).contiguous().to(dev) for t in (ty[0], ty[1], tx[0], tx[1])),
                                  tuple(t.contiguous().to(dev) for t in (ty[2], ty[3], tx[2], tx[3])))
        return self._runners[key]

    def _plan(self, shapes, keep):
        """shapes: the NCHW shape of every input map (one, or the maps 'resize_concat' resizes to the first one's
        grid and joins along the channels)."""
        B, _, H, W = shapes[0]
        C = sum(s[1] for s in shapes)
        key = (shapes, keep, self.training)
        if key not in self._plans:
            hr = self._runner((H, W))
            ws = self._runners.setdefault(('ws', keep), Workspace(self._store.device))
            io = dict(xs=[ws.f32(f'x_nchw{i}', *s) for i, s in enumerate(shapes)], feat=ws.bf16('feat_cl', B * H * W, C),
                      heat=ws.f32('heat', B, hr.K, *hr.out_grid))
            fwd = Plan()
            if len(shapes) == 1:
                fwd.nchw_to_cl(io['xs'][0], io['feat'], B, C, H * W, C)
            else:
                coff = 0
                for x, s in zip(io['xs'], shapes):
                    fwd.resize_to_cl(x, io['feat'], self._resize_tabs(s[2:], (H, W))[0], B, s[1], s[2], s[3], H, W, C, coff)
                    coff += s[1]
            # eval mode with a gradient wanted: the deconv stack keeps its intermediates as in training, but its
            # BatchNorms normalise with (and do not update) the running statistics
            frozen_bn = keep and not self.training and not hr.simple and hr.nl + hr.nconv > 0
            if frozen_bn:
                hr.record_forward(fwd, ws, io['feat'], B, True, io['heat'], running_stats=True)
            else:
                hr.record_forward(fwd, ws, io['feat'], B, keep and self.training, io['heat'])
            bwd = None
            if keep:
                io['dheat'] = ws.f32('dheat_in', B, hr.K, *hr.out_grid)
                io['dxs'] = [ws.f32(f'dx_nchw{i}', *s) for i, s in enumerate(shapes)]
                bwd = Plan()
                g = hr.record_backward_nchw(bwd, ws, io['feat'], io['dheat'], B, running_stats=frozen_bn)
                if len(shapes) == 1:
                    bwd.strided_cast(g, io['dxs'][0], (B, C, H * W, 1), (H * W * C, 1, C, 0))
                else:
                    coff = 0
                    for dx, s in zip(io['dxs'], shapes):
                        bwd.resize_to_cl_bwd(g, dx, self._resize_tabs(s[2:], (H, W))[1], B, s[1], s[2], s[3], H, W, C, coff)
                        coff += s[1]
            self._plans[key] = (fwd, bwd, io)
        return self._plans[key]

    def _refresh(self, grid):
        self._runner(grid)
        v = _param_versions(self.parameters())
        if v != self._seen_version:
            self._runners[('prep', grid)].run()
            self._seen_version = v

    def _run(self, xs, keep):
        shapes = tuple(tuple(x.shape) for x in xs)
        if sum(s[1] for s in shapes) != self._cfg['in_channels']:
            raise ValueError(f'expected {self._cfg["in_channels"]} input channels, got {[s[1] for s in shapes]}')
        if any(s[0] != shapes[0][0] or s[1] % 8 for s in shapes[1:]):
            raise ValueError(f'resize_concat inputs need one batch size and channel counts in multiples of 8: {shapes}')
        self._refresh(shapes[0][2:])
        fwd, _, io = self._plan(shapes, keep)
        for buf, x in zip(io['xs'], xs):
            buf.copy_(x)
        fwd.run()
        if keep and self.training:
            self._bn_steps_pending += 1
        return io['heat'].clone()

    def _run_backward(self, dheat, shapes):
        _, bwd, io = self._plan(shapes, True)
        stash = self._stash_live_grads()
        io['dheat'].copy_(dheat)
        bwd.run()
        self._publish_grads(stash)
        return [dx.clone() for dx in io['dxs']]

    def forward(self, x):
        x = self._transform_inputs(x)
        xs = [t.float() for t in x] if isinstance(x, list) else [x.float()]
        for t in xs:
            _need_hip(t, 'TopdownHeatmapSimpleHead.forward')
        if self._store is None:
            raise RuntimeError('move the module to the HIP device first (model.cuda())')
        params = list(self.parameters())
        if torch.is_grad_enabled() and (any(t.requires_grad for t in xs) or any(p.requires_grad for p in params)):
            return _HeadFunction.apply(self, len(xs), *xs, *params)
        return self._run(xs, keep=False)


# =====================================================================================
# detector
# =====================================================================================
class _FusedTrainFunction(torch.autograd.Function):
    """forward = the forward half of the fused PoseEngine plan (backbone + head + loss), backward =
    its backward half; gradients land in the flat buffer the parameters' .grad alias."""

    @staticmethod
    def forward(ctx, mod, img, target, tw, *params):
        ctx.mod, ctx.B = mod, img.shape[0]
        return mod._fused_forward(img, target, tw)

    @staticmethod
    def backward(ctx, g):
        ctx.mod._fused_backward(ctx.B, g)
        return (None,) * len(ctx.needs_input_grad)


@POSENETS.register_module()
class TopDown(nn.Module):
    """Top-down pose detector container (top_down.py:22-218, detectors/base.py:38-126)."""

    def __init__(self, backbone, neck=None, keypoint_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 loss_pose=None):
        super().__init__()
        self.fp16_enabled = False
        self.backbone = build_backbone(backbone)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self._src = None
        if neck is not None:
            self.neck = build_neck(neck)
        if keypoint_head is not None:
            keypoint_head['train_cfg'] = train_cfg
            keypoint_head['test_cfg'] = test_cfg
            if 'loss_keypoint' not in keypoint_head and loss_pose is not None:
                warnings.warn('`loss_pose` for TopDown is deprecated, use `loss_keypoint` for heads instead.',
                              DeprecationWarning)
                keypoint_head['loss_keypoint'] = loss_pose
            self.keypoint_head = build_head(keypoint_head)
        self._engine = None
        self.init_weights(pretrained=pretrained)

    @property
    def with_neck(self):
        return hasattr(self, 'neck')

    @property
    def with_keypoint(self):
        return hasattr(self, 'keypoint_head')

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained)
        if self.with_neck:
            self.neck.init_weights()
        if self.with_keypoint:
            self.keypoint_head.init_weights()

    # ---- fused storage / plan ----
    def _heads(self):
        """(state-dict prefix, head) of every head: the main one, then TopDownMoE's associate heads."""
        out = [('keypoint_head.', self.keypoint_head)] if self.with_keypoint else []
        return out + [(f'associate_keypoint_heads.{j}.', h)
                      for j, h in enumerate(getattr(self, 'associate_keypoint_heads', []))]

    def _src_of(self, img_metas):
        """Per-image dataset indices (TopDownMoE only)."""
        return None

    def _backbone(self, img, img_metas=None):
        return self.backbone(img)

    def _fusable(self):
        return (isinstance(self.backbone, ViT) and not self.with_neck and self.with_keypoint and
                all(isinstance(h, TopdownHeatmapSimpleHead) and isinstance(h.loss, JointsMSELoss)
                    for _, h in self._heads()))

    def _apply(self, fn, *a, **k):
        if self._fusable():
            self.backbone._owned = True
            for _, h in self._heads():
                h._owned = True
        out = super()._apply(fn, *a, **k)
        if self._fusable():
            self._bind()
        return out

    def _bind(self):
        from .pose_engine import PoseEngine
        self._engine = None
        bb, hd = self.backbone, self.keypoint_head
        heads = self._heads()
        mods = [bb] + [h for _, h in heads]
        for m in mods:
            m._store, m._runners, m._plans = None, {}, {}
            m._prefix = ''
        dev = bb.pos_embed.device
        if dev.type != 'cuda':
            for m in mods:
                m._owned = False
            return
        try:
            eng = PoseEngine(bb._cfg, hd._cfg, dev, loss_weight=hd.loss.loss_weight,
                             use_target_weight=hd.loss.use_target_weight,
                             assoc_heads=[(h._cfg, h.loss.loss_weight, h.loss.use_target_weight)
                                          for _, h in heads[1:]])
        except NotImplementedError:
            # a shape outside the fused plan's kernels: fall back to module-by-module HIP execution
            for m in mods:
                m._owned = False
                m._bind()
            return
        st = eng.store
        for n, p in self.named_parameters():
            st.view(n).copy_(p.data)
            p.data = st.view(n)
            p.grad = None
        for n, b in self.named_buffers():
            if n in eng.buffers:
                eng.buffers[n].copy_(b.data)
                b.data = eng.buffers[n]
        # the sub-modules run from the same store when called on their own
        for pre, m in [('backbone.', bb)] + heads:
            m._store, m._prefix, m._seen_version = st, pre, None
        self._grad_views = [(p, st.gview(n)) for n, p in self.named_parameters()]
        self._engine = eng
        # bf16 mirror and re-laid-out weights follow the f32 masters from the start (the engine is usable directly)
        eng.refresh_weights()
        self._seen_version = _param_versions(self.parameters())

    def _refresh(self):
        v = _param_versions(self.parameters())
        if v != self._seen_version:
            self._engine.refresh_weights()
            self._seen_version = v

    def _fused_forward(self, img, target, tw):
        eng = self._engine
        B = img.shape[0]
        self._refresh()
        plan, io = eng._train_plan(B)
        eng.load_batch(io, img, target, tw, self._src)
        vr = eng.vit
        if self.backbone.training and max(vr.dp_rates) > 0:
            vr.sample_drop_path(io['dp'], active=[blk.training for blk in self.backbone.blocks])
        else:
            io['dp'].fill_(1.0)
        self._stash = self._stash_live()
        plan.run(first=0, last=eng.n_fwd_calls)
        for _, h in self._heads():
            if h.training:
                h._bn_steps_pending += 1          # num_batches_tracked catches up when somebody looks (state_dict)
        self._heat, self._heats = io['heat'], io['heats']
        if len(io['losses']) > 1:
            return torch.cat(io['losses'])       # one loss per head
        return io['loss'].reshape(()).clone()

    def _stash_live(self):
        for p, gv in self._grad_views:
            if p.grad is not None and p.grad.data_ptr() == gv.data_ptr():
                return self._engine.store.grad.clone()
        return None

    def _fused_backward(self, B, g):
        eng = self._engine
        plan, io = eng._train_plan(B)
        # every parameter gradient is linear in the heat-map gradient the forward half left behind:
        # seeding with the incoming dL/dloss there costs one small kernel and no host sync
        gs = g.reshape(-1).float()
        for j, dh in enumerate(io['dheats']):
            eng.scale_dheat(dh, gs[j:j + 1] if gs.numel() > 1 else gs)
        plan.run(first=eng.n_fwd_calls)
        st = eng.store
        if self._stash is not None:
            st.grad.add_(self._stash)
            self._stash = None
        for p, gv in self._grad_views:
            if p.requires_grad:
                if p.grad is None or p.grad.data_ptr() == gv.data_ptr():
                    p.grad = gv
                else:
                    p.grad.add_(gv)

    # ---- reference call contract ----
    def forward(self, img, target=None, target_weight=None, img_metas=None, return_loss=True,
                return_heatmap=False, **kwargs):
        if return_loss:
            return self.forward_train(img, target, target_weight, img_metas, **kwargs)
        return self.forward_test(img, img_metas, return_heatmap=return_heatmap, **kwargs)

    def forward_train(self, img, target, target_weight, img_metas, **kwargs):
        """top_down.py:143-161."""
        if self._engine is not None and self.training and torch.is_grad_enabled():
            _need_hip(img, 'TopDown.forward_train')
            assert target.dim() == 4 and target_weight.dim() == 3
            # the zero-size token makes autograd record the node although the plan owns the weights
            loss = _FusedTrainFunction.apply(self, img.float(), target.float(), target_weight.float(),
                                             *self.parameters())
            losses = dict(heatmap_loss=loss)
            if not (self.train_cfg or {}).get('skip_accuracy', False):
                losses.update(self.keypoint_head.get_accuracy(self._heat, target, target_weight))
            return losses
        output = self.backbone(img)
        if self.with_neck:
            output = self.neck(output)
        if self.with_keypoint:
            output = self.keypoint_head(output)
        losses = dict()
        if self.with_keypoint:
            losses.update(self.keypoint_head.get_loss(output, target, target_weight))
            losses.update(self.keypoint_head.get_accuracy(output, target, target_weight))
        return losses

    def _exact(self):
        """test_cfg['fp32_exact'] (extension): evaluate at the reference's fp32 accuracy instead of bf16."""
        if self._engine is not None:
            # test_cfg['fold_layernorm'] (extension, bf16 evaluation only): norm1 / norm2 folded into qkv / fc1
            self._engine.fold_layernorm = bool((self.test_cfg or {}).get('fold_layernorm', False))
        return bool((self.test_cfg or {}).get('fp32_exact', False))

    def _infer_heatmap(self, img, flip_pairs=None, img_metas=None):
        """backbone + head.inference_model (-> host array); one fused plan when available."""
        if self._engine is not None and not self.training:
            _need_hip(img, 'TopDown.forward_test')
            self._refresh()
            heat = self._engine.infer(img.float(), copy_out=False, exact=self._exact(),
                                      dataset_source=self._src_of(img_metas)).cpu().numpy()
            if flip_pairs is not None:
                heat = flip_back(heat, flip_pairs, target_type=self.keypoint_head.target_type)
                if self.keypoint_head.test_cfg.get('shift_heatmap', False):
                    heat[:, :, :, 1:] = heat[:, :, :, :-1]
            return heat
        features = self._backbone(img, img_metas)
        if self.with_neck:
            features = self.neck(features)
        return self.keypoint_head.inference_model(features, flip_pairs=flip_pairs)

    @torch.no_grad()
    def forward_test(self, img, img_metas, return_heatmap=False, **kwargs):
        """top_down.py:163-200.  With the fused HIP plan the heat maps stay on the GPU through the flip-test
        average and the decode; they are copied to the host only when `return_heatmap` asks for them."""
        assert img.size(0) == len(img_metas)
        batch_size, _, img_height, img_width = img.shape
        if batch_size > 1:
            assert 'bbox_id' in img_metas[0]
        result = {}
        head = self.keypoint_head if self.with_keypoint else None
        flip = self.test_cfg.get('flip_test', True)
        on_device = (self._engine is not None and not self.training and head is not None
                     and head.target_type == 'GaussianHeatmap')
        if on_device:
            _need_hip(img, 'TopDown.forward_test')
            self._refresh()
            x = img.float()
            src = self._src_of(img_metas)
            heat = self._engine.infer(x, exact=self._exact(), dataset_source=src)   # a copy: the plan's buffer is reused
            if flip:
                hf = self._engine.infer(x.flip(3), copy_out=False, exact=self._exact(), dataset_source=src)
                heat = flip_back_avg_device(heat, hf, img_metas[0]['flip_pairs'],
                                            head.test_cfg.get('shift_heatmap', False))
            result.update(head.decode(img_metas, heat, img_size=[img_width, img_height]))
            result['output_heatmap'] = heat.cpu().numpy() if return_heatmap else None
            return result
        output_heatmap = self._infer_heatmap(img, img_metas=img_metas)
        if flip:
            output_flipped_heatmap = self._infer_heatmap(img.flip(3), img_metas[0]['flip_pairs'], img_metas)
            output_heatmap = (output_heatmap + output_flipped_heatmap) * 0.5
        if self.with_keypoint:
            result.update(self.keypoint_head.decode(img_metas, output_heatmap, img_size=[img_width, img_height]))
            if not return_heatmap:
                output_heatmap = None
            result['output_heatmap'] = output_heatmap
        return result

    def forward_dummy(self, img):
        """top_down.py:202-218."""
        if self._engine is not None and not self.training and not torch.is_grad_enabled():
            self._refresh()
            return self._engine.infer(img.float(), exact=self._exact())
        output = self.backbone(img)
        if self.with_neck:
            output = self.neck(output)
        if self.with_keypoint:
            output = self.keypoint_head(output)
        return output

    @staticmethod
    def _parse_losses(losses):
        """detectors/base.py:38-76."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, float):
                log_vars[name] = value
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors or float')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        # The reference all-reduces and reads back (.item()) every logged scalar by itself (base.py:66-72): two
        # collectives and two host syncs per iteration for ViTPose.  Here all tensor-valued entries travel as ONE small
        # vector: one collective (only when a process group exists), one copy to the host.
        names = [k for k, v in log_vars.items() if not isinstance(v, float)]
        if names:
            from .ddp import reduce_scalars
            dev = loss.device if isinstance(loss, torch.Tensor) else log_vars[names[0]].device
            vals = torch.stack([log_vars[k].detach().to(dev, torch.float32).reshape(()) for k in names])
            if dist.is_available() and dist.is_initialized():
                vals = reduce_scalars(vals)
            for k, v in zip(names, vals.tolist()):
                log_vars[k] = v
        return loss, log_vars

    def train_step(self, data_batch, optimizer, **kwargs):
        """detectors/base.py:78-113."""
        losses = self.forward(**data_batch)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    def val_step(self, data_batch, optimizer, **kwargs):
        """detectors/base.py:115-126."""
        return dict(results=self.forward(return_loss=False, **data_batch))

    # ---- optimizer / data parallel (the reference builds these in mmpose/apis/train.py:116-200) ----
    def build_optimizer(self, optimizer_cfg: dict, grad_clip: Optional[dict] = None):
        """Fused layer-decay AdamW over the flat store from the reference's `optimizer=dict(type='AdamW',
        ..., constructor='LayerDecayOptimizerConstructor', paramwise_cfg=dict(num_layers, layer_decay_rate))`
        and `optimizer_config=dict(grad_clip=dict(max_norm, norm_type=2))` blocks."""
        if self._engine is None:
            raise RuntimeError('build_optimizer needs the fused model on a HIP device (model.cuda())')
        cfg = dict(optimizer_cfg)
        if cfg.pop('type', 'AdamW') != 'AdamW':
            raise NotImplementedError('the HIP optimizer implements AdamW')
        pw = cfg.get('paramwise_cfg') or {}
        if grad_clip and grad_clip.get('norm_type', 2) != 2:
            raise NotImplementedError('grad_clip norm_type must be 2')
        frozen = [n for n, p in self.named_parameters() if not p.requires_grad]
        self._engine.setup_optimizer(lr=cfg.get('lr', 1e-3), betas=tuple(cfg.get('betas', (0.9, 0.999))),
                                     weight_decay=cfg.get('weight_decay', 1e-2),
                                     layer_decay_rate=pw.get('layer_decay_rate', 1.0),
                                     num_layers=pw.get('num_layers', self.backbone.depth),
                                     max_grad_norm=(grad_clip or {}).get('max_norm', 0.0), frozen=frozen)
        return _FlatOptimizer(self)

    def enable_ddp(self, process_group=None, blocks_per_bucket=4):
        """Batch-sharded data parallelism over RCCL: gradient buckets are all-reduced from inside the
        backward plan (replaces MMDistributedDataParallel, mmpose/apis/train.py:116-133)."""
        if self._engine is None:
            raise RuntimeError('enable_ddp needs the fused model on a HIP device (model.cuda())')
        return self._engine.enable_ddp(process_group, blocks_per_bucket)


@POSENETS.register_module()
class TopDownMoE(TopDown):
    """ViTPose+ detector (top_down_moe.py:26-262): a ViTMoE backbone routed by img_metas[i]['dataset_idx'], the
    main head for dataset 0 and one associate head per further dataset; every head sees every image, its loss
    only the images of its own dataset (targets and target weights of the others are zeroed)."""

    def __init__(self, backbone, neck=None, keypoint_head=None, associate_keypoint_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None, loss_pose=None):
        super().__init__(backbone, neck, keypoint_head, train_cfg, test_cfg, pretrained, loss_pose)
        heads = []
        if associate_keypoint_head is not None:
            if not isinstance(associate_keypoint_head, list):
                associate_keypoint_head = [associate_keypoint_head]
            for cfg in associate_keypoint_head:
                cfg['train_cfg'] = train_cfg
                cfg['test_cfg'] = test_cfg
                heads.append(build_head(cfg))
        self.associate_keypoint_heads = nn.ModuleList(heads)
        self.keypoint_heads_cnt = 1 + len(heads)
        for h in heads:
            h.init_weights()

    def init_weights(self, pretrained=None):
        super().init_weights(pretrained)
        for h in getattr(self, 'associate_keypoint_heads', []):
            h.init_weights()

    def _src_of(self, img_metas):
        return [int(m['dataset_idx']) for m in img_metas]

    def _backbone(self, img, img_metas=None):
        if img_metas is None:
            return self.backbone(img)
        return self.backbone(img, torch.as_tensor(self._src_of(img_metas), device=img.device))

    def forward_train(self, img, target, target_weight, img_metas, **kwargs):
        """top_down_moe.py:166-203."""
        src_host = self._src_of(img_metas)
        src = torch.as_tensor(src_host, device=img.device)
        names = [('main_stream_loss', 'main_stream_acc')] + \
                [(f'{i}_loss', f'{i}_acc') for i in range(1, self.keypoint_heads_cnt)]
        heads = [h for _, h in self._heads()]
        losses = dict()
        fused = self._engine is not None and self.training and torch.is_grad_enabled()
        if fused:
            _need_hip(img, 'TopDownMoE.forward_train')
            assert target.dim() == 4 and target_weight.dim() == 3
            self._src = src_host
            try:
                loss = _FusedTrainFunction.apply(self, img.float(), target.float(), target_weight.float(),
                                                 *self.parameters())
            finally:
                self._src = None
            loss = loss.reshape(-1)
            outputs = self._heats
        else:
            feat = self._backbone(img, img_metas)
            if self.with_neck:
                feat = self.neck(feat)
        skip_acc = (self.train_cfg or {}).get('skip_accuracy', False)
        for i, (head, (ln, an)) in enumerate(zip(heads, names)):
            sel = (src == i)
            t_sel = target * sel.view(-1, 1, 1, 1)
            w_sel = target_weight * sel.view(-1, 1, 1)
            if fused:
                losses[ln] = loss[i]
                out = outputs[i]
            else:
                out = head(feat)
                losses[ln] = head.get_loss(out, t_sel, w_sel)['heatmap_loss']
            if not (fused and skip_acc):
                losses[an] = head.get_accuracy(out, t_sel, w_sel)['acc_pose']
        return losses

    def forward_dummy(self, img):
        """top_down_moe.py:246-262 (the reference passes no dataset index there; dataset 0 is used)."""
        return super().forward_dummy(img)


class _FlatOptimizer:
    """step()/zero_grad() facade over the fused AdamW kernel (grad clip + layer decay inside)."""

    def __init__(self, model: TopDown):
        self.model = model

    def zero_grad(self, set_to_none: bool = True):
        for p in self.model.parameters():
            p.grad = None

    def step(self, lr: Optional[float] = None):
        # the kernel updates the master weights and their bf16 mirror in place (below torch's version
        # counters), and optimizer_step re-derives the re-laid-out head weights / combined pos-embed
        self.model._engine.optimizer_step(lr)

    def state_dict(self):
        """The fused AdamW's state + the position of the DropPath random stream (the sampling kernel is keyed by
        torch.initial_seed() and a per-model step counter: a resumed run must not replay the masks of its first steps)."""
        sd = self.model._engine.optimizer.state_dict()
        sd['drop_path_counter'] = int(self.model._engine.vit._dp_counter)
        return sd

    def load_state_dict(self, sd):
        self.model._engine.optimizer.load_state_dict(sd)
        if 'drop_path_counter' in sd:
            self.model._engine.vit._dp_counter = int(sd['drop_path_counter'])
