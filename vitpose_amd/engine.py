"""Host-side sequencing of the HIP kernels: parameter storage, workspaces and the
recorded launch plans of the ViT backbone and the top-down heatmap head.

Data layout in HBM (one process per GPU):
  * parameters: one flat f32 buffer (master weights, 256-byte aligned tensors), a flat
    bf16 mirror (MFMA operands) and a flat f32 gradient buffer with the same offsets;
  * activations: token-major / channels-last (`[B*tokens][C]`): the f32 residual stream
    and bf16 GEMM operands; the 16x12 token grid IS the NHWC feature map of the head;
  * everything a backward pass needs is kept per block (HBM is 288 GB, the whole
    ViTPose-B activation set at 64 images is ~4 GB), nothing is recomputed except
    the attention probabilities.
"""
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .plan import Plan

ALIGN = 64  # elements; keeps every tensor 256-byte aligned in the f32 buffers
POISON_WORKSPACES = os.environ.get('VITPOSE_POISON_WORKSPACES', '0') == '1'


def _round_up(n, a):
    return (n + a - 1) // a * a


class ParamStore:
    """Flat master / bf16 / gradient buffers with named views."""

    def __init__(self, named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], device):
        self.device = torch.device(device)
        self.shapes = dict(named_shapes)
        self.names = [n for n, _ in named_shapes]
        self.offsets = {}
        off = 0
        for n, s in named_shapes:
            self.offsets[n] = off
            off += _round_up(int(math.prod(s)), ALIGN)
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.half = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.exp_avg = None
        self.exp_avg_sq = None
        # fp32-accurate evaluation: the hi / lo pair array of `data` (2 bf16 per value, groups of 32; on demand)
        self.pair = None

    def ensure_pair(self):
        if self.pair is None:
            self.pair = torch.zeros(2 * self.numel, dtype=torch.bfloat16, device=self.device)

    def _v(self, buf, name):
        o = self.offsets[name]
        s = self.shapes[name]
        return buf[o:o + int(math.prod(s))].view(s)

    def view(self, name):
        return self._v(self.data, name)

    def gview(self, name):
        return self._v(self.grad, name)

    def hview(self, name):
        return self._v(self.half, name)

    def pview(self, name):
        """[rows][2 * K] pair array of a weight [rows][K...] (K = product of the trailing extents, % 32 == 0)."""
        o = self.offsets[name]
        s = self.shapes[name]
        n = int(math.prod(s))
        if (n // s[0]) % 32:
            raise ValueError(f'{name}: rows of {n // s[0]} values have no pair layout (multiple of 32 needed)')
        return self.pair[2 * o:2 * (o + n)].view(s[0], 2 * (n // s[0]))

    def load(self, sd: Dict[str, torch.Tensor], strict=True):
        for n in self.names:
            if n in sd:
                self.view(n).copy_(sd[n].to(self.device, torch.float32))
            elif strict:
                raise KeyError(n)

    def state_dict(self):
        return {n: self.view(n).clone() for n in self.names}


class Workspace:
    """Named device buffers (allocated once per shape, owned by torch)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs = {}

    def get(self, name, shape, dtype, zero=False):
        key = name
        t = self.bufs.get(key)
        shape = tuple(int(s) for s in shape)
        if t is None or t.shape != shape or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            if POISON_WORKSPACES and not zero and t.is_floating_point():
                t.fill_(float('nan'))       # debugging aid: a kernel that reads a buffer nobody wrote shows up as NaN
            self.bufs[key] = t
        return t

    def bf16(self, name, *shape, zero=False):
        return self.get(name, shape, torch.bfloat16, zero)

    def f32(self, name, *shape, zero=False):
        return self.get(name, shape, torch.float32, zero)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


SPLITK_TARGET_BLOCKS = 512  # ~2 workgroups per CU
# Blocks whose weight gradients share one grouped launch (0: one split-K launch per weight instead)
# Blocks per grouped weight-gradient launch.  VITPOSE_WGRAD_GROUP pins it; unset, a pass with gradient buckets (N > 1)
# uses 4 -- the launches are the hand-over points of the buckets, and 4 Blocks of ViTPose-B are exactly three rounds of
# the chip -- and a pass without any (one GPU) defers EVERY Block's weight gradients to one launch at the end of the
# backward pass: 2304 tiles = nine rounds with one tail instead of three (0.500 against 0.489 of the MFMA peak).
_WGRAD_GROUP_ENV = os.environ.get('VITPOSE_WGRAD_GROUP')
WGRAD_GROUP_BLOCKS = int(_WGRAD_GROUP_ENV) if _WGRAD_GROUP_ENV else 4


def _split_k(M, N, K, target_blocks=None):
    """K slices of a weight-gradient GEMM so that tiles * slices covers the chip."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    nk = K // 64
    return max(1, min(nk, (target_blocks or SPLITK_TARGET_BLOCKS) // tiles))


def _slab_elems(out_elems, K, target_blocks=None):
    """Upper bound of split_k * out_elems over outputs of at most `out_elems` elements."""
    tb = target_blocks or SPLITK_TARGET_BLOCKS
    # split <= tb / tiles and tiles >= out_elems / 128^2  =>  split * out_elems <= tb * 128^2 (+ one slab);
    # the same bound with 256 x 256 tiles (tb workgroups of 65536 outputs) is 4x that
    return int(4 * tb * 128 * 128 + out_elems)


def _long_first(problems):
    """Order of the problems of one grouped weight-gradient launch that holds expert (image-masked) problems: the
    full-length ones first, the short ones last.  The masked kernel deals the tile list out to the XCDs round-robin,
    so every XCD works through the same mix, longest jobs first (994 -> 818 us per launch of 4 ViTPose+-B Blocks)."""
    full = [p for p in problems if len(p) <= 8 or p[8] is None]
    short = [p for p in problems if len(p) > 8 and p[8] is not None]
    return full + short


# =====================================================================================
# ViT backbone
# =====================================================================================
def vit_param_shapes(cfg) -> List[Tuple[str, Tuple[int, ...]]]:
    """State-dict keys/shapes of the backbone (identical to the reference ViT's; with `part_features`
    those of ViTMoE, vit_moe.py:78-96: fc2 keeps E - part_features outputs, every expert adds a
    Linear(hidden, part_features))."""
    E = cfg['embed_dim']
    P = cfg.get('part_features') or 0
    nexp = cfg.get('num_expert', 1) if P else 0
    Hd = int(E * cfg.get('mlp_ratio', 4))
    p = cfg['patch_size']
    H, W = cfg['img_size']
    ratio = cfg.get('ratio', 1)
    npos = (H // p) * (W // p) * ratio ** 2 + 1
    out = [('pos_embed', (1, npos, E)),
           ('patch_embed.proj.weight', (E, cfg.get('in_chans', 3), p, p)),
           ('patch_embed.proj.bias', (E,))]
    for i in range(cfg['depth']):
        b = f'blocks.{i}.'
        out += [(b + 'norm1.weight', (E,)), (b + 'norm1.bias', (E,)),
                (b + 'attn.qkv.weight', (3 * E, E))]
        if cfg.get('qkv_bias', True):
            out += [(b + 'attn.qkv.bias', (3 * E,))]
        out += [(b + 'attn.proj.weight', (E, E)), (b + 'attn.proj.bias', (E,)),
                (b + 'norm2.weight', (E,)), (b + 'norm2.bias', (E,)),
                (b + 'mlp.fc1.weight', (Hd, E)), (b + 'mlp.fc1.bias', (Hd,)),
                (b + 'mlp.fc2.weight', (E - P, Hd)), (b + 'mlp.fc2.bias', (E - P,))]
        for j in range(nexp):
            out += [(b + f'mlp.experts.{j}.weight', (P, Hd)), (b + f'mlp.experts.{j}.bias', (P,))]
    if cfg.get('last_norm', True):
        out += [('last_norm.weight', (E,)), ('last_norm.bias', (E,))]
    return out


class VitRunner:
    """Records forward / backward plans of the ViT backbone over a ParamStore."""

    def __init__(self, cfg: dict, store: ParamStore, prefix: str = ''):
        self.cfg, self.store, self.pre = cfg, store, prefix
        self.E = cfg['embed_dim']
        self.depth = cfg['depth']
        self.heads = cfg['num_heads']
        self.d = self.E // self.heads
        self.hidden = int(self.E * cfg.get('mlp_ratio', 4))
        self.patch = cfg['patch_size']
        ratio = cfg.get('ratio', 1)
        self.stride = self.patch // ratio
        self.pad = 4 + 2 * (ratio // 2 - 1)
        H, W = cfg['img_size']
        self.img_hw = (H, W)
        self.C = cfg.get('in_chans', 3)
        self.Hp = (H + 2 * self.pad - self.patch) // self.stride + 1
        self.Wp = (W + 2 * self.pad - self.patch) // self.stride + 1
        self.T = self.Hp * self.Wp
        self.scale = cfg.get('qk_scale') or self.d ** -0.5
        self.Kpe = self.C * self.patch * self.patch
        if self.T + 1 != store.shapes[prefix + 'pos_embed'][1]:
            raise ValueError('pos_embed length does not match the patch grid')
        if self.T != 192 or self.d not in (32, 64, 80):
            raise NotImplementedError(
                f'attention kernel covers 192 tokens and head_dim 32/64/80 (got {self.T}, {self.d})')
        if self.E % 64 or self.hidden % 64 or self.Kpe % 64:
            raise NotImplementedError('channel counts must be multiples of 64')
        dev = store.device
        self.pos_comb = torch.empty(self.T, self.E, dtype=torch.float32, device=dev)
        rates = torch.linspace(0, cfg.get('drop_path_rate', 0.), self.depth)
        self.dp_rates = [r.item() for r in rates]
        # per-Block rates the sampling kernel reads (a Block in eval mode -- frozen_stages -- samples nothing: 0)
        self.dp_rates_dev = torch.tensor(self.dp_rates, dtype=torch.float32, device=dev)
        self._dp_active = tuple([True] * self.depth)
        self._dp_counter = 0
        # mixture-of-experts FFN (ViTMoE): the last `part` output channels of fc2 come from the expert Linear of
        # the image's dataset.  The GEMM kernels read the expert's rows `delta` elements beyond where the shared
        # matrix would hold them (offsets inside the flat parameter buffers, identical for f32 / bf16 / gradients).
        self.part = cfg.get('part_features') or 0
        self.nexp = cfg.get('num_expert', 1) if self.part else 0
        self.moe_delta, self.moe_bdelta = [], []
        if self.part:
            if self.T != 192:
                raise NotImplementedError('expert routing needs 192 tokens per image (one GEMM row tile each)')
            Ns, off = self.E - self.part, store.offsets
            if Ns % 48 or self.part % 8:
                raise NotImplementedError('embed_dim - part_features must be a multiple of 48')
            for i in range(self.depth):
                b = f'{prefix}blocks.{i}.mlp.'
                self.moe_delta.append(torch.tensor(
                    [off[b + f'experts.{j}.weight'] - off[b + 'fc2.weight'] - Ns * self.hidden
                     for j in range(self.nexp)], dtype=torch.int64, device=dev))
                self.moe_bdelta.append(torch.tensor(
                    [off[b + f'experts.{j}.bias'] - off[b + 'fc2.bias'] - Ns for j in range(self.nexp)],
                    dtype=torch.int64, device=dev))

    def route_buffers(self, ws: 'Workspace', B: int):
        """Plan-owned routing state of one batch size: expert [B] (the images' dataset indices) and, for the
        expert weight gradients, per expert the bit mask of its images (64 images per int64 word)."""
        return (ws.get(f'moe_expert.{B}', (B,), torch.int32, zero=True),
                ws.get(f'moe_mask.{B}', (max(self.nexp, 1), (B + 63) // 64), torch.int64, zero=True))

    def set_routing(self, ws: 'Workspace', B: int, dataset_source):
        """Fill the routing buffers from the per-image dataset indices (host list / array / tensor)."""
        import numpy as np
        src = dataset_source.detach().cpu().numpy() if isinstance(dataset_source, torch.Tensor) else np.asarray(dataset_source)
        src = src.astype(np.int64).reshape(-1)
        if src.shape[0] != B:
            raise ValueError(f'{src.shape[0]} dataset indices for a batch of {B} images')
        if src.min() < 0 or src.max() >= self.nexp:
            raise ValueError(f'dataset index outside [0, {self.nexp}) (num_expert)')
        expert, mask = self.route_buffers(ws, B)
        words = np.zeros((self.nexp, (B + 63) // 64), np.uint64)
        for b, e in enumerate(src):
            words[e, b // 64] |= np.uint64(1) << np.uint64(b % 64)
        expert.copy_(torch.from_numpy(src.astype(np.int32)))
        mask.copy_(torch.from_numpy(words.view(np.int64)))

    def _route(self, ws, B, i, with_bias=True):
        if not self.part:
            return None
        expert = self.route_buffers(ws, B)[0]
        return (expert, self.moe_delta[i], self.moe_bdelta[i] if with_bias else None, self.E - self.part)

    # -- names --
    def w(self, n):
        return self.store.view(self.pre + n)

    def h(self, n):
        return self.store.hview(self.pre + n)

    def g(self, n):
        return self.store.gview(self.pre + n)

    def has(self, n):
        return (self.pre + n) in self.store.shapes

    def refresh_derived(self):
        """Weight-derived constants (after every weight update)."""
        if getattr(self, '_derived_plan', None) is None:
            self._derived_plan = Plan()
            self._derived_plan.pos_combine(self.w('pos_embed')[0], self.pos_comb, self.T, self.E)
        self._derived_plan.run()

    def sample_drop_path(self, dp: torch.Tensor, active=None):
        """Fill dp [depth][2][B] with per-sample DropPath scales floor(keep + U) / keep (one HIP launch; U from a
        counter-based generator keyed by torch's seed and a step counter).  active: per-Block flags -- a Block whose
        module is in eval mode (vit.py:249-284 puts frozen stages there) keeps scale 1."""
        import ctypes
        active = tuple(bool(a) for a in active) if active is not None else tuple([True] * self.depth)
        if active != self._dp_active:
            self.dp_rates_dev.copy_(torch.tensor([r if a else 0.0 for r, a in zip(self.dp_rates, active)]))
            self._dp_active = active
        self._dp_counter += 1
        rc = _lib.lib().vp_drop_path_scales(dp.data_ptr(), self.dp_rates_dev.data_ptr(), self.depth, dp[0].numel(),
                                            torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, self._dp_counter,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f'vp_drop_path_scales failed with status {rc}')

    # -- forward --
    def record_forward(self, plan: Plan, ws: Workspace, img: torch.Tensor, B: int, training: bool,
                       dp: Optional[torch.Tensor] = None, tokens_f32: Optional[torch.Tensor] = None,
                       fold_ln: bool = False):
        """img [B,C,H,W] f32 -> returns the final token-major bf16 feature [B*T][E].
        fold_ln (evaluation only; the caller keeps the folded weights current with ensure_ln_fold): norm1 / norm2 of
        every Block are folded into the Linear behind them (see record_forward_folded)."""
        if fold_ln and not training and self.ln_fold_supported():
            return self.record_forward_folded(plan, ws, img, B, tokens_f32)
        E, T, M, Hd = self.E, self.T, B * self.T, self.hidden
        H, W = self.img_hw
        col = ws.bf16('pe_col', M, self.Kpe)
        plan.patch_im2col(img, col, B, self.C, H, W, self.patch, self.stride, self.pad)
        # training keeps every block's tensors (suffix .i); inference reuses one set and
        # ping-pongs the residual stream between two buffers
        sfx = (lambda i: f'.{i}') if training else (lambda i: '')
        x = ws.f32('x_in.0' if training else 'x_a', M, E)
        plan.gemm(col, self.h('patch_embed.proj.weight'), x, M, E, self.Kpe, ldx=self.Kpe, ldy=self.Kpe, ldo=E,
                  epilogue=_lib.VP_EPI_RESID_F32, bias=self.w('patch_embed.proj.bias'), pos=self.pos_comb,
                  pos_rows=T)
        for i in range(self.depth):
            b = f'blocks.{i}.'
            s = sfx(i)
            xn = ws.bf16('xn1' + s, M, E)
            mean1, rstd1 = ws.f32('mean1' + s, M), ws.f32('rstd1' + s, M)
            plan.layernorm_fwd(x, self.w(b + 'norm1.weight'), self.w(b + 'norm1.bias'), xn, None,
                               mean1 if training else None, rstd1 if training else None, M, E)
            qkv = ws.bf16('qkv' + s, M, 3 * E)
            plan.gemm(xn, self.h(b + 'attn.qkv.weight'), qkv, M, 3 * E, E, ldx=E, ldy=E, ldo=3 * E,
                      bias=self.w(b + 'attn.qkv.bias') if self.has(b + 'attn.qkv.bias') else None)
            ao = ws.bf16('ao' + s, M, E)
            lse = ws.f32('lse' + s, B * self.heads, T)
            plan.attention_fwd(qkv, ao, lse if training else None, B, self.heads, self.d, T, self.scale)
            xm = ws.f32('x_mid' + s if training else 'x_b', M, E)
            use_dp = training and dp is not None and self.dp_rates[i] > 0
            plan.gemm(ao, self.h(b + 'attn.proj.weight'), xm, M, E, E, ldx=E, ldy=E, ldo=E,
                      epilogue=_lib.VP_EPI_RESID_F32, bias=self.w(b + 'attn.proj.bias'), resid=x,
                      rowscale=dp[i, 0] if use_dp else None, rows_per_scale=T)
            xn2 = ws.bf16('xn2' + s, M, E)
            mean2, rstd2 = ws.f32('mean2' + s, M), ws.f32('rstd2' + s, M)
            plan.layernorm_fwd(xm, self.w(b + 'norm2.weight'), self.w(b + 'norm2.bias'), xn2, None,
                               mean2 if training else None, rstd2 if training else None, M, E)
            hact = ws.bf16('h' + s, M, Hd)
            u = ws.bf16('u' + s, M, Hd) if training else None
            plan.gemm(xn2, self.h(b + 'mlp.fc1.weight'), hact, M, Hd, E, ldx=E, ldy=E, ldo=Hd,
                      epilogue=_lib.VP_EPI_GELU, out2=u, bias=self.w(b + 'mlp.fc1.bias'))
            xo = ws.f32(f'x_in.{i + 1}' if training else 'x_a', M, E)
            plan.gemm(hact, self.h(b + 'mlp.fc2.weight'), xo, M, E, Hd, ldx=Hd, ldy=Hd, ldo=E,
                      epilogue=_lib.VP_EPI_RESID_F32, bias=self.w(b + 'mlp.fc2.bias'), resid=xm,
                      rowscale=dp[i, 1] if use_dp else None, rows_per_scale=T, route=self._route(ws, B, i))
            x = xo
        feat = ws.bf16('feat', M, E)
        if self.cfg.get('last_norm', True):
            mean, rstd = ws.f32('mean_l', M), ws.f32('rstd_l', M)
            plan.layernorm_fwd(x, self.w('last_norm.weight'), self.w('last_norm.bias'), feat, tokens_f32,
                               mean if training else None, rstd if training else None, M, E)
        else:
            plan.cast_f32_bf16(x, feat, M * E)
            if tokens_f32 is not None:
                plan.strided_cast(x, tokens_f32, (1, 1, 1, M * E), (0, 0, 0, 1))
        self._x_last = x
        return feat

    # -- evaluation forward with the Blocks' LayerNorms folded into qkv / fc1 --
    def ln_fold_supported(self):
        """The routed FFN of ViTMoE keeps the plain path."""
        return not self.part

    def _fold_buffers(self):
        """Folded weights of every Block's qkv and fc1 (bf16 W diag(gamma)), their column sums s and shifted biases
        c = W beta + b; one prep plan that rebuilds all of them from the f32 masters."""
        if getattr(self, '_fold', None) is None:
            E, Hd, dev = self.E, self.hidden, self.store.device
            rows = 3 * E + Hd
            fw = torch.empty(self.depth, rows, E, dtype=torch.bfloat16, device=dev)
            fs = torch.empty(self.depth, rows, dtype=torch.float32, device=dev)
            fc = torch.empty(self.depth, rows, dtype=torch.float32, device=dev)
            prep = Plan()
            for i in range(self.depth):
                b = f'blocks.{i}.'
                for r0, n, lin, norm in ((0, 3 * E, 'attn.qkv', 'norm1'), (3 * E, Hd, 'mlp.fc1', 'norm2')):
                    bias = self.w(b + lin + '.bias') if self.has(b + lin + '.bias') else None
                    prep.ln_fold_weights(self.w(b + lin + '.weight'), self.w(b + norm + '.weight'), self.w(b + norm + '.bias'),
                                         bias, fw[i, r0:r0 + n], fs[i, r0:r0 + n], fc[i, r0:r0 + n], n, E)
            self._fold = (fw, fs, fc, prep)
            self._fold_version = None
        return self._fold

    def ensure_ln_fold(self, weights_version):
        """Rebuild the folded weights when the f32 masters have changed since they were built (2 x depth small launches:
        once per weight version, not per forward pass)."""
        if getattr(self, '_fold', None) is not None and self._fold_version != weights_version:
            self._fold[3].run()
            self._fold_version = weights_version

    def record_forward_folded(self, plan: Plan, ws: Workspace, img: torch.Tensor, B: int,
                              tokens_f32: Optional[torch.Tensor] = None):
        """Evaluation forward without the Blocks' LayerNorm launches (vit.py:137-140: norm1 feeds attn.qkv only,
        norm2 feeds mlp.fc1 only).  With W' = W diag(gamma):  LN(x) W^T + b = rstd (x W'^T - mean s) + c.  Every GEMM
        that writes the residual stream (patch embedding, proj, fc2) also writes its bf16 copy and per-row sums / sums
        of squares per column tile; qkv and fc1 read that copy against the folded weights and finish the
        normalisation in their epilogues.  The final LayerNorm (its consumer is the head) stays a launch."""
        E, T, M, Hd = self.E, self.T, B * self.T, self.hidden
        H, W = self.img_hw
        fw, fs, fc, _ = self._fold_buffers()
        RES = _lib.VP_EPI_RESID_F32
        col = ws.bf16('pe_col', M, self.Kpe)
        plan.patch_im2col(img, col, B, self.C, H, W, self.patch, self.stride, self.pad)
        slots_max = (E + 127) // 128
        xs = [(ws.f32('x_a', M, E), ws.bf16('xb_a', M, E), ws.f32('xst_a', M, slots_max, 2)),
              (ws.f32('x_b', M, E), ws.bf16('xb_b', M, E), ws.f32('xst_b', M, slots_max, 2))]
        x, xb, st = xs[0]
        slots = plan.gemm(col, self.h('patch_embed.proj.weight'), x, M, E, self.Kpe, ldx=self.Kpe, ldy=self.Kpe, ldo=E,
                          epilogue=RES, bias=self.w('patch_embed.proj.bias'), pos=self.pos_comb, pos_rows=T,
                          out2=xb, ln_stats_out=st)
        for i in range(self.depth):
            b = f'blocks.{i}.'
            last = i == self.depth - 1
            qkv = ws.bf16('qkv', M, 3 * E)
            plan.gemm(xb, fw[i, :3 * E], qkv, M, 3 * E, E, ldx=E, ldy=E, ldo=3 * E, colscale=fs[i, :3 * E],
                      bias=fc[i, :3 * E], ln_stats=(st, slots))
            ao = ws.bf16('ao', M, E)
            plan.attention_fwd(qkv, ao, None, B, self.heads, self.d, T, self.scale)
            xm, xmb, stm = xs[1]
            slots_m = plan.gemm(ao, self.h(b + 'attn.proj.weight'), xm, M, E, E, ldx=E, ldy=E, ldo=E, epilogue=RES,
                                bias=self.w(b + 'attn.proj.bias'), resid=x, out2=xmb, ln_stats_out=stm)
            hact = ws.bf16('h', M, Hd)
            plan.gemm(xmb, fw[i, 3 * E:], hact, M, Hd, E, ldx=E, ldy=E, ldo=Hd, epilogue=_lib.VP_EPI_GELU,
                      colscale=fs[i, 3 * E:], bias=fc[i, 3 * E:], ln_stats=(stm, slots_m))
            # (the last Block's output goes to the final LayerNorm launch: no copy, no statistics)
            slots = plan.gemm(hact, self.h(b + 'mlp.fc2.weight'), x, M, E, Hd, ldx=Hd, ldy=Hd, ldo=E, epilogue=RES,
                              bias=self.w(b + 'mlp.fc2.bias'), resid=xm, out2=None if last else xb,
                              ln_stats_out=None if last else st)
        feat = ws.bf16('feat', M, E)
        if self.cfg.get('last_norm', True):
            plan.layernorm_fwd(x, self.w('last_norm.weight'), self.w('last_norm.bias'), feat, tokens_f32, None, None, M, E)
        else:
            plan.cast_f32_bf16(x, feat, M * E)
            if tokens_f32 is not None:
                plan.strided_cast(x, tokens_f32, (1, 1, 1, M * E), (0, 0, 0, 1))
        self._x_last = x
        return feat

    # -- fp32-accurate forward (evaluation only) --
    def record_forward_exact(self, plan: Plan, ws: Workspace, img: torch.Tensor, B: int, relu_out: bool = False):
        """Same network at the reference's fp32 accuracy: every Linear is one split3 GEMM on hi / lo pair operands
        (three MFMAs per fragment pair), everything else f32, fused into the producing kernel.  Returns the
        pair array [B*T][2E] of the last LayerNorm's output (of its ReLU with `relu_out`: the simple decoder's
        input transform) -- the operand of the head's first product."""
        E, T, M, Hd = self.E, self.T, B * self.T, self.hidden
        H, W = self.img_hw
        st = self.store
        F32, RES, PAIR = _lib.VP_EPI_F32, _lib.VP_EPI_RESID_F32, _lib.VP_EPI_PAIR

        def wp(n):
            return st.pview(self.pre + n)
        col = ws.bf16('x_pe_pair', M, 2 * self.Kpe)
        plan.patch_im2col_pair(img, col, B, self.C, H, W, self.patch, self.stride, self.pad)
        x, xm = ws.f32('x_xa', M, E), ws.f32('x_xb', M, E)
        plan.gemm(col, wp('patch_embed.proj.weight'), x, M, E, self.Kpe, ldx=2 * self.Kpe, ldy=2 * self.Kpe, ldo=E,
                  epilogue=RES, bias=self.w('patch_embed.proj.bias'), pos=self.pos_comb, pos_rows=T, split3=True)
        y = ws.bf16('x_ln_pair', M, 2 * E)
        qkv = ws.f32('x_qkv', M, 3 * E)
        ao = ws.bf16('x_ao_pair', M, 2 * E)
        hh = ws.bf16('x_h_pair', M, 2 * Hd)
        route2 = None
        for i in range(self.depth):
            b = f'blocks.{i}.'
            plan.layernorm_fwd_pair(x, self.w(b + 'norm1.weight'), self.w(b + 'norm1.bias'), y, M, E)
            plan.gemm(y, wp(b + 'attn.qkv.weight'), qkv, M, 3 * E, E, ldx=2 * E, ldy=2 * E, ldo=3 * E, epilogue=F32,
                      bias=self.w(b + 'attn.qkv.bias') if self.has(b + 'attn.qkv.bias') else None, split3=True)
            plan.attention_split(qkv, ao, B, self.heads, self.d, T, self.scale)
            plan.gemm(ao, wp(b + 'attn.proj.weight'), xm, M, E, E, ldx=2 * E, ldy=2 * E, ldo=E, epilogue=RES,
                      bias=self.w(b + 'attn.proj.bias'), resid=x, split3=True)
            plan.layernorm_fwd_pair(xm, self.w(b + 'norm2.weight'), self.w(b + 'norm2.bias'), y, M, E)
            plan.gemm(y, wp(b + 'mlp.fc1.weight'), hh, M, Hd, E, ldx=2 * E, ldy=2 * E, ldo=2 * Hd, epilogue=PAIR,
                      pair_act=1, bias=self.w(b + 'mlp.fc1.bias'), split3=True)
            if self.part:
                # expert rows sit `delta` elements behind the shared matrix in the flat buffers: twice that in the pair array
                expert = self.route_buffers(ws, B)[0]
                route2 = (expert, self.moe_delta[i] * 2, self.moe_bdelta[i], self.E - self.part)
            plan.gemm(hh, wp(b + 'mlp.fc2.weight'), x, M, E, Hd, ldx=2 * Hd, ldy=2 * Hd, ldo=E, epilogue=RES,
                      bias=self.w(b + 'mlp.fc2.bias'), resid=xm, split3=True, route=route2)
        feat = ws.bf16('x_feat_pair', M, 2 * E)
        if self.cfg.get('last_norm', True):
            plan.layernorm_fwd_pair(x, self.w('last_norm.weight'), self.w('last_norm.bias'), feat, M, E, relu=relu_out)
        else:
            if relu_out:
                raise NotImplementedError('last_norm=False with a ReLU-first head in the fp32-accurate path')
            plan.split_pair(x, feat, M * E)
        return feat

    # -- backward (training workspaces of record_forward(training=True) required) --
    def record_backward(self, plan: Plan, ws: Workspace, dfeat: torch.Tensor, B: int,
                        dp: Optional[torch.Tensor] = None, after_block=None, sync_blocks=(),
                        group_blocks: Optional[int] = None):
        """dfeat = bf16 [B*T][E] gradient wrt the final feature.  Writes the gradients into store.grad
        (zeroed by the caller).  `after_block(plan, i)` is called once block i is finished and every
        gradient of blocks >= i is final for each i in `sync_blocks` (the all-reduce trigger points)."""
        E, T, M, Hd = self.E, self.T, B * self.T, self.hidden

        def dp_scale(i, j):
            if dp is not None and i >= 0 and self.dp_rates[i] > 0:
                return dp[i, j]
            return None

        dx_a, dx_b = ws.f32('dx_a', M, E), ws.f32('dx_b', M, E)
        dxn = ws.bf16('dxn', M, E)
        dao = ws.bf16('dao', M, E)
        # LayerNorm backward leaves its per-workgroup column sums (d gamma, d beta, the next bias gradient) in a
        # workspace of its own; they are added to the gradient buffer by ONE launch per group of Blocks (with the
        # grouped weight gradients) instead of one small launch per LayerNorm
        ln_pending = []
        ln_count = [0]

        def ln_bwd(dy, x, mean, rstd, gamma, dres, dx, dxb, rowscale, dgamma, dbeta, dcolsum):
            wsp = ws.f32(f'ln_bwd_partials.{ln_count[0]}', 3 * E * 512)
            ln_count[0] += 1
            plan.layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dxb, rowscale, T, None, None, None, M, E,
                               workspace=wsp)
            ln_pending.append((wsp, dgamma, dbeta, dcolsum))
        # The weight-gradient GEMMs (and the bias column sums) are off the critical path dx -> dx of the
        # backward pass: they are recorded on the plan's side stream, where they fill the CUs the
        # dependent chain leaves idle (kernel tails, HBM-bound LayerNorm / attention phases).  Their
        # inputs (dxb, du, dqkv) therefore get one buffer per use, so the main stream never has to wait
        # before overwriting; fork() makes the side stream wait for the producer.
        def dxb_buf(tag):
            return ws.bf16(f'dxb.{tag}', M, E)
        # None of a Block's four weight gradients has enough output tiles to fill 256 CUs without cutting
        # the token reduction into slices (f32 partial slabs + a reduce pass).  They are not needed before
        # the optimizer (or the block's all-reduce bucket), so they are deferred and issued as ONE grouped
        # launch per `group_blocks` Blocks: 4 Blocks of ViTPose-B are 768 tiles of 192 x 192 = 3 whole
        # rounds of the chip, no slices, no slabs.
        G = group_blocks if group_blocks is not None else WGRAD_GROUP_BLOCKS
        if group_blocks is None and not _WGRAD_GROUP_ENV and after_block is None and not self.part:
            G = self.depth
        P, Ns = self.part, self.E - self.part
        if P:
            G = max(G, 1)     # the expert weight gradients exist in the grouped launch only
            _, img_mask = self.route_buffers(ws, B)
        slab = ws.f32('splitk_slab', _slab_elems(max(E * Hd, 3 * E * E, E * self.Kpe), M)) if G <= 0 else None
        pending = []

        def defer_fc2(dxb_t, h_t, b):
            """fc2 of a mixture-of-experts FFN: the shared rows reduce over every token, expert j's rows over the
            tokens of its images; the bias gradients are the row sums of the same problems."""
            pending.append((dxb_t, h_t, self.g(b + 'mlp.fc2.weight'), Ns, Hd, E, Hd, self.g(b + 'mlp.fc2.bias')))
            xe = dxb_t.view(-1)[Ns:]
            for j in range(self.nexp):
                pending.append((xe, h_t, self.g(b + f'mlp.experts.{j}.weight'), P, Hd, E, Hd,
                                self.g(b + f'mlp.experts.{j}.bias'), img_mask[j], self.T // 64))

        def fc2_bias_grad(i):
            """Target of the LayerNorm-backward column sums of the gradient that enters Block i's fc2."""
            return None if P else self.g(f'blocks.{i}.mlp.fc2.bias')   # expert routing: row sums of the grouped launch

        def defer(x, y, name, Mw, Nw, bias=None):
            """Weight gradient x^T y (and, with `bias`, that Linear's bias gradient = column sums of x)."""
            if G <= 0:
                plan.wgrad(x, y, self.g(name), Mw, Nw, M, ldx=Mw, ldy=Nw, split_k=_split_k(Mw, Nw, M), slab=slab)
                if bias is not None:
                    plan.memset0(self.g(bias))
                    plan.colsum(x, self.g(bias), M, Mw, Mw)
            else:
                # the grouped kernel sums the columns of x on the side (one extra MFMA per fragment)
                pending.append((x, y, self.g(name), Mw, Nw, Mw, Nw, self.g(bias) if bias is not None else None))

        def flush():
            if ln_pending:
                # every target receives exactly one sum per pass: stored, not added (no zeroed gradient buffer needed)
                plan.ln_finalize(list(ln_pending), M, E, overwrite=True)
                ln_pending.clear()
            if pending:
                plan.fork()
                with plan.on_side():
                    plan.wgrad_group(_long_first(list(pending)), M)
                pending.clear()
        last = self.depth - 1
        x_last = ws.bufs[f'x_in.{self.depth}']
        dxb = dxb_buf(f'{last}.mlp')
        if self.cfg.get('last_norm', True):
            ln_bwd(dfeat, x_last, ws.bufs['mean_l'], ws.bufs['rstd_l'], self.w('last_norm.weight'),
                   None, dx_a, dxb, dp_scale(last, 1), self.g('last_norm.weight'),
                   self.g('last_norm.bias'), fc2_bias_grad(last))
        else:
            # no final LayerNorm (vit.py:257: nn.Identity): the feature gradient is the residual-stream gradient
            plan.identity_norm_bwd(dfeat, dp_scale(last, 1), T, dx_a, dxb, M, E)
            if fc2_bias_grad(last) is not None:
                plan.memset0(fc2_bias_grad(last))
                plan.colsum(dxb, fc2_bias_grad(last), M, E, E)
        dx = dx_a
        other = dx_b
        for i in range(self.depth - 1, -1, -1):
            b = f'blocks.{i}.'
            s = f'.{i}'
            g = ws.bufs
            du = ws.bf16('du' + s, M, Hd)
            dqkv = ws.bf16('dqkv' + s, M, 3 * E)
            # ---- MLP branch: x_out = x_mid + s * (fc2(gelu(fc1(LN2(x_mid)))))
            # du = (dxb W2) * gelu'(u)
            plan.gemm(dxb, self.h(b + 'mlp.fc2.weight'), du, M, Hd, E, ldx=E, ldy=Hd, ldo=Hd, y_kmajor=True,
                      epilogue=_lib.VP_EPI_DGELU, aux=g['u' + s], ldaux=Hd, route=self._route(ws, B, i, False))
            plan.fork()
            with plan.on_side():
                # dW2 [E][Hd] = dxb^T h ; dW1 [Hd][E] = du^T xn2 ; db1 = colsum(du)
                if P:
                    defer_fc2(dxb, g['h' + s], b)
                else:
                    defer(dxb, g['h' + s], b + 'mlp.fc2.weight', E, Hd)
                defer(du, g['xn2' + s], b + 'mlp.fc1.weight', Hd, E, bias=b + 'mlp.fc1.bias')
            # dxn2 = du W1
            plan.gemm(du, self.h(b + 'mlp.fc1.weight'), dxn, M, E, Hd, ldx=Hd, ldy=E, ldo=E, y_kmajor=True)
            # LN2 backward, residual add, scaled bf16 copy for the attention branch, proj bias grad
            dxb = dxb_buf(f'{i}.attn')
            ln_bwd(dxn, g['x_mid' + s], g['mean2' + s], g['rstd2' + s], self.w(b + 'norm2.weight'),
                   dx, other, dxb, dp_scale(i, 0), self.g(b + 'norm2.weight'),
                   self.g(b + 'norm2.bias'), self.g(b + 'attn.proj.bias'))
            dx, other = other, dx
            # ---- attention branch
            plan.gemm(dxb, self.h(b + 'attn.proj.weight'), dao, M, E, E, ldx=E, ldy=E, ldo=E, y_kmajor=True)
            plan.attention_bwd(g['qkv' + s], g['ao' + s], dao, g['lse' + s], dqkv, B, self.heads, self.d, T,
                               self.scale)
            plan.fork()
            with plan.on_side():
                defer(dxb, g['ao' + s], b + 'attn.proj.weight', E, E)
                defer(dqkv, g['xn1' + s], b + 'attn.qkv.weight', 3 * E, E,
                      bias=b + 'attn.qkv.bias' if self.has(b + 'attn.qkv.bias') else None)
            plan.gemm(dqkv, self.h(b + 'attn.qkv.weight'), dxn, M, E, 3 * E, ldx=3 * E, ldy=E, ldo=E, y_kmajor=True)
            # LN1 backward: bf16 copy is scaled for the previous block's MLP branch (or unscaled for
            # the patch embedding), its column sum is that branch's fc2 bias (or the patch bias) grad
            nxt_bias = fc2_bias_grad(i - 1) if i > 0 else self.g('patch_embed.proj.bias')
            dxb = dxb_buf(f'{i - 1}.mlp' if i > 0 else 'patch')
            ln_bwd(dxn, g['x_in' + s], g['mean1' + s], g['rstd1' + s], self.w(b + 'norm1.weight'),
                   dx, other, dxb, dp_scale(i - 1, 1), self.g(b + 'norm1.weight'),
                   self.g(b + 'norm1.bias'), nxt_bias)
            dx, other = other, dx
            sync = after_block is not None and i > 0 and i in sync_blocks
            if i > 0 and (sync or (G > 0 and i % G == 0)):
                flush()
            if sync:
                # every gradient of blocks >= i is final here except fc2.bias of block i-1
                plan.join()
                after_block(plan, i)
        # ---- patch embedding + position embedding
        # (on its own: 16 more tiles would push the last group of 4 Blocks, 768 tiles = 3 whole rounds of the
        # chip, into a fourth round)
        flush()
        plan.fork()
        with plan.on_side():
            pslab = ws.f32('splitk_slab_patch', 32 * E * self.Kpe)
            plan.wgrad(dxb, ws.bufs['pe_col'], self.g('patch_embed.proj.weight'), E, self.Kpe, M, ldx=E, ldy=self.Kpe,
                       split_k=_split_k(E, self.Kpe, M), slab=pslab)
        # pos_embed gradient: rows 1 .. T = sum over images of dx, row 0 (added to every token) = their sum
        plan.pos_embed_grad(dx, self.g('pos_embed')[0], B, T, E)
        plan.join()
        self._dx0 = dx
