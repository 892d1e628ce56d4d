"""Launch plans of the top-down heatmap head on channels-last activations.

Classic decoder: each ConvTranspose2d(k=4, s=2, p=1, bias=False) is four sub-pixel
GEMMs (one per output parity), BatchNorm is a column reduction + folded scale/shift,
the final 1x1 conv a skinny GEMM (17 output channels padded to 64 columns).

Simple decoder (`upsample` > 0 and / or a 3x3 final conv): ReLU -> bilinear resize -> conv is
evaluated as a GEMM on the low-resolution grid (one column block per conv tap) followed by an
interpolate-shift-sum kernel; the up-sampled feature map is never materialised.
"""
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from .engine import ParamStore, Workspace, _slab_elems, _split_k
from .plan import Plan

FUSED_DECONV = os.environ.get('VITPOSE_FUSED_DECONV', '1') != '0'   # 0: explicit im2col + one GEMM per parity
# VITPOSE_DECONV_WGRAD_GATHER=0: materialise the 16-tap matrix for the deconv weight gradients instead of gathering it in the GEMM loader
WGRAD_GATHER = os.environ.get('VITPOSE_DECONV_WGRAD_GATHER', '1') != '0'
KPAD = 64  # heat-map channels (17 COCO ... 133 COCO-WholeBody) are padded to whole 64-wide MFMA K steps


def head_param_shapes(cfg) -> Tuple[List[Tuple[str, Tuple[int, ...]]], List[Tuple[str, Tuple[int, ...]]]]:
    """(parameters, buffers) with the reference head's state-dict keys/shapes."""
    params, bufs = [], []
    cin = cfg['in_channels']
    for li in range(cfg.get('num_deconv_layers', 0)):
        k = cfg['num_deconv_kernels'][li]
        co = cfg['num_deconv_filters'][li]
        params.append((f'deconv_layers.{3 * li}.weight', (cin, co, k, k)))
        bn = f'deconv_layers.{3 * li + 1}.'
        params += [(bn + 'weight', (co,)), (bn + 'bias', (co,))]
        bufs += [(bn + 'running_mean', (co,)), (bn + 'running_var', (co,))]
        cin = co
    extra = cfg.get('extra') or {}
    fk = extra.get('final_conv_kernel', 1)
    if fk:
        # optional [Conv2d, BN, ReLU] x num_conv_layers in front of the final conv: final_layer is an nn.Sequential then
        nconv = extra.get('num_conv_layers', 0)
        kernels = extra.get('num_conv_kernels', [1] * nconv)
        for i in range(nconv):
            params += [(f'final_layer.{3 * i}.weight', (cin, cin, kernels[i], kernels[i])),
                       (f'final_layer.{3 * i}.bias', (cin,))]
            bn = f'final_layer.{3 * i + 1}.'
            params += [(bn + 'weight', (cin,)), (bn + 'bias', (cin,))]
            bufs += [(bn + 'running_mean', (cin,)), (bn + 'running_var', (cin,))]
        fl = f'final_layer.{3 * nconv}.' if nconv else 'final_layer.'
        params += [(fl + 'weight', (cfg['out_channels'], cin, fk, fk)), (fl + 'bias', (cfg['out_channels'],))]
    return params, bufs


def bilinear_tables(n_in: int, n_out: int, scale_factor, align_corners: bool):
    """Source index pairs / weights of torch's bilinear F.interpolate along one axis
    (the reference's `resize`, mmpose/models/utils/ops.py:7-29), and the transposed lists.
    Returns (idx [n_out][2] int32, w [n_out][2] f32, tidx [n_in][L] int32, tw [n_in][L] f32)."""
    idx = torch.zeros(n_out, 2, dtype=torch.int32)
    w = torch.zeros(n_out, 2, dtype=torch.float32)
    for u in range(n_out):
        if n_out == n_in:
            src = float(u)
        elif align_corners:
            src = u * (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        else:
            sc = (1.0 / scale_factor) if scale_factor else n_in / n_out
            src = max((u + 0.5) * sc - 0.5, 0.0)
        i0 = min(int(math.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        l1 = src - i0
        idx[u, 0], idx[u, 1] = i0, i1
        w[u, 0], w[u, 1] = 1.0 - l1, l1
    lists = [[] for _ in range(n_in)]
    for u in range(n_out):
        for a in range(2):
            if w[u, a] != 0:
                lists[int(idx[u, a])].append((u, float(w[u, a])))
    L = max(1, max(len(l) for l in lists))
    tidx = torch.full((n_in, L), -1, dtype=torch.int32)
    tw = torch.zeros(n_in, L, dtype=torch.float32)
    for j, l in enumerate(lists):
        for a, (u, c) in enumerate(l):
            tidx[j, a], tw[j, a] = u, c
    return idx, w, tidx, tw


class HeadRunner:
    def __init__(self, cfg: dict, store: ParamStore, buffers: dict, grid_hw: Tuple[int, int], prefix: str = ''):
        """`buffers`: name -> f32 device tensor for the BN running statistics."""
        self.cfg, self.store, self.pre, self.buffers = cfg, store, prefix, buffers
        self.nl = cfg.get('num_deconv_layers', 0)
        self.K = cfg['out_channels']
        self.Hg, self.Wg = grid_hw
        self.cin = cfg['in_channels']
        extra = cfg.get('extra') or {}
        self.fk = extra.get('final_conv_kernel', 1)
        self.upsample = cfg.get('upsample', 0)
        # optional [Conv2d(C, C, k), BN, ReLU] layers in front of the final conv (topdown_heatmap_simple_head.py:112-138)
        self.nconv = int(extra.get('num_conv_layers', 0))
        self.ck = [int(k) for k in extra.get('num_conv_kernels', [1] * self.nconv)][:self.nconv]
        if any(k not in (1, 3) for k in self.ck) or len(self.ck) != self.nconv:
            raise NotImplementedError('conv layers of the head: kernel size 1 or 3, one per layer')
        self.fname = f'final_layer.{3 * self.nconv}.' if self.nconv else 'final_layer.'
        if self.nl:
            if any(k not in (2, 3, 4) for k in cfg['num_deconv_kernels'][:self.nl]):
                raise ValueError(f"Not supported num_kernels ({cfg['num_deconv_kernels']}).")
        if (self.nl or self.nconv) and self.upsample:
            raise NotImplementedError('deconv / conv layers behind a bilinear upsample')
        if self.fk not in (1, 3):
            raise NotImplementedError('HIP head needs a final conv (final_conv_kernel 1 or 3)')
        self.kpad = (self.K + KPAD - 1) // KPAD * KPAD
        dev = store.device
        self.simple = self.nl == 0 and self.nconv == 0 and (self.fk == 3 or self.upsample > 0)
        if self.simple:
            self._init_simple(dev)
            return
        self.chans = [self.cin] + [cfg['num_deconv_filters'][i] for i in range(self.nl)]
        # compute-layout weight copies (bf16): sub-pixel phase matrices and the 16-tap matrix
        self.wph = [torch.empty(4, self.chans[l + 1], 4 * self.chans[l], dtype=torch.bfloat16, device=dev)
                    for l in range(self.nl)]
        self.wd = [torch.empty(self.chans[l], 16 * self.chans[l + 1], dtype=torch.bfloat16, device=dev)
                   for l in range(self.nl)]
        self.dwd = [torch.zeros(self.chans[l], 16 * self.chans[l + 1], dtype=torch.float32, device=dev)
                    for l in range(self.nl)]
        # kernel sizes 3 and 2 run as the 4 x 4 kernel that computes the same layer (vp_deconv_weight_pad4): an f32
        # copy of the weight with the missing taps zero, refreshed with the other compute layouts
        self.dk = [int(k) for k in cfg['num_deconv_kernels'][:self.nl]]
        self.w4 = [torch.zeros(self.chans[l], self.chans[l + 1], 4, 4, dtype=torch.float32, device=dev)
                   if self.dk[l] != 4 else None for l in range(self.nl)]
        cl = self.chans[-1]
        if cl % 64:
            raise NotImplementedError('the head\'s conv channels must be a multiple of 64')
        ft = self.fk * self.fk                         # taps of the final conv: its GEMM reduces over (tap, channel)
        self.wf = torch.zeros(self.kpad, ft * cl, dtype=torch.bfloat16, device=dev)
        self.bf = torch.zeros(self.kpad, dtype=torch.float32, device=dev)
        self.dwf = torch.zeros(self.kpad, ft * cl, dtype=torch.float32, device=dev)
        self.dbf = torch.zeros(self.kpad, dtype=torch.float32, device=dev)
        # BatchNorm state: the deconv layers' first, then the conv layers'
        bnc = self.chans[1:] + [cl] * self.nconv
        self.bn_names = [f'deconv_layers.{3 * l + 1}.' for l in range(self.nl)] + \
                        [f'final_layer.{3 * i + 1}.' for i in range(self.nconv)]
        self.bn_scale = [torch.empty(c, dtype=torch.float32, device=dev) for c in bnc]
        self.bn_shift = [torch.empty(c, dtype=torch.float32, device=dev) for c in bnc]
        self.bn_mean = [torch.empty(c, dtype=torch.float32, device=dev) for c in bnc]
        self.bn_invstd = [torch.empty(c, dtype=torch.float32, device=dev) for c in bnc]
        self.bn_sums = [torch.zeros(2 * c, dtype=torch.float32, device=dev) for c in bnc]
        # conv layers: weight as the GEMM operand [Cout][(tap, Cin)] (bf16) and its f32 gradient
        self.wcv = [torch.zeros(cl, k * k * cl, dtype=torch.bfloat16, device=dev) for k in self.ck]
        self.dwcv = [torch.zeros(cl, k * k * cl, dtype=torch.float32, device=dev) for k in self.ck]

    def _init_simple(self, dev):
        cfg = self.cfg
        self.chans = [self.cin]
        s = self.upsample if self.upsample > 0 else 1
        self.Ho, self.Wo = int(self.Hg * s), int(self.Wg * s)
        self.taps = self.fk * self.fk
        if self.K > 64:
            raise NotImplementedError('simple decoder with more than 64 keypoints')
        self.kp = 20 if self.K <= 20 else (32 if self.K <= 32 else 64)   # per-tap columns of G
        self.ng = self.taps * self.kp                                  # valid columns of G
        self.ngp = (self.ng + 63) // 64 * 64                           # as a reduction extent
        ac = bool(cfg.get('align_corners', False))
        ty = bilinear_tables(self.Hg, self.Ho, self.upsample or None, ac)
        tx = bilinear_tables(self.Wg, self.Wo, self.upsample or None, ac)
        self.tabs = tuple(t.reshape(-1).contiguous().to(dev) for t in (ty[0], ty[1], tx[0], tx[1]))
        self.ttabs = tuple(t.contiguous().to(dev) for t in (ty[2], ty[3], tx[2], tx[3]))
        # wcat[t*kp + k][c] = W[k][c][tap t] (zero rows elsewhere); dwcat its f32 gradient
        self.wcat = torch.zeros(self.ngp, self.cin, dtype=torch.bfloat16, device=dev)
        self.dwcat = torch.zeros(self.ngp, self.cin, dtype=torch.float32, device=dev)

    def w(self, n):
        return self.store.view(self.pre + n)

    def g(self, n):
        return self.store.gview(self.pre + n)

    def buf(self, n):
        return self.buffers[self.pre + n]

    def deconv_w4(self, l):
        """The [Cin][Cout][4][4] f32 weight the compute layouts of layer l are built from."""
        return self.w4[l] if self.w4[l] is not None else self.w(f'deconv_layers.{3 * l}.weight')

    @property
    def out_grid(self):
        """(H, W) of the heat maps."""
        if self.simple:
            return self.Ho, self.Wo
        return self.Hg * 2 ** self.nl, self.Wg * 2 ** self.nl

    # -- weight re-layout (recorded once, replayed after every weight update) --
    def record_weight_prep(self, plan: Plan):
        if self.simple:
            W = self.w('final_layer.weight')  # [K][C][fk][fk] f32
            for t in range(self.taps):
                plan.strided_cast(W.view(-1)[t:], self.wcat[t * self.kp:], (1, 1, self.K, self.cin),
                                  (0, 0, self.cin * self.taps, self.taps))
            return
        for l in range(self.nl):
            if self.w4[l] is not None:
                plan.deconv_weight_pad4(self.w(f'deconv_layers.{3 * l}.weight'), self.w4[l],
                                        self.chans[l] * self.chans[l + 1], self.dk[l])
            # wph[phase][co][(ty,tx,ci)] = W[ci][co][(1-py)+2ty][(1-px)+2tx]; wd[ci][(ky,kx,co)] = W[ci][co][ky][kx]
            plan.deconv_weight_prep(self.deconv_w4(l), self.wph[l], self.wd[l], self.chans[l], self.chans[l + 1])
        cl = self.chans[-1]
        for i, k in enumerate(self.ck):
            # wcv[co][t * C + ci] = W[co][ci][tap t]
            plan.strided_cast(self.w(f'final_layer.{3 * i}.weight'), self.wcv[i], (1, cl, k * k, cl), (0, cl * k * k, 1, k * k))
        Wf = self.w(self.fname + 'weight')  # [K][C][fk][fk] -> wf[k][t * C + c]
        ft = self.fk * self.fk
        plan.strided_cast(Wf, self.wf, (1, self.K, ft, cl), (0, cl * ft, 1, ft))
        plan.strided_cast(self.w(self.fname + 'bias'), self.bf, (1, 1, 1, self.K), (0, 0, 0, 1))

    # -- fp32-accurate evaluation: pair arrays of the compute-layout weights + the forward pass --
    def record_weight_prep_exact(self, plan: Plan):
        """hi / lo pair arrays of the re-laid-out weights of record_weight_prep, built from the f32 masters."""
        dev = self.store.device
        f32 = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        if self.simple:
            self._wcat32 = f32(self.ngp, self.cin)
            self.wcat_pair = torch.zeros(self.ngp, 2 * self.cin, dtype=torch.bfloat16, device=dev)
            W = self.w('final_layer.weight')
            for t in range(self.taps):
                plan.strided_cast(W.view(-1)[t:], self._wcat32[t * self.kp:], (1, 1, self.K, self.cin),
                                  (0, 0, self.cin * self.taps, self.taps))
            plan.split_pair(self._wcat32, self.wcat_pair, self._wcat32.numel())
            return
        if self.nconv or self.fk != 1:
            raise NotImplementedError('fp32-accurate evaluation of a head with conv layers or a 3 x 3 conv behind deconvs')
        self._wph32 = [f32(4, self.chans[l + 1], 4 * self.chans[l]) for l in range(self.nl)]
        self.wph_pair = [torch.zeros(4, self.chans[l + 1], 8 * self.chans[l], dtype=torch.bfloat16, device=dev)
                         for l in range(self.nl)]
        for l in range(self.nl):
            ci, co = self.chans[l], self.chans[l + 1]
            if self.w4[l] is not None:
                plan.deconv_weight_pad4(self.w(f'deconv_layers.{3 * l}.weight'), self.w4[l], ci * co, self.dk[l])
            W = self.deconv_w4(l)
            for py in range(2):
                for px in range(2):
                    # wph[phase][co][(ty, tx, ci)] = W[ci][co][(1 - py) + 2 ty][(1 - px) + 2 tx]
                    src = W.view(-1)[(1 - py) * 4 + (1 - px):]
                    plan.strided_cast(src, self._wph32[l][py * 2 + px], (co, 2, 2, ci), (16, 8, 2, co * 16))
            plan.split_pair(self._wph32[l], self.wph_pair[l], self._wph32[l].numel())
        cl = self.chans[-1]
        self._wf32 = f32(self.kpad, cl)
        self.wf_pair = torch.zeros(self.kpad, 2 * cl, dtype=torch.bfloat16, device=dev)
        Wf = self.w('final_layer.weight')
        plan.strided_cast(Wf, self._wf32, (1, 1, 1, Wf.numel()), (0, 0, 0, 1))
        plan.split_pair(self._wf32, self.wf_pair, self._wf32.numel())

    def record_forward_exact(self, plan: Plan, ws: Workspace, feat_pair: torch.Tensor, B: int, heat_nchw: torch.Tensor):
        """feat_pair: hi / lo pair array [B*Hg*Wg][2 Cin] of the feature map (of its ReLU for the simple decoder
        with `upsample`); eval-mode BatchNorm; writes heat_nchw f32."""
        H, W = self.Hg, self.Wg
        M0 = B * H * W
        if self.simple:
            G = ws.f32('xh_G', M0, self.ng)
            plan.gemm(feat_pair, self.wcat_pair, G, M0, self.ng, self.cin, ldx=2 * self.cin, ldy=2 * self.cin,
                      ldo=self.ng, epilogue=_lib.VP_EPI_F32, split3=True)
            plan.upconv_fwd(G, self.ng, self.w('final_layer.bias'), heat_nchw, self.tabs, B, self.K, self.kp,
                            self.Hg, self.Wg, self.Ho, self.Wo, self.fk)
            return
        a = feat_pair
        for l in range(self.nl):
            ci, co = self.chans[l], self.chans[l + 1]
            Mout = B * 4 * H * W
            z = ws.bf16(f'xh_z{l}_pair', Mout, 2 * co)
            bn = f'deconv_layers.{3 * l + 1}.'
            plan.bn_finalize(None, Mout, co, self.w(bn + 'weight'), self.w(bn + 'bias'),
                             self.buf(bn + 'running_mean'), self.buf(bn + 'running_var'), 0.1, 1e-5,
                             self.bn_scale[l], self.bn_shift[l], None, None)
            plan.deconv_fwd_split(a, self.wph_pair[l], z, B, H, W, ci, co, self.bn_scale[l], self.bn_shift[l])
            a = z
            H, W = 2 * H, 2 * W
        Mh = B * H * W
        cl = self.chans[-1]
        heat = ws.f32('xh_heat_cl', Mh, self.kpad)
        plan.gemm(a, self.wf_pair, heat, Mh, self.kpad, cl, ldx=2 * cl, ldy=2 * cl, ldo=self.kpad,
                  epilogue=_lib.VP_EPI_F32, bias=self.bf, split3=True)
        plan.heatmap_loss(heat, self.kpad, heat_nchw, None, None, None, None, B, self.K, H * W, 1.0)

    @property
    def exact_relu_input(self):
        """The fp32-accurate plan feeds this head the ReLU of the feature map (simple decoder with `upsample`)."""
        return bool(self.simple and self.upsample > 0)

    # -- forward --
    def record_forward(self, plan: Plan, ws: Workspace, feat: torch.Tensor, B: int, training: bool,
                       heat_nchw: torch.Tensor, target=None, tw=None, loss=None, want_grad=False,
                       loss_weight: float = 1.0, running_stats: bool = False):
        """feat: bf16 channels-last [B*Hg*Wg][Cin].  Writes heat_nchw [B][K][4Hg*4Wg...] f32.
        training: keep what the backward pass reads (pre-BatchNorm maps, activations).  running_stats (with training):
        BatchNorm normalises with its running statistics and leaves them alone -- the module is in eval mode but a
        gradient is wanted."""
        if self.simple:
            return self._record_forward_simple(plan, ws, feat, B, heat_nchw, target, tw, loss, want_grad, loss_weight)
        H, W = self.Hg, self.Wg
        a = feat
        sfx = (lambda l: f'.{l}') if training else (lambda l: '')
        # the accumulators of this pass (BatchNorm sums, the loss scalar) start from zero: one launch for all
        plan.zero_small(([s_ for s_ in self.bn_sums] if training and not running_stats else []) +
                        ([loss] if loss is not None else []))
        for l in range(self.nl):
            ci, co = self.chans[l], self.chans[l + 1]
            Min, Mout = B * H * W, B * 4 * H * W
            z = ws.bf16('z' + sfx(l) + ('' if training else f'_{l}'), Mout, co)
            bn = f'deconv_layers.{3 * l + 1}.'
            if not training:
                plan.bn_finalize(None, Mout, co, self.w(bn + 'weight'), self.w(bn + 'bias'),
                                 self.buf(bn + 'running_mean'), self.buf(bn + 'running_var'), 0.1, 1e-5,
                                 self.bn_scale[l], self.bn_shift[l], None, None)
            if ci % 64 == 0 and FUSED_DECONV:
                # the four output parities in one launch; the sub-pixel im2col is gathered inside the GEMM
                plan.deconv_fwd(a, self.wph[l], z, B, H, W, ci, co,
                                epilogue=_lib.VP_EPI_BF16 if training else _lib.VP_EPI_SCALE_RELU,
                                colscale=None if training else self.bn_scale[l],
                                shift=None if training else self.bn_shift[l])
            else:
                col = ws.bf16('dc_col', Min * 4 * ci)
                for py in range(2):
                    for px in range(2):
                        plan.deconv_phase_im2col(a, col, B, H, W, ci, py, px)
                        plan.gemm(col, self.wph[l][py * 2 + px], z, Min, co, 4 * ci, ldx=4 * ci, ldy=4 * ci, ldo=co,
                                  epilogue=_lib.VP_EPI_BF16 if training else _lib.VP_EPI_SCALE_RELU,
                                  colscale=None if training else self.bn_scale[l],
                                  bias=None if training else self.bn_shift[l],
                                  row_map=(W, 4 * W, 2, py * 2 * W + px))
            if training:
                if not running_stats:
                    plan.bn_stats(z, self.bn_sums[l], Mout, co)
                plan.bn_finalize(None if running_stats else self.bn_sums[l], Mout, co, self.w(bn + 'weight'), self.w(bn + 'bias'),
                                 self.buf(bn + 'running_mean'), self.buf(bn + 'running_var'), 0.1, 1e-5,
                                 self.bn_scale[l], self.bn_shift[l], self.bn_mean[l], self.bn_invstd[l])
                act = ws.bf16(f'a.{l}', Mout, co)
                plan.bn_apply_relu(z, act, self.bn_scale[l], self.bn_shift[l], Mout, co)
                a = act
            else:
                a = z
            H, W = 2 * H, 2 * W
        Mh = B * H * W
        cl = self.chans[-1]
        for i, k in enumerate(self.ck):
            # Conv2d(C, C, k, padding k // 2) + BatchNorm + ReLU on the output grid: a GEMM over (tap, channel)
            j = self.nl + i
            bn = self.bn_names[j]
            x, kd = a, cl
            if k == 3:
                x, kd = ws.bf16('cv_col', Mh, 9 * cl), 9 * cl
                plan.im2col3x3(a, x, B, H, W, cl)
            z = ws.bf16(f'cz.{i}' if training else f'cz_{i}', Mh, cl)
            plan.gemm(x, self.wcv[i], z, Mh, cl, kd, ldx=kd, ldy=kd, ldo=cl, bias=self.w(f'final_layer.{3 * i}.bias'))
            if training and not running_stats:
                plan.bn_stats(z, self.bn_sums[j], Mh, cl)
            plan.bn_finalize(self.bn_sums[j] if training and not running_stats else None, Mh, cl,
                             self.w(bn + 'weight'), self.w(bn + 'bias'),
                             self.buf(bn + 'running_mean'), self.buf(bn + 'running_var'), 0.1, 1e-5,
                             self.bn_scale[j], self.bn_shift[j],
                             self.bn_mean[j] if training else None, self.bn_invstd[j] if training else None)
            act = ws.bf16(f'ca.{i}', Mh, cl) if training else z
            plan.bn_apply_relu(z, act, self.bn_scale[j], self.bn_shift[j], Mh, cl)
            a = act
        heat = ws.f32('heat_cl', Mh, self.kpad)
        if self.fk == 3:
            col = ws.bf16('cv_col', Mh, 9 * cl)
            plan.im2col3x3(a, col, B, H, W, cl)
            plan.gemm(col, self.wf, heat, Mh, self.kpad, 9 * cl, ldx=9 * cl, ldy=9 * cl, ldo=self.kpad,
                      epilogue=_lib.VP_EPI_F32, bias=self.bf)
        else:
            plan.gemm(a, self.wf, heat, Mh, self.kpad, cl, ldx=cl, ldy=cl, ldo=self.kpad, epilogue=_lib.VP_EPI_F32, bias=self.bf)
        dheat = ws.bf16('dheat', Mh, self.kpad) if want_grad else None
        plan.heatmap_loss(heat, self.kpad, heat_nchw, target, tw, loss, dheat, B, self.K, H * W, loss_weight)
        self._last_act = a
        self.out_hw = (H, W)
        return dheat

    def _record_forward_simple(self, plan, ws, feat, B, heat_nchw, target, tw, loss, want_grad, loss_weight):
        M, C = B * self.Hg * self.Wg, self.cin
        if self.upsample > 0:
            r = ws.bf16('relu_feat', M, C)
            plan.relu_bf16(feat, r, M * C)
        else:
            r = feat
        G = ws.f32('upconv_G', M, self.ng)
        plan.gemm(r, self.wcat, G, M, self.ng, C, ldx=C, ldy=C, ldo=self.ng, epilogue=_lib.VP_EPI_F32)
        plan.upconv_fwd(G, self.ng, self.w('final_layer.bias'), heat_nchw, self.tabs, B, self.K, self.kp,
                        self.Hg, self.Wg, self.Ho, self.Wo, self.fk)
        self._relu_feat = r
        self.out_hw = (self.Ho, self.Wo)
        dpred = None
        if target is not None:
            dpred = ws.f32('dheat_nchw', B, self.K, self.Ho, self.Wo) if want_grad else None
            plan.memset0(loss)
            plan.mse_loss_nchw(heat_nchw, target, tw, loss, dpred, B, self.K, self.Ho * self.Wo, loss_weight)
        return dpred

    def _record_backward_simple(self, plan, ws, feat, dpred, B):
        """dpred: NCHW f32 gradient wrt the heat maps."""
        M, C = B * self.Hg * self.Wg, self.cin
        plan.memset0(self.g('final_layer.bias'))       # vp_plane_sum_f32 accumulates
        plan.plane_sum(dpred, self.g('final_layer.bias'), B, self.K, self.Ho * self.Wo)
        dG = ws.bf16('upconv_dG', M, self.ngp)
        plan.upconv_bwd(dpred, dG, self.ngp, self.ttabs, B, self.K, self.kp, self.Hg, self.Wg, self.Ho, self.Wo,
                        self.fk)
        slab = ws.f32('splitk_slab_head', _slab_elems(self.ngp * C, M))
        plan.wgrad(dG, self._relu_feat, self.dwcat, self.ngp, C, M, ldx=self.ngp, ldy=C,
                   split_k=_split_k(self.ngp, C, M), slab=slab)
        # dwcat[t * kp + k][c] -> the parameter's [k][c][tap t] (stored)
        plan.strided_cast(self.dwcat, self.g('final_layer.weight'), (self.K, C, self.taps, 1), (C, 1, self.kp * C, 0))
        g = ws.bf16('hg', M, C)
        if self.upsample > 0:
            plan.gemm(dG, self.wcat, g, M, C, self.ngp, ldx=self.ngp, ldy=C, ldo=C, y_kmajor=True,
                      epilogue=_lib.VP_EPI_DRELU, aux=feat, ldaux=C)
        else:
            plan.gemm(dG, self.wcat, g, M, C, self.ngp, ldx=self.ngp, ldy=C, ldo=C, y_kmajor=True)
        return g

    def record_backward_nchw(self, plan: Plan, ws: Workspace, feat: torch.Tensor, dheat_nchw: torch.Tensor, B: int,
                             running_stats: bool = False):
        """Backward from an NCHW f32 heat-map gradient (stand-alone head module)."""
        if self.simple:
            return self._record_backward_simple(plan, ws, feat, dheat_nchw, B)
        H, W = self.out_hw
        dheat = ws.bf16('dheat', B * H * W, self.kpad)
        plan.nchw_to_cl(dheat_nchw, dheat, B, self.K, H * W, self.kpad)
        return self.record_backward(plan, ws, feat, dheat, B, running_stats)

    # -- backward: returns the bf16 gradient wrt `feat` --
    def record_backward(self, plan: Plan, ws: Workspace, feat: torch.Tensor, dheat: torch.Tensor, B: int,
                        running_stats: bool = False):
        if self.simple:
            return self._record_backward_simple(plan, ws, feat, dheat, B)
        H, W = self.out_hw
        Mh = B * H * W
        cl = self.chans[-1]
        def act_in(i):
            """(input of conv layer i -- i == nconv: of the final conv --, whether it is the output of a ReLU)"""
            if i > 0:
                return ws.bufs[f'ca.{i - 1}'], True
            return (ws.bufs[f'a.{self.nl - 1}'], True) if self.nl else (feat, False)
        a_last, relu_last = act_in(self.nconv)
        ft = self.fk * self.fk
        slab = ws.f32('splitk_slab_head', _slab_elems(max([self.chans[l] * 16 * self.chans[l + 1] for l in range(self.nl)] +
                                                          [cl * k * k * cl for k in self.ck] + [self.kpad * ft * cl]), Mh))
        # vp_colsum_bf16 and vp_bn_bwd accumulate: their targets start from zero (one launch)
        plan.zero_small([self.dbf] + [self.g(n + k) for n in self.bn_names for k in ('weight', 'bias')] +
                        [self.g(f'final_layer.{3 * i}.bias') for i in range(self.nconv)])

        def conv_bwd(dy, ldy_, nout, wmat, dwmat, k, x_in, relu_in, tag):
            """Backward of y = conv_k(x_in) as a GEMM over (tap, channel): the weight gradient into dwmat [nout][k*k*cl]
            and the gradient wrt x_in (times the ReLU mask when x_in is a ReLU's output)."""
            kd = k * k * cl
            if k == 3:
                col = ws.bf16('cv_col', Mh, kd)
                plan.im2col3x3(x_in, col, B, H, W, cl)
            else:
                col = x_in
            plan.wgrad(dy, col, dwmat, nout, kd, Mh, ldx=ldy_, ldy=kd, split_k=_split_k(nout, kd, Mh), slab=slab)
            gin = ws.bf16('hg' + tag, Mh, cl)
            if k == 3:
                dcol = ws.bf16('cv_col', Mh, kd)           # the im2col rows are dead once the weight gradient has run
                plan.gemm(dy, wmat, dcol, Mh, kd, nout, ldx=ldy_, ldy=kd, ldo=kd, y_kmajor=True)
                plan.col2im3x3(dcol, gin, B, H, W, cl, relu_aux=x_in if relu_in else None)
            elif relu_in:
                plan.gemm(dy, wmat, gin, Mh, cl, nout, ldx=ldy_, ldy=cl, ldo=cl, y_kmajor=True,
                          epilogue=_lib.VP_EPI_DRELU, aux=x_in, ldaux=cl)
            else:
                plan.gemm(dy, wmat, gin, Mh, cl, nout, ldx=ldy_, ldy=cl, ldo=cl, y_kmajor=True)
            return gin

        # final conv
        plan.colsum(dheat, self.dbf, Mh, self.kpad, self.kpad)
        g = conv_bwd(dheat, self.kpad, self.kpad, self.wf, self.dwf, self.fk, a_last, relu_last, '')
        # the first K rows of the padded gradients are the final layer's (stored: no zeroed gradient buffer needed)
        gW, gb, K = self.g(self.fname + 'weight'), self.g(self.fname + 'bias'), self.K
        plan.strided_cast(self.dwf, gW, (K, cl, ft, 1), (ft * cl, 1, cl, 0))
        plan.strided_cast(self.dbf, gb, (1, 1, 1, K), (0, 0, 0, 1))
        for i in range(self.nconv - 1, -1, -1):
            j, k = self.nl + i, self.ck[i]
            bn = self.bn_names[j]
            dz = ws.bf16('hdz', Mh, cl)
            plan.bn_bwd(g, ws.bufs[f'cz.{i}'], self.bn_mean[j], self.bn_invstd[j], self.w(bn + 'weight'),
                        self.g(bn + 'weight'), self.g(bn + 'bias'), dz, Mh, cl, running_stats=running_stats)
            plan.colsum(dz, self.g(f'final_layer.{3 * i}.bias'), Mh, cl, cl)
            x_in, relu_in = act_in(i)
            g = conv_bwd(dz, cl, cl, self.wcv[i], self.dwcv[i], k, x_in, relu_in, '_c' if i % 2 else '_c2')
            # [co][(tap, ci)] -> the parameter's [co][ci][tap] (stored)
            plan.strided_cast(self.dwcv[i], self.g(f'final_layer.{3 * i}.weight'), (cl, cl, k * k, 1), (k * k * cl, 1, cl, 0))
        if self.nl == 0:
            return g
        for l in range(self.nl - 1, -1, -1):
            ci, co = self.chans[l], self.chans[l + 1]
            Hi, Wi = H // 2, W // 2
            Mout, Min = B * H * W, B * Hi * Wi
            bn = f'deconv_layers.{3 * l + 1}.'
            dz = ws.bf16('hdz', Mout, co)
            plan.bn_bwd(g, ws.bufs[f'z.{l}'], self.bn_mean[l], self.bn_invstd[l], self.w(bn + 'weight'),
                        self.g(bn + 'weight'), self.g(bn + 'bias'), dz, Mout, co, running_stats=running_stats)
            a_in = ws.bufs[f'a.{l - 1}'] if l > 0 else feat
            implicit = co % 128 == 0 and FUSED_DECONV and WGRAD_GATHER and Wi <= 64 and 64 // Wi + 1 <= Hi and Min % 64 == 0
            if implicit:
                # dWd [ci][16co] = a_in^T G with G, the 16-tap gather of dz, read through the loader (never stored)
                plan.wgrad(a_in, dz, self.dwd[l], ci, 16 * co, Min, ldx=ci, ldy=co,
                           split_k=_split_k(ci, 16 * co, Min), slab=slab, gather16=(Hi, Wi, co))
            else:
                G = ws.bf16('hG', Min * 16 * co)
                plan.deconv_out_gather16(dz, G, B, Hi, Wi, co)
                plan.wgrad(a_in, G, self.dwd[l], ci, 16 * co, Min, ldx=ci, ldy=16 * co,
                           split_k=_split_k(ci, 16 * co, Min), slab=slab)
            # [ci][(ky, kx)][co] -> the parameter's [ci][co][ky][kx] (stored)
            k = self.dk[l]
            if k == 4:
                plan.strided_cast(self.dwd[l], self.g(f'deconv_layers.{3 * l}.weight'), (ci, co, 16, 1), (16 * co, 1, co, 0))
            else:
                # the k x k taps of the 4 x 4 gradient (taps 0..2 for k = 3, 1..2 for k = 2); the rest multiplied zeros
                off = 1 if k == 2 else 0
                plan.strided_cast(self.dwd[l].view(-1)[(off * 4 + off) * co:], self.g(f'deconv_layers.{3 * l}.weight'),
                                  (ci, co, k, k), (16 * co, 1, 4 * co, co))
            # data gradient wrt the layer input: G Wd^T  (+ ReLU mask of the previous activation)
            gin = ws.bf16('hg_in' if l % 2 else 'hg_in2', Min, ci)
            if co % 64 == 0 and FUSED_DECONV:
                # implicit GEMM: the 16-tap rows are gathered from dz by the loader (G is only read by the
                # weight gradient above)
                plan.deconv_dgrad(dz, self.wd[l], gin, B, Hi, Wi, co, ci, relu_aux=a_in if l > 0 else None)
            elif l > 0:
                plan.gemm(G, self.wd[l], gin, Min, ci, 16 * co, ldx=16 * co, ldy=16 * co, ldo=ci,
                          epilogue=_lib.VP_EPI_DRELU, aux=a_in, ldaux=ci)
            else:
                plan.gemm(G, self.wd[l], gin, Min, ci, 16 * co, ldx=16 * co, ldy=16 * co, ldo=ci)
            g = gin
            H, W = Hi, Wi
        return g
