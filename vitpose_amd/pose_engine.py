"""PoseEngine: the fused ViT backbone + heatmap head path on one GPU.

Owns the flat parameter store, records one launch plan per (batch size, mode) and
replays it.  The token-major bf16 output of the backbone's last LayerNorm is handed
to the head as its channels-last feature map directly (no NCHW round trip).
"""
import os
from typing import Dict, Optional

import torch

from . import _lib
from .engine import ParamStore, VitRunner, Workspace, vit_param_shapes
from .head_engine import HeadRunner, head_param_shapes
from .plan import Plan


class PoseEngine:
    def __init__(self, vit_cfg: dict, head_cfg: dict, device='cuda', loss_weight: float = 1.0,
                 use_target_weight: bool = True):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('PoseEngine runs on an MI355X (HIP device) only; there is no CPU path')
        _lib.lib()  # fail loudly now if the HIP library is missing
        self.vit_cfg, self.head_cfg = dict(vit_cfg), dict(head_cfg)
        self.loss_weight, self.use_target_weight = loss_weight, use_target_weight
        hp, hb = head_param_shapes(head_cfg)
        named = [('backbone.' + n, s) for n, s in vit_param_shapes(vit_cfg)] + \
                [('keypoint_head.' + n, s) for n, s in hp]
        self.store = ParamStore(named, self.device)
        self.buffers = {'keypoint_head.' + n: (torch.ones if n.endswith('var') else torch.zeros)(
            s, dtype=torch.float32, device=self.device) for n, s in hb}
        self.vit = VitRunner(vit_cfg, self.store, 'backbone.')
        self.head = HeadRunner(head_cfg, self.store, self.buffers, (self.vit.Hp, self.vit.Wp), 'keypoint_head.')
        self.K = head_cfg['out_channels']
        self.ws_train = Workspace(self.device)
        self.ws_infer = Workspace(self.device)
        self.plans = {}
        self.prep = Plan()
        self.prep.cast_f32_bf16(self.store.data, self.store.half, self.store.numel)
        self.head.record_weight_prep(self.prep)
        self.prep_derived = Plan()  # after an optimizer step the bf16 mirror is already fresh
        self.head.record_weight_prep(self.prep_derived)
        self.prep_exact = None      # lo halves of the weights, built on the first exact evaluation
        self.optimizer = None
        self.comm = None
        # VITPOSE_SIDE_STREAM=1 puts the weight-gradient launches on a second HIP stream.  Off by default:
        # once the weight gradients of several Blocks share one grouped launch that fills the chip, running
        # it beside the dependent chain only makes both wait for each other's workgroups (measured 0.6 % slower).
        self.side_stream_wgrad = os.environ.get('VITPOSE_SIDE_STREAM', '0') == '1'

    def enable_ddp(self, process_group=None, blocks_per_bucket=4):
        """Batch-sharded data parallelism: all-reduce the flat gradient buffer in buckets launched
        from inside the backward plan (call before the first training step)."""
        from .ddp import GradBuckets
        self.comm = GradBuckets(self.store, depth=self.vit.depth, blocks_per_bucket=blocks_per_bucket,
                                process_group=process_group)
        self.plans = {k: v for k, v in self.plans.items() if k[0] != 'train'}
        return self.comm

    # ---- weights ----
    def named_shapes(self):
        return [(n, self.store.shapes[n]) for n in self.store.names]

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        self.store.load(sd, strict=strict)
        for n, t in self.buffers.items():
            if n in sd:
                t.copy_(sd[n].to(self.device, torch.float32))
        self.refresh_weights()

    def state_dict(self):
        out = self.store.state_dict()
        out.update({n: t.clone() for n, t in self.buffers.items()})
        return out

    def refresh_weights(self, after_optimizer_step: bool = False):
        (self.prep_derived if after_optimizer_step else self.prep).run()
        self.vit.refresh_derived()

    # ---- inference ----
    def _infer_plan(self, B):
        key = ('infer', B)
        if key not in self.plans:
            ws = self.ws_infer
            H, W = self.vit.img_hw
            img = ws.f32('img', B, self.vit.C, H, W)
            plan = Plan()
            feat = self.vit.record_forward(plan, ws, img, B, training=False)
            oh, ow = self.head.out_grid
            heat = ws.f32('heat', B, self.K, oh, ow)
            self.head.record_forward(plan, ws, feat, B, False, heat)
            self.plans[key] = (plan, img, heat)
        return self.plans[key]

    def _exact_plan(self, B):
        """fp32-accurate evaluation (three bf16 MFMA GEMMs per Linear / conv on hi + lo operand pairs,
        everything else in f32): parity with the reference's fp32 forward at ~1e-5 relative, ~4x the cost."""
        key = ('exact', B)
        if key not in self.plans:
            if self.prep_exact is None:
                self.store.ensure_lo()
                self.prep_exact = Plan()
                self.prep_exact.split_f32(self.store.data, None, self.store.half_lo, self.store.data_lo,
                                          self.store.numel)
                self.head.record_weight_prep_exact(self.prep_exact)
            ws = self.ws_infer
            H, W = self.vit.img_hw
            img = ws.f32('img', B, self.vit.C, H, W)
            plan = Plan()
            feat = self.vit.record_forward_exact(plan, ws, img, B)
            oh, ow = self.head.out_grid
            heat = ws.f32('heat', B, self.K, oh, ow)
            self.head.record_forward_exact(plan, ws, feat, B, heat)
            self.plans[key] = (plan, img, heat)
        return self.plans[key]

    @torch.no_grad()
    def infer(self, img: torch.Tensor, copy_out: bool = True, exact: bool = False) -> torch.Tensor:
        """img [B,3,H,W] f32 on the device -> heatmaps [B,K,4Hp,4Wp] f32 (eval-mode BN).
        exact=True evaluates at fp32 accuracy (see _exact_plan)."""
        B = img.shape[0]
        plan, img_in, heat = self._exact_plan(B) if exact else self._infer_plan(B)
        if exact:
            self.prep_exact.run()
        if img.data_ptr() != img_in.data_ptr():
            img_in.copy_(img)
        plan.run()
        return heat.clone() if copy_out else heat

    # ---- training ----
    def _train_plan(self, B):
        key = ('train', B)
        if key not in self.plans:
            ws = self.ws_train
            H, W = self.vit.img_hw
            oh, ow = self.head.out_grid
            io = dict(img=ws.f32('img', B, self.vit.C, H, W), target=ws.f32('target', B, self.K, oh, ow),
                      tw=ws.f32('tw', B, self.K), heat=ws.f32('heat', B, self.K, oh, ow),
                      loss=ws.f32('loss', 1), dp=ws.f32('dp', self.vit.depth, 2, B))
            io['dp'].fill_(1.0)
            plan = Plan()
            if self.side_stream_wgrad:
                plan.enable_side_stream(self.device)
            plan.memset0(self.store.grad)
            feat = self.vit.record_forward(plan, ws, io['img'], B, training=True, dp=io['dp'])
            dheat = self.head.record_forward(plan, ws, feat, B, True, io['heat'], io['target'],
                                             io['tw'] if self.use_target_weight else None, io['loss'],
                                             want_grad=True, loss_weight=self.loss_weight)
            self.n_fwd_calls = len(plan)
            io['dheat'] = dheat
            dfeat = self.head.record_backward(plan, ws, feat, dheat, B)
            comm = self.comm
            self.vit.record_backward(plan, ws, dfeat, B, dp=io['dp'],
                                     after_block=comm.after_block if comm is not None else None,
                                     sync_blocks=tuple(comm.triggers) if comm is not None else ())
            if comm is not None:
                comm.finish(plan)
            self.plans[key] = (plan, io)
        return self.plans[key]

    def forward_backward(self, img, target, target_weight, drop_path: bool = True, dp_scales=None,
                         generator=None, timers=None):
        """One training forward + backward.  Gradients land in self.store.grad (zeroed first);
        returns (loss [1] device tensor, heatmaps [B,K,H,W] device tensor; both are plan-owned)."""
        B = img.shape[0]
        plan, io = self._train_plan(B)
        for k, t in (('img', img), ('target', target), ('tw', target_weight.reshape(B, self.K))):
            if t.data_ptr() != io[k].data_ptr():
                io[k].copy_(t)
        if dp_scales is not None:
            io['dp'].copy_(dp_scales)
        elif drop_path and max(self.vit.dp_rates) > 0:
            self.vit.sample_drop_path(io['dp'], generator)
        else:
            io['dp'].fill_(1.0)
        if timers is not None:
            plan.run_timed(timers)
        else:
            plan.run()
        return io['loss'], io['heat']

    def setup_optimizer(self, lr=5e-4, betas=(0.9, 0.999), weight_decay=0.1, layer_decay_rate=0.75,
                        num_layers=None, max_grad_norm=1.0, frozen=()):
        from .optim import FlatAdamW, layer_decay_param_groups
        groups = layer_decay_param_groups(self.named_shapes(), num_layers or self.vit.depth, layer_decay_rate,
                                          lr, weight_decay)
        self.optimizer = FlatAdamW(self.store, groups, lr, betas, 1e-8, max_grad_norm, frozen=frozen)
        return self.optimizer

    def optimizer_step(self, lr=None, grad_scale=None):
        if grad_scale is None:
            grad_scale = self.comm.grad_scale if self.comm is not None else 1.0
        self.optimizer.step(lr, grad_scale)
        self.refresh_weights(after_optimizer_step=True)
