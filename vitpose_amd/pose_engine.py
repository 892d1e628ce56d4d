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
                 use_target_weight: bool = True, assoc_heads=()):
        """assoc_heads: [(head_cfg, loss_weight, use_target_weight)] = the `associate_keypoint_head` list of
        TopDownMoE (top_down_moe.py:75-89): head j + 1 is trained on the images whose dataset index is j + 1."""
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('PoseEngine runs on an MI355X (HIP device) only; there is no CPU path')
        _lib.lib()  # fail loudly now if the HIP library is missing
        self.vit_cfg, self.head_cfg = dict(vit_cfg), dict(head_cfg)
        self.loss_weight, self.use_target_weight = loss_weight, use_target_weight
        head_specs = [('keypoint_head.', dict(head_cfg), loss_weight, use_target_weight)] + \
                     [(f'associate_keypoint_heads.{j}.', dict(c), lw, utw) for j, (c, lw, utw) in enumerate(assoc_heads)]
        named = [('backbone.' + n, s) for n, s in vit_param_shapes(vit_cfg)]
        self.buffers = {}
        for pre, cfg, _, _ in head_specs:
            hp, hb = head_param_shapes(cfg)
            named += [(pre + n, s) for n, s in hp]
            self.buffers.update({pre + n: (torch.ones if n.endswith('var') else torch.zeros)(
                s, dtype=torch.float32, device=self.device) for n, s in hb})
        self.store = ParamStore(named, self.device)
        self.vit = VitRunner(vit_cfg, self.store, 'backbone.')
        self.heads = [HeadRunner(cfg, self.store, self.buffers, (self.vit.Hp, self.vit.Wp), pre)
                      for pre, cfg, _, _ in head_specs]
        self.head_loss = [(lw, utw) for _, _, lw, utw in head_specs]
        self.head = self.heads[0]
        self.K = head_cfg['out_channels']
        self.routed = self.vit.part > 0          # ViTMoE backbone: every pass needs the images' dataset indices
        self.multi = self.routed or len(self.heads) > 1
        self.ws_train = Workspace(self.device)
        self.ws_heads = [self.ws_train] + [Workspace(self.device) for _ in self.heads[1:]]
        self.ws_infer = Workspace(self.device)
        self.plans = {}
        self.fold_layernorm = False    # evaluation default (TopDown sets it from test_cfg['fold_layernorm'])
        self.prep = Plan()
        self.prep.cast_f32_bf16(self.store.data, self.store.half, self.store.numel)
        self.prep_derived = Plan()  # after an optimizer step the bf16 mirror is already fresh
        for h in self.heads:
            h.record_weight_prep(self.prep)
            h.record_weight_prep(self.prep_derived)
        self.prep_exact = None      # pair arrays of the weights, built on the first exact evaluation
        self._weights_version, self._exact_version = 0, None
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
        self._weights_version += 1
        (self.prep_derived if after_optimizer_step else self.prep).run()
        self.vit.refresh_derived()

    # ---- inference ----
    def _infer_plan(self, B, fold=False):
        """The evaluation plan of batch size B.  fold = True (opt-in: `test_cfg['fold_layernorm']`, infer(fold_ln=True)):
        VitRunner.record_forward_folded -- the GEMMs that write the residual stream also emit its bf16 copy and row
        statistics, qkv / fc1 finish norm1 / norm2 in their epilogues, the LayerNorm launches disappear.  Measured on one
        MI355X: ViTPose-B at batch 64 +3.5 %, ViTPose-L at 32 +4.1 %; ViTPose-S at 64 -0.9 %, ViTPose-H at 128 -6.8 % (its
        LayerNorm launch already runs at 5 TB/s and the 256-wide tiles have no registers to spare in their epilogues),
        ViTPose-B at batch 8 -14 % (small grids want the schedules the producers of the fold may not choose).  Not the
        default: the plain plan is also the one whose result for an image is bit-identical at every batch size."""
        fold = bool(fold) and self.vit.ln_fold_supported()
        key = ('infer', B, True) if fold else ('infer', B)
        if key not in self.plans:
            ws = self.ws_infer
            H, W = self.vit.img_hw
            img = ws.f32('img', B, self.vit.C, H, W)
            plan = Plan()
            feat = self.vit.record_forward(plan, ws, img, B, training=False, fold_ln=fold)
            oh, ow = self.head.out_grid
            heat = ws.f32('heat', B, self.K, oh, ow)
            self.head.record_forward(plan, ws, feat, B, False, heat)
            self.plans[key] = (plan, img, heat)
        return self.plans[key]

    def _exact_plan(self, B):
        """fp32-accurate evaluation (one split3 GEMM per Linear / conv on hi + lo pair operands, everything else in
        f32): parity with the reference's fp32 forward at ~1e-5 relative."""
        key = ('exact', B)
        if key not in self.plans:
            if self.prep_exact is None:
                self.store.ensure_pair()
                self.prep_exact = Plan()
                self.prep_exact.split_pair(self.store.data, self.store.pair, self.store.numel)
                self.head.record_weight_prep_exact(self.prep_exact)
                self._exact_version = None
            ws = self.ws_infer
            H, W = self.vit.img_hw
            img = ws.f32('img', B, self.vit.C, H, W)
            plan = Plan()
            feat = self.vit.record_forward_exact(plan, ws, img, B, relu_out=self.head.exact_relu_input)
            oh, ow = self.head.out_grid
            heat = ws.f32('heat', B, self.K, oh, ow)
            self.head.record_forward_exact(plan, ws, feat, B, heat)
            self.plans[key] = (plan, img, heat)
        return self.plans[key]

    @torch.no_grad()
    def infer(self, img: torch.Tensor, copy_out: bool = True, exact: bool = False,
              dataset_source=None, fold_ln: Optional[bool] = None) -> torch.Tensor:
        """img [B,3,H,W] f32 on the device -> heatmaps [B,K,4Hp,4Wp] f32 (eval-mode BN) of the main head.
        exact=True evaluates at fp32 accuracy (see _exact_plan).  dataset_source: per-image dataset indices
        (ViTMoE backbone: selects the FFN expert, vit_moe.py:98-115; None = dataset 0)."""
        B = img.shape[0]
        if B < 1:
            raise ValueError('empty batch')
        fold = (self.fold_layernorm if fold_ln is None else fold_ln) and not exact and self.vit.ln_fold_supported()
        plan, img_in, heat = self._exact_plan(B) if exact else self._infer_plan(B, fold)
        if self.routed:
            self.vit.set_routing(self.ws_infer, B, dataset_source if dataset_source is not None else [0] * B)
        if exact and self._exact_version != self._weights_version:
            self.prep_exact.run()               # the pair arrays follow the f32 masters (refresh_weights bumps the version)
            self._exact_version = self._weights_version
        if fold:
            self.vit.ensure_ln_fold(self._weights_version)      # folded LayerNorm weights follow the f32 masters
        if img.data_ptr() != img_in.data_ptr():
            img_in.copy_(img)
        plan.run()
        return heat.clone() if copy_out else heat

    # ---- training ----
    def _train_plan(self, B, _tuned=False):
        key = ('train', B)
        if key not in self.plans:
            ws = self.ws_train
            H, W = self.vit.img_hw
            oh, ow = self.head.out_grid
            io = dict(img=ws.f32('img', B, self.vit.C, H, W), target=ws.f32('target', B, self.K, oh, ow),
                      tw=ws.f32('tw', B, self.K), dp=ws.f32('dp', self.vit.depth, 2, B))
            io['dp'].fill_(1.0)
            nh = len(self.heads)
            # per head: heat maps, loss, (masked) target weights; head 0's are also io['heat'] / ['loss'] / ['tw']
            io['heats'] = [hw.f32('heat', B, h.K, oh, ow) for h, hw in zip(self.heads, self.ws_heads)]
            io['losses'] = [hw.f32('loss', 1) for hw in self.ws_heads]
            io['tws'] = [io['tw']] + [hw.f32('tw_sel', B, self.K) for hw in self.ws_heads[1:]]
            # a loss without target weights sees the masked TARGET (top_down_moe.py:181,193), so it needs a copy
            io['targets'] = [io['target'] if (utw or not self.multi) else hw.f32('target_sel', B, self.K, oh, ow)
                             for (_, utw), hw in zip(self.head_loss, self.ws_heads)]
            io['heat'], io['loss'] = io['heats'][0], io['losses'][0]
            plan = Plan()
            if self.side_stream_wgrad:
                plan.enable_side_stream(self.device)
            # (no gradient memset: every element of store.grad is stored by exactly one kernel of the backward half)
            feat = self.vit.record_forward(plan, ws, io['img'], B, training=True, dp=io['dp'])
            dheats = []
            for j, (h, hw) in enumerate(zip(self.heads, self.ws_heads)):
                if h.K != self.K or h.out_grid != (oh, ow):
                    raise ValueError(f'head {j} emits {h.K} heat maps of {h.out_grid}; the batch carries one target of '
                                     f'{self.K} x {(oh, ow)} for every head (top_down_moe.py:181-197)')
                lw, utw = self.head_loss[j]
                dheats.append(h.record_forward(plan, hw, feat, B, True, io['heats'][j], io['targets'][j],
                                               io['tws'][j] if utw else None, io['losses'][j],
                                               want_grad=True, loss_weight=lw))
            self.n_fwd_calls = len(plan)
            io['dheats'] = dheats
            io['dheat'] = dheats[0]
            dfeats = [h.record_backward(plan, hw, feat, dh, B) for h, hw, dh in zip(self.heads, self.ws_heads, dheats)]
            dfeat = dfeats[0]
            if nh > 1:
                # every head reads the same feature map: its gradient is the sum over the heads (a small
                # elementwise add, f32 accumulation)
                dsum = ws.bf16('dfeat_sum', *dfeat.shape)
                if nh > 8:
                    raise NotImplementedError('more than 8 heads on one feature map')
                plan.sum_bf16(dfeats, dsum)
                dfeat = dsum
            comm = self.comm
            self.vit.record_backward(plan, ws, dfeat, B, dp=io['dp'],
                                     after_block=comm.after_block if comm is not None else None,
                                     sync_blocks=tuple(comm.triggers) if comm is not None else ())
            if comm is not None:
                comm.finish(plan)
            self.plans[key] = (plan, io)
            if comm is not None and not _tuned:
                # all ranks run the tile schedules rank 0 measured (see ddp.sync_tune_cache); a rank whose own picks
                # differed records the plan once more, from the shared picks
                from .ddp import sync_tune_cache
                if sync_tune_cache(self.device, comm.group):
                    del self.plans[key]
                    return self._train_plan(B, _tuned=True)
        return self.plans[key]

    def scale_dheat(self, dheat: torch.Tensor, scale: torch.Tensor):
        """dheat *= scale[0] (device scalar): the incoming dL/dloss of an autograd backward.  Heat-map gradients are
        bf16 channels-last (classic head) or f32 NCHW (simple decoder)."""
        import ctypes
        fn = _lib.lib().vp_scale_bf16 if dheat.dtype == torch.bfloat16 else _lib.lib().vp_scale_f32
        rc = fn(dheat.data_ptr(), dheat.numel(), scale.data_ptr(),
                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f'scaling the heat-map gradient failed with status {rc}')

    def load_batch(self, io, img, target, target_weight, dataset_source=None):
        """Copy one batch into the plan's input buffers; with several heads / a routed backbone also the
        per-head selections of TopDownMoE.forward_train (top_down_moe.py:177-197): head j is trained on the
        images whose dataset index is j, every other image's target weights (and targets) are zeroed."""
        B = img.shape[0]
        plan = self.plans[('train', B)][0]
        # argument slots of the three inputs, taken while each still holds its own plan buffer's address: a caller's
        # tensor may later sit at an address another input was bound to before (the caching allocator reuses freed
        # blocks), so inputs are never matched by address again
        slots = io.get('_slots')
        if slots is None:
            slots = io['_slots'] = {k: plan.slots_of(io[k]) for k in ('img', 'target', 'tw')}
            if not all(slots.values()):
                raise RuntimeError('an input of the training plan has no recorded argument slot')
        for k, t in (('img', img), ('target', target), ('tw', target_weight.reshape(B, self.K))):
            if (not self.multi and t.dtype == torch.float32 and t.is_contiguous() and t.device == io[k].device
                    and t.shape == io[k].shape and t.data_ptr() % 16 == 0):
                # the caller's tensor has the plan's layout: the forward kernels read it in place (they are the only
                # readers; stream order keeps the memory valid until they have run)
                plan.rebind(slots[k], t.data_ptr())
            else:
                plan.rebind(slots[k], io[k].data_ptr())
                io[k].copy_(t)
        if not self.multi:
            return
        src = dataset_source if dataset_source is not None else [0] * B
        if self.routed:
            self.vit.set_routing(self.ws_train, B, src)
        import ctypes
        src_t = torch.as_tensor(src, dtype=torch.int32).reshape(B).to(self.device)      # (input data: one upload)
        L, s_ = _lib.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        # head 0 masks its weights in place last: every other head reads the unmasked ones first
        for j in list(range(1, len(self.heads))) + [0]:
            if io['targets'][j] is not io['target']:
                n = io['target'][0].numel()
                rc = L.vp_mask_rows_f32(io['target'].data_ptr(), src_t.data_ptr(), j, io['targets'][j].data_ptr(), B, n, s_)
                if rc != 0:
                    raise RuntimeError(f'vp_mask_rows_f32 failed with status {rc}')
            rc = L.vp_mask_rows_f32(io['tw'].data_ptr(), src_t.data_ptr(), j, io['tws'][j].data_ptr(), B, self.K, s_)
            if rc != 0:
                raise RuntimeError(f'vp_mask_rows_f32 failed with status {rc}')
        io['_src_keep'] = src_t            # stays alive until the kernels that read it have run

    def forward_backward(self, img, target, target_weight, drop_path: bool = True, dp_scales=None,
                         timers=None, dataset_source=None):
        """One training forward + backward.  Gradients land in self.store.grad (zeroed first);
        returns (loss [1] device tensor, heatmaps [B,K,H,W] device tensor; both are plan-owned) -- with
        associate heads, the lists of both over the heads."""
        B = img.shape[0]
        if B < 1:
            raise ValueError('empty batch')
        plan, io = self._train_plan(B)
        self.load_batch(io, img, target, target_weight, dataset_source)
        if dp_scales is not None:
            io['dp'].copy_(dp_scales)
        elif drop_path and max(self.vit.dp_rates) > 0:
            self.vit.sample_drop_path(io['dp'])
        else:
            io['dp'].fill_(1.0)
        if timers is not None:
            plan.run_timed(timers)
        else:
            plan.run()
        if len(self.heads) > 1:
            return io['losses'], io['heats']
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
