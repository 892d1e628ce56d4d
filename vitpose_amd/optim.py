"""Optimizer-side host logic: the layer-wise lr-decay parameter groups of the
reference's LayerDecayOptimizerConstructor
(mmcv_custom/layer_decay_optimizer_constructor.py:6-78), the step / warm-up lr
schedule of the ViTPose configs, and the chunk table that drives the fused AdamW
kernel over the flat parameter buffer.
"""
from typing import Dict, List, Sequence, Tuple

import torch

CHUNK = 1 << 11  # elements per AdamW workgroup: two float4 per lane; measured 596 us per step for 90 M
# parameters (norm + update) against 637 us at 1 << 16, whose 1472 workgroups leave an uneven last round


def get_num_layer_for_vit(var_name: str, num_max_layer: int) -> int:
    """Layer id of a TopDown parameter name (layer_decay_optimizer_constructor.py:6-15)."""
    if var_name in ('backbone.cls_token', 'backbone.mask_token', 'backbone.pos_embed'):
        return 0
    if var_name.startswith('backbone.patch_embed'):
        return 0
    if var_name.startswith('backbone.blocks'):
        return int(var_name.split('.')[2]) + 1
    return num_max_layer - 1


def layer_decay_param_groups(named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], num_layers: int,
                             layer_decay_rate: float, base_lr: float, base_wd: float) -> Dict[str, dict]:
    """Per-parameter {lr_scale, lr, weight_decay, group_name}: the groups `add_params` builds
    (same file, :32-65).  `num_layers` is the config value (depth); the constructor adds 2."""
    n = num_layers + 2
    out = {}
    for name, shape in named_shapes:
        if len(shape) == 1 or name.endswith('.bias') or 'pos_embed' in name:
            gname, wd = 'no_decay', 0.
        else:
            gname, wd = 'decay', base_wd
        lid = get_num_layer_for_vit(name, n)
        scale = layer_decay_rate ** (n - lid - 1)
        out[name] = dict(group_name=f'layer_{lid}_{gname}', lr_scale=scale, lr=scale * base_lr, weight_decay=wd)
    return out


def step_lr(base_lr: float, it: int, epoch: int, steps=(170, 200), gamma=0.1, warmup_iters=500,
            warmup_ratio=0.001) -> float:
    """mmcv StepLrUpdaterHook with linear warm-up, as configured at
    configs/.../ViTPose_base_coco_256x192.py:32-38."""
    lr = base_lr * gamma ** sum(epoch >= s for s in steps)
    if it < warmup_iters:
        k = (1 - it / warmup_iters) * (1 - warmup_ratio)
        lr = lr * (1 - k)
    return lr


class FlatAdamW:
    """torch.optim.AdamW semantics over a ParamStore, one fused kernel per step."""

    def __init__(self, store, groups: Dict[str, dict], lr: float, betas=(0.9, 0.999), eps=1e-8,
                 max_grad_norm: float = 0.0, frozen: Sequence[str] = ()):
        self.store, self.lr, self.betas, self.eps = store, lr, betas, eps
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        self.frozen = tuple(frozen)
        dev = store.device
        starts, lens, scales, wds = [], [], [], []
        for name in store.names:
            if name in frozen:
                continue
            grp = groups[name]
            n = store.view(name).numel()
            n_pad = (n + 3) // 4 * 4  # padding inside the 64-element alignment gap: zeros stay zeros
            off = store.offsets[name]
            for c0 in range(0, n_pad, CHUNK):
                starts.append(off + c0)
                lens.append(min(CHUNK, n_pad - c0))
                scales.append(grp['lr_scale'])
                wds.append(grp['weight_decay'])
        self.nchunks = len(starts)
        self.chunk_start = torch.tensor(starts, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(lens, dtype=torch.int32, device=dev)
        self.chunk_lr_scale = torch.tensor(scales, dtype=torch.float32, device=dev)
        self.chunk_wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(store.data)
        self.exp_avg_sq = torch.zeros_like(store.data)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        # per-workgroup partial sums of the gradient norm: summed in a fixed order, so that every data-parallel
        # replica derives the same clip factor, bit for bit, from the same all-reduced gradients
        self.gnorm_partials = torch.zeros(2048, dtype=torch.float32, device=dev)

    def step(self, lr: float = None, grad_scale: float = 1.0):
        from . import _lib
        from ctypes import c_void_p
        L = _lib.lib()
        s = c_void_p(torch.cuda.current_stream().cuda_stream)
        st = self.store
        self.step_count += 1
        gn = None
        if self.max_grad_norm > 0:
            # the norm runs over the optimizer's own chunks: parameters it does not update (frozen ones) stay out of
            # the clip factor, as torch's clip_grad_norm_ over the parameters with requires_grad leaves them out
            rc = L.vp_sumsq_chunks_det(c_void_p(st.grad.data_ptr()), c_void_p(self.chunk_start.data_ptr()),
                                       c_void_p(self.chunk_len.data_ptr()), self.nchunks,
                                       c_void_p(self.gnorm_sq.data_ptr()), c_void_p(self.gnorm_partials.data_ptr()),
                                       self.gnorm_partials.numel(), s)
            if rc:
                raise RuntimeError(f'vp_sumsq_chunks_det failed: {rc}')
            gn = c_void_p(self.gnorm_sq.data_ptr())
        rc = L.vp_adamw_step(
            c_void_p(st.data.data_ptr()), c_void_p(st.grad.data_ptr()), c_void_p(self.exp_avg.data_ptr()),
            c_void_p(self.exp_avg_sq.data_ptr()), c_void_p(st.half.data_ptr()),
            c_void_p(self.chunk_start.data_ptr()), c_void_p(self.chunk_len.data_ptr()),
            c_void_p(self.chunk_lr_scale.data_ptr()), c_void_p(self.chunk_wd.data_ptr()), self.nchunks,
            self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps, self.step_count, gn,
            self.max_grad_norm, grad_scale, s)
        if rc:
            raise RuntimeError(f'vp_adamw_step failed: {rc}')

    # -- checkpointing (the reference saves optimizer.state_dict() next to the weights,
    #    mmcv_custom/checkpoint.py:490-552; runner.resume restores it) --
    def state_dict(self):
        """Adam moments per parameter NAME (CPU tensors) + the step count and hyper-parameters."""
        st = self.store
        names = [n for n in st.names if n not in self.frozen]
        return {'format': 'vitpose_amd.FlatAdamW/1', 'step': self.step_count,
                'hyper': dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, max_grad_norm=self.max_grad_norm),
                'state': {n: {'exp_avg': st._v(self.exp_avg, n).detach().cpu().clone(),
                              'exp_avg_sq': st._v(self.exp_avg_sq, n).detach().cpu().clone()} for n in names}}

    def load_state_dict(self, sd):
        if sd.get('format') != 'vitpose_amd.FlatAdamW/1':
            raise ValueError('not a FlatAdamW state dict (expected format "vitpose_amd.FlatAdamW/1")')
        st = self.store
        names = [n for n in st.names if n not in self.frozen]
        missing = [n for n in names if n not in sd['state']]
        unexpected = [n for n in sd['state'] if n not in names]
        if missing or unexpected:
            raise KeyError(f'optimizer state does not match the model: missing {missing[:3]}, unexpected {unexpected[:3]}')
        for n in names:
            for key, buf in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                v = sd['state'][n][key]
                if tuple(v.shape) != tuple(st.shapes[n]):
                    raise ValueError(f'{n}.{key}: shape {tuple(v.shape)} != {tuple(st.shapes[n])}')
                st._v(buf, n).copy_(v.to(st.device, torch.float32))
        self.step_count = int(sd['step'])
        h = sd.get('hyper', {})
        self.lr = h.get('lr', self.lr)
        self.betas = tuple(h.get('betas', self.betas))
        self.eps = h.get('eps', self.eps)
        self.max_grad_norm = h.get('max_grad_norm', self.max_grad_norm)
