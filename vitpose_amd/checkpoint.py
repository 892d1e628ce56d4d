"""Checkpoint interop with the reference (mmcv_custom/checkpoint.py:312-408, :490-552).

Load-time host work (runs once, on the CPU copy of the state dict, before the weights enter the
flat HIP store): key-prefix stripping, zero-padding / resizing of a 14x14 MAE patch-embedding
kernel to the model's 16x16, bicubic resize of the position embedding to the model's patch grid.
Files are read with `torch.load(weights_only=True)` (tensors only, nothing from the file is
executed).  `save_checkpoint` writes the reference's {'meta', 'state_dict', 'optimizer'} layout.
"""
import time
from collections import OrderedDict

import torch
import torch.nn.functional as F


def _state_dict_of(checkpoint, filename):
    if not isinstance(checkpoint, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    for key in ('state_dict', 'model', 'module'):
        if key in checkpoint:
            return checkpoint[key]
    return checkpoint


def adapt_backbone_state_dict(model, state_dict, patch_padding='pad'):
    """The shape adaptations of mmcv_custom/checkpoint.py:360-394 for a ViT `model`."""
    sd = OrderedDict(state_dict)
    if 'patch_embed.proj.weight' in sd:
        w = sd['patch_embed.proj.weight']
        orig, cur = tuple(w.shape[2:]), tuple(model.patch_embed.proj.weight.shape[2:])
        if orig != cur:
            pad = cur[0] - orig[0]
            lo, hi = pad // 2, pad - pad // 2
            if 'pad' in patch_padding:
                w = F.pad(w, (lo, hi, lo, hi))
            elif 'bilinear' in patch_padding:
                w = F.interpolate(w, size=cur, mode='bilinear', align_corners=False)
            elif 'bicubic' in patch_padding:
                w = F.interpolate(w, size=cur, mode='bicubic', align_corners=False)
            sd['patch_embed.proj.weight'] = w
    if 'pos_embed' in sd:
        pe = sd['pos_embed']
        E = pe.shape[-1]
        H, W = model.patch_embed.patch_shape
        extra = model.pos_embed.shape[-2] - model.patch_embed.num_patches
        orig = int((pe.shape[-2] - extra) ** 0.5)
        tokens = pe[:, extra:].reshape(-1, orig, orig, E).permute(0, 3, 1, 2)
        tokens = F.interpolate(tokens.float(), size=(H, W), mode='bicubic', align_corners=False)
        sd['pos_embed'] = torch.cat((pe[:, :extra], tokens.permute(0, 2, 3, 1).flatten(1, 2).to(pe.dtype)), dim=1)
    return sd


def split_expert_state_dict(model, state_dict, part_features):
    """mmcv_custom/checkpoint.py:396-404: a plain-ViT state dict enters a ViTMoE model.  Every fc2 tensor of the file
    has E rows; the model's shared fc2 takes the first E - part_features of them and each of its experts a copy of
    the last part_features."""
    out = OrderedDict(state_dict)
    for key in model.state_dict():
        block, is_expert, rest = key.partition('.mlp.experts.')
        if is_expert:
            leaf = rest.split('.', 1)[1]                      # 'weight' / 'bias' behind the expert's number
            out[key] = state_dict[f'{block}.mlp.fc2.{leaf}'][-part_features:]
        elif '.mlp.fc2.' in key:
            out[key] = state_dict[key][:-part_features]
    return out


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None, patch_padding='pad',
                    part_features=None):
    """Load a local checkpoint file into `model` (a ViT backbone or a whole TopDown)."""
    checkpoint = torch.load(filename, map_location=map_location, weights_only=True)
    sd = _state_dict_of(checkpoint, filename)
    if list(sd.keys())[0].startswith('module.'):
        sd = {k[7:]: v for k, v in sd.items()}
    if sorted(sd.keys())[0].startswith('encoder'):
        sd = {k.replace('encoder.', ''): v for k, v in sd.items() if k.startswith('encoder.')}
    if hasattr(model, 'patch_embed'):
        sd = adapt_backbone_state_dict(model, sd, patch_padding)
    if part_features is not None:
        sd = split_expert_state_dict(model, sd, part_features)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    if msgs:
        msg = 'The model and loaded state dict do not match exactly\n' + '\n'.join(msgs)
        if strict:
            raise RuntimeError(msg)
        (logger.warning if logger is not None else print)(msg)
    return checkpoint


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """{'meta', 'state_dict', 'optimizer'} with CPU tensors (mmcv_custom/checkpoint.py:490-552)."""
    meta = dict(meta or {})
    meta.update(time=time.asctime())
    if hasattr(model, 'module'):
        model = model.module
    ckpt = {'meta': meta,
            'state_dict': OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items())}
    if optimizer is not None:
        if not hasattr(optimizer, 'state_dict'):
            raise TypeError(f'{type(optimizer).__name__} has no state_dict(): its state cannot be saved '
                            '(pass optimizer=None to save the weights alone)')
        ckpt['optimizer'] = optimizer.state_dict()
    torch.save(ckpt, filename)
