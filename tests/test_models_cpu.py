"""Host-side behaviour of the drop-in model classes, mirroring the reference's own unit tests for
this path (tests/test_models/test_top_down_head.py:139-233: constructor errors and layer layout)
plus the registry / config / state-dict contract.  No GPU: forward on CPU must raise."""
import os

import pytest
import torch

from vitpose_amd import models  # noqa: F401  (registers the classes)
from vitpose_amd.builder import BACKBONES, HEADS, LOSSES, POSENETS, build_posenet
from vitpose_amd.models import JointsMSELoss, TopDown, TopdownHeatmapSimpleHead, ViT

LOSS = dict(type='JointsMSELoss', use_target_weight=True)
REF_CFG_DIR = '/root/reference/configs/body/2d_kpt_sview_rgb_img/topdown_heatmap/coco'


def test_registry_is_one_object_with_the_reference_names():
    assert BACKBONES is HEADS is LOSSES is POSENETS
    for name in ('ViT', 'TopdownHeatmapSimpleHead', 'JointsMSELoss', 'TopDown'):
        assert name in BACKBONES


def test_simple_head_constructor_contract():
    with pytest.raises(TypeError):
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, extra=[], loss_keypoint=LOSS)
    with pytest.raises(TypeError):   # loss_keypoint missing
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, extra={'final_conv_kernel': 1})
    with pytest.raises(ValueError):
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, num_deconv_layers=-1, loss_keypoint=LOSS)
    TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, num_deconv_layers=0, loss_keypoint=LOSS)
    with pytest.raises(ValueError):
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, num_deconv_layers=3, num_deconv_filters=(256, 256),
                                 num_deconv_kernels=(4, 4), loss_keypoint=LOSS)
    with pytest.raises(ValueError):
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, num_deconv_layers=3,
                                 num_deconv_filters=(256, 256, 256), num_deconv_kernels=(4, 4), loss_keypoint=LOSS)
    for bad in ((3, 2, 0), (4, 4, -1)):
        with pytest.raises(ValueError):
            TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, num_deconv_layers=3,
                                     num_deconv_filters=(256, 256, 256), num_deconv_kernels=bad, loss_keypoint=LOSS)
    head = TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, extra={'final_conv_kernel': 3}, loss_keypoint=LOSS)
    head.init_weights()
    assert head.final_layer.padding == (1, 1)
    head = TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, extra={'final_conv_kernel': 1}, loss_keypoint=LOSS)
    assert head.final_layer.padding == (0, 0)
    TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, extra={'final_conv_kernel': 0}, loss_keypoint=LOSS)
    head = TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, loss_keypoint=LOSS,
                                    extra=dict(final_conv_kernel=1, num_conv_layers=1, num_conv_kernels=(1,)))
    assert len(head.final_layer) == 4
    with pytest.raises(AssertionError):
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, extra={'final_conv_kernel': 2}, loss_keypoint=LOSS)
    with pytest.raises(AssertionError):
        TopdownHeatmapSimpleHead(out_channels=3, in_channels=512, input_transform='bogus', loss_keypoint=LOSS)


def test_state_dict_keys_and_parameter_counts():
    vit = ViT(img_size=(256, 192), patch_size=16, embed_dim=384, depth=12, num_heads=12, ratio=1, mlp_ratio=4,
              qkv_bias=True, drop_path_rate=0.1)
    assert sum(p.numel() for p in vit.parameters()) == 21663744          # ViTPose-S backbone
    assert vit.pos_embed.shape == (1, 193, 384)
    assert vit.patch_embed.patch_shape == (16, 12) and vit.patch_embed.num_patches == 192
    assert tuple(vit.patch_embed.proj.padding) == (2, 2) and tuple(vit.patch_embed.proj.stride) == (16, 16)
    keys = set(vit.state_dict().keys())
    for k in ('pos_embed', 'patch_embed.proj.weight', 'patch_embed.proj.bias', 'blocks.0.norm1.weight',
              'blocks.11.attn.qkv.bias', 'blocks.3.attn.proj.weight', 'blocks.5.mlp.fc1.bias',
              'blocks.5.mlp.fc2.weight', 'last_norm.bias'):
        assert k in keys
    assert vit.get_num_layers() == 12 and vit.no_weight_decay() == {'pos_embed', 'cls_token'}
    head = TopdownHeatmapSimpleHead(in_channels=384, out_channels=17, num_deconv_layers=2,
                                    num_deconv_filters=(256, 256), num_deconv_kernels=(4, 4),
                                    extra=dict(final_conv_kernel=1), loss_keypoint=LOSS)
    assert sum(p.numel() for p in head.parameters()) == 4096 * 384 + 1053969
    hk = set(head.state_dict().keys())
    assert {'deconv_layers.0.weight', 'deconv_layers.1.running_mean', 'deconv_layers.1.num_batches_tracked',
            'deconv_layers.3.weight', 'deconv_layers.4.bias', 'final_layer.weight', 'final_layer.bias'} <= hk
    assert head.state_dict()['deconv_layers.0.weight'].shape == (384, 256, 4, 4)
    simple = TopdownHeatmapSimpleHead(in_channels=768, out_channels=17, num_deconv_layers=0, num_deconv_filters=[],
                                      num_deconv_kernels=[], upsample=4, extra=dict(final_conv_kernel=3),
                                      loss_keypoint=LOSS)
    assert sum(p.numel() for p in simple.parameters()) == 153 * 768 + 17


def test_vit_init_and_freeze_contract():
    vit = ViT(img_size=(256, 192), embed_dim=64, depth=3, num_heads=2, qkv_bias=True, frozen_stages=1)
    assert not vit.patch_embed.proj.weight.requires_grad and not vit.blocks[1].mlp.fc1.weight.requires_grad
    assert vit.blocks[2].mlp.fc1.weight.requires_grad and vit.blocks[0].mlp.fc1.weight.requires_grad
    vit.init_weights(None)
    assert float(vit.blocks[0].norm1.weight.min()) == 1.0 and float(vit.blocks[0].attn.qkv.bias.abs().max()) == 0.0
    assert 0.015 < float(vit.blocks[0].mlp.fc1.weight.std()) < 0.025
    with pytest.raises(TypeError):
        vit.init_weights(pretrained=3)
    v2 = ViT(img_size=(256, 192), embed_dim=64, depth=2, num_heads=2, freeze_ffn=True)
    assert not v2.pos_embed.requires_grad and not v2.blocks[1].norm2.weight.requires_grad
    assert v2.blocks[1].attn.qkv.weight.requires_grad
    with pytest.raises(NotImplementedError):
        ViT(hybrid_backbone=object())


def test_no_cpu_execution_path():
    vit = ViT(img_size=(256, 192), embed_dim=64, depth=1, num_heads=2)
    with pytest.raises(RuntimeError, match='no CPU path'):
        vit(torch.zeros(1, 3, 256, 192))
    head = TopdownHeatmapSimpleHead(in_channels=64, out_channels=17, num_deconv_layers=0, loss_keypoint=LOSS)
    with pytest.raises(RuntimeError, match='no CPU path'):
        head(torch.zeros(1, 64, 16, 12))
    with pytest.raises(RuntimeError, match='no CPU path'):
        JointsMSELoss()(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), torch.ones(1, 2, 1))


@pytest.mark.skipif(not os.path.isdir(REF_CFG_DIR), reason='reference configs not present on this machine')
@pytest.mark.parametrize('name,nparams', [('ViTPose_small_coco_256x192.py', 21663744 + 2626833),
                                          ('ViTPose_base_coco_256x192.py', 85794816 + 4199697),
                                          ('ViTPose_base_simple_coco_256x192.py', 85794816 + 117521)])
def test_reference_configs_build_unchanged(name, nparams):
    from vitpose_amd.config import Config
    cfg = Config.fromfile(os.path.join(REF_CFG_DIR, name))
    assert cfg.model.type == 'TopDown' and cfg.data_cfg.heatmap_size == [48, 64]
    model = build_posenet(cfg.model)
    assert isinstance(model, TopDown) and isinstance(model.backbone, ViT)
    assert sum(p.numel() for p in model.parameters()) == nparams
    assert model.keypoint_head.test_cfg['flip_test'] is True
    assert all(k.startswith(('backbone.', 'keypoint_head.')) for k in model.state_dict())


def test_checkpoint_adaptation_pad_and_pos_resize(tmp_path):
    """mmcv_custom/checkpoint.py:360-394: a 14x14 patch kernel is zero-padded to 16x16 and a 14x14(+cls)
    position table is bicubically resized to the 16x12 grid; save/load round trip."""
    from vitpose_amd.checkpoint import load_checkpoint, save_checkpoint
    vit = ViT(img_size=(256, 192), embed_dim=32, depth=1, num_heads=2, qkv_bias=True)
    sd = {k: v.clone() for k, v in vit.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    sd['patch_embed.proj.weight'] = torch.randn(32, 3, 14, 14, generator=g)
    sd['pos_embed'] = torch.randn(1, 14 * 14 + 1, 32, generator=g)
    f = str(tmp_path / 'mae.pth')
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}}, f)
    load_checkpoint(vit, f, strict=True)
    w = vit.patch_embed.proj.weight
    assert torch.equal(w[:, :, 1:15, 1:15], sd['patch_embed.proj.weight']) and float(w[:, :, 0].abs().max()) == 0
    ref = torch.nn.functional.interpolate(sd['pos_embed'][:, 1:].reshape(1, 14, 14, 32).permute(0, 3, 1, 2),
                                          size=(16, 12), mode='bicubic', align_corners=False)
    assert torch.allclose(vit.pos_embed[:, 1:], ref.permute(0, 2, 3, 1).flatten(1, 2))
    assert torch.equal(vit.pos_embed[:, :1], sd['pos_embed'][:, :1])
    f2 = str(tmp_path / 'out.pth')
    save_checkpoint(vit, f2, meta=dict(epoch=3))
    ck = torch.load(f2, weights_only=True)
    assert set(ck) == {'meta', 'state_dict'} and ck['meta']['epoch'] == 3
    assert torch.equal(ck['state_dict']['pos_embed'], vit.pos_embed.detach())


def test_layer_decay_groups_and_lr_schedule_match_reference_recipe(golden_dir):
    import json
    from vitpose_amd.optim import layer_decay_param_groups, step_lr
    m = json.load(open(os.path.join(golden_dir, 'misc.json')))
    model = build_posenet(dict(
        type='TopDown', backbone=dict(type='ViT', img_size=(256, 192), embed_dim=384, depth=12, num_heads=12,
                                      qkv_bias=True, drop_path_rate=0.1),
        keypoint_head=dict(type='TopdownHeatmapSimpleHead', in_channels=384, out_channels=17, num_deconv_layers=2,
                           num_deconv_filters=(256, 256), num_deconv_kernels=(4, 4), extra=dict(final_conv_kernel=1),
                           loss_keypoint=LOSS), train_cfg=dict(), test_cfg=dict()))
    named = [(n, tuple(p.shape)) for n, p in model.named_parameters()]
    got = layer_decay_param_groups(named, 12, 0.8, 5e-4, 0.1)
    ref = m['layer_decay_S']
    assert set(got) == set(ref)
    for n, g in got.items():
        assert ref[n] == pytest.approx([g['lr_scale'], g['lr'], g['weight_decay']], rel=1e-12), n
    # mmcv StepLrUpdaterHook + linear warm-up (configs/.../ViTPose_base_coco_256x192.py:32-38)
    assert step_lr(5e-4, 0, 0) == pytest.approx(5e-4 * 0.001)
    assert step_lr(5e-4, 250, 0) == pytest.approx(5e-4 * (1 - 0.5 * 0.999))
    assert step_lr(5e-4, 500, 0) == 5e-4
    assert step_lr(5e-4, 10 ** 5, 170) == pytest.approx(5e-5) and step_lr(5e-4, 10 ** 5, 205) == pytest.approx(5e-6)


def test_reference_config_files_build_unchanged():
    """Every model family of the reference's ViTPose config files (S/B/L/H, classic and simple decoder, COCO /
    AIC / MPII / AP-10K / WholeBody keypoint sets) goes through the config reader and the registry as it is.
    Reads the reference tree, so it only runs where that tree exists (the build container)."""
    import glob
    import os
    root = os.environ.get('VITPOSE_REFERENCE_ROOT', '/root/reference')
    if not os.path.isdir(os.path.join(root, 'configs')):
        pytest.skip('reference configs not available here')
    from vitpose_amd.builder import build_posenet
    from vitpose_amd.config import Config
    files = sorted(glob.glob(os.path.join(root, 'configs', '**', 'ViTPose_*_256x192.py'), recursive=True))
    picked = {}
    for f in files:                      # one file per (directory, size, decoder) combination
        name = os.path.basename(f)
        key = (os.path.dirname(f), name.split('_')[1], 'simple' in name)
        picked.setdefault(key, f)
    assert len(picked) >= 12
    built = 0
    order = {'small': 0, 'base': 1, 'large': 2, 'huge': 3}
    todo = sorted(picked.items(), key=lambda kv: (order.get(kv[0][1], 9), kv[1]))
    light = [f for k, f in todo if k[1] in ('small', 'base')][:12] + [f for k, f in todo if k[1] == 'large'][:1]
    for f in light:      # (every file was built once by hand; the big models only cost start-up time here)
        cfg = Config.fromfile(f)
        model_cfg = dict(cfg.model)
        if model_cfg.get('pretrained'):
            model_cfg['pretrained'] = None   # points at a checkpoint file that is not part of the tree
        m = build_posenet(model_cfg)
        hk = cfg.model['keypoint_head']
        assert m.keypoint_head.final_layer.weight.shape[0] == hk['out_channels']
        assert m.backbone.pos_embed.shape[1] == 193
        pw = cfg.optimizer.get('paramwise_cfg')
        if pw and 'num_layers' in pw:        # layer-decay AdamW configs
            assert pw['num_layers'] == cfg.model['backbone']['depth']
        built += 1
    assert built >= 10


def test_vit_moe_contract_and_reference_plus_config():
    """ViTMoE / TopDownMoE: state-dict keys and shapes of vit_moe.py:78-96, the constructor error, the routing
    tables, the expert split of a plain-ViT checkpoint (mmcv_custom/checkpoint.py:396-404), and the reference's
    ViTPose+ config file built unchanged (where the reference tree exists)."""
    import os
    import numpy as np
    from vitpose_amd.builder import build_backbone, build_posenet
    from vitpose_amd.checkpoint import split_expert_state_dict
    with pytest.raises(TypeError):
        build_backbone(dict(type='ViTMoE', img_size=(256, 192), embed_dim=64, depth=1, num_heads=2, num_expert=2))
    m = build_backbone(dict(type='ViTMoE', img_size=(256, 192), patch_size=16, embed_dim=192, depth=2, num_heads=6,
                            mlp_ratio=4, qkv_bias=True, num_expert=3, part_features=48))
    sd = m.state_dict()
    assert sd['blocks.1.mlp.fc2.weight'].shape == (144, 768) and sd['blocks.1.mlp.fc2.bias'].shape == (144,)
    assert sd['blocks.0.mlp.experts.2.weight'].shape == (48, 768) and sd['blocks.0.mlp.experts.2.bias'].shape == (48,)
    assert 'blocks.0.mlp.experts.3.weight' not in sd
    # oracle and engine agree on the parameter list
    from oracle import vitpose_oracle as O
    from vitpose_amd.engine import ParamStore, VitRunner, Workspace, vit_param_shapes
    assert vit_param_shapes(m._cfg) == O.vit_param_shapes(m._cfg)
    assert [k for k, _ in vit_param_shapes(m._cfg)] == list(sd.keys())
    # a plain ViT state dict enters the experts: shared rows first, the last part_features rows to every expert
    plain = build_backbone(dict(type='ViT', img_size=(256, 192), patch_size=16, embed_dim=192, depth=2, num_heads=6,
                                mlp_ratio=4, qkv_bias=True))
    psd = {k: torch.randn_like(v) for k, v in plain.state_dict().items()}
    new = split_expert_state_dict(m, psd, 48)
    assert torch.equal(new['blocks.1.mlp.fc2.weight'], psd['blocks.1.mlp.fc2.weight'][:144])
    assert torch.equal(new['blocks.1.mlp.experts.1.weight'], psd['blocks.1.mlp.fc2.weight'][144:])
    assert torch.equal(new['blocks.0.mlp.experts.2.bias'], psd['blocks.0.mlp.fc2.bias'][144:])
    missing, unexpected = m.load_state_dict(new, strict=True)
    # routing tables: per expert the bit mask of its images, and the offsets of the expert rows
    st = ParamStore(vit_param_shapes(m._cfg), 'cpu')
    vr = VitRunner(m._cfg, st)
    ws = Workspace('cpu')
    vr.set_routing(ws, 4, [2, 0, 2, 1])
    expert, mask = vr.route_buffers(ws, 4)
    assert expert.tolist() == [2, 0, 2, 1] and mask.tolist() == [[0b0010], [0b1000], [0b0101]]
    off = st.offsets
    assert int(vr.moe_delta[1][2]) == off['blocks.1.mlp.experts.2.weight'] - off['blocks.1.mlp.fc2.weight'] - 144 * 768
    assert int(vr.moe_bdelta[0][0]) == off['blocks.0.mlp.experts.0.bias'] - off['blocks.0.mlp.fc2.bias'] - 144
    assert all(int(d) % 8 == 0 for d in vr.moe_delta[0])
    with pytest.raises(ValueError):
        vr.set_routing(ws, 4, [0, 1, 3, 0])          # dataset index outside num_expert
    with pytest.raises(ValueError):
        vr.set_routing(ws, 4, [0, 1])
    root = os.environ.get('VITPOSE_REFERENCE_ROOT', '/root/reference')
    f = os.path.join(root, 'configs/body/2d_kpt_sview_rgb_img/topdown_heatmap/coco',
                     'vitPose+_small_coco+aic+mpii+ap10k+apt36k+wholebody_256x192_udp.py')
    if not os.path.isfile(f):
        return
    from vitpose_amd.config import Config
    cfg = Config.fromfile(f)
    det = build_posenet(dict(cfg.model))
    assert type(det).__name__ == 'TopDownMoE' and det.keypoint_heads_cnt == 6
    assert [h.final_layer.weight.shape[0] for h in det.associate_keypoint_heads] == [14, 16, 17, 17, 133]
    part = cfg.model['backbone']['part_features']
    assert det.backbone.blocks[0].mlp.experts[5].weight.shape == (part, 1536)
    assert det.backbone.blocks[0].mlp.fc2.weight.shape == (384 - part, 1536)
    names = [n for n, _ in det.named_parameters()]
    assert 'associate_keypoint_heads.4.final_layer.weight' in names
