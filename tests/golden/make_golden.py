"""Generate the golden fixtures in tests/golden/ by EXECUTING THE REFERENCE'S OWN
SOURCE FILES (vit.py, topdown_heatmap_simple_head.py, mse_loss.py,
top_down_eval.py, post_transforms.py, layer_decay_optimizer_constructor.py)
from /root/reference via tests/golden/ref_loader.py.

Run in the build container only:   python tests/golden/make_golden.py
Inputs and weights come from the seeded recipes in oracle/vitpose_oracle.py
(synth_*), so the fixtures store only (sub-sampled) expected outputs.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from oracle import vitpose_oracle as O  # noqa: E402

torch.set_num_threads(8)
ref = ref_loader.load_reference()

TOY = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=64, depth=2, num_heads=2,
           mlp_ratio=4, qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0)


def build_ref_vit(cfg, sd):
    m = ref.ViT(img_size=cfg['img_size'], patch_size=cfg['patch_size'], embed_dim=cfg['embed_dim'],
                depth=cfg['depth'], num_heads=cfg['num_heads'], ratio=cfg['ratio'],
                use_checkpoint=False, mlp_ratio=cfg['mlp_ratio'], qkv_bias=cfg['qkv_bias'],
                drop_path_rate=cfg['drop_path_rate'])
    m.init_weights(None)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m


def build_ref_head(cfg, sd):
    h = ref.TopdownHeatmapSimpleHead(
        in_channels=cfg['in_channels'], out_channels=cfg['out_channels'],
        num_deconv_layers=cfg['num_deconv_layers'], num_deconv_filters=cfg['num_deconv_filters'],
        num_deconv_kernels=cfg['num_deconv_kernels'], extra=cfg['extra'], upsample=cfg['upsample'],
        loss_keypoint=dict(type='JointsMSELoss', use_target_weight=True))
    h.init_weights()
    h.load_state_dict(sd, strict=False)  # num_batches_tracked stays at its default
    return h


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB')


def toy_vit():
    sd = O.synth_vit_weights(TOY, seed=10)
    m = build_ref_vit(TOY, sd)
    m.eval()  # the reference's ViT.train() returns None, so no chaining
    img, _, _ = O.synth_batch(2, seed=11)
    taps = {}
    blk = m.blocks[0]
    def tap(**named):
        def hook(mod, i, o):
            for k, which in named.items():
                taps[k] = i[0] if which == 'in' else o
            return None  # a non-None return would replace the module's output
        return hook
    hooks = [
        blk.norm1.register_forward_hook(tap(ln1='out')),
        blk.attn.qkv.register_forward_hook(tap(qkv='out')),
        blk.attn.proj.register_forward_hook(tap(attn_o='in', proj='out')),
        blk.mlp.fc1.register_forward_hook(tap(fc1_pre='out')),
        blk.mlp.act.register_forward_hook(tap(fc1_act='out')),
        blk.register_forward_hook(tap(block_in='in', block_out='out')),
    ]
    with torch.no_grad():
        feat = m(img)
    for h in hooks:
        h.remove()
    save('toy_vit.npz', feat=feat, **{k: v[:1] for k, v in taps.items()})  # taps: image 0 only


def vitpose_s_eval():
    vcfg = O.vit_cfg('S')
    bb = O.synth_vit_weights(vcfg, seed=20)
    m = build_ref_vit(vcfg, bb)
    m.eval()
    # classic decoder, batch 2
    hc = O.classic_head_cfg(384)
    hd = O.synth_head_weights(hc, seed=21)
    h = build_ref_head(hc, hd).eval()
    img, _, _ = O.synth_batch(2, seed=22)
    with torch.no_grad():
        feat = m(img)
        heat = h(feat)
    save('vitpose_s_classic_eval.npz', feat_sub=feat[:, ::8], heat=heat)
    # simple decoder, batch 1 (the CPU-runnable configuration of the headline benchmark)
    hs = O.simple_head_cfg(384)
    hsd = O.synth_head_weights(hs, seed=23)
    h2 = build_ref_head(hs, hsd).eval()
    with torch.no_grad():
        heat1 = h2(m(img[:1]))
    save('vitpose_s_simple_eval.npz', heat=heat1)
    # flip-test path of TopDown.forward_test (top_down.py:179-188) on the classic head
    pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    with torch.no_grad():
        out = h.inference_model(m(img), flip_pairs=None)
        outf = h.inference_model(m(img.flip(3)), flip_pairs=pairs)
    save('vitpose_s_classic_fliptest.npz', heat=(out + outf) * 0.5)


def vitpose_s_train():
    """Train-mode forward + backward: BN batch statistics, DropPath masks recorded,
    loss via head.get_loss, grads of a spread of parameters."""
    vcfg = O.vit_cfg('S', drop_path_rate=0.3)
    bb = O.synth_vit_weights(vcfg, seed=30)
    m = build_ref_vit(vcfg, bb)
    m.train()
    hc = O.classic_head_cfg(384)
    hd = O.synth_head_weights(hc, seed=31)
    h = build_ref_head(hc, hd).train()
    img, target, tw = O.synth_batch(4, seed=32)
    # record the uniform draws of every DropPath call (the stub drop_path uses torch.rand)
    draws = []
    real_rand = torch.rand

    def rec_rand(*a, **k):
        t = real_rand(*a, **k)
        draws.append(t.flatten().clone())
        return t
    torch.manual_seed(33)
    torch.rand = rec_rand
    try:
        feat = m(img)
    finally:
        torch.rand = real_rand
    heat = h(feat)
    loss = h.get_loss(heat, target, tw)['heatmap_loss']
    acc = h.get_accuracy(heat, target, tw)['acc_pose']
    loss.backward()
    # block 0 uses nn.Identity (rate 0): draws exist for blocks 1..depth-1, two per block
    assert len(draws) == 2 * (vcfg['depth'] - 1), len(draws)
    g = {}
    named = dict(m.named_parameters())
    for k in ['pos_embed', 'patch_embed.proj.bias', 'blocks.0.norm1.weight', 'blocks.0.norm1.bias',
              'blocks.0.attn.qkv.bias', 'blocks.5.attn.proj.bias', 'blocks.11.mlp.fc2.bias',
              'blocks.11.mlp.fc1.bias', 'last_norm.weight', 'last_norm.bias']:
        g['g.' + k] = named[k].grad
    for k in ['patch_embed.proj.weight', 'blocks.0.attn.qkv.weight', 'blocks.6.attn.proj.weight',
              'blocks.6.mlp.fc1.weight', 'blocks.11.mlp.fc2.weight']:
        g['gs.' + k] = named[k].grad.flatten(1)[::16, ::4]  # sub-sampled
    hn = dict(h.named_parameters())
    for k in ['final_layer.weight', 'final_layer.bias', 'deconv_layers.1.weight',
              'deconv_layers.1.bias', 'deconv_layers.4.weight', 'deconv_layers.4.bias']:
        g['gh.' + k] = hn[k].grad
    for k in ['deconv_layers.0.weight', 'deconv_layers.3.weight']:
        g['ghs.' + k] = hn[k].grad.flatten(1)[::8, ::16]
    hb = dict(h.named_buffers())
    save('vitpose_s_train.npz', heat_sub=heat[:, ::4], loss=loss.detach(), acc=np.float64(acc),
         dp_draws=torch.stack(draws),
         bn1_mean=hb['deconv_layers.1.running_mean'], bn1_var=hb['deconv_layers.1.running_var'],
         bn4_mean=hb['deconv_layers.4.running_mean'], bn4_var=hb['deconv_layers.4.running_var'],
         **g)


def block_widths():
    out = {}
    for name in ('B', 'L', 'H'):
        cfg = O.vit_cfg(name, depth=1, drop_path_rate=0.0)
        sd = O.synth_vit_weights(cfg, seed=40)
        m = build_ref_vit(cfg, sd)
        m.eval()
        g = torch.Generator().manual_seed(41)
        x = torch.randn(1, 192, cfg['embed_dim'], generator=g)
        with torch.no_grad():
            y = m.blocks[0](x)
        out[f'{name}.block_out_sub'] = y[:, ::8]
    save('block_widths.npz', **out)


def misc():
    res = {}
    # JointsMSELoss known answers (same inputs as tests/test_losses/test_top_down_losses.py:27-43)
    L = ref.JointsMSELoss(use_target_weight=True)
    fo, ft, fw = torch.zeros((1, 3, 64, 64)), torch.zeros((1, 3, 64, 64)), torch.tensor([1, 0, 1]).reshape(1, 3, 1)
    res['mse_zero'] = float(L(fo, ft, fw))
    res['mse_ones'] = float(L(torch.ones((1, 3, 64, 64)), ft, torch.tensor([1, 0, 1]).reshape(1, 3, 1).float()))
    fo = torch.zeros((1, 2, 64, 64)); fo[0, 0] += 1
    res['mse_half'] = float(L(fo, torch.zeros((1, 2, 64, 64)), torch.tensor([1, 1]).reshape(1, 2, 1).float()))
    g = torch.Generator().manual_seed(50)
    o, t, w = torch.randn(3, 17, 64, 48, generator=g), torch.rand(3, 17, 64, 48, generator=g), \
        (torch.rand(3, 17, 1, generator=g) > 0.3).float()
    res['mse_rand'] = float(L(o, t, w))
    res['mse_rand_noweight'] = float(ref.JointsMSELoss(use_target_weight=False, loss_weight=0.5)(o, t, w))
    # PCK (tests/test_evaluation/test_top_down_eval.py:11-26 shape of case + a random one)
    out = np.zeros((1, 5, 64, 64), dtype=np.float32)
    tgt = np.zeros((1, 5, 64, 64), dtype=np.float32)
    out[0, 0, 0, 0] = 1; out[0, 1, 0, 0] = 1; out[0, 2, 0, 1] = 1; out[0, 3, 0, 1] = 1; out[0, 4, 0, 0] = 1
    tgt[0, 0, 0, 0] = 1; tgt[0, 1, 0, 0] = 1; tgt[0, 2, 0, 0] = 1; tgt[0, 3, 0, 0] = 1; tgt[0, 4, 63, 63] = 1
    mask = np.array([[True, True, False, False, False]])
    acc, avg, cnt = ref.pose_pck_accuracy(out, tgt, mask, 0.5)
    res['pck_unit'] = dict(acc=[float(a) for a in acc], avg=float(avg), cnt=int(cnt))
    acc, avg, cnt = ref.pose_pck_accuracy(o.numpy(), t.numpy(), (w.numpy().squeeze(-1) > 0))
    res['pck_rand'] = dict(acc=[float(a) for a in acc], avg=float(avg), cnt=int(cnt))
    # flip_back
    pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    fb = ref.flip_back(o.numpy().copy(), pairs)
    res['flip_back_checksum'] = [float(fb[0, 1, 3, 5]), float(fb[2, 16, 63, 47]), float(fb.astype(np.float64).sum()),
                                 float((fb.astype(np.float64) * np.arange(fb.size).reshape(fb.shape)).sum())]
    # layer-decay groups: the reference's add_params driven on a TopDown-shaped module
    ld = ref_loader.load_layer_id_fn()

    class TD(nn.Module):
        def __init__(self, bb, hd):
            super().__init__()
            self.backbone, self.keypoint_head = bb, hd
    for name, rate in (('S', 0.8), ('B', 0.75)):
        vcfg = O.vit_cfg(name)
        m = ref.ViT(img_size=(256, 192), patch_size=16, embed_dim=vcfg['embed_dim'], depth=vcfg['depth'],
                    num_heads=vcfg['num_heads'], ratio=1, mlp_ratio=4, qkv_bias=True,
                    drop_path_rate=vcfg['drop_path_rate'])
        h = build_ref_head(O.classic_head_cfg(vcfg['embed_dim']), {})
        ctor = object.__new__(ld.LayerDecayOptimizerConstructor)
        ctor.paramwise_cfg = dict(num_layers=vcfg['depth'], layer_decay_rate=rate)
        ctor.base_lr, ctor.base_wd = 5e-4, 0.1
        params = []
        with contextlib.redirect_stdout(io.StringIO()):
            ctor.add_params(params, TD(m, h))
        flat = {}
        for grp in params:
            for pn in grp['param_names']:
                flat[pn] = [grp['lr_scale'], grp['lr'], grp['weight_decay']]
        res[f'layer_decay_{name}'] = flat
    with open(os.path.join(HERE, 'misc.json'), 'w') as f:
        json.dump(res, f, indent=0)
    print('misc.json written')


def synth_heatmaps(N, K, H, W, seed):
    """Smooth positive heat maps: one Gaussian bump per joint at a random sub-pixel centre + a noise floor
    (bumps near the border included).  Same recipe as oracle/decode_oracle.py: synth_heatmaps."""
    from oracle import decode_oracle as D
    return D.synth_heatmaps(N, K, H, W, seed)


def decode():
    """keypoints_from_heatmaps / _taylor / post_dark_udp / transform_preds of the reference on seeded maps.
    UDP-DARK is evaluated with modulate kernel 1 (cv2.GaussianBlur is then the identity; no cv2 here)."""
    N, K, H, W = 4, 17, 64, 48
    hm = synth_heatmaps(N, K, H, W, seed=70)
    g = np.random.default_rng(71)
    center = (g.random((N, 2)) * 200 + 50).astype(np.float32)
    scale = (g.random((N, 2)) + 0.8).astype(np.float32)
    out = {}
    p, m = ref.keypoints_from_heatmaps(hm, center, scale)
    out['default_preds'], out['default_maxvals'] = p, m
    p, m = ref.keypoints_from_heatmaps(hm, center, scale, post_process=None)
    out['none_preds'] = p
    p, m = ref.keypoints_from_heatmaps(hm, center, scale, post_process='default', use_udp=True, kernel=1,
                                       target_type='GaussianHeatmap')
    out['udp_k1_preds'], out['udp_k1_maxvals'] = p, m
    # post_dark_udp alone (heat-map pixel units), identity blur
    coords, _ = ref._get_max_preds(hm)
    out['dark_udp_k1_coords'] = ref.post_dark_udp(coords.copy(), hm.copy(), kernel=1)
    # _taylor on the log maps at the arg-max
    lg = np.log(np.maximum(hm, 1e-10))
    out['taylor_coords'] = np.stack([np.stack([ref._taylor(lg[n][k], coords[n][k].copy()) for k in range(K)])
                                     for n in range(N)])
    out['transform_udp'] = ref.transform_preds(coords[0].copy(), center[0], scale[0], [W, H], use_udp=True)
    out['transform_plain'] = ref.transform_preds(coords[0].copy(), center[0], scale[0], [W, H], use_udp=False)
    out['center'], out['scale'] = center, scale
    save('decode.npz', **out)


def pipeline():
    """get_warp_matrix / warp_affine_joints / TopDownGenerateTarget (UDP, MSRA, MSRA unbiased) of the reference
    on the seeded samples of oracle/pipeline_oracle.py: synth_samples."""
    from oracle import pipeline_oracle as P
    K, image_size, heatmap_size = 17, np.array([192, 256]), np.array([48, 64])
    samples = P.synth_samples(6, K=K, seed=80)
    jw = np.linspace(1.0, 1.5, K).astype(np.float32).reshape(K, 1)
    out = {}
    for b, r in enumerate(samples):
        trans = ref.get_warp_matrix(r['rotation'], r['center'] * 2.0, image_size - 1.0, r['scale'] * 200.0)
        j = r['joints_3d'].copy()
        j[:, 0:2] = ref.warp_affine_joints(j[:, 0:2].copy(), trans)
        out[f'trans_{b}'], out[f'joints_{b}'] = trans, j
        vis = np.stack([r['visible']] * 3, 1)
        for tag, kw, diff in (('udp', dict(sigma=2, encoding='UDP'), False), ('msra', dict(sigma=2, encoding='MSRA'), True),
                              ('msra_unb', dict(sigma=3, encoding='MSRA', unbiased_encoding=True), False)):
            gen = ref.TopDownGenerateTarget(**kw)
            res = gen(dict(joints_3d=j.copy(), joints_3d_visible=vis.copy(),
                           ann_info=dict(num_joints=K, image_size=image_size, heatmap_size=heatmap_size,
                                         joint_weights=jw, use_different_joint_weights=diff)))
            out[f'{tag}_target_{b}'], out[f'{tag}_weight_{b}'] = res['target'], res['target_weight']
    save('pipeline.npz', **out)


MOE = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=192, depth=3, num_heads=6, mlp_ratio=4,
           qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0, num_expert=3, part_features=48)
MOE_SOURCES = [0, 2, 1, 0, 2, 0]      # dataset index of each image of the fixture batch


def vitpose_plus():
    """ViTPose+ (ViTMoE + TopDownMoE): the reference detector class itself, built through the stub registry,
    train-mode forward_train losses + gradients and the eval-mode main-head heat maps."""
    refm = ref_loader.load_reference_moe()
    vcfg = dict(MOE)
    bb = O.synth_vit_weights(vcfg, seed=90)
    hc = O.classic_head_cfg(vcfg['embed_dim'])
    heads = [O.synth_head_weights(hc, seed=91 + j) for j in range(3)]

    def hcfg():
        return dict(type='TopdownHeatmapSimpleHead', in_channels=hc['in_channels'], out_channels=hc['out_channels'],
                    num_deconv_layers=hc['num_deconv_layers'], num_deconv_filters=hc['num_deconv_filters'],
                    num_deconv_kernels=hc['num_deconv_kernels'], extra=hc['extra'],
                    loss_keypoint=dict(type='JointsMSELoss', use_target_weight=True))
    det = refm.TopDownMoE(
        backbone=dict(type='ViTMoE', img_size=vcfg['img_size'], patch_size=16, embed_dim=vcfg['embed_dim'],
                      depth=vcfg['depth'], num_heads=vcfg['num_heads'], ratio=1, use_checkpoint=False, mlp_ratio=4,
                      qkv_bias=True, drop_path_rate=0.0, num_expert=vcfg['num_expert'],
                      part_features=vcfg['part_features']),
        keypoint_head=hcfg(), associate_keypoint_head=[hcfg(), hcfg()], train_cfg=dict(),
        test_cfg=dict(flip_test=False, post_process='default', shift_heatmap=False, modulate_kernel=11))
    det.backbone.load_state_dict(bb, strict=True)
    det.keypoint_head.load_state_dict(heads[0], strict=False)
    for j in range(2):
        det.associate_keypoint_heads[j].load_state_dict(heads[1 + j], strict=False)
    img, target, tw = O.synth_batch(len(MOE_SOURCES), seed=95)
    metas = [dict(dataset_idx=s) for s in MOE_SOURCES]
    det.train()
    losses = det.forward_train(img, target, tw, metas)
    total = losses['main_stream_loss'] + losses['1_loss'] + losses['2_loss']
    total.backward()
    out = dict(loss=torch.stack([losses['main_stream_loss'], losses['1_loss'], losses['2_loss']]).detach(),
               acc=np.array([losses['main_stream_acc'], losses['1_acc'], losses['2_acc']], np.float64),
               sources=np.array(MOE_SOURCES))
    named = dict(det.named_parameters())
    for k in ['backbone.blocks.0.mlp.fc2.bias', 'backbone.blocks.2.mlp.fc2.bias', 'backbone.blocks.1.mlp.experts.0.bias',
              'backbone.blocks.1.mlp.experts.1.bias', 'backbone.blocks.2.mlp.experts.2.bias',
              'backbone.blocks.0.mlp.fc1.bias', 'backbone.pos_embed', 'keypoint_head.final_layer.bias',
              'associate_keypoint_heads.0.final_layer.bias', 'associate_keypoint_heads.1.deconv_layers.1.weight']:
        out['g.' + k] = named[k].grad
    for k in ['backbone.blocks.0.mlp.fc2.weight', 'backbone.blocks.2.mlp.fc2.weight',
              'backbone.blocks.0.mlp.experts.0.weight', 'backbone.blocks.1.mlp.experts.1.weight',
              'backbone.blocks.2.mlp.experts.2.weight', 'backbone.blocks.1.mlp.fc1.weight',
              'backbone.blocks.0.attn.qkv.weight', 'associate_keypoint_heads.1.final_layer.weight']:
        out['gs.' + k] = named[k].grad.flatten(1)[::4, ::4]
    det.eval()
    with torch.no_grad():
        src = torch.tensor(MOE_SOURCES)
        feat = det.backbone(img, src)
        out['feat_sub'] = feat[:, ::4]
        out['heat_main_sub'] = det.keypoint_head(feat)[:, ::4]
        out['heat_assoc1_sub'] = det.associate_keypoint_heads[1](feat)[:, ::4]
    save('vitpose_plus.npz', **out)


if __name__ == '__main__':
    if sys.argv[1:] == ['vitpose_plus']:
        vitpose_plus()
        sys.exit(0)
    toy_vit()
    vitpose_s_eval()
    vitpose_s_train()
    block_widths()
    misc()
    decode()
    pipeline()
    vitpose_plus()
