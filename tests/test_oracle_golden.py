"""The CPU oracle (oracle/vitpose_oracle.py) against the golden vectors produced by
the reference's own source files (tests/golden/make_golden.py), and against the
known answers in the reference's unit tests.  Tolerances: fp32 torch CPU ops on
both sides, so only summation-order noise is allowed (1e-5 relative)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vitpose_oracle as O

torch.set_num_threads(4)
TOY = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=64, depth=2, num_heads=2,
           mlp_ratio=4, qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0)


def close(a, b, rtol=2e-5, atol=2e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_toy_vit_stages(golden_dir):
    g = np.load(os.path.join(golden_dir, 'toy_vit.npz'))
    sd = O.synth_vit_weights(TOY, seed=10)
    img, _, _ = O.synth_batch(2, seed=11)
    with torch.no_grad():
        close(O.vit_forward(sd, img, TOY), g['feat'])
        x, _ = O.patch_embed(sd, img, TOY)
        x = x + sd['pos_embed'][:, 1:] + sd['pos_embed'][:, :1]
        close(x[:1], g['block_in'])
        taps = {}
        y = O.block(sd, 0, x, TOY, taps=taps)
        close(y[:1], g['block_out'])
        close(taps['ln1'][:1], g['ln1'])
        close(taps['attn_o'][:1], g['attn_o'])
        close(taps['fc1_pre'][:1], g['fc1_pre'])
        close(taps['fc1_act'][:1], g['fc1_act'])
        # reference qkv tap is the raw Linear output [B,N,3E]
        qkv = taps['qkv'].permute(1, 3, 0, 2, 4).reshape(2, 192, -1)
        close(qkv[:1], g['qkv'])


def test_vitpose_s_eval(golden_dir):
    vcfg = O.vit_cfg('S')
    bb = O.synth_vit_weights(vcfg, seed=20)
    img, _, _ = O.synth_batch(2, seed=22)
    g = np.load(os.path.join(golden_dir, 'vitpose_s_classic_eval.npz'))
    hc = O.classic_head_cfg(384)
    hd = O.synth_head_weights(hc, seed=21)
    with torch.no_grad():
        feat = O.vit_forward(bb, img, vcfg)
        close(feat[:, ::8], g['feat_sub'], 1e-4, 1e-4)
        heat = O.head_forward(hd, feat, hc)
        assert np.abs(g['heat']).max() > 0.5  # non-degenerate heatmaps
        close(heat, g['heat'], 1e-4, 1e-4)
        g2 = np.load(os.path.join(golden_dir, 'vitpose_s_simple_eval.npz'))
        hs = O.simple_head_cfg(384)
        close(O.head_forward(O.synth_head_weights(hs, seed=23), feat[:1], hs), g2['heat'], 1e-4, 1e-4)
        # flip test: average of the plain pass and the flipped-back pass of the mirrored image
        g3 = np.load(os.path.join(golden_dir, 'vitpose_s_classic_fliptest.npz'))
        pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
        hf = O.head_forward(hd, O.vit_forward(bb, img.flip(3), vcfg), hc).numpy()
        close((heat.numpy() + O.flip_back(hf, pairs)) * 0.5, g3['heat'], 1e-4, 1e-4)


def test_vitpose_s_train_grads(golden_dir):
    g = np.load(os.path.join(golden_dir, 'vitpose_s_train.npz'))
    vcfg = O.vit_cfg('S', drop_path_rate=0.3)
    bb = {k: v.requires_grad_() for k, v in O.synth_vit_weights(vcfg, seed=30).items()}
    hc = O.classic_head_cfg(384)
    hd = O.synth_head_weights(hc, seed=31)
    for k, v in hd.items():
        if 'running' not in k:
            v.requires_grad_()
    img, target, tw = O.synth_batch(4, seed=32)
    rates = O.drop_path_rates(vcfg)
    draws = torch.from_numpy(g['dp_draws'])
    ones = torch.ones(4)
    scales = [(ones, ones)] + [(O.drop_path_scale(draws[2 * (i - 1)], rates[i]),
                                O.drop_path_scale(draws[2 * (i - 1) + 1], rates[i]))
                               for i in range(1, vcfg['depth'])]
    assert any((s[0] == 0).any() or (s[1] == 0).any() for s in scales)  # some paths dropped
    feat = O.vit_forward(bb, img, vcfg, scales)
    heat = O.head_forward(hd, feat, hc, training=True)
    loss = O.joints_mse_loss(heat, target, tw)
    close(heat[:, ::4], g['heat_sub'], 1e-4, 1e-4)
    close(loss, g['loss'], 1e-5, 1e-7)
    acc = O.pck_accuracy(heat.detach().numpy(), target.numpy(), tw.numpy().squeeze(-1) > 0)[1]
    assert abs(acc - float(g['acc'])) < 1e-9
    loss.backward()
    for k in g.files:
        if k.startswith('g.'):
            close(bb[k[2:]].grad, g[k], 2e-4, 1e-7)
        elif k.startswith('gs.'):
            close(bb[k[3:]].grad.flatten(1)[::16, ::4], g[k], 2e-4, 1e-7)
        elif k.startswith('gh.'):
            close(hd[k[3:]].grad, g[k], 2e-4, 1e-7)
        elif k.startswith('ghs.'):
            close(hd[k[4:]].grad.flatten(1)[::8, ::16], g[k], 2e-4, 1e-7)
    close(hd['deconv_layers.1.running_mean'], g['bn1_mean'])
    close(hd['deconv_layers.1.running_var'], g['bn1_var'])
    close(hd['deconv_layers.4.running_mean'], g['bn4_mean'])
    close(hd['deconv_layers.4.running_var'], g['bn4_var'])


def test_block_widths(golden_dir):
    g = np.load(os.path.join(golden_dir, 'block_widths.npz'))
    for name in ('B', 'L', 'H'):
        cfg = O.vit_cfg(name, depth=1, drop_path_rate=0.0)
        sd = O.synth_vit_weights(cfg, seed=40)
        x = torch.randn(1, 192, cfg['embed_dim'], generator=torch.Generator().manual_seed(41))
        with torch.no_grad():
            close(O.block(sd, 0, x, cfg)[:, ::8], g[f'{name}.block_out_sub'], 1e-4, 1e-4)


def test_loss_pck_flip_known_answers(golden_dir):
    m = json.load(open(os.path.join(golden_dir, 'misc.json')))
    # the reference's own unit-test answers (tests/test_losses/test_top_down_losses.py:32-43)
    z = torch.zeros((1, 3, 64, 64))
    w = torch.tensor([1., 0., 1.]).reshape(1, 3, 1)
    assert float(O.joints_mse_loss(z, z, w)) == 0.0 == m['mse_zero']
    assert float(O.joints_mse_loss(torch.ones_like(z), z, w)) == pytest.approx(2 / 3) == pytest.approx(m['mse_ones'])
    fo = torch.zeros((1, 2, 64, 64)); fo[0, 0] += 1
    assert float(O.joints_mse_loss(fo, torch.zeros_like(fo), torch.ones(1, 2, 1))) == 0.5 == m['mse_half']
    g = torch.Generator().manual_seed(50)
    o, t = torch.randn(3, 17, 64, 48, generator=g), torch.rand(3, 17, 64, 48, generator=g)
    w = (torch.rand(3, 17, 1, generator=g) > 0.3).float()
    assert float(O.joints_mse_loss(o, t, w)) == pytest.approx(m['mse_rand'], rel=1e-6)
    assert float(O.joints_mse_loss(o, t, w, use_target_weight=False, loss_weight=0.5)) == \
        pytest.approx(m['mse_rand_noweight'], rel=1e-6)
    # PCK: the reference's unit case (tests/test_evaluation/test_top_down_eval.py:11-26)
    out = np.zeros((1, 5, 64, 64), dtype=np.float32); tgt = np.zeros_like(out)
    out[0, 0, 0, 0] = out[0, 1, 0, 0] = out[0, 2, 0, 1] = out[0, 3, 0, 1] = out[0, 4, 0, 0] = 1
    tgt[0, 0, 0, 0] = tgt[0, 1, 0, 0] = tgt[0, 2, 0, 0] = tgt[0, 3, 0, 0] = tgt[0, 4, 63, 63] = 1
    acc, avg, cnt = O.pck_accuracy(out, tgt, np.array([[True, True, False, False, False]]), 0.5)
    assert list(acc) == [1.0, 1.0, -1.0, -1.0, -1.0] == m['pck_unit']['acc']
    assert avg == 1.0 and cnt == 2
    acc, avg, cnt = O.pck_accuracy(o.numpy(), t.numpy(), w.numpy().squeeze(-1) > 0)
    np.testing.assert_allclose(acc, m['pck_rand']['acc'])
    assert avg == pytest.approx(m['pck_rand']['avg']) and cnt == m['pck_rand']['cnt']
    pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    fb = O.flip_back(o.numpy().copy(), pairs)
    chk = [float(fb[0, 1, 3, 5]), float(fb[2, 16, 63, 47]), float(fb.astype(np.float64).sum()),
           float((fb.astype(np.float64) * np.arange(fb.size).reshape(fb.shape)).sum())]
    np.testing.assert_allclose(chk, m['flip_back_checksum'], rtol=1e-12)


def test_layer_decay_groups(golden_dir):
    m = json.load(open(os.path.join(golden_dir, 'misc.json')))
    for name, rate in (('S', 0.8), ('B', 0.75)):
        vcfg = O.vit_cfg(name)
        named = [('backbone.' + k, s) for k, s in O.vit_param_shapes(vcfg)]
        named += [('keypoint_head.' + k, s) for k, s in O.head_param_shapes(O.classic_head_cfg(vcfg['embed_dim']))
                  if 'running' not in k]
        got = O.layer_decay_groups(named, vcfg['depth'], rate, 5e-4, 0.1)
        ref = m[f'layer_decay_{name}']
        assert len(got) == len(ref)
        for pn, lid, scale, lr, wd in got:
            assert ref[pn] == pytest.approx([scale, lr, wd], rel=1e-12), pn


MOE = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=192, depth=3, num_heads=6, mlp_ratio=4,
           qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0, num_expert=3, part_features=48)


def test_vitpose_plus(golden_dir):
    """ViTMoE + TopDownMoE: the oracle against the reference detector's own forward_train / backward
    (vit_moe.py, top_down_moe.py executed from source by make_golden.py: vitpose_plus)."""
    g = np.load(os.path.join(golden_dir, 'vitpose_plus.npz'))
    src = torch.tensor(g['sources'])
    bb = {k: v.requires_grad_() for k, v in O.synth_vit_weights(MOE, seed=90).items()}
    hc = O.classic_head_cfg(MOE['embed_dim'])
    heads = [{k: (v.requires_grad_() if 'running' not in k else v) for k, v in O.synth_head_weights(hc, seed=91 + j).items()}
             for j in range(3)]
    img, target, tw = O.synth_batch(len(src), seed=95)
    losses, _ = O.topdown_moe_losses(bb, heads, img, target, tw, src, MOE, [hc] * 3, training=True)
    close(torch.stack(losses), g['loss'], 1e-5, 1e-7)
    sum(losses).backward()
    prefixes = {'backbone.': bb, 'keypoint_head.': heads[0], 'associate_keypoint_heads.0.': heads[1],
                'associate_keypoint_heads.1.': heads[2]}
    checked = 0
    for k in g.files:
        if not (k.startswith('g.') or k.startswith('gs.')):
            continue
        name = k.split('.', 1)[1]
        pre = next(p for p in prefixes if name.startswith(p))
        grad = prefixes[pre][name[len(pre):]].grad
        if k.startswith('gs.'):
            grad = grad.flatten(1)[::4, ::4]
        close(grad, g[k], 2e-4, 1e-7)
        checked += 1
    assert checked == 18
    # an expert only receives gradient from its own dataset's images: expert 1 of block 1 is non-zero here
    assert np.abs(g['gs.backbone.blocks.1.mlp.experts.1.weight']).max() > 0
    with torch.no_grad():
        # eval mode after one training pass: the BatchNorm running statistics carry that pass's update
        heads_e = [{k: v.detach() for k, v in h.items()} for h in heads]
        feat = O.vit_forward({k: v.detach() for k, v in bb.items()}, img, MOE, dataset_source=src)
        close(feat[:, ::4], g['feat_sub'], 1e-4, 1e-4)
        close(O.head_forward(heads_e[0], feat, hc)[:, ::4], g['heat_main_sub'], 1e-4, 1e-4)
        close(O.head_forward(heads_e[2], feat, hc)[:, ::4], g['heat_assoc1_sub'], 1e-4, 1e-4)
