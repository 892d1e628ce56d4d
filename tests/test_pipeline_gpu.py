"""Device input side (vp_warp_affine_normalize, vp_generate_target through vitpose_amd.pipelines) against the
oracle and the reference-produced vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as P

pytestmark = pytest.mark.gpu
K, IMG, HM = 17, np.array([192, 256]), np.array([48, 64])
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
PIPE = [dict(type='LoadImageFromFile'), dict(type='TopDownRandomFlip', flip_prob=0.5),
        dict(type='TopDownAffine', use_udp=True), dict(type='ToTensor'),
        dict(type='NormalizeTensor', mean=MEAN, std=STD),
        dict(type='TopDownGenerateTarget', sigma=2, encoding='UDP', target_type='GaussianHeatmap')]
DATA = dict(image_size=[192, 256], heatmap_size=[48, 64], num_joints=K)


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def as_results(samples):
    return [dict(img=s['img'], center=s['center'], scale=s['scale'], rotation=s['rotation'],
                 joints_3d=s['joints_3d'].copy(), joints_3d_visible=np.stack([s['visible']] * 3, 1)) for s in samples]


def test_udp_pipeline_batch_vs_oracle_and_reference_vectors(dev, golden_dir):
    from vitpose_amd.pipelines import TopDownBatchTransform
    gold = np.load(os.path.join(golden_dir, 'pipeline.npz'))
    samples = P.synth_samples(6, K=K, seed=80)
    tf = TopDownBatchTransform.from_cfg(PIPE, DATA, device=dev)
    assert tf.use_udp and tf.mode == 0
    out = tf(as_results(samples))
    assert out['img'].shape == (6, 3, 256, 192) and out['target'].shape == (6, K, 64, 48)
    for b, s in enumerate(samples):
        trans = gold[f'trans_{b}']
        want = P.to_tensor_normalize(P.warp_affine_u8(s['img'], trans, 192, 256), MEAN, STD)
        got = out['img'][b].cpu().numpy()
        assert np.array_equal(got, want), np.abs(got - want).max()                  # integer warp: bit-exact
        np.testing.assert_allclose(out['joints_3d'][b], gold[f'joints_{b}'], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(out['target'][b].cpu().numpy(), gold[f'udp_target_{b}'], atol=2e-6)
        assert np.array_equal(out['target_weight'][b].cpu().numpy(), gold[f'udp_weight_{b}'])


@pytest.mark.parametrize('tag,kw', [('msra', dict(sigma=2, encoding='MSRA', use_different_joint_weights=True)),
                                    ('msra_unb', dict(sigma=3, encoding='MSRA', unbiased_encoding=True))])
def test_msra_targets_vs_reference_vectors(dev, golden_dir, tag, kw):
    from vitpose_amd.pipelines import TopDownBatchTransform
    gold = np.load(os.path.join(golden_dir, 'pipeline.npz'))
    samples = P.synth_samples(6, K=K, seed=80)
    jw = np.linspace(1.0, 1.5, K).astype(np.float32).reshape(K, 1)
    tf = TopDownBatchTransform(IMG, HM, K, use_udp=True, joint_weights=jw, device=dev, **kw)
    out = tf(as_results(samples))
    for b in range(6):
        np.testing.assert_allclose(out['target'][b].cpu().numpy(), gold[f'{tag}_target_{b}'], atol=2e-6)
        np.testing.assert_allclose(out['target_weight'][b].cpu().numpy(), gold[f'{tag}_weight_{b}'], atol=1e-7)


def test_non_udp_affine_vs_oracle(dev):
    from vitpose_amd.pipelines import TopDownBatchTransform
    samples = P.synth_samples(4, K=K, seed=81)
    tf = TopDownBatchTransform(IMG, HM, K, use_udp=False, mean=MEAN, std=STD, sigma=2, encoding='MSRA', device=dev)
    out = tf(as_results(samples), with_target=False)
    for b, s in enumerate(samples):
        # (the two three-point solves agree to ~1e-5, tests/test_pipeline_cpu.py; the byte-exact comparison of
        # the warp itself uses one matrix on both sides)
        trans = tf.matrix(dict(center=s['center'], scale=s['scale'], rotation=s['rotation']))
        want = P.to_tensor_normalize(P.warp_affine_u8(s['img'], trans, 192, 256), MEAN, STD)
        assert np.array_equal(out['img'][b].cpu().numpy(), want)
        ref_m = P.get_affine_transform(s['center'], s['scale'], s['rotation'], IMG)
        other = P.to_tensor_normalize(P.warp_affine_u8(s['img'], ref_m, 192, 256), MEAN, STD)
        assert np.mean(out['img'][b].cpu().numpy() != other) < 0.02      # a few pixels move by one grey level


def test_pipeline_feeds_the_model(dev):
    """Device-born images, targets and weights go straight into TopDown.forward_train."""
    from vitpose_amd.builder import build_posenet
    import vitpose_amd.models  # noqa: F401  (registers the model classes)
    from vitpose_amd.pipelines import TopDownBatchTransform
    from oracle import vitpose_oracle as O
    toy = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=64, depth=2, num_heads=2, mlp_ratio=4,
               qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0)
    m = build_posenet(dict(type='TopDown', backbone=dict(toy, type='ViT'),
                           keypoint_head=dict(O.classic_head_cfg(64), type='TopdownHeatmapSimpleHead',
                                              loss_keypoint=dict(type='JointsMSELoss', use_target_weight=True)),
                           train_cfg=dict(), test_cfg=dict())).to(dev).train()
    out = TopDownBatchTransform.from_cfg(PIPE, DATA, device=dev)(as_results(P.synth_samples(4, K=K, seed=82)))
    losses = m.forward_train(out['img'], out['target'], out['target_weight'], None)
    losses['heatmap_loss'].backward()
    assert torch.isfinite(losses['heatmap_loss']) and losses['heatmap_loss'].item() > 0
