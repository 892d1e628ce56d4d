"""Device-side decoding (vp_decode_heatmaps, vp_decode_blurred, vp_flip_back_avg through the C ABI) against the
oracle, the reference-produced vectors and the known answers of the reference's unit test.  The product decodes
every post_process branch on the GPU; a host array is uploaded first."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_oracle as D
from oracle import vitpose_oracle as O

pytestmark = pytest.mark.gpu
N, K, H, W = 4, 17, 64, 48
PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'decode.npz'))


def test_device_decode_vs_reference_vectors_and_oracle(dev, gold):
    from vitpose_amd.models import keypoints_from_heatmaps_device as dec
    maps = D.synth_heatmaps(N, K, H, W, seed=70)
    hm = torch.from_numpy(maps).to(dev)
    c, s = gold['center'], gold['scale']
    p, m = dec(hm, c, s, mode=1)
    np.testing.assert_allclose(p, gold['default_preds'], atol=1e-4)
    assert np.array_equal(m, gold['default_maxvals'])                       # scores: bit-exact
    p, _ = dec(hm, c, s, mode=0)
    np.testing.assert_allclose(p, gold['none_preds'], atol=1e-4)
    p, m = dec(hm, c, s, mode=2, kernel=1, use_udp=True)                  # reference's DARK-UDP, identity blur
    np.testing.assert_allclose(p, gold['udp_k1_preds'], atol=2e-3)
    assert np.array_equal(m, gold['udp_k1_maxvals'])
    for ks in (3, 11, 17):                                                   # blurred variants vs the oracle
        p, m = dec(hm, c, s, mode=2, kernel=ks, use_udp=True)
        po, mo = D.keypoints_from_heatmaps(maps, c, s, use_udp=True, kernel=ks)
        np.testing.assert_allclose(p, po, atol=5e-3, err_msg=f'kernel {ks}')
        assert np.array_equal(m, mo)
    # ties: the FIRST maximum wins (np.argmax); an all-non-positive map is invalid (-1, -1)
    t = torch.zeros(1, 3, H, W, device=dev)
    t[0, 0, 10, 7] = t[0, 0, 40, 3] = 1.0
    t[0, 1] = -1.0
    t[0, 2, H - 1, W - 1] = 0.5
    p, m = dec(t, np.zeros((1, 2), np.float32), np.full((1, 2), W / 200.0, np.float32), mode=0)
    po, mo = D.keypoints_from_heatmaps(t.cpu().numpy(), np.zeros((1, 2), np.float32),
                                       np.full((1, 2), W / 200.0, np.float32), post_process=None)
    np.testing.assert_allclose(p, po, atol=1e-5)
    assert np.array_equal(m, mo)


def test_reference_unit_case_on_device(dev):
    """tests/test_evaluation/test_top_down_eval.py:29-89 (test_keypoints_from_heatmaps), through the product's
    keypoints_from_heatmaps on host arrays and on device tensors."""
    from vitpose_amd.models import keypoints_from_heatmaps as f
    hm = np.ones((1, 1, 64, 64), dtype=np.float32)
    hm[0, 0, 31, 31] = 2
    c, s = np.array([[127, 127]]), np.array([[64 / 200.0, 64 / 200.0]])
    for x in (hm, torch.from_numpy(hm).to(dev)):
        p, m = f(x, c, s)
        np.testing.assert_array_almost_equal(p, np.array([[[126, 126]]]), decimal=4)
        np.testing.assert_array_almost_equal(m, np.array([[[2]]]), decimal=4)
        assert isinstance(p, np.ndarray) and isinstance(m, np.ndarray)
        p, m = f(x, c, s, post_process='unbiased')
        np.testing.assert_array_almost_equal(p, np.array([[[126, 126]]]), decimal=4)
        np.testing.assert_array_almost_equal(m, np.array([[[2]]]), decimal=4)
    assert hm[0, 0, 31, 31] == 2 and hm.sum() == 64 * 64 + 1            # the caller's array is not modified
    udp = np.ones((32, 17, 64, 64), dtype=np.float32)
    udp[:, :, 31, 31] = 2
    p, m = f(udp, np.tile([127, 127], (32, 1)), np.tile([32, 32], (32, 1)), post_process='default', use_udp=True)
    np.testing.assert_array_almost_equal(p, np.tile([76, 76], (32, 17, 1)), decimal=0)
    np.testing.assert_array_almost_equal(m, np.tile([2], (32, 17, 1)), decimal=4)


def test_blurred_branches_on_device_vs_oracle(dev, gold):
    """post_process 'unbiased' (DARK) and 'megvii', and UDP 'CombinedTarget': whole-map blurs on the GPU against the
    oracle's scipy restatement (f64 blur: the f32 summation order differs).  The Taylor step is pinned by the
    reference's own vectors (taylor_coords: `_taylor` of the reference on the log maps) through a 1-tap blur."""
    from vitpose_amd.models import keypoints_from_heatmaps as f, keypoints_from_heatmaps_device as dec
    maps = D.synth_heatmaps(N, K, H, W, seed=70)
    hm = torch.from_numpy(maps).to(dev)
    c, s = gold['center'], gold['scale']
    for kw in (dict(post_process='unbiased', kernel=11), dict(post_process='unbiased', kernel=3),
               dict(post_process='megvii', kernel=5), dict(post_process='megvii', kernel=11)):
        p, m = f(hm, c, s, **kw)
        po, mo = D.keypoints_from_heatmaps(maps, c, s, **kw)
        np.testing.assert_allclose(p, po, atol=5e-3, err_msg=str(kw))          # image pixels
        np.testing.assert_allclose(m, mo, atol=1e-6, err_msg=str(kw))
    # a 1-tap blur inside a zero border of 0 pixels is the identity, the rescale factor is 1: the 'unbiased' branch
    # is then the reference's `_taylor` on log(max(map, 1e-10)) -- its own vectors
    unit = (np.full((N, 2), [W / 2.0, H / 2.0], np.float32), np.full((N, 2), [W / 200.0, H / 200.0], np.float32))   # identity map to the image
    p, _ = dec(hm, *unit, mode=3, kernel=1)
    np.testing.assert_allclose(p, gold['taylor_coords'], atol=1e-4)
    # CombinedTarget: three maps per joint (heat, x offset, y offset)
    g = np.random.default_rng(9)
    ct = np.zeros((N, 3 * 6, H, W), dtype=np.float32)
    ct[:, 0::3] = maps[:, :6]
    ct[:, 1::3] = g.uniform(-1, 1, (N, 6, H, W)).astype(np.float32)
    ct[:, 2::3] = g.uniform(-1, 1, (N, 6, H, W)).astype(np.float32)
    for ks in (3, 11):
        p, m = f(torch.from_numpy(ct).to(dev), c, s, use_udp=True, target_type='CombinedTarget', kernel=ks)
        po, mo = D.combined_target_decode(ct, c, s, kernel=ks)
        assert p.shape == (N, 6, 2) and m.shape == (N, 6, 1)
        np.testing.assert_allclose(p, po, atol=5e-3, err_msg=f'CombinedTarget kernel {ks}')
        np.testing.assert_allclose(m, mo, atol=1e-6)
    # joints without a positive maximum: the reference's arg-max is (-1, -1) and its flattened offset lookup
    # (top_down_eval.py:582-585) then reads the previous joint's maps -- the first joint of the batch wraps to the last
    dead = ct.copy()
    dead[0, 0] = -np.abs(dead[0, 0])             # first joint of the batch (the wrap-around case)
    dead[1, 3 * 2] = 0.0                         # an all-zero map in the middle
    dead[N - 1, 3 * 5] = -1.0                    # the last joint
    p, m = f(torch.from_numpy(dead).to(dev), c, s, use_udp=True, target_type='CombinedTarget', kernel=11)
    po, mo = D.combined_target_decode(dead, c, s, kernel=11)
    assert mo[0, 0, 0] <= 0 and mo[1, 2, 0] <= 0 and mo[N - 1, 5, 0] <= 0
    np.testing.assert_allclose(p, po, atol=5e-3, err_msg='CombinedTarget, joints without a positive maximum')
    np.testing.assert_allclose(m, mo, atol=1e-6)


def test_head_decode_every_test_cfg_on_device(dev, gold):
    """TopdownHeatmapSimpleHead.decode (topdown_heatmap_base_head.py:40-103) with each post-processing test_cfg:
    same keypoints as the oracle, boxes / ids / paths as the reference lays them out."""
    from vitpose_amd.builder import build_head
    maps = D.synth_heatmaps(N, K, H, W, seed=70)
    c, s = gold['center'], gold['scale']
    metas = [dict(center=c[i], scale=s[i], image_file=f'{i}.jpg', bbox_id=10 + i, bbox_score=0.5 + 0.1 * i)
             for i in range(N)]
    for tc, okw in ((dict(post_process='default'), dict(post_process='default')),
                    (dict(post_process='unbiased', modulate_kernel=11), dict(post_process='unbiased', kernel=11)),
                    (dict(post_process='megvii', modulate_kernel=5), dict(post_process='megvii', kernel=5)),
                    (dict(post_process='default', use_udp=True, modulate_kernel=11), dict(use_udp=True, kernel=11)),
                    (dict(unbiased_decoding=True), dict(post_process='unbiased', kernel=11))):
        head = build_head(dict(type='TopdownHeatmapSimpleHead', in_channels=64, out_channels=K, num_deconv_layers=0,
                               extra=dict(final_conv_kernel=1), loss_keypoint=dict(type='JointsMSELoss'),
                               test_cfg=dict(tc)))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', DeprecationWarning)
            for x in (maps, torch.from_numpy(maps).to(dev)):
                res = head.decode(metas, x)
        po, mo = D.keypoints_from_heatmaps(maps, c, s, **okw)
        np.testing.assert_allclose(res['preds'][:, :, :2], po, atol=5e-3, err_msg=str(tc))
        np.testing.assert_allclose(res['preds'][:, :, 2:3], mo, atol=1e-6)
        assert res['preds'].dtype == np.float32 and res['boxes'].shape == (N, 6)
        np.testing.assert_allclose(res['boxes'][:, 0:2], c)
        np.testing.assert_allclose(res['boxes'][:, 2:4], s)
        np.testing.assert_allclose(res['boxes'][:, 4], np.prod(s * 200.0, axis=1), rtol=1e-6)
        np.testing.assert_allclose(res['boxes'][:, 5], [0.5, 0.6, 0.7, 0.8], rtol=1e-6)
        assert res['image_paths'] == [f'{i}.jpg' for i in range(N)] and res['bbox_ids'] == [10, 11, 12, 13]


@pytest.mark.parametrize('shift', [False, True])
def test_flip_back_avg_device(dev, shift):
    from vitpose_amd.models import flip_back_avg_device
    g = np.random.default_rng(5)
    a, b = g.random((3, K, H, W), dtype=np.float32), g.random((3, K, H, W), dtype=np.float32)
    got = flip_back_avg_device(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), PAIRS, shift).cpu().numpy()
    assert np.array_equal(got, D.flip_back_avg(a, b, PAIRS, shift))


def test_forward_test_udp_config_device_decode_equals_host_decode(dev):
    """The ViTPose test_cfg (flip_test, post_process='default', use_udp, modulate_kernel 11): the
    device-resident path (flip average + decode on the GPU) returns what the host recipe returns on the
    same heat maps."""
    from vitpose_amd.builder import build_posenet
    from vitpose_amd.models import keypoints_from_heatmaps
    vcfg, hc = O.vit_cfg('S', depth=2), O.classic_head_cfg(384)
    tc = dict(flip_test=True, post_process='default', shift_heatmap=False, target_type='GaussianHeatmap',
              modulate_kernel=11, use_udp=True)
    m = build_posenet(dict(type='TopDown', backbone=dict(vcfg, type='ViT'),
                           keypoint_head=dict(hc, type='TopdownHeatmapSimpleHead',
                                              loss_keypoint=dict(type='JointsMSELoss', use_target_weight=True)),
                           train_cfg=dict(), test_cfg=tc))
    sd = {'backbone.' + k: v for k, v in O.synth_vit_weights(vcfg, seed=20).items()}
    sd.update({'keypoint_head.' + k: v for k, v in O.synth_head_weights(hc, seed=21).items()})
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    img, _, _ = O.synth_batch(3, seed=22)
    metas = [dict(center=np.array([100. + i, 120.]), scale=np.array([1.2, 1.6]), image_file=f'{i}.jpg', bbox_id=i,
                  flip_pairs=PAIRS, bbox_score=0.9) for i in range(3)]
    res = m(img=img.to(dev), img_metas=metas, return_loss=False, return_heatmap=True)
    hm = res['output_heatmap']
    assert isinstance(hm, np.ndarray) and hm.shape == (3, 17, 64, 48)
    c = np.stack([mt['center'] for mt in metas]).astype(np.float32)
    s = np.stack([mt['scale'] for mt in metas]).astype(np.float32)
    p, mv = keypoints_from_heatmaps(hm, c, s, post_process='default', kernel=11, use_udp=True)
    # random-weight heat maps are not Gaussian bumps: some Newton steps are ill-conditioned and amplify the
    # f32 summation-order difference of the two blurs (0.012 image px measured; 1 heat-map px = 5 image px)
    np.testing.assert_allclose(res['preds'][:, :, :2], p, atol=5e-2)
    np.testing.assert_allclose(res['preds'][:, :, 2:3], mv, atol=0)
    # heat maps themselves: plain + flipped passes averaged as the oracle does it
    with torch.no_grad():
        h0 = m.forward_dummy(img.to(dev)).cpu().numpy()
        h1 = m.forward_dummy(img.flip(3).to(dev)).cpu().numpy()
    np.testing.assert_allclose(hm, D.flip_back_avg(h0, h1, PAIRS, False), atol=1e-6)
    assert m(img=img.to(dev), img_metas=metas, return_loss=False)['output_heatmap'] is None
