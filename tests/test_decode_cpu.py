"""Heat-map decoding on the host: the oracle (oracle/decode_oracle.py) against the vectors the reference's
own functions produced (tests/golden/decode.npz) and against the known answers of the reference's unit
test (tests/test_evaluation/test_top_down_eval.py:29-89); then the product's host implementation
(vitpose_amd.models.keypoints_from_heatmaps, every post_process branch) against both."""
import os

import numpy as np
import pytest

from oracle import decode_oracle as D

N, K, H, W = 4, 17, 64, 48


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'decode.npz'))


@pytest.fixture(scope='module')
def maps():
    return D.synth_heatmaps(N, K, H, W, seed=70)


def test_oracle_matches_reference_vectors(gold, maps):
    c, s = gold['center'], gold['scale']
    p, m = D.keypoints_from_heatmaps(maps, c, s, post_process='default')
    np.testing.assert_allclose(p, gold['default_preds'], atol=1e-4)
    np.testing.assert_allclose(m, gold['default_maxvals'], atol=0)
    p, _ = D.keypoints_from_heatmaps(maps, c, s, post_process=None)
    np.testing.assert_allclose(p, gold['none_preds'], atol=1e-4)
    p, m = D.keypoints_from_heatmaps(maps, c, s, use_udp=True, kernel=1)
    np.testing.assert_allclose(p, gold['udp_k1_preds'], atol=2e-3)     # image units (scale ~4 px per heat-map px)
    p, _ = D.keypoints_from_heatmaps(maps, c, s, use_udp=True, kernel=1, image_space=False)
    np.testing.assert_allclose(p, gold['dark_udp_k1_coords'], atol=5e-4)
    lg = np.log(np.maximum(maps, 1e-10))
    for n in range(N):
        for k in range(K):
            c0, _ = D.max_pred(maps[n, k])
            np.testing.assert_allclose(D.taylor_one(lg[n, k], c0), gold['taylor_coords'][n, k], atol=1e-4)
    c0 = np.stack([D.max_pred(maps[0, k])[0] for k in range(K)])
    for udp, key in ((True, 'transform_udp'), (False, 'transform_plain')):
        got = np.stack([D.to_image(c0[k], c[0], s[0], W, H, udp) for k in range(K)])
        np.testing.assert_allclose(got, gold[key], rtol=1e-6, atol=1e-4)


def _unit_case():
    hm = np.ones((1, 1, 64, 64), dtype=np.float32)
    hm[0, 0, 31, 31] = 2
    return hm, np.array([[127, 127]]), np.array([[64 / 200.0, 64 / 200.0]])


@pytest.mark.parametrize('impl', ['oracle', 'product'])
def test_reference_unit_test_known_answers(impl):
    if impl == 'oracle':
        f = D.keypoints_from_heatmaps
    else:
        from vitpose_amd.models import keypoints_from_heatmaps as f
    hm, c, s = _unit_case()
    p, m = f(hm, c, s)
    np.testing.assert_array_almost_equal(p, np.array([[[126, 126]]]), decimal=4)
    np.testing.assert_array_almost_equal(m, np.array([[[2]]]), decimal=4)
    p, m = f(hm, c, s, post_process='unbiased')
    np.testing.assert_array_almost_equal(p, np.array([[[126, 126]]]), decimal=4)
    udp = np.ones((32, 17, 64, 64), dtype=np.float32)
    udp[:, :, 31, 31] = 2
    p, m = f(udp, np.tile([127, 127], (32, 1)), np.tile([32, 32], (32, 1)), post_process='default', use_udp=True)
    np.testing.assert_array_almost_equal(p, np.tile([76, 76], (32, 17, 1)), decimal=0)
    np.testing.assert_array_almost_equal(m, np.tile([2], (32, 17, 1)), decimal=4)


def test_product_host_decode_all_branches(gold, maps):
    from vitpose_amd.models import gaussian_blur2d, gaussian_kernel1d, keypoints_from_heatmaps
    c, s = gold['center'], gold['scale']
    with pytest.raises(AssertionError):
        keypoints_from_heatmaps(maps, c, s, post_process='unbiased', kernel=0)
    np.testing.assert_allclose(keypoints_from_heatmaps(maps, c, s)[0], gold['default_preds'], atol=1e-4)
    np.testing.assert_allclose(keypoints_from_heatmaps(maps, c, s, post_process=None)[0], gold['none_preds'], atol=1e-4)
    np.testing.assert_allclose(keypoints_from_heatmaps(maps, c, s, use_udp=True, kernel=1)[0], gold['udp_k1_preds'],
                               atol=2e-3)
    # blur: two independent restatements of OpenCV's filter agree (f32 separable vs f64 scipy)
    for ks in (3, 5, 7, 11, 17):
        np.testing.assert_allclose(gaussian_kernel1d(ks), D.gaussian_kernel(ks), rtol=1e-6)
        np.testing.assert_allclose(gaussian_blur2d(maps[0, 0], ks), D.blur(maps[0, 0], ks), atol=1e-6)
    for kw in (dict(use_udp=True, kernel=11), dict(use_udp=True, kernel=3), dict(post_process='unbiased', kernel=11),
               dict(post_process='megvii', kernel=5)):
        p, m = keypoints_from_heatmaps(maps, c, s, **kw)
        po, mo = D.keypoints_from_heatmaps(maps, c, s, **kw)
        np.testing.assert_allclose(p, po, atol=5e-3, err_msg=str(kw))
        np.testing.assert_allclose(m, mo, atol=1e-6)
    # the input array is not modified (the reference copies it first)
    before = maps.copy()
    keypoints_from_heatmaps(maps, c, s, use_udp=True, kernel=11)
    assert np.array_equal(before, maps)
