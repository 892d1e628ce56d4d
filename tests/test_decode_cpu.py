"""Heat-map decoding on the host: the oracle (oracle/decode_oracle.py) against the vectors the reference's
own functions produced (tests/golden/decode.npz) and against the known answers of the reference's unit
test (tests/test_evaluation/test_top_down_eval.py:29-89); and the host-side pieces the product keeps (argument
rules, transform_preds, flip_back, the OpenCV blur taps).  The decoding itself runs on the GPU
(tests/test_decode_gpu.py)."""
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


def test_reference_unit_test_known_answers():
    f = D.keypoints_from_heatmaps
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


def test_product_host_helpers(gold, maps):
    """What stays on the host in the product: the argument rules of keypoints_from_heatmaps, transform_preds
    (vs the reference's vectors), flip_back (vs the oracle), the blur taps (vs the oracle's f64 formula)."""
    import warnings

    from oracle import vitpose_oracle as O
    from vitpose_amd import models as M
    for ks in (1, 3, 5, 7, 11, 17, 23):
        np.testing.assert_allclose(M.gaussian_kernel1d(ks), D.gaussian_kernel(ks), rtol=1e-6)
    c0 = np.stack([D.max_pred(maps[0, k])[0] for k in range(K)])
    for udp, key in ((True, 'transform_udp'), (False, 'transform_plain')):
        got = M.transform_preds(c0.copy(), gold['center'][0], gold['scale'][0], [W, H], use_udp=udp)
        np.testing.assert_allclose(got, gold[key], rtol=1e-6, atol=1e-4)
    pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
    assert np.array_equal(M.flip_back(maps.copy(), pairs), O.flip_back(maps.copy(), pairs))
    # CombinedTarget: joints are triples of maps, the x-offset map changes sign (post_transforms.py:133-137)
    ct = np.random.default_rng(3).random((2, 9, 8, 6), dtype=np.float32)
    fb = M.flip_back(ct.copy(), [[0, 2]], target_type='CombinedTarget')
    want = ct.copy()
    want[:, 1::3] *= -1
    want = want.reshape(2, 3, 3, 8, 6)[:, [2, 1, 0]].reshape(2, 9, 8, 6)[..., ::-1]
    assert np.array_equal(fb, want)
    # argument rules (top_down_eval.py:517-560)
    mode = M.device_decode_mode
    assert [mode(post_process=p) for p in (None, 'default', 'unbiased', 'megvii')] == [0, 1, 3, 4]
    assert mode(use_udp=True) == 2 and mode(use_udp=True, target_type='CombinedTarget') == 5
    for bad in (dict(unbiased=True, post_process=None), dict(unbiased=True, post_process='megvii'),
                dict(post_process='unbiased', kernel=0), dict(post_process='megvii', kernel=0),
                dict(use_udp=True, post_process='megvii')):
        with pytest.raises(AssertionError):
            mode(**bad)
    with pytest.raises(ValueError):
        mode(use_udp=True, target_type='Other')
    for legacy, want_mode in ((dict(post_process=False), 0), (dict(post_process=True), 1),
                              (dict(post_process=True, unbiased=True), 3), (dict(unbiased=True), 3)):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            assert mode(**legacy) == want_mode
        assert any(issubclass(x.category, DeprecationWarning) for x in w)
