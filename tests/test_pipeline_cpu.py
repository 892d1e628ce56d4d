"""Input-side oracle (oracle/pipeline_oracle.py) against the vectors the reference's own functions produced
(tests/golden/pipeline.npz: get_warp_matrix, warp_affine_joints, TopDownGenerateTarget UDP / MSRA / MSRA
unbiased), and the host half of the product (vitpose_amd.pipelines: matrices, joint mapping) against both."""
import os

import numpy as np
import pytest

from oracle import pipeline_oracle as P

K, IMG, HM = 17, np.array([192, 256]), np.array([48, 64])


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'pipeline.npz'))


def test_oracle_matches_reference_vectors(gold):
    jw = np.linspace(1.0, 1.5, K).astype(np.float32)
    for b, r in enumerate(P.synth_samples(6, K=K, seed=80)):
        trans = P.get_warp_matrix(r['rotation'], r['center'] * 2.0, IMG - 1.0, r['scale'] * 200.0)
        np.testing.assert_allclose(trans, gold[f'trans_{b}'], rtol=1e-6, atol=1e-5)
        j = r['joints_3d'].copy()
        j[:, :2] = P.warp_affine_joints(j[:, :2], gold[f'trans_{b}'])
        np.testing.assert_allclose(j, gold[f'joints_{b}'], rtol=1e-5, atol=1e-3)
        j = gold[f'joints_{b}']
        for tag, kw in (('udp', dict(sigma=2, encoding='UDP')), ('msra', dict(sigma=2, encoding='MSRA', joint_weights=jw)),
                        ('msra_unb', dict(sigma=3, encoding='MSRA', unbiased=True))):
            t, w = P.generate_target(j, r['visible'], IMG, HM, **kw)
            np.testing.assert_allclose(w, gold[f'{tag}_weight_{b}'], atol=1e-7, err_msg=tag)
            np.testing.assert_allclose(t, gold[f'{tag}_target_{b}'], atol=2e-6, err_msg=tag)
        assert gold[f'udp_weight_{b}'][K - 1, 0] == 0          # the far-away joint is dropped


def test_product_host_matrices_match_reference_and_oracle(gold):
    from vitpose_amd import pipelines as V
    for b, r in enumerate(P.synth_samples(6, K=K, seed=80)):
        trans = V.get_warp_matrix(r['rotation'], r['center'] * 2.0, IMG - 1.0, r['scale'] * 200.0)
        assert trans.dtype == np.float32 and np.array_equal(trans, gold[f'trans_{b}'])   # bit for bit: the image warp is integer work
        np.testing.assert_allclose(V.warp_affine_joints(r['joints_3d'][:, :2], gold[f'trans_{b}']),
                                   gold[f'joints_{b}'][:, :2], rtol=1e-5, atol=1e-3)
        a = V.get_affine_transform(r['center'], r['scale'], r['rotation'], IMG)
        np.testing.assert_allclose(a, P.get_affine_transform(r['center'], r['scale'], r['rotation'], IMG), atol=1e-4)  # f32 control points
        # the box centre lands on the centre of the crop
        np.testing.assert_allclose(a @ np.array([r['center'][0], r['center'][1], 1.0]), IMG / 2, atol=1e-3)
        np.testing.assert_allclose(V._invert_affine(trans).reshape(2, 3), P.invert_affine(trans), atol=0)


def test_warp_oracle_properties():
    r = P.synth_samples(2, seed=3)[0]
    img = r['img']
    ident = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float32)
    assert np.array_equal(P.warp_affine_u8(img, ident, 100, 90), img[:90, :100])
    shift = np.array([[1, 0, 7], [0, 1, -3]], dtype=np.float32)          # integer shift: exact copy + zero border
    w = P.warp_affine_u8(img, shift, 64, 48)
    assert np.array_equal(w[:, 7:], img[3:51, :57]) and not w[:, :7].any()
    half = np.array([[1, 0, 0.5], [0, 1, 0]], dtype=np.float32)           # half-pixel shift: rounded mean of neighbours
    w = P.warp_affine_u8(img, half, 50, 20)
    want = (img[:20, :49].astype(np.int64) + img[:20, 1:50] + 1) >> 1
    assert np.array_equal(w[:, 1:], want.astype(np.uint8))
