"""CPU restatement of the input side of the reference's top-down pipeline (TEST INFRASTRUCTURE ONLY).

Follows mmpose/core/post_processing/post_transforms.py (`get_warp_matrix` :291-320, `warp_affine_joints`
:323-342, `get_affine_transform` :196-252 with its three-point construction), mmpose/datasets/pipelines/
top_down_transform.py (`TopDownAffine` :312-363, `TopDownGenerateTarget` `_msra_generate_target` :418-492 and
`_udp_generate_target` :541-661) and shared_transform.py (`ToTensor`, `NormalizeTensor` :21-64), sample by
sample with plain loops.

Pinning: tests/golden/pipeline.npz holds the outputs of the reference's own `get_warp_matrix`,
`warp_affine_joints`, `_udp_generate_target` and `_msra_generate_target` on seeded inputs.  The image warp is
cv2.warpAffine in the reference and cv2 is not installed here: `warp_affine_u8` restates OpenCV's fixed-point
bilinear scheme (imgwarp.cpp: AB_BITS 10, INTER_BITS 5, INTER_REMAP_COEF_BITS 15, BORDER_CONSTANT 0) from its
published source; parity with cv2's output bytes is unpinned.  `get_affine_transform` solves the three point
pairs cv2.getAffineTransform would be given (unpinned for the same reason; the ViTPose configs use the UDP
matrix, which is pinned).
"""
import math

import numpy as np


def get_warp_matrix(theta, size_input, size_dst, size_target):
    th = np.deg2rad(theta)
    m = np.zeros((2, 3), dtype=np.float32)
    sx, sy = size_dst[0] / size_target[0], size_dst[1] / size_target[1]
    c, s = math.cos(th), math.sin(th)
    m[0] = [c * sx, -s * sx, sx * (-0.5 * size_input[0] * c + 0.5 * size_input[1] * s + 0.5 * size_target[0])]
    m[1] = [s * sy, c * sy, sy * (-0.5 * size_input[0] * s - 0.5 * size_input[1] * c + 0.5 * size_target[1])]
    return m


def warp_affine_joints(joints, mat):
    out = np.empty_like(np.asarray(joints))
    for i, (x, y) in enumerate(np.asarray(joints).reshape(-1, 2)):
        out.reshape(-1, 2)[i] = [mat[0, 0] * x + mat[0, 1] * y + mat[0, 2], mat[1, 0] * x + mat[1, 1] * y + mat[1, 2]]
    return out


def get_affine_transform(center, scale, rot, output_size):
    scale_tmp = np.asarray(scale, dtype=np.float64) * 200.0
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    r = np.pi * rot / 180
    sn, cs = np.sin(r), np.cos(r)
    src_dir = np.array([0. * cs - (src_w * -0.5) * sn, 0. * sn + (src_w * -0.5) * cs])
    dst_dir = np.array([0., dst_w * -0.5])
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0] = center
    src[1] = np.asarray(center) + src_dir
    d = src[0] - src[1]
    src[2] = src[1] + np.array([-d[1], d[0]], dtype=np.float32)
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    d = dst[0] - dst[1]
    dst[2] = dst[1] + np.array([-d[1], d[0]], dtype=np.float32)
    # the 2x3 matrix that maps the three src points onto the three dst points
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def invert_affine(m):
    """cv::invertAffineTransform in double (what warpAffine does to the matrix it is given)."""
    m = np.asarray(m, dtype=np.float64)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = m[1, 1] * D, m[0, 0] * D, -m[0, 1] * D, -m[1, 0] * D
    b1 = -A11 * m[0, 2] - A12 * m[1, 2]
    b2 = -A21 * m[0, 2] - A22 * m[1, 2]
    return np.array([[A11, A12, b1], [A21, A22, b2]], dtype=np.float64)


def warp_affine_u8(img, trans, out_w, out_h):
    """cv2.warpAffine(img, trans, (out_w, out_h), flags=cv2.INTER_LINEAR) for an [H, W, 3] uint8 image."""
    H, W = img.shape[:2]
    m = invert_affine(trans).reshape(-1)
    out = np.zeros((out_h, out_w, 3), dtype=np.uint8)
    src = img.astype(np.int64)
    xs = np.arange(out_w)
    adelta = np.rint(m[0] * xs * 1024).astype(np.int64)
    bdelta = np.rint(m[3] * xs * 1024).astype(np.int64)
    for y in range(out_h):
        X0 = int(np.rint((m[1] * y + m[2]) * 1024)) + 16
        Y0 = int(np.rint((m[4] * y + m[5]) * 1024)) + 16
        X, Y = (X0 + adelta) >> 5, (Y0 + bdelta) >> 5
        sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
        ax, ay = X & 31, Y & 31
        acc = np.zeros((out_w, 3), dtype=np.int64)
        for dy, dx, w in ((0, 0, (32 - ax) * (32 - ay) * 32), (0, 1, ax * (32 - ay) * 32),
                          (1, 0, (32 - ax) * ay * 32), (1, 1, ax * ay * 32)):
            px, py = sx + dx, sy + dy
            ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
            vals = src[np.clip(py, 0, H - 1), np.clip(px, 0, W - 1)]
            acc += np.where(ok[:, None], vals * w[:, None], 0)
        out[y] = ((acc + 16384) >> 15).astype(np.uint8)
    return out


def to_tensor_normalize(img_u8, mean, std):
    t = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    return (t - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]


def generate_target(joints_3d, visible, image_size, heatmap_size, sigma=2, encoding='UDP', unbiased=False,
                    joint_weights=None):
    """One sample.  joints_3d [K, 3], visible [K] (= joints_3d_visible[:, 0]) -> target [K, H, W], weight [K, 1]."""
    K = joints_3d.shape[0]
    W, H = heatmap_size
    image_size = np.asarray(image_size, dtype=np.float64)
    target = np.zeros((K, H, W), dtype=np.float32)
    weight = np.asarray(visible, dtype=np.float32).reshape(K, 1).copy()
    tmp = sigma * 3
    size = 2 * tmp + 1
    for k in range(K):
        if encoding == 'UDP':
            fs = (image_size - 1.0) / (np.array([W, H]) - 1.0)
        else:
            fs = image_size / np.array([W, H])
        mxa, mya = joints_3d[k, 0] / fs[0], joints_3d[k, 1] / fs[1]
        if encoding == 'MSRA' and unbiased:
            if mxa - tmp >= W or mya - tmp >= H or mxa + tmp + 1 < 0 or mya + tmp + 1 < 0:
                weight[k] = 0
            if weight[k] > 0.5:
                x = np.arange(W, dtype=np.float32)
                y = np.arange(H, dtype=np.float32)[:, None]
                target[k] = np.exp(-((x - np.float32(mxa)) ** 2 + (y - np.float32(mya)) ** 2) / (2 * sigma ** 2))
            continue
        mux, muy = int(mxa + 0.5), int(mya + 0.5)
        ul = (int(mux - tmp), int(muy - tmp))
        br = (int(mux + tmp + 1), int(muy + tmp + 1))
        if ul[0] >= W or ul[1] >= H or br[0] < 0 or br[1] < 0:
            weight[k] = 0
            continue
        if weight[k] <= 0.5:
            continue
        x0 = y0 = size // 2
        if encoding == 'UDP':
            x0, y0 = x0 + (mxa - mux), y0 + (mya - muy)
        for y in range(max(0, ul[1]), min(br[1], H)):
            for x in range(max(0, ul[0]), min(br[0], W)):
                target[k, y, x] = math.exp(-(((x - ul[0]) - x0) ** 2 + ((y - ul[1]) - y0) ** 2) / (2.0 * sigma ** 2))
    if joint_weights is not None:
        weight = weight * np.asarray(joint_weights, dtype=np.float32).reshape(K, 1)
    return target, weight


def synth_samples(B, K=17, seed=0, image_size=(192, 256)):
    """Seeded 'dataset samples': uint8 RGB images of different sizes, a person box (center, scale), a rotation,
    keypoints around the box and visibility flags (some joints far outside, some invisible)."""
    g = np.random.default_rng(seed)
    out = []
    for b in range(B):
        H, W = int(g.integers(120, 400)), int(g.integers(120, 500))
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
        img = (img.astype(np.int64) + g.integers(0, 40, img.shape)).clip(0, 255).astype(np.uint8)
        c = np.array([g.uniform(0.3, 0.7) * W, g.uniform(0.3, 0.7) * H], dtype=np.float32)
        bw = g.uniform(0.3, 0.9) * W
        s = np.array([bw, bw * image_size[1] / image_size[0]], dtype=np.float32) / 200.0 * 1.25
        r = float(g.uniform(-40, 40)) if b % 3 else 0.0
        j = np.zeros((K, 3), dtype=np.float32)
        j[:, 0] = c[0] + g.uniform(-0.6, 0.6, K) * s[0] * 200
        j[:, 1] = c[1] + g.uniform(-0.6, 0.6, K) * s[1] * 200
        j[K - 1, :2] = [-5000., 9000.]                  # far outside every map
        v = (g.random(K) > 0.2).astype(np.float32)
        out.append(dict(img=img, center=c, scale=s, rotation=r, joints_3d=j, visible=v))
    return out
