"""CPU restatement of the reference's heat-map decoding (TEST INFRASTRUCTURE ONLY: imported by tests/ and
by tests/golden/make_golden.py for its input recipe, never by the product).

Follows mmpose/core/evaluation/top_down_eval.py (`_get_max_preds` :63-95, `_taylor` :298-332,
`post_dark_udp` :335-396, `_gaussian_blur` :399-438, `keypoints_from_heatmaps` :474-622) and
mmpose/core/post_processing/post_transforms.py (`transform_preds` :150-194, `flip_back` :54-95), written
map-by-map with explicit loops and scipy.ndimage for the blur, i.e. independently of the HIP kernels that
decode in the product (vitpose_amd/csrc/decode.hip).

Pinning: tests/golden/decode.npz holds the outputs of the reference's own functions for post_process None /
'default', `_taylor`, `transform_preds` and DARK-UDP with modulate kernel 1 (identity blur).  The blur itself
is cv2.GaussianBlur in the reference and cv2 is not installed here: the kernel coefficients follow OpenCV's
documented getGaussianKernel formula and are pinned only through the reference's unit-test known answers
(tests/test_evaluation/test_top_down_eval.py:29-89) — parity with OpenCV's exact f32 rounding is unpinned.
"""
import numpy as np
from scipy import ndimage


def synth_heatmaps(N, K, H, W, seed):
    g = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    hm = np.zeros((N, K, H, W), dtype=np.float32)
    for n in range(N):
        for k in range(K):
            # every fifth joint sits at / near the border to exercise the edge handling
            if k % 5 == 4:
                cx, cy = g.choice([0.3, 1.2, W - 1.4, W - 2.2]), g.choice([0.4, 1.1, H - 1.3, H - 2.4])
            else:
                cx, cy = g.uniform(3, W - 4), g.uniform(3, H - 4)
            amp = g.uniform(0.3, 1.0)
            hm[n, k] = amp * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * 2.0 ** 2)) + \
                0.02 * g.random((H, W)).astype(np.float32)
    return hm


def gaussian_kernel(ksize):
    """OpenCV getGaussianKernel(ksize, sigma <= 0): fixed tables up to 7, else the sampled Gaussian with
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8 (the formula quoted at top_down_eval.py:401-405)."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize in small:
        return np.array(small[ksize], dtype=np.float64)
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize) - (ksize - 1) / 2
    k = np.exp(-(x * x) / (2 * sigma * sigma))
    return k / k.sum()


def blur(img, ksize):
    """cv2.GaussianBlur(img, (ksize, ksize), 0): separable, BORDER_REFLECT_101 (scipy's 'mirror')."""
    k = gaussian_kernel(ksize)
    a = ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode='mirror')
    return ndimage.correlate1d(a, k, axis=0, mode='mirror').astype(np.float32)


def max_pred(hm):
    """_get_max_preds for one map: first arg-max as (x, y), value; (-1, -1) when the value is <= 0."""
    idx = int(np.argmax(hm))
    v = float(hm.reshape(-1)[idx])
    W = hm.shape[1]
    return (np.array([idx % W, idx // W], dtype=np.float32) if v > 0 else np.array([-1, -1], dtype=np.float32)), v


def dark_udp_one(hm, coord, ksize):
    """post_dark_udp for one map: blur, clip, log, Newton step on the edge-padded map."""
    H, W = hm.shape
    L = np.log(np.clip(blur(hm, ksize), 0.001, 50)).astype(np.float32)
    P = np.pad(L, 1, mode='edge')
    x, y = int(coord[0]) + 1, int(coord[1]) + 1
    i_, ix1, iy1, ix1y1 = P[y, x], P[y, x + 1], P[y + 1, x], P[y + 1, x + 1]
    ix1_y1_, ix1_, iy1_ = P[y - 1, x - 1], P[y, x - 1], P[y - 1, x]
    dx, dy = np.float32(0.5) * (ix1 - ix1_), np.float32(0.5) * (iy1 - iy1_)
    dxx, dyy = ix1 - 2 * i_ + ix1_, iy1 - 2 * i_ + iy1_
    dxy = np.float32(0.5) * (ix1y1 - ix1 - iy1 + i_ + i_ - ix1_ - iy1_ + ix1_y1_)
    Hm = np.array([[dxx, dxy], [dxy, dyy]], dtype=np.float64) + np.finfo(np.float32).eps * np.eye(2)
    step = np.linalg.inv(Hm) @ np.array([dx, dy], dtype=np.float64)
    return (coord.astype(np.float64) - step).astype(np.float32)


def taylor_one(L, coord):
    """_taylor for one (log) map."""
    H, W = L.shape
    px, py = int(coord[0]), int(coord[1])
    if 1 < px < W - 2 and 1 < py < H - 2:
        dx = 0.5 * (L[py, px + 1] - L[py, px - 1])
        dy = 0.5 * (L[py + 1, px] - L[py - 1, px])
        dxx = 0.25 * (L[py, px + 2] - 2 * L[py, px] + L[py, px - 2])
        dxy = 0.25 * (L[py + 1, px + 1] - L[py - 1, px + 1] - L[py + 1, px - 1] + L[py - 1, px - 1])
        dyy = 0.25 * (L[py + 2, px] - 2 * L[py, px] + L[py - 2, px])
        if dxx * dyy - dxy ** 2 != 0:
            step = np.linalg.inv(np.array([[dxx, dxy], [dxy, dyy]])) @ np.array([dx, dy])
            return (coord - step).astype(coord.dtype)
    return coord


def modulate_one(hm, ksize):
    """_gaussian_blur for one map: blur inside a zero border of (ksize - 1) / 2, restore the maximum."""
    b = (ksize - 1) // 2
    big = np.zeros((hm.shape[0] + 2 * b, hm.shape[1] + 2 * b), dtype=np.float32)
    big[b:-b, b:-b] = hm
    out = blur(big, ksize)[b:-b, b:-b]
    return out * (hm.max() / out.max())


def to_image(coord, center, scale, W, H, use_udp):
    """transform_preds for one point."""
    sx = scale[0] * 200.0 / ((W - 1.0) if use_udp else W)
    sy = scale[1] * 200.0 / ((H - 1.0) if use_udp else H)
    return np.array([coord[0] * sx + center[0] - scale[0] * 100.0, coord[1] * sy + center[1] - scale[1] * 100.0],
                    dtype=np.float32)


def keypoints_from_heatmaps(heatmaps, center, scale, post_process='default', kernel=11, use_udp=False,
                            image_space=True):
    """'GaussianHeatmap' targets; post_process None / 'default' / 'unbiased' / 'megvii', or UDP (DARK-UDP)."""
    N, K, H, W = heatmaps.shape
    preds = np.zeros((N, K, 2), dtype=np.float32)
    maxvals = np.zeros((N, K, 1), dtype=np.float32)
    for n in range(N):
        for k in range(K):
            hm = heatmaps[n, k].astype(np.float32)
            if post_process == 'megvii':
                hm = modulate_one(hm, kernel)
            c, v = max_pred(hm)
            if use_udp:
                c = dark_udp_one(hm, c, kernel)
            elif post_process == 'unbiased':
                c = taylor_one(np.log(np.maximum(modulate_one(hm, kernel), 1e-10)), c)
            elif post_process is not None:
                px, py = int(c[0]), int(c[1])
                if 1 < px < W - 1 and 1 < py < H - 1:
                    c = c + 0.25 * np.sign([hm[py, px + 1] - hm[py, px - 1], hm[py + 1, px] - hm[py - 1, px]])
                    if post_process == 'megvii':
                        c = c + 0.5
            preds[n, k] = to_image(c, center[n], scale[n], W, H, use_udp) if image_space else c
            maxvals[n, k] = v / 255.0 + 0.5 if post_process == 'megvii' else v
    return preds, maxvals


def combined_target_decode(heatmaps, center, scale, kernel=11, valid_radius_factor=0.0546875, image_space=True):
    """UDP decoding of 'CombinedTarget' maps (top_down_eval.py:571-585): three maps per joint (heat, x offset,
    y offset); the heat map is blurred with 2 * kernel + 1 taps, the offsets with `kernel` taps; the arg-max of the
    blurred heat map is moved by valid_radius_factor * H times the blurred offsets at that pixel.  The offsets are
    looked up the way the reference does it, in the FLATTENED offset maps of the whole batch at index
    x + y * W + W * H * j: a joint without a positive maximum has (x, y) = (-1, -1), which lands on pixel
    (H - 2, W - 1) of the previous joint's maps (and, for j = 0, wraps to the last joint: a negative numpy index)."""
    N, C, H, W = heatmaps.shape
    K = C // 3
    radius = np.float32(valid_radius_factor * H)
    hm = np.zeros((N, K, H, W), dtype=np.float32)
    ox, oy = np.zeros_like(hm), np.zeros_like(hm)
    for n in range(N):
        for k in range(K):
            hm[n, k] = blur(heatmaps[n, 3 * k].astype(np.float32), 2 * kernel + 1)
            ox[n, k] = blur(heatmaps[n, 3 * k + 1].astype(np.float32), kernel)
            oy[n, k] = blur(heatmaps[n, 3 * k + 2].astype(np.float32), kernel)
    ox, oy = ox.reshape(-1) * radius, oy.reshape(-1) * radius
    preds = np.zeros((N, K, 2), dtype=np.float32)
    maxvals = np.zeros((N, K, 1), dtype=np.float32)
    for n in range(N):
        for k in range(K):
            c, v = max_pred(hm[n, k])
            index = int(c[0] + c[1] * W + W * H * (n * K + k))
            c = c + np.array([ox[index], oy[index]], dtype=np.float32)
            preds[n, k] = to_image(c, center[n], scale[n], W, H, True) if image_space else c
            maxvals[n, k] = v
    return preds, maxvals


def flip_back_avg(heat, heat_flipped, flip_pairs, shift):
    """(heat + flip_back(heat_flipped)) * 0.5 with the optional one-pixel shift (top_down.py:181-189,
    topdown_heatmap_simple_head.py:252-262)."""
    fb = heat_flipped.copy()
    for a, b in flip_pairs:
        fb[:, a], fb[:, b] = heat_flipped[:, b], heat_flipped[:, a]
    fb = fb[..., ::-1].copy()
    if shift:
        fb[..., 1:] = fb[..., :-1].copy()
    return (heat + fb) * 0.5
