"""Input side of the top-down training / test pipeline on the MI355X.

The reference runs, per sample on CPU workers: TopDownAffine (cv2.warpAffine of the person box,
datasets/pipelines/top_down_transform.py:312-363), ToTensor + NormalizeTensor (shared_transform.py:21-64) and
TopDownGenerateTarget (top_down_transform.py:366-740).  Here the random augmentations stay on the host (they
only change a box centre, a scale, an angle and a few joint coordinates), and the two heavy stages — the image
warp + normalisation and the heat-map painting — run for the whole batch in two HIP kernels; images, targets
and weights are born on the device, where the model wants them.
"""
import ctypes
import math
from typing import List, Sequence

import numpy as np
import torch

from . import _lib


def get_warp_matrix(theta, size_input, size_dst, size_target):
    """The unbiased-data-processing warp (core/post_processing/post_transforms.py:291-320): a source point p goes to
    diag(s) (R(theta) (p - size_input / 2) + size_target / 2) with s = size_dst / size_target, i.e. the 2 x 3 matrix
    diag(s) [R | size_target / 2 - R size_input / 2].  The terms are summed in the order and in the precision the
    reference's scalar expressions have (the arguments' own dtype -- f32 in the pipeline -- with cos / sin rounded to
    it), so the matrix, and with it the integer warp of the image, is the reference's bit for bit."""
    theta = np.deg2rad(theta)
    c, s = math.cos(theta), math.sin(theta)
    size_input, size_dst, size_target = np.asarray(size_input), np.asarray(size_dst), np.asarray(size_target)
    half_in, half_target = 0.5 * size_input, 0.5 * size_target
    scale = size_dst / size_target
    shift = ((-half_in[0]) * np.array([c, s], dtype=half_in.dtype) + half_in[1] * np.array([s, -c], dtype=half_in.dtype)
             + half_target)
    matrix = np.empty((2, 3), dtype=np.float32)
    matrix[:, :2] = np.array([[c, -s], [s, c]], dtype=scale.dtype) * scale[:, None]
    matrix[:, 2] = scale * shift
    return matrix


def warp_affine_joints(joints, mat):
    """post_transforms.py:323-342: every (x, y) pair of `joints` (any leading shape) through the 2 x 3 matrix."""
    joints = np.asarray(joints)
    mat = np.asarray(mat)
    return joints @ mat[:, :2].T + mat[:, 2]


def get_affine_transform(center, scale, rot, output_size):
    """post_transforms.py:196-252 (shift 0, inv False); the three point pairs are solved directly
    (the reference hands them to cv2.getAffineTransform)."""
    scale_tmp = np.asarray(scale) * 200.0
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    src_dir = np.array([-(src_w * -0.5) * sn, (src_w * -0.5) * cs])
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center) + src_dir
    src[2, :] = src[1] + np.array([-(src[0] - src[1])[1], (src[0] - src[1])[0]], dtype=np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0., dst_w * -0.5])
    dst[2, :] = dst[1] + np.array([-(dst[0] - dst[1])[1], (dst[0] - dst[1])[0]], dtype=np.float32)
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def _invert_affine(m):
    """What cv2.warpAffine does to `trans` before sampling (cv::invertAffineTransform, double)."""
    m = np.asarray(m, dtype=np.float64)
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22, A12, A21 = m[1, 1] * D, m[0, 0] * D, -m[0, 1] * D, -m[1, 0] * D
    return np.array([A11, A12, -A11 * m[0, 2] - A12 * m[1, 2], A21, A22, -A21 * m[0, 2] - A22 * m[1, 2]])


_ENCODINGS = {('UDP', False): 0, ('MSRA', False): 1, ('MSRA', True): 2}


class TopDownBatchTransform:
    """[TopDownAffine, ToTensor, NormalizeTensor, TopDownGenerateTarget] of a reference pipeline for a batch.

    Arguments carry the reference's names: `use_udp` (TopDownAffine), `mean` / `std` (NormalizeTensor),
    `sigma`, `encoding`, `unbiased_encoding`, `target_type` (TopDownGenerateTarget); `image_size` (w, h),
    `heatmap_size` (w, h), `num_joints`, `joint_weights`, `use_different_joint_weights` come from `data_cfg` /
    `ann_info`.  Each sample is a `results` dict as it leaves the random augmentations: 'img' ([H, W, 3] uint8
    RGB), 'center', 'scale', 'rotation', 'joints_3d', 'joints_3d_visible'.
    """

    def __init__(self, image_size, heatmap_size, num_joints, use_udp=False, mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), sigma=2, encoding='MSRA', unbiased_encoding=False,
                 target_type='GaussianHeatmap', joint_weights=None, use_different_joint_weights=False,
                 device='cuda'):
        if (encoding, bool(unbiased_encoding)) not in _ENCODINGS or target_type.lower() != 'gaussianheatmap':
            raise NotImplementedError("device targets cover encoding 'UDP' / 'MSRA' with 'GaussianHeatmap'")
        if isinstance(sigma, (list, tuple)):
            raise NotImplementedError('multi-sigma targets')
        self.image_size = np.asarray(image_size)
        self.heatmap_size = np.asarray(heatmap_size)
        self.K, self.use_udp = int(num_joints), bool(use_udp)
        self.mean, self.std, self.sigma = tuple(mean), tuple(std), float(sigma)
        self.mode = _ENCODINGS[(encoding, bool(unbiased_encoding))]
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('TopDownBatchTransform runs on the GPU only; there is no CPU path')
        self.jw = None
        if use_different_joint_weights:
            self.jw = torch.as_tensor(np.asarray(joint_weights, dtype=np.float32).reshape(-1)).to(self.device)

    @classmethod
    def from_cfg(cls, pipeline: Sequence[dict], data_cfg: dict, **kw):
        """Read the parameters out of a reference `train_pipeline` / `val_pipeline` list and its `data_cfg`."""
        args = dict(image_size=data_cfg['image_size'], heatmap_size=data_cfg['heatmap_size'],
                    num_joints=data_cfg['num_joints'])
        for t in pipeline:
            if t['type'] == 'TopDownAffine':
                args['use_udp'] = t.get('use_udp', False)
            elif t['type'] == 'NormalizeTensor':
                args.update(mean=t['mean'], std=t['std'])
            elif t['type'] == 'TopDownGenerateTarget':
                args.update(sigma=t.get('sigma', 2), encoding=t.get('encoding', 'MSRA'),
                            unbiased_encoding=t.get('unbiased_encoding', False),
                            target_type=t.get('target_type', 'GaussianHeatmap'))
        args.update(kw)
        return cls(**args)

    def matrix(self, results):
        c, s, r = results['center'], results['scale'], results['rotation']
        if self.use_udp:
            return get_warp_matrix(r, c * 2.0, self.image_size - 1.0, s * 200.0)
        return get_affine_transform(c, s, r, self.image_size)

    def __call__(self, batch: List[dict], with_target: bool = True):
        L = _lib.lib()
        B, (Wo, Ho) = len(batch), (int(self.image_size[0]), int(self.image_size[1]))
        offs, hw, minv, chunks, joints = [], [], [], [], []
        pos = 0
        for res in batch:
            img = np.ascontiguousarray(res['img'])
            if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
                raise ValueError('images must be [H, W, 3] uint8')
            trans = self.matrix(res)
            j = np.array(res['joints_3d'], dtype=np.float32, copy=True)
            if self.use_udp:
                j[:, 0:2] = warp_affine_joints(j[:, 0:2].copy(), trans)
            else:
                vis = np.asarray(res['joints_3d_visible'])[:, 0] > 0.0
                j[vis, 0:2] = (np.asarray(trans) @ np.concatenate([j[vis, 0:2], np.ones((int(vis.sum()), 1))], 1).T).T
            res['joints_3d'] = j
            joints.append(j)
            offs.append(pos)
            hw.append(img.shape[:2])
            minv.append(_invert_affine(trans))
            chunks.append(img.reshape(-1))
            pos += (img.size + 15) // 16 * 16
        packed = np.zeros(pos, dtype=np.uint8)
        for o, ch in zip(offs, chunks):
            packed[o:o + ch.size] = ch
        dev = self.device
        d_img = torch.from_numpy(packed).to(dev)
        d_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        d_hw = torch.tensor(np.asarray(hw, dtype=np.int32), device=dev)
        d_m = torch.tensor(np.asarray(minv, dtype=np.float64), device=dev)
        out = torch.empty(B, 3, Ho, Wo, dtype=torch.float32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        mean, std = (ctypes.c_float * 3)(*self.mean), (ctypes.c_float * 3)(*self.std)
        rc = L.vp_warp_affine_normalize(d_img.data_ptr(), d_off.data_ptr(), d_hw.data_ptr(), d_m.data_ptr(),
                                        out.data_ptr(), B, Ho, Wo, mean, std, st)
        if rc != 0:
            raise RuntimeError(f'vp_warp_affine_normalize failed with status {rc}')
        result = dict(img=out, joints_3d=np.stack(joints))
        if with_target:
            W, H = int(self.heatmap_size[0]), int(self.heatmap_size[1])
            d_j = torch.from_numpy(np.stack(joints)).to(dev)
            d_v = torch.from_numpy(np.stack([np.asarray(r['joints_3d_visible'], dtype=np.float32)[:, 0]
                                             for r in batch])).to(dev)
            target = torch.empty(B, self.K, H, W, dtype=torch.float32, device=dev)
            tw = torch.empty(B, self.K, 1, dtype=torch.float32, device=dev)
            rc = L.vp_generate_target(d_j.data_ptr(), d_v.data_ptr(), self.jw.data_ptr() if self.jw is not None else None,
                                      target.data_ptr(), tw.data_ptr(), B, self.K, H, W, Wo, Ho, self.sigma, self.mode, st)
            if rc != 0:
                raise RuntimeError(f'vp_generate_target failed with status {rc}')
            result.update(target=target, target_weight=tw)
        return result
