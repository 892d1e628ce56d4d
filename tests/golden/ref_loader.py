"""Load the reference's hot-path source files *in this build container only*.

The reference package cannot be imported as a whole (mmcv / timm / cv2 are not
installed), so the individual source files are executed from where they lie
under /root/reference with a handful of import stubs for the third-party
layer builders they call.  All arithmetic then comes from torch CPU ops driven
by the reference's own code, which is what the golden vectors pin.

This module is test-fixture tooling: it is only used by make_golden.py.  It
never runs on the GPU box (the reference tree does not exist there) and
nothing in the product imports it.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('VITPOSE_REFERENCE_ROOT', '/root/reference')


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave like a package
    sys.modules[name] = m
    return m


class _Registry:
    """Minimal registry with the mmcv `register_module()` / `build()` shape."""

    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop('type')](**cfg)


def _install_stubs():
    if 'mmpose.models.builder' in sys.modules and getattr(
            sys.modules['mmpose.models.builder'], '_vp_stub', False):
        return
    # timm.models.layers: drop_path / to_2tuple / trunc_normal_
    _mod('timm')
    _mod('timm.models')
    tl = _mod('timm.models.layers')

    def drop_path(x, drop_prob=0., training=False):
        # timm 0.4.x semantics: per-sample Bernoulli(keep) mask, scaled 1/keep
        if drop_prob == 0. or not training:
            return x
        keep_prob = 1 - drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = keep_prob + torch.rand(shape, dtype=x.dtype, device=x.device)
        random_tensor.floor_()
        return x.div(keep_prob) * random_tensor

    tl.drop_path = drop_path
    tl.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tl.trunc_normal_ = torch.nn.init.trunc_normal_

    # mmcv_custom.checkpoint.load_checkpoint: never called with pretrained=None
    _mod('mmcv_custom')
    ck = _mod('mmcv_custom.checkpoint')

    def load_checkpoint(*a, **k):
        raise RuntimeError('checkpoint loading is not available in the stub loader')
    ck.load_checkpoint = load_checkpoint

    # cv2 is not installed.  The decode fixtures call the reference's UDP branch with kernel size 1, for
    # which cv2.GaussianBlur is the identity (its 1-tap kernel is [1.0]); any other size is refused, so no
    # fixture ever contains a blur that did not come from OpenCV.
    cv2 = _mod('cv2')

    def GaussianBlur(src, ksize, sigma, dst=None):
        if tuple(ksize) != (1, 1):
            raise RuntimeError('cv2 is not available: only the identity blur (ksize 1) can be evaluated')
        if dst is not None and dst is not src:
            dst[...] = src
        return src if dst is None else dst
    cv2.GaussianBlur = GaussianBlur

    # mmcv.cnn layer builders and initialisers
    _mod('mmcv')
    cnn = _mod('mmcv.cnn')

    def build_conv_layer(cfg, *args, **kwargs):
        assert cfg is None or cfg.get('type', 'Conv2d') == 'Conv2d'
        return nn.Conv2d(*args, **kwargs)

    def build_norm_layer(cfg, num_features, postfix=''):
        assert cfg['type'] == 'BN'
        return 'bn' + str(postfix), nn.BatchNorm2d(num_features)

    def build_upsample_layer(cfg, *args, **kwargs):
        assert cfg['type'] == 'deconv'
        return nn.ConvTranspose2d(*args, **kwargs)

    def normal_init(module, mean=0, std=1, bias=0):
        if hasattr(module, 'weight') and module.weight is not None:
            nn.init.normal_(module.weight, mean, std)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    def constant_init(module, val, bias=0):
        if hasattr(module, 'weight') and module.weight is not None:
            nn.init.constant_(module.weight, val)
        if hasattr(module, 'bias') and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    cnn.build_conv_layer = build_conv_layer
    cnn.build_norm_layer = build_norm_layer
    cnn.build_upsample_layer = build_upsample_layer
    cnn.normal_init = normal_init
    cnn.constant_init = constant_init

    # mmpose package skeleton + builder with one shared registry
    for n in ('mmpose', 'mmpose.models', 'mmpose.models.backbones', 'mmpose.models.heads',
              'mmpose.models.losses', 'mmpose.models.utils', 'mmpose.core'):
        _mod(n)
    b = _mod('mmpose.models.builder')
    b._vp_stub = True
    reg = _Registry('models')
    b.MODELS = b.BACKBONES = b.NECKS = b.HEADS = b.LOSSES = b.POSENETS = reg
    b.build_backbone = b.build_head = b.build_loss = b.build_neck = b.build_posenet = reg.build


def _load(modname, relpath):
    if modname in sys.modules and hasattr(sys.modules[modname], '__file__'):
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Return a namespace with the reference classes/functions on the hot path."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f'reference tree not found at {REF_ROOT}')
    _install_stubs()
    post = _load('mmpose.core.post_processing.post_transforms',
                 'mmpose/core/post_processing/post_transforms.py')
    pp = _mod('mmpose.core.post_processing')
    for k in ('flip_back', 'transform_preds', 'fliplr_joints', 'fliplr_regression',
              'get_affine_transform', 'affine_transform', 'rotate_point',
              'get_warp_matrix', 'warp_affine_joints'):
        if hasattr(post, k):
            setattr(pp, k, getattr(post, k))
    ev = _load('mmpose.core.evaluation.top_down_eval',
               'mmpose/core/evaluation/top_down_eval.py')
    evp = _mod('mmpose.core.evaluation')
    for k in ('pose_pck_accuracy', 'keypoints_from_heatmaps', 'keypoint_pck_accuracy',
              '_get_max_preds'):
        setattr(evp, k, getattr(ev, k))
    sys.modules['mmpose.core.evaluation.top_down_eval'] = ev
    ops = _load('mmpose.models.utils.ops', 'mmpose/models/utils/ops.py')
    _load('mmpose.models.backbones.base_backbone', 'mmpose/models/backbones/base_backbone.py')
    vit = _load('mmpose.models.backbones.vit', 'mmpose/models/backbones/vit.py')
    loss = _load('mmpose.models.losses.mse_loss', 'mmpose/models/losses/mse_loss.py')
    _load('mmpose.models.heads.topdown_heatmap_base_head',
          'mmpose/models/heads/topdown_heatmap_base_head.py')
    head = _load('mmpose.models.heads.topdown_heatmap_simple_head',
                 'mmpose/models/heads/topdown_heatmap_simple_head.py')
    # input-side transforms (numpy only, apart from the cv2 calls the fixtures never reach)
    db = _mod('mmpose.datasets')
    dbb = _mod('mmpose.datasets.builder')
    dbb.PIPELINES = _Registry('pipeline')
    db.builder = dbb
    tdt = _load('mmpose.datasets.pipelines.top_down_transform', 'mmpose/datasets/pipelines/top_down_transform.py')
    ns = types.SimpleNamespace(
        get_warp_matrix=post.get_warp_matrix, warp_affine_joints=post.warp_affine_joints,
        TopDownGenerateTarget=tdt.TopDownGenerateTarget,
        ViT=vit.ViT, TopdownHeatmapSimpleHead=head.TopdownHeatmapSimpleHead,
        JointsMSELoss=loss.JointsMSELoss, pose_pck_accuracy=ev.pose_pck_accuracy,
        flip_back=post.flip_back, resize=ops.resize, _get_max_preds=ev._get_max_preds,
        keypoints_from_heatmaps=ev.keypoints_from_heatmaps, _taylor=ev._taylor, post_dark_udp=ev.post_dark_udp,
        transform_preds=post.transform_preds)
    return ns


def load_reference_moe():
    """ViTPose+ classes: ViTMoE (vit_moe.py) and TopDownMoE (top_down_moe.py).  The detector file imports
    mmcv's image / visualisation helpers and decorators at module level; they are stubbed (the decorators as
    identities, the drawing helpers as functions that raise) because forward_train / forward_test never
    reach them."""
    ns = load_reference()
    vm = _load('mmpose.models.backbones.vit_moe', 'mmpose/models/backbones/vit_moe.py')

    def _ident_deco(*a, **k):
        return lambda f: f

    def _unavailable(*a, **k):
        raise RuntimeError('drawing helpers are not available in the stub loader')
    sys.modules['mmcv'].imread = _unavailable
    _mod('mmcv.image').imwrite = _unavailable
    _mod('mmcv.utils')
    _mod('mmcv.utils.misc').deprecated_api_warning = _ident_deco
    _mod('mmcv.visualization')
    _mod('mmcv.visualization.image').imshow = _unavailable
    if 'mmcv.runner' not in sys.modules:
        _mod('mmcv.runner')
    sys.modules['mmcv.runner'].auto_fp16 = _ident_deco
    core = sys.modules['mmpose.core']
    core.imshow_bboxes = core.imshow_keypoints = _unavailable
    _mod('mmpose.models.detectors')
    sys.modules['mmpose.models'].builder = sys.modules['mmpose.models.builder']
    _load('mmpose.models.detectors.base', 'mmpose/models/detectors/base.py')
    td = _load('mmpose.models.detectors.top_down_moe', 'mmpose/models/detectors/top_down_moe.py')
    ns.ViTMoE, ns.TopDownMoE = vm.ViTMoE, td.TopDownMoE
    return ns


# Layer-decay parameter groups: the constructor file subclasses an mmcv class, so
# only its pure function `get_num_layer_for_vit` is executed from source; the
# group assembly loop is re-driven with the same arithmetic by make_golden.py.
def load_layer_id_fn():
    _install_stubs()
    runner = _mod('mmcv.runner')

    class _DOC:
        pass
    runner.DefaultOptimizerConstructor = _DOC
    runner.OPTIMIZER_BUILDERS = _Registry('optimizer builder')
    runner.get_dist_info = lambda: (0, 1)
    m = _load('mmcv_custom.layer_decay_optimizer_constructor',
              'mmcv_custom/layer_decay_optimizer_constructor.py')
    return m
