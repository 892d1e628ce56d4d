"""ViTPose+ on the HIP path: the per-image expert routing of the GEMM kernels (through the C ABI) against
plain torch, the routed PoseEngine against the CPU oracle, and the TopDownMoE detector against the golden
vectors the reference's own vit_moe.py / top_down_moe.py produced (tests/golden/make_golden.py: vitpose_plus).

Tolerances: integer-valued data makes the kernel checks exact (f32 sums of small-integer products are exact);
the model-level checks are relative, bf16 operands against the fp32 oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import vitpose_oracle as O

pytestmark = pytest.mark.gpu

MOE = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=192, depth=3, num_heads=6, mlp_ratio=4,
           qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0, num_expert=3, part_features=48)


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from vitpose_amd import _lib
    _lib.lib()
    return torch.device('cuda:0')


def run(build):
    from vitpose_amd.plan import Plan
    p = Plan()
    build(p)
    p.run()
    torch.cuda.synchronize()


def ints(shape, dev, lo=-3, hi=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float().to(dev)


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('E,P,Hd,nexp', [(192, 48, 256, 3), (384, 96, 128, 2), (768, 192, 192, 6)])
def test_routed_gemm_exact(dev, E, P, Hd, nexp):
    """Forward (k-contiguous Y: routed output columns + bias) and data gradient (k-major Y: routed reduction rows)
    with the expert matrices scattered behind the shared one, as they lie in the flat parameter buffer."""
    from vitpose_amd import _lib
    B, T, Ns = 5, 192, E - P
    M = B * T
    expert = torch.tensor([(3 * b + 1) % nexp for b in range(B)], dtype=torch.int32, device=dev)
    gap = 64                                     # elements between the tensors
    flat = torch.zeros(Ns * Hd + nexp * (P * Hd + gap) + gap, device=dev)
    Wsh = ints((Ns, Hd), dev, seed=1)
    flat[:Ns * Hd] = Wsh.flatten()
    We, off = [], []
    o = Ns * Hd + gap
    for e in range(nexp):
        w = ints((P, Hd), dev, seed=10 + e)
        flat[o:o + P * Hd] = w.flatten()
        We.append(w), off.append(o)
        o += P * Hd + gap
    delta = torch.tensor([x - Ns * Hd for x in off], dtype=torch.int64, device=dev)
    bflat = torch.zeros(Ns + nexp * (P + 16) + 16, device=dev)
    bsh = ints((Ns,), dev, seed=2)
    bflat[:Ns] = bsh
    be, boff = [], []
    o = Ns + 16
    for e in range(nexp):
        b_ = ints((P,), dev, seed=20 + e)
        bflat[o:o + P] = b_
        be.append(b_), boff.append(o)
        o += P + 16
    bdelta = torch.tensor([x - Ns for x in boff], dtype=torch.int64, device=dev)
    half = flat.bfloat16()
    X = ints((M, Hd), dev, seed=3)
    resid = ints((M, E), dev, seed=4)
    out = torch.full((M, E), float('nan'), device=dev)
    run(lambda p: p.gemm(X.bfloat16(), half, out, M, E, Hd, ldx=Hd, ldy=Hd, ldo=E, epilogue=_lib.VP_EPI_RESID_F32,
                         bias=bflat, resid=resid, route=(expert, delta, bdelta, Ns)))
    ref = torch.empty(M, E, device=dev)
    for b in range(B):
        e = int(expert[b])
        W = torch.cat([Wsh, We[e]])
        ref[b * T:(b + 1) * T] = X[b * T:(b + 1) * T] @ W.t() + torch.cat([bsh, be[e]]) + resid[b * T:(b + 1) * T]
    assert torch.equal(out, ref)
    # data gradient: dX[i][k] = sum_j dY[i][j] W_eff(image of i)[j][k]
    dY = ints((M, E), dev, seed=5)
    dx = torch.full((M, Hd), float('nan'), device=dev, dtype=torch.bfloat16)
    ones = torch.zeros(M, Hd, device=dev, dtype=torch.bfloat16)     # gelu'(0) = 0.5 (to the 8e-6 of the epilogue's polynomial)
    run(lambda p: p.gemm(dY.bfloat16(), half, dx, M, Hd, E, ldx=E, ldy=Hd, ldo=Hd, y_kmajor=True,
                         epilogue=_lib.VP_EPI_DGELU, aux=ones, ldaux=Hd, route=(expert, delta, None, Ns)))
    ref = torch.empty(M, Hd, device=dev)
    for b in range(B):
        W = torch.cat([Wsh, We[int(expert[b])]])
        ref[b * T:(b + 1) * T] = dY[b * T:(b + 1) * T] @ W
    torch.testing.assert_close(dx.float(), 0.5 * ref, rtol=2.0 ** -7, atol=0.26)   # bf16 rounding of integer halves


def test_grouped_wgrad_kmap_exact(dev):
    """Expert weight gradients: every expert reduces over the K tiles of its own images only (an expert without
    images gets zeros), next to an ordinary problem in the same launch; row sums = the bias gradients.  70 images:
    the image masks span two 64-bit words."""
    E, P, Hd, nexp, B, T = 192, 48, 128, 4, 70, 192
    Ns, M = E - P, B * T
    src = [(2, 0, 2, 1, 0, 2, 2)[b % 7] if b != 64 else 1 for b in range(B)]     # expert 3 has no image
    dxb = ints((M, E), dev, seed=1).bfloat16()
    h = ints((M, Hd), dev, seed=2).bfloat16()
    words = np.zeros((nexp, 2), np.uint64)
    for b, e in enumerate(src):
        words[e, b // 64] |= np.uint64(1) << np.uint64(b % 64)
    mask = torch.from_numpy(words.view(np.int64)).to(dev)
    g_sh = torch.full((Ns, Hd), float('nan'), device=dev)
    b_sh = torch.full((Ns,), float('nan'), device=dev)
    g_e = [torch.full((P, Hd), float('nan'), device=dev) for _ in range(nexp)]
    b_e = [torch.full((P,), float('nan'), device=dev) for _ in range(nexp)]
    xe = dxb.view(-1)[Ns:]
    probs = [(dxb, h, g_sh, Ns, Hd, E, Hd, b_sh)] + \
            [(xe, h, g_e[e], P, Hd, E, Hd, b_e[e], mask[e], T // 64) for e in range(nexp)]
    run(lambda p: p.wgrad_group(probs, M))
    d, hf = dxb.float(), h.float()
    assert torch.equal(g_sh, d[:, :Ns].t() @ hf)
    assert torch.equal(b_sh, d[:, :Ns].sum(0))
    for e in range(nexp):
        rows = torch.cat([torch.arange(b * T, (b + 1) * T) for b, s in enumerate(src) if s == e] or
                         [torch.zeros(0, dtype=torch.long)]).to(dev)
        assert torch.equal(g_e[e], d[rows][:, Ns:].t() @ hf[rows]), e
        assert torch.equal(b_e[e], d[rows][:, Ns:].sum(0)), e


def make_moe_engine(dev, n_heads=3):
    from vitpose_amd.pose_engine import PoseEngine
    hc = O.classic_head_cfg(MOE['embed_dim'])
    bb = O.synth_vit_weights(MOE, seed=90)
    heads = [O.synth_head_weights(hc, seed=91 + j) for j in range(n_heads)]
    eng = PoseEngine(MOE, hc, dev, assoc_heads=[(hc, 1.0, True)] * (n_heads - 1))
    sd = {'backbone.' + k: v for k, v in bb.items()}
    sd.update({'keypoint_head.' + k: v for k, v in heads[0].items()})
    for j in range(1, n_heads):
        sd.update({f'associate_keypoint_heads.{j - 1}.' + k: v for k, v in heads[j].items()})
    eng.load_state_dict(sd)
    return eng, bb, heads, hc


def test_moe_engine_vs_oracle(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'vitpose_plus.npz'))
    src = [int(s) for s in g['sources']]
    eng, bb, heads, hc = make_moe_engine(dev)
    img, target, tw = O.synth_batch(len(src), seed=95)
    losses, heats = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False,
                                         dataset_source=src)
    torch.cuda.synchronize()
    got = torch.cat([l.cpu() for l in losses])
    # the reference detector's own losses
    np.testing.assert_allclose(got.numpy(), g['loss'], rtol=3e-2)
    # gradients against the reference's (sub-sampled as in the fixture)
    worst = 0.0
    for k in g.files:
        if not (k.startswith('g.') or k.startswith('gs.')):
            continue
        name = k.split('.', 1)[1]
        grad = eng.store.gview(name).cpu()
        if k.startswith('gs.'):
            grad = grad.flatten(1)[::4, ::4]
        r = relerr(grad, g[k])
        worst = max(worst, r)
        assert r < 0.12, (k, r)
    # an expert that saw no image of its dataset keeps an exactly zero gradient
    eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False, dataset_source=[0, 0, 1, 0, 1, 0])
    torch.cuda.synchronize()
    for i in range(MOE['depth']):
        assert float(eng.store.gview(f'backbone.blocks.{i}.mlp.experts.2.weight').abs().max()) == 0.0
        assert float(eng.store.gview(f'backbone.blocks.{i}.mlp.experts.2.bias').abs().max()) == 0.0
        assert float(eng.store.gview(f'backbone.blocks.{i}.mlp.experts.1.weight').abs().max()) > 0.0
    # eval: routed inference, bf16 and fp32-accurate, against the oracle's routed forward
    with torch.no_grad():
        srct = torch.tensor(src)
        feat = O.vit_forward(bb, img, MOE, dataset_source=srct)
        hd_eval = {k: v.clone() for k, v in O.synth_head_weights(hc, seed=91).items()}
        eng.load_state_dict({**{'backbone.' + k: v for k, v in bb.items()},
                             **{'keypoint_head.' + k: v for k, v in hd_eval.items()}}, strict=False)
        ref = O.head_forward(hd_eval, feat, hc)
        out = eng.infer(img.to(dev), dataset_source=src).cpu()
        assert relerr(out, ref) < 2e-2
        out32 = eng.infer(img.to(dev), exact=True, dataset_source=src).cpu()
        assert float((out32 - ref).abs().max()) < 1e-3, float((out32 - ref).abs().max())
        # routing matters: another assignment gives other heat maps
        other = eng.infer(img.to(dev), dataset_source=[1] * len(src)).cpu()
        assert relerr(other, ref) > 1e-2


def test_topdown_moe_detector_vs_reference_golden(dev, golden_dir):
    """The drop-in classes built from a ViTPose+-style config dict: forward_train losses / accuracies and the
    gradients autograd publishes, against the reference detector's."""
    from vitpose_amd.builder import build_posenet
    g = np.load(os.path.join(golden_dir, 'vitpose_plus.npz'))
    src = [int(s) for s in g['sources']]
    hc = O.classic_head_cfg(MOE['embed_dim'])

    def hcfg():
        return dict(type='TopdownHeatmapSimpleHead', in_channels=hc['in_channels'], out_channels=hc['out_channels'],
                    num_deconv_layers=hc['num_deconv_layers'], num_deconv_filters=hc['num_deconv_filters'],
                    num_deconv_kernels=hc['num_deconv_kernels'], extra=hc['extra'],
                    loss_keypoint=dict(type='JointsMSELoss', use_target_weight=True))
    model = build_posenet(dict(
        type='TopDownMoE', pretrained=None,
        backbone=dict(type='ViTMoE', img_size=(256, 192), patch_size=16, embed_dim=MOE['embed_dim'], depth=MOE['depth'],
                      num_heads=MOE['num_heads'], ratio=1, use_checkpoint=False, mlp_ratio=4, qkv_bias=True,
                      drop_path_rate=0.0, num_expert=MOE['num_expert'], part_features=MOE['part_features']),
        keypoint_head=hcfg(), associate_keypoint_head=[hcfg(), hcfg()], train_cfg=dict(),
        test_cfg=dict(flip_test=False, post_process='default', shift_heatmap=False, modulate_kernel=11)))
    sd = {'backbone.' + k: v for k, v in O.synth_vit_weights(MOE, seed=90).items()}
    for pre, seed in (('keypoint_head.', 91), ('associate_keypoint_heads.0.', 92), ('associate_keypoint_heads.1.', 93)):
        sd.update({pre + k: v for k, v in O.synth_head_weights(hc, seed=seed).items()})
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all('num_batches_tracked' in m for m in missing), (missing, unexpected)
    model = model.to(dev).train()
    assert model._engine is not None and len(model._engine.heads) == 3
    img, target, tw = O.synth_batch(len(src), seed=95)
    metas = [dict(dataset_idx=s) for s in src]
    losses = model.forward_train(img.to(dev), target.to(dev), tw.to(dev), metas)
    assert set(losses) == {'main_stream_loss', 'main_stream_acc', '1_loss', '1_acc', '2_loss', '2_acc'}
    got = torch.stack([losses['main_stream_loss'], losses['1_loss'], losses['2_loss']]).detach().cpu().numpy()
    np.testing.assert_allclose(got, g['loss'], rtol=3e-2)
    np.testing.assert_allclose([losses['main_stream_acc'], losses['1_acc'], losses['2_acc']], g['acc'], atol=0.1)
    loss, log_vars = model._parse_losses(losses)
    assert log_vars['loss'] == pytest.approx(float(g['loss'].sum()), rel=3e-2)
    loss.backward()
    named = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('g.'):
            assert relerr(named[k[2:]].grad, g[k]) < 0.12, k
        elif k.startswith('gs.'):
            assert relerr(named[k[3:]].grad.flatten(1)[::4, ::4], g[k]) < 0.12, k
    # eval through forward_test: the routed backbone + main head, decoded on the device
    model.eval()
    metas = [dict(dataset_idx=s, center=np.array([96., 128.]), scale=np.array([0.96, 1.28]), image_file='x',
                  bbox_id=i, flip_pairs=[]) for i, s in enumerate(src)]
    res = model.forward_test(img.to(dev), metas, return_heatmap=True)
    assert relerr(res['output_heatmap'][:, ::4], g['heat_main_sub']) < 2e-2
