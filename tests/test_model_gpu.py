"""End-to-end parity of the HIP path (PoseEngine, through the C ABI) against
  (a) the golden vectors produced by the reference's own source files, and
  (b) the CPU oracle on the same seeded inputs.

Tolerances.  The path computes in bf16 (MFMA operands, bf16 activations between
kernels; f32 accumulation, f32 residual stream / LayerNorm / softmax statistics).
The north-star tolerance "heatmaps within 1e-3 max-abs of the reference CPU forward"
is asserted on the reference's own weight initialisation (trunc_normal 0.02 backbone,
std=0.001 head: heatmap magnitudes ~1e-4..1e-3).  On the non-degenerate synthetic
weights of the golden fixtures (heatmaps O(1)) bf16 rounding gives ~1e-2 max-abs; that
bound is asserted separately as a relative error so the number is visible.

Every bound below is the value measured on an MI355X + 50 % (`meas` logs the values when VITPOSE_TEST_LOG is
set).  Heat maps: 6e-3 relative.  Gradients of a model with the classic (deconv + BatchNorm + ReLU) head: 3 - 9 %
per tensor, all of it from the ReLU masks of the head, which differ from the fp32 forward's on the 0.1 - 0.3 % of
pre-activations that lie within bf16 rounding of zero (tests/test_grad_stages_gpu.py walks the backward pass stage
by stage: no other stage adds more than 0.5 %); with the simple decoder the same gradients are within 0.5 %.
"""
import os

import numpy as np
import pytest
import torch

from conftest import meas

from oracle import vitpose_oracle as O

pytestmark = pytest.mark.gpu

TOY = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=64, depth=2, num_heads=2,
           mlp_ratio=4, qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0)


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def make_engine(vcfg, hcfg, bb, hd, dev):
    from vitpose_amd.pose_engine import PoseEngine
    eng = PoseEngine(vcfg, hcfg, dev)
    sd = {'backbone.' + k: v for k, v in bb.items()}
    sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
    eng.load_state_dict(sd)
    return eng


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item(), (a - b).abs().max().item()


def test_toy_backbone_tokens(dev, golden_dir):
    from vitpose_amd.engine import ParamStore, VitRunner, Workspace, vit_param_shapes
    from vitpose_amd.plan import Plan
    g = np.load(os.path.join(golden_dir, 'toy_vit.npz'))
    sd = O.synth_vit_weights(TOY, seed=10)
    img, _, _ = O.synth_batch(2, seed=11)
    st = ParamStore(vit_param_shapes(TOY), dev)
    st.load(sd)
    st.half.copy_(st.data)
    vr = VitRunner(TOY, st)
    vr.refresh_derived()
    ws = Workspace(dev)
    tok = torch.empty(2 * 192, 64, device=dev)
    plan = Plan()
    vr.record_forward(plan, ws, img.to(dev), 2, training=True, tokens_f32=tok)
    plan.run()
    torch.cuda.synchronize()
    feat = tok.view(2, 192, 64).permute(0, 2, 1).reshape(2, 64, 16, 12).cpu()
    # stage-by-stage against the reference taps of block 0, image 0
    for name, buf, width in (('block_in', 'x_in.0', 64), ('ln1', 'xn1.0', 64), ('qkv', 'qkv.0', 192),
                             ('attn_o', 'ao.0', 64), ('fc1_pre', 'u.0', 256), ('fc1_act', 'h.0', 256),
                             ('block_out', 'x_in.1', 64)):
        got = ws.bufs[buf].float().view(2, 192, width)[0].cpu()
        rel, mx = relerr(got, g[name][0])
        assert meas(f'toy fwd stage {name} rel', rel) < 6e-3, (name, rel, mx)
    rel, mx = relerr(feat, g['feat'])
    assert meas('toy fwd feat rel', rel) < 5e-3, (rel, mx)


def test_vitpose_s_classic_eval_vs_reference_golden(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'vitpose_s_classic_eval.npz'))
    vcfg, hc = O.vit_cfg('S'), O.classic_head_cfg(384)
    eng = make_engine(vcfg, hc, O.synth_vit_weights(vcfg, seed=20), O.synth_head_weights(hc, seed=21), dev)
    img, _, _ = O.synth_batch(2, seed=22)
    heat = eng.infer(img.to(dev)).cpu()
    assert heat.shape == (2, 17, 64, 48)
    rel, mx = relerr(heat, g['heat'])
    print(f'ViTPose-S classic eval, O(1) heatmaps (|ref|max {np.abs(g["heat"]).max():.2f}): rel {rel:.2e} maxabs {mx:.2e}')
    assert meas('S classic eval rel', rel) < 9e-3 and meas('S classic eval maxabs', mx) < 5e-2, (rel, mx)


def test_reference_init_heatmaps_within_1e3(dev):
    """North-star gate: max-abs 1e-3 against the reference CPU forward (here the oracle,
    which the goldens pin to the reference) with the reference's own initialisation."""
    vcfg, hc = O.vit_cfg('S'), O.classic_head_cfg(384)
    g = torch.Generator().manual_seed(7)
    bb = {}
    for n, s in O.vit_param_shapes(vcfg):
        if n.endswith('norm1.weight') or n.endswith('norm2.weight') or n == 'last_norm.weight':
            bb[n] = torch.ones(s)
        elif n.endswith('.bias'):
            bb[n] = torch.zeros(s)
        elif n == 'patch_embed.proj.weight':   # nn.Conv2d default init (kaiming uniform a=sqrt(5))
            bound = 1.0 / np.sqrt(s[1] * s[2] * s[3])
            bb[n] = (torch.rand(s, generator=g) * 2 - 1) * bound
        else:                                  # trunc_normal_(std=.02) (vit.py:245,297)
            bb[n] = torch.nn.init.trunc_normal_(torch.empty(s), std=.02, generator=g)
    hd = {}
    for n, s in O.head_param_shapes(hc):       # normal_init(std=0.001), BN (1, 0) (simple_head.py:339-350)
        if n.endswith('running_var') or (n.endswith('.weight') and len(s) == 1):
            hd[n] = torch.ones(s)
        elif len(s) == 1:
            hd[n] = torch.zeros(s)
        else:
            hd[n] = torch.randn(s, generator=g) * 0.001
    eng = make_engine(vcfg, hc, bb, hd, dev)
    img, _, _ = O.synth_batch(2, seed=8)
    heat = eng.infer(img.to(dev)).cpu()
    with torch.no_grad():
        ref = O.topdown_forward(bb, hd, img, vcfg, hc)
    mx = (heat - ref).abs().max().item()
    print(f'reference-init heatmaps: |ref|max {ref.abs().max():.2e}, max-abs diff {mx:.2e}')
    assert mx < 1e-3
    assert mx < 0.05 * ref.abs().max().item()  # and not vacuously: within 5 % of the signal


def test_vitpose_s_train_step_vs_reference_golden(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'vitpose_s_train.npz'))
    vcfg = O.vit_cfg('S', drop_path_rate=0.3)
    hc = O.classic_head_cfg(384)
    eng = make_engine(vcfg, hc, O.synth_vit_weights(vcfg, seed=30), O.synth_head_weights(hc, seed=31), dev)
    img, target, tw = O.synth_batch(4, seed=32)
    rates = O.drop_path_rates(vcfg)
    draws = torch.from_numpy(g['dp_draws'])
    dp = torch.ones(vcfg['depth'], 2, 4)
    for i in range(1, vcfg['depth']):
        dp[i, 0] = O.drop_path_scale(draws[2 * (i - 1)], rates[i])
        dp[i, 1] = O.drop_path_scale(draws[2 * (i - 1) + 1], rates[i])
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), dp_scales=dp.to(dev))
    torch.cuda.synchronize()
    rel, mx = relerr(heat[:, ::4], g['heat_sub'])
    print(f'train heat: rel {rel:.2e} max {mx:.2e}; loss {loss.item():.6f} vs {float(g["loss"]):.6f}')
    assert meas('S train heat rel', rel) < 9e-3
    assert meas('S train loss rel', abs(loss.item() - float(g['loss'])) / float(g['loss'])) < 1e-3
    worst = {}
    for k in g.files:
        if k.startswith('g.'):
            got = eng.store.gview('backbone.' + k[2:])
        elif k.startswith('gs.'):
            got = eng.store.gview('backbone.' + k[3:]).flatten(1)[::16, ::4]
        elif k.startswith('gh.'):
            got = eng.store.gview('keypoint_head.' + k[3:])
        elif k.startswith('ghs.'):
            got = eng.store.gview('keypoint_head.' + k[4:]).flatten(1)[::8, ::16]
        else:
            continue
        worst[k] = relerr(got.reshape(g[k].shape), g[k])[0]
    print('grad rel errors:', {k: f'{v:.1e}' for k, v in worst.items()})
    assert meas('S train worst grad rel', max(worst.values())) < 7.5e-2, worst
    # BN running statistics were updated with the batch statistics
    for n, k in (('deconv_layers.1.running_mean', 'bn1_mean'), ('deconv_layers.1.running_var', 'bn1_var'),
                 ('deconv_layers.4.running_mean', 'bn4_mean'), ('deconv_layers.4.running_var', 'bn4_var')):
        rel, _ = relerr(eng.buffers['keypoint_head.' + n], g[k])
        assert meas(f'S train {n} rel', rel) < 2e-3, (n, rel)


def test_train_step_updates_weights_like_torch_adamw(dev):
    """One full step (fwd + bwd + clip + layer-decay AdamW) on the toy model against the oracle
    differentiated by torch autograd and stepped by torch.optim.AdamW with the same groups."""
    from vitpose_amd.optim import layer_decay_param_groups
    hc = dict(O.classic_head_cfg(64))
    bb, hd = O.synth_vit_weights(TOY, seed=1), O.synth_head_weights(hc, seed=2)
    eng = make_engine(TOY, hc, bb, hd, dev)
    eng.setup_optimizer(lr=1e-3, weight_decay=0.1, layer_decay_rate=0.75, max_grad_norm=1.0)
    img, target, tw = O.synth_batch(2, seed=3)
    eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    eng.optimizer_step()
    torch.cuda.synchronize()
    params = {('backbone.' + k): v.clone().requires_grad_() for k, v in bb.items()}
    hparams = {k: v.clone() for k, v in hd.items()}
    for k, v in hparams.items():
        if 'running' not in k:
            params['keypoint_head.' + k] = v.requires_grad_()
    bbp = {k[len('backbone.'):]: v for k, v in params.items() if k.startswith('backbone.')}
    heat = O.head_forward(hparams, O.vit_forward(bbp, img, TOY), hc, training=True)
    O.joints_mse_loss(heat, target, tw).backward()
    groups = layer_decay_param_groups([(n, tuple(p.shape)) for n, p in params.items()], TOY['depth'], 0.75, 1e-3, 0.1)
    topt = torch.optim.AdamW([dict(params=[p], lr=groups[n]['lr'], weight_decay=groups[n]['weight_decay'])
                              for n, p in params.items()], lr=1e-3)
    before = {n: p.detach().clone() for n, p in params.items()}
    torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
    topt.step()
    # compare the UPDATE direction (Adam's first step is lr * sign-like, so compare deltas)
    num = den = 0.0
    for n, p in params.items():
        d_ref = (p.detach() - before[n]).flatten()
        d_got = (eng.store.view(n).cpu() - before[n]).flatten()
        num += (d_ref - d_got).pow(2).sum().item()
        den += d_ref.pow(2).sum().item()
    rel = (num / den) ** 0.5
    print(f'relative error of the first AdamW update over all parameters: {rel:.3e}')
    # Adam's first update is lr * g / (|g| + eps): sign-like, so the bf16 gradient noise flips near-zero entries;
    # 0.17 measured, a wrong gradient anywhere gives ~1.4
    assert meas('toy adamw first-update rel', rel) < 0.25


def test_loss_trajectory_over_steps_tracks_the_oracle(dev):
    """Eight optimizer steps on two alternating batches (fwd + bwd + clip + layer-decay AdamW, BatchNorm statistics moving):
    the loss of every step stays within a few per cent of the oracle trained by torch autograd + torch.optim.AdamW with the
    same groups -- errors do not compound into a different trajectory -- and the loss goes down."""
    from vitpose_amd.optim import layer_decay_param_groups
    hc = dict(O.classic_head_cfg(64))
    bb, hd = O.synth_vit_weights(TOY, seed=21), O.synth_head_weights(hc, seed=22)
    eng = make_engine(TOY, hc, bb, hd, dev)
    eng.setup_optimizer(lr=2e-4, weight_decay=0.1, layer_decay_rate=0.75, max_grad_norm=1.0)
    batches = [O.synth_batch(2, seed=23), O.synth_batch(2, seed=24)]
    params = {('backbone.' + k): v.clone().requires_grad_() for k, v in bb.items()}
    hparams = {k: v.clone() for k, v in hd.items()}
    for k, v in hparams.items():
        if 'running' not in k:
            params['keypoint_head.' + k] = v.requires_grad_()
    bbp = {k[len('backbone.'):]: v for k, v in params.items() if k.startswith('backbone.')}
    groups = layer_decay_param_groups([(n, tuple(p.shape)) for n, p in params.items()], TOY['depth'], 0.75, 2e-4, 0.1)
    topt = torch.optim.AdamW([dict(params=[p], lr=groups[n]['lr'], weight_decay=groups[n]['weight_decay'])
                              for n, p in params.items()], lr=2e-4)
    got, ref = [], []
    for step in range(8):
        img, target, tw = batches[step % 2]
        loss, _ = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
        got.append(loss.item())
        eng.optimizer_step()
        topt.zero_grad()
        rl = O.joints_mse_loss(O.head_forward(hparams, O.vit_forward(bbp, img, TOY), hc, training=True), target, tw)
        rl.backward()
        torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
        topt.step()
        ref.append(rl.item())
    worst = max(abs(a - b) / b for a, b in zip(got, ref))
    print('loss per step, HIP:', [f'{v:.5f}' for v in got], 'oracle:', [f'{v:.5f}' for v in ref])
    assert meas('toy 8-step loss trajectory worst rel', worst) < 3e-3      # 7e-4 measured
    assert got[-2] < got[0] and got[-1] < got[1]
    # BatchNorm running statistics followed the same path
    rm = eng.buffers['keypoint_head.deconv_layers.1.running_mean'].cpu()
    assert meas('toy 8-step running_mean rel', relerr(rm, hparams['deconv_layers.1.running_mean'])[0]) < 2e-2


def test_vitpose_s_simple_decoder_vs_reference_golden(dev, golden_dir):
    """The CPU-runnable benchmark configuration (ViTPose-S + simple decoder, batch 1): the reference source ran it on the
    CPU (golden), the HIP path runs it on the GPU.  O(1) heat maps, bf16 operands: relative bound."""
    g = np.load(os.path.join(golden_dir, 'vitpose_s_simple_eval.npz'))
    vcfg, hs = O.vit_cfg('S'), O.simple_head_cfg(384)
    eng = make_engine(vcfg, hs, O.synth_vit_weights(vcfg, seed=20), O.synth_head_weights(hs, seed=23), dev)
    img, _, _ = O.synth_batch(2, seed=22)
    heat = eng.infer(img[:1].to(dev)).cpu()
    assert heat.shape == (1, 17, 64, 48)
    rel, mx = relerr(heat, g['heat'])
    print(f'ViTPose-S simple decoder, |ref|max {np.abs(g["heat"]).max():.2f}: rel {rel:.2e} maxabs {mx:.2e}')
    assert meas('S simple eval rel', rel) < 9e-3, (rel, mx)


def test_simple_decoder_train_step_vs_oracle(dev):
    """Forward + backward of backbone + simple decoder against the oracle under torch autograd."""
    hc = O.simple_head_cfg(64)
    bb, hd = O.synth_vit_weights(TOY, seed=5), O.synth_head_weights(hc, seed=6)
    eng = make_engine(TOY, hc, bb, hd, dev)
    img, target, tw = O.synth_batch(2, seed=7)
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: v.clone().requires_grad_() for k, v in hd.items()}
    ref = O.head_forward(hdr, O.vit_forward(bbr, img, TOY), hc, training=True)
    rl = O.joints_mse_loss(ref, target, tw)
    rl.backward()
    assert meas('toy simple train heat rel', relerr(heat, ref.detach())[0]) < 6e-3
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    for n in ('final_layer.weight', 'final_layer.bias'):
        assert meas(f'toy simple grad {n}', relerr(eng.store.gview('keypoint_head.' + n), hdr[n].grad)[0]) < 5e-3, n
    for n in ('blocks.1.mlp.fc2.weight', 'blocks.0.attn.qkv.weight', 'patch_embed.proj.weight', 'pos_embed'):
        assert meas(f'toy simple grad {n}', relerr(eng.store.gview('backbone.' + n), bbr[n].grad)[0]) < 3.6e-2, n   # (patch embedding 2.4e-2: ReLU sign flips in front of the decoder; the rest 5e-3)


def test_no_last_norm_train_step_and_exact_eval_vs_oracle(dev):
    """ViT(last_norm=False) (vit.py:257: the final LayerNorm is an Identity): forward + backward against the oracle
    under torch autograd -- the feature gradient enters the residual stream as it is -- and the fp32-accurate
    evaluation path at 1e-3."""
    vcfg = dict(TOY, last_norm=False)
    hc = O.classic_head_cfg(TOY['embed_dim'])
    bb, hd = O.synth_vit_weights(vcfg, seed=15), O.synth_head_weights(hc, seed=16)
    assert 'last_norm.weight' not in bb
    eng = make_engine(vcfg, hc, bb, hd, dev)
    img, target, tw = O.synth_batch(2, seed=17)
    # evaluation, fp32-accurate path (before the training step moves the BatchNorm running statistics)
    ref_eval = O.head_forward({k: v.clone() for k, v in hd.items()}, O.vit_forward(bb, img, vcfg), hc, training=False)
    got = eng.infer(img.to(dev), exact=True)
    assert meas('no last_norm exact max abs', (got.float().cpu() - ref_eval).abs().max().item()) < EXACT_TOL
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: v.clone() for k, v in hd.items()}
    ref = O.head_forward(hdr, O.vit_forward(bbr, img, vcfg), hc, training=True)
    rl = O.joints_mse_loss(ref, target, tw)
    rl.backward()
    assert meas('no last_norm heat rel', relerr(heat, ref.detach())[0]) < 9e-3
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    for n in ('blocks.1.mlp.fc2.weight', 'blocks.1.mlp.fc2.bias', 'blocks.1.mlp.fc1.weight', 'blocks.0.attn.qkv.weight',
              'patch_embed.proj.weight', 'pos_embed'):
        assert meas(f'no last_norm grad {n}', relerr(eng.store.gview('backbone.' + n), bbr[n].grad)[0]) < 0.1, n


def test_head_with_conv_layers_whole_model_train_step_vs_oracle(dev):
    """The whole training step with `extra=dict(num_conv_layers=1, num_conv_kernels=(3,), final_conv_kernel=3)` on
    the classic head (topdown_heatmap_simple_head.py:112-138): loss, heat maps, head and backbone gradients against
    the oracle under torch autograd, then bf16 evaluation."""
    hc = dict(O.classic_head_cfg(TOY['embed_dim']), extra=dict(final_conv_kernel=3, num_conv_layers=1, num_conv_kernels=(3,)))
    bb, hd = O.synth_vit_weights(TOY, seed=21), O.synth_head_weights(hc, seed=22)
    eng = make_engine(TOY, hc, bb, hd, dev)
    img, target, tw = O.synth_batch(2, seed=23)
    ref_eval = O.head_forward({k: v.clone() for k, v in hd.items()}, O.vit_forward(bb, img, TOY), hc, training=False)
    got = eng.infer(img.to(dev))
    assert meas('conv-layer head eval heat rel', relerr(got, ref_eval)[0]) < 2e-2
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: (v.clone().requires_grad_() if 'running' not in k else v.clone()) for k, v in hd.items()}
    ref = O.head_forward(hdr, O.vit_forward(bbr, img, TOY), hc, training=True)
    rl = O.joints_mse_loss(ref, target, tw)
    rl.backward()
    assert meas('conv-layer head heat rel', relerr(heat, ref.detach())[0]) < 1.5e-2
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    for n in ('final_layer.3.weight', 'final_layer.3.bias', 'final_layer.0.weight', 'final_layer.1.weight', 'deconv_layers.3.weight'):
        assert meas(f'conv-layer head grad {n}', relerr(eng.store.gview('keypoint_head.' + n), hdr[n].grad)[0]) < 0.1, n
    for n in ('blocks.1.mlp.fc2.weight', 'blocks.0.attn.qkv.weight', 'patch_embed.proj.weight'):
        assert meas(f'conv-layer head grad {n}', relerr(eng.store.gview('backbone.' + n), bbr[n].grad)[0]) < 0.18, n


@pytest.mark.parametrize('name', ['L', 'H'])
def test_large_and_huge_widths_one_block_train_step(dev, name):
    """ViTPose-L / -H widths (E 1024 / 1280, head_dim 64 / 80) through one block + classic head,
    forward and backward, against the oracle (whose block arithmetic at these widths is pinned by
    tests/golden/block_widths.npz)."""
    vcfg = O.vit_cfg(name, depth=1, drop_path_rate=0.0)
    hc = O.classic_head_cfg(vcfg['embed_dim'])
    bb, hd = O.synth_vit_weights(vcfg, seed=40), O.synth_head_weights(hc, seed=41)
    eng = make_engine(vcfg, hc, bb, hd, dev)
    img, target, tw = O.synth_batch(2, seed=42)
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: v.clone() for k, v in hd.items()}
    ref = O.head_forward(hdr, O.vit_forward(bbr, img, vcfg), hc, training=True)
    rl = O.joints_mse_loss(ref, target, tw)
    rl.backward()
    assert meas('large_and_huge_widths_one_block_tra L256', relerr(heat, ref.detach())[0]) < 8e-3
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    for n in ('blocks.0.attn.qkv.weight', 'blocks.0.attn.proj.weight', 'blocks.0.mlp.fc1.weight', 'blocks.0.norm1.weight',
              'patch_embed.proj.weight'):
        assert meas(f'large_and_huge_widths_one_block_tra {n}', relerr(eng.store.gview('backbone.' + n), bbr[n].grad)[0]) < 0.1, n


# ---------------------------------------------------------------------------------------------
# fp32-accurate evaluation: the 1e-3 max-abs heat-map tolerance on NON-degenerate weights
# (O(1) heat maps), against the vectors the reference's own source produced.
# ---------------------------------------------------------------------------------------------
EXACT_TOL = 1e-3   # max-abs on heat maps, the tolerance the headline configuration states


def test_exact_eval_classic_vs_reference_golden_1e3(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'vitpose_s_classic_eval.npz'))
    vcfg, hc = O.vit_cfg('S'), O.classic_head_cfg(384)
    eng = make_engine(vcfg, hc, O.synth_vit_weights(vcfg, seed=20), O.synth_head_weights(hc, seed=21), dev)
    img, _, _ = O.synth_batch(2, seed=22)
    heat = eng.infer(img.to(dev), exact=True).cpu()
    rel, mx = relerr(heat, g['heat'])
    print(f'exact eval, classic head, |ref|max {np.abs(g["heat"]).max():.2f}: rel {rel:.2e} maxabs {mx:.2e}')
    assert mx < EXACT_TOL, (rel, mx)
    # the bf16 plan of the same engine still works afterwards and is the looser one
    heat_b = eng.infer(img.to(dev)).cpu()
    assert relerr(heat_b, g['heat'])[1] > mx


def test_exact_eval_simple_decoder_vs_reference_golden_1e3(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, 'vitpose_s_simple_eval.npz'))
    vcfg, hs = O.vit_cfg('S'), O.simple_head_cfg(384)
    eng = make_engine(vcfg, hs, O.synth_vit_weights(vcfg, seed=20), O.synth_head_weights(hs, seed=23), dev)
    img, _, _ = O.synth_batch(2, seed=22)
    heat = eng.infer(img[:1].to(dev), exact=True).cpu()
    rel, mx = relerr(heat, g['heat'])
    print(f'exact eval, simple decoder, |ref|max {np.abs(g["heat"]).max():.2f}: rel {rel:.2e} maxabs {mx:.2e}')
    assert mx < EXACT_TOL, (rel, mx)


@pytest.mark.parametrize('name', ['B', 'H'])
def test_exact_eval_wide_models_vs_oracle_1e3(dev, name):
    """ViTPose-B (head_dim 64) and -H widths (head_dim 80), two blocks, against the oracle."""
    vcfg = O.vit_cfg(name, depth=2, drop_path_rate=0.0)
    hc = O.classic_head_cfg(vcfg['embed_dim'])
    bb, hd = O.synth_vit_weights(vcfg, seed=50), O.synth_head_weights(hc, seed=51)
    eng = make_engine(vcfg, hc, bb, hd, dev)
    img, _, _ = O.synth_batch(3, seed=52)
    heat = eng.infer(img.to(dev), exact=True).cpu()
    with torch.no_grad():
        ref = O.topdown_forward(bb, hd, img, vcfg, hc)
    rel, mx = relerr(heat, ref)
    print(f'exact eval ViTPose-{name} (2 blocks), |ref|max {ref.abs().max():.2f}: rel {rel:.2e} maxabs {mx:.2e}')
    assert mx < EXACT_TOL * max(1.0, ref.abs().max().item()), (rel, mx)


@pytest.mark.parametrize('bf16', [False, True])
def test_ddp_plan_on_gpu_single_rank_group(dev, bf16, monkeypatch):
    """The batch-sharded training plan with gradient buckets (RCCL all-reduce launched from inside the
    backward plan on the communication stream) over a one-rank process group: same gradients as the plan
    without buckets.  The collective is forced on (world treated as 2) so the async launches, the stream
    hand-over and the bucket trigger points all execute; more ranks need more GPUs than this box has."""
    import torch.distributed as dist
    hc = O.classic_head_cfg(64)
    vcfg = dict(TOY, depth=6)
    bb, hd = O.synth_vit_weights(vcfg, seed=5), O.synth_head_weights(hc, seed=6)
    img, target, tw = O.synth_batch(2, seed=7)
    ref_eng = make_engine(vcfg, hc, bb, hd, dev)
    ref_eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        monkeypatch.setenv('VITPOSE_DDP_BF16', '1' if bf16 else '0')   # bf16: buckets staged, sent as bf16, widened back
        eng = make_engine(vcfg, hc, bb, hd, dev)
        comm = eng.enable_ddp(blocks_per_bucket=2)
        assert sorted(comm.triggers) == [2, 4] and comm.overlap and comm.bf16 == bf16
        comm.world = 2                      # force the all-reduce launches (sum over one rank = identity)
        comm.measure = True
        for _ in range(2):                  # twice: the plan is replayed, events and streams are reused
            eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
        torch.cuda.synchronize()
        assert comm.launched == 0 and not comm.staged
        times = comm.exposed_ms()           # (exposed, busy) per step from the HIP events around the final wait
        assert len(times) == 2 and all(0 <= e <= b + 0.05 for e, b in times), times
        # two runs of the same plan already differ by ~1e-3 per tensor (float atomics in the batch-norm and
        # loss reductions feed bf16 roundings downstream), so "same gradients" is a per-tensor norm bound
        for n in eng.store.names:
            got, ref = eng.store.gview(n), ref_eng.store.gview(n)
            assert meas(f'ddp_plan_on_gpu_single_rank_group {n}', relerr(got, ref)[0]) < (1.2e-2 if bf16 else 7e-3), n
    finally:
        if created:
            dist.destroy_process_group()


def test_wholebody_133_keypoints_train_step_and_eval(dev):
    """COCO-WholeBody heads (133 heat maps, configs/wholebody/.../ViTPose_*_wholebody_256x192.py): the channel
    padding follows the keypoint count.  Forward + backward and the eval forward against the oracle."""
    hc = dict(O.classic_head_cfg(64), out_channels=133)
    bb, hd = O.synth_vit_weights(TOY, seed=5), O.synth_head_weights(hc, seed=6)
    eng = make_engine(TOY, hc, bb, hd, dev)
    img, target, tw = O.synth_batch(2, K=133, seed=7)
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    assert heat.shape == (2, 133, 64, 48)
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: v.clone().requires_grad_() if v.dtype.is_floating_point and 'running' not in k else v.clone()
           for k, v in hd.items()}
    ref = O.head_forward(hdr, O.vit_forward(bbr, img, TOY), hc, training=True)
    rl = O.joints_mse_loss(ref, target, tw)
    rl.backward()
    assert meas('wholebody_133_keypoints_train_step_ L364', relerr(heat, ref.detach())[0]) < 9e-3
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    for n in ('final_layer.weight', 'final_layer.bias', 'deconv_layers.3.weight'):
        assert meas(f'wholebody_133_keypoints_train_step_ {n}', relerr(eng.store.gview('keypoint_head.' + n), hdr[n].grad)[0]) < 1.7e-2, n
    assert meas('wholebody_133_keypoints_train_step_ L368', relerr(eng.store.gview('backbone.blocks.1.mlp.fc2.weight'), bbr['blocks.1.mlp.fc2.weight'].grad)[0]) < 3.7e-2
    with torch.no_grad():
        ref_eval = O.topdown_forward(bb, hd, img, TOY, hc)
    ev = make_engine(TOY, hc, bb, hd, dev)      # fresh BatchNorm running statistics (the train step moved eng's)
    assert meas('wholebody_133_keypoints_train_step_ L372', relerr(ev.infer(img.to(dev)), ref_eval)[0]) < 8e-3
    assert (ev.infer(img.to(dev), exact=True).cpu() - ref_eval).abs().max().item() < 1e-3 * max(1.0, ref_eval.abs().max().item())


@pytest.mark.parametrize('B', [1, 3, 7])
def test_odd_batch_sizes_train_and_eval_vs_oracle(dev, B):
    """Ragged batches (1, 3, 7 images: partial GEMM tiles in every product, BatchNorm statistics over a single image
    at B = 1), one engine re-recorded per size: loss, heat maps and a spread of gradients against the oracle, and the
    eval forward of the same engine afterwards."""
    hc = O.classic_head_cfg(64)
    bb, hd = O.synth_vit_weights(TOY, seed=1), O.synth_head_weights(hc, seed=2)
    eng = make_engine(TOY, hc, bb, hd, dev)
    img, target, tw = O.synth_batch(B, seed=100 + B)
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: (v.clone().requires_grad_() if 'running' not in k else v.clone()) for k, v in hd.items()}
    ref = O.head_forward(hdr, O.vit_forward(bbr, img, TOY), hc, training=True)
    rl = O.joints_mse_loss(ref, target, tw)
    rl.backward()
    assert meas('odd_batch_sizes_train_and_eval_vs_o L392', relerr(heat, ref.detach())[0]) < 9e-3
    assert abs(loss.item() - rl.item()) < 3e-2 * rl.item()
    for n in ('blocks.1.mlp.fc2.weight', 'blocks.0.attn.qkv.weight', 'blocks.0.norm1.weight', 'pos_embed',
              'patch_embed.proj.weight', 'blocks.1.mlp.fc1.bias'):
        assert meas(f'odd_batch_sizes_train_and_eval_vs_o {n}', relerr(eng.store.gview('backbone.' + n), bbr[n].grad)[0]) < 0.135, n
    for n in ('final_layer.weight', 'deconv_layers.0.weight', 'deconv_layers.3.weight', 'deconv_layers.4.bias'):
        assert meas(f'odd_batch_sizes_train_and_eval_vs_o {n}', relerr(eng.store.gview('keypoint_head.' + n), hdr[n].grad)[0]) < 7.2e-2, n
    with torch.no_grad():
        ref_eval = O.head_forward({k: v.detach() for k, v in hdr.items()},
                                  O.vit_forward({k: v.detach() for k, v in bbr.items()}, img, TOY), hc)
    # eval after one training pass: the engine's running statistics carry that pass's update, like the oracle's
    assert meas('odd_batch_sizes_train_and_eval_vs_o L403', relerr(eng.infer(img.to(dev)), ref_eval)[0]) < 8e-3


def test_empty_batch_is_rejected(dev):
    hc = O.classic_head_cfg(64)
    eng = make_engine(TOY, hc, O.synth_vit_weights(TOY, seed=1), O.synth_head_weights(hc, seed=2), dev)
    with pytest.raises(ValueError):
        eng.infer(torch.zeros(0, 3, 256, 192, device=dev))
    with pytest.raises(ValueError):
        eng.forward_backward(torch.zeros(0, 3, 256, 192, device=dev), torch.zeros(0, 17, 64, 48, device=dev),
                             torch.zeros(0, 17, 1, device=dev))


@pytest.mark.parametrize('name,B', [('S', 2), ('B', 4), ('H', 3)])
def test_folded_layernorm_eval_matches_plain_path_and_oracle(dev, name, B):
    """Evaluation with norm1 / norm2 of every Block folded into qkv / fc1 (infer(fold_ln=True), test_cfg fold_layernorm):
    the same network -- heat maps as close to the fp32 oracle as the plain path's, and within bf16 rounding of them; a
    weight update reaches the folded weights; an image's result does not depend on its batch (the statistics are summed
    in an order the width alone fixes)."""
    vcfg = O.vit_cfg(name)
    hc = O.classic_head_cfg(vcfg['embed_dim'])
    bb, hd = O.synth_vit_weights(vcfg, seed=30), O.synth_head_weights(hc, seed=31)
    img, _, _ = O.synth_batch(max(B, 8), seed=32)
    with torch.no_grad():
        ref = O.topdown_forward(bb, hd, img[:B], vcfg, hc)
    eng = make_engine(vcfg, hc, bb, hd, dev)
    x = img.to(dev)
    plain, folded = eng.infer(x[:B]).cpu(), eng.infer(x[:B], fold_ln=True).cpu()
    # no LayerNorm launch but the last one
    names = [c[0] for c in eng.plans[('infer', B, True)][0].calls]
    assert names.count('vp_layernorm_fwd') == 1 and [c[0] for c in eng.plans[('infer', B)][0].calls].count('vp_layernorm_fwd') > 2
    # batch-independent, bit for bit: batch 1 (no timing-based schedule choice applies) and batch 8 (it does)
    assert torch.equal(eng.infer(x[:1], fold_ln=True).cpu(), folded[:1])
    assert torch.equal(eng.infer(x, fold_ln=True).cpu()[:B], folded)
    r_plain, r_fold = relerr(plain, ref)[0], relerr(folded, ref)[0]
    print(f'ViTPose-{name} x {B}: heat maps vs oracle: plain {r_plain:.2e}, folded {r_fold:.2e}; '
          f'folded vs plain {relerr(folded, plain)[0]:.2e}')
    assert meas(f'{name} folded eval rel', r_fold) < max(1.25 * r_plain, 9e-3) and relerr(folded, plain)[0] < 8e-3
    # new weights reach the folded copies
    sd = eng.state_dict()
    sd['backbone.blocks.0.norm1.weight'] = sd['backbone.blocks.0.norm1.weight'] * 1.5
    sd['backbone.blocks.1.mlp.fc1.bias'] = sd['backbone.blocks.1.mlp.fc1.bias'] + 0.3
    eng.load_state_dict(sd)
    changed, want = eng.infer(x[:B], fold_ln=True).cpu(), eng.infer(x[:B]).cpu()
    assert relerr(changed, folded)[0] > 1e-2 and relerr(changed, want)[0] < 8e-3
