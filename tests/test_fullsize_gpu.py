"""The headline configuration at FULL size (ViTPose-B, classic decoder, 64 images per GPU) through the
C ABI: a direct comparison with the CPU oracle (which takes ~10 s of host time for one forward + backward at
this size), and size-independent properties: per-image independence and permutation equivariance of the eval
forward, zero gradients under zero target weights, DropPath scaling with the same masks on both sides."""
import numpy as np
import pytest
import torch

from conftest import meas

from oracle import vitpose_oracle as O

pytestmark = pytest.mark.gpu
B = 64


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def setup(dev):
    from vitpose_amd.pose_engine import PoseEngine
    vcfg, hc = O.vit_cfg('B'), O.classic_head_cfg(768)
    bb, hd = O.synth_vit_weights(vcfg, seed=60), O.synth_head_weights(hc, seed=61)

    def make():
        eng = PoseEngine(vcfg, hc, dev)
        sd = {'backbone.' + k: v for k, v in bb.items()}
        sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
        eng.load_state_dict(sd)
        return eng
    img, target, tw = O.synth_batch(B, seed=62)
    return dict(vcfg=vcfg, hc=hc, bb=bb, hd=hd, make=make, img=img, target=target, tw=tw)


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


def test_full_size_train_step_vs_oracle(dev, setup):
    s = setup
    eng = s['make']()
    depth = s['vcfg']['depth']
    g = torch.Generator().manual_seed(63)
    rates = O.drop_path_rates(s['vcfg'])
    u = torch.rand(depth, 2, B, generator=g)
    dp = torch.stack([torch.stack([O.drop_path_scale(u[i, j], rates[i]) for j in range(2)]) for i in range(depth)])
    assert (dp == 0).any() and (dp > 1).any()          # some samples dropped, the rest scaled up
    loss, heat = eng.forward_backward(s['img'].to(dev), s['target'].to(dev), s['tw'].to(dev), dp_scales=dp.to(dev))
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    bbr = {k: v.clone().requires_grad_() for k, v in s['bb'].items()}
    hdr = {k: (v.clone().requires_grad_() if 'running' not in k and v.dtype.is_floating_point else v.clone())
           for k, v in s['hd'].items()}
    feat = O.vit_forward(bbr, s['img'], s['vcfg'], [(dp[i, 0], dp[i, 1]) for i in range(depth)])
    ref = O.head_forward(hdr, feat, s['hc'], training=True)
    rl = O.joints_mse_loss(ref, s['target'], s['tw'])
    rl.backward()
    r_heat = rel(heat, ref.detach())
    print(f'full size: loss {loss.item():.6f} vs {rl.item():.6f}, heat rel {r_heat:.2e}')
    assert meas('fullsize heat rel', r_heat) < 1.1e-2
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    worst = 0.0
    for n in ('blocks.11.mlp.fc2.weight', 'blocks.6.attn.qkv.weight', 'blocks.6.attn.qkv.bias', 'blocks.0.mlp.fc1.weight',
              'blocks.0.mlp.fc1.bias', 'blocks.0.norm1.weight', 'patch_embed.proj.weight', 'pos_embed', 'last_norm.bias'):
        r = rel(eng.store.gview('backbone.' + n), bbr[n].grad)
        worst = max(worst, r)
        # 64 images average the head's ReLU sign flips out: 1e-2 per tensor (4.3e-2 on the patch embedding)
        assert meas(f'fullsize grad {n}', r) < (6.5e-2 if n.startswith('patch_embed') else 1.6e-2), (n, r)
    for n in ('final_layer.weight', 'deconv_layers.0.weight', 'deconv_layers.4.weight'):
        r = rel(eng.store.gview('keypoint_head.' + n), hdr[n].grad)
        worst = max(worst, r)
        assert meas(f'fullsize grad {n}', r) < 1.5e-2, (n, r)
    print(f'full size: worst per-tensor gradient rel err {worst:.2e}')


def test_full_size_eval_is_per_image_and_permutation_equivariant(dev, setup):
    s = setup
    eng = s['make']()
    img = s['img'].to(dev)
    full = eng.infer(img)
    # (a) an image's heat maps do not depend on its batch: batch 64 (large-tile schedules) vs batch 2
    small = eng.infer(img[:2])
    assert meas('fullsize batch 64 vs batch 2', rel(full[:2], small)) < 1e-3
    with torch.no_grad():
        ref = O.topdown_forward(s['bb'], s['hd'], s['img'][:2], s['vcfg'], s['hc'])
    assert meas('fullsize eval rel (batch 2)', rel(small, ref)) < 9e-3 and meas('fullsize eval rel', rel(full[:2], ref)) < 9e-3
    # (b) permuting the batch permutes the output, exactly (same kernels, same tiles per row)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(dev)
    assert torch.equal(eng.infer(img[perm]), full[perm])
    # (c) the fp32-accurate path at full size against the oracle on the same two images
    ex = eng.infer(img, exact=True)
    assert (ex[:2].cpu() - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())


def test_full_size_zero_target_weight_gives_zero_gradients(dev, setup):
    s = setup
    eng = s['make']()
    loss, _ = eng.forward_backward(s['img'].to(dev), s['target'].to(dev), torch.zeros_like(s['tw']).to(dev),
                                   drop_path=False)
    torch.cuda.synchronize()
    assert loss.item() == 0.0
    assert torch.count_nonzero(eng.store.grad).item() == 0
