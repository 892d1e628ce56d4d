"""The two wide configurations at FULL size through the C ABI: ViTPose-L, 32 images per GPU, training step;
ViTPose-H, 128 images, inference.  The CPU oracle cannot run these whole in seconds, so each is pinned by
(a) the oracle on the first two images of the batch (eval-mode BatchNorm makes an image's heat maps independent of
its batch), (b) size-independent properties: batch-permutation equivariance, zero target weights -> zero
gradients, linearity of the gradients in the loss weight; and for the training step (c) the oracle's loss on
heat maps of the HIP forward (the loss kernel at this size) and one Block-level gradient through torch autograd
from the kernel's own saved activations."""
import pytest
import torch

from conftest import meas

from oracle import vitpose_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def rel(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm()).item()


def make(name, dev, seed, loss_weight=1.0):
    from vitpose_amd.pose_engine import PoseEngine
    vcfg = O.vit_cfg(name)
    hc = O.classic_head_cfg(vcfg['embed_dim'])
    bb, hd = O.synth_vit_weights(vcfg, seed=seed), O.synth_head_weights(hc, seed=seed + 1)
    eng = PoseEngine(vcfg, hc, dev, loss_weight=loss_weight)
    sd = {'backbone.' + k: v for k, v in bb.items()}
    sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
    eng.load_state_dict(sd)
    return eng, vcfg, hc, bb, hd


def test_vitpose_h_inference_128_images(dev):
    B = 128
    eng, vcfg, hc, bb, hd = make('H', dev, 70)
    img, _, _ = O.synth_batch(B, seed=72)
    x = img.to(dev)
    full = eng.infer(x)
    assert full.shape == (B, 17, 64, 48) and torch.isfinite(full).all()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = O.topdown_forward(bb, hd, img[:2], vcfg, hc)
    assert ref.abs().max() > 0.5                       # non-degenerate heat maps
    r = rel(full[:2], ref)
    print(f'ViTPose-H x 128: heat maps of images 0-1 vs oracle, rel {r:.2e}')
    assert meas('H x128 eval rel', r) < 1e-2
    assert rel(eng.infer(x[:2]), full[:2]) < 1e-2      # an image does not depend on its batch
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(7)).to(dev)
    assert torch.equal(eng.infer(x[perm]), full[perm])


def test_vitpose_l_train_step_32_images(dev):
    B = 32
    eng, vcfg, hc, bb, hd = make('L', dev, 80)
    img, target, tw = O.synth_batch(B, seed=82)
    x, t, w = img.to(dev), target.to(dev), tw.to(dev)
    # eval forward of the first two images against the oracle
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = O.topdown_forward(bb, hd, img[:2], vcfg, hc)
    r = rel(eng.infer(x)[:2], ref)
    print(f'ViTPose-L x 32: eval heat maps of images 0-1 vs oracle, rel {r:.2e}')
    assert meas('L x32 eval rel', r) < 1e-2
    # training step (DropPath off so that the properties below are deterministic)
    loss, heat = eng.forward_backward(x, t, w, drop_path=False)
    torch.cuda.synchronize()
    assert torch.isfinite(heat).all()
    # the loss kernel at this size: JointsMSELoss of the oracle on the heat maps the HIP forward produced
    assert loss.item() == pytest.approx(float(O.joints_mse_loss(heat.cpu(), target, tw)), rel=1e-4)
    g1 = eng.store.grad.clone()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    # the last Block's fc2 weight gradient through torch autograd from the kernels' own saved tensors:
    # dW2 = (DropPath-scaled d x_out)^T h, with d x_out = gradient entering the Block's output
    ws = eng.ws_train.bufs
    last = vcfg['depth'] - 1
    dxb, h = ws[f'dxb.{last}.mlp'].float(), ws[f'h.{last}'].float()
    assert meas('vitpose_l_train_step_32_images L84', rel(eng.store.gview(f'backbone.blocks.{last}.mlp.fc2.weight'), dxb.t() @ h)) < 1e-3
    assert meas('vitpose_l_train_step_32_images L85', rel(eng.store.gview(f'backbone.blocks.{last}.mlp.fc2.bias'), dxb.sum(0))) < 1e-3
    # zero target weights: zero loss, exactly zero gradients
    loss0, _ = eng.forward_backward(x, t, torch.zeros_like(w), drop_path=False)
    torch.cuda.synchronize()
    assert loss0.item() == 0.0 and torch.count_nonzero(eng.store.grad).item() == 0
    # gradients are linear in the loss weight (same batch statistics, same masks)
    eng2, *_ = make('L', dev, 80, loss_weight=0.5)
    eng2.forward_backward(x, t, w, drop_path=False)
    torch.cuda.synchronize()
    assert meas('vitpose_l_train_step_32_images L94', rel(eng2.store.grad, 0.5 * g1)) < 2e-2
