"""Where the gradient error of the bf16 training path comes from: the backward plan of a small model is replayed ONE
CALL AT A TIME and the tensor every stage leaves behind is compared with the gradient torch autograd holds for
the same intermediate of the fp32 oracle (oracle/vitpose_oracle.py `taps`).  Per stage:
  accumulated - error of the stage's output as the plan computes it (inherits the error of everything upstream);
  own         - error when the stage's INPUT gradient is replaced by the oracle's (bf16-rounded) one: what the
                stage itself adds (bf16 operands, bf16 output rounding, bf16 probabilities ...);
  same-mask   - (the two stages behind a ReLU) `own` over the elements whose ReLU mask agrees with the oracle's.
Finding: every stage adds 0.04 - 0.5 % of its own, EXCEPT the two ReLU derivatives of the head, which add ~5 %
each.  There the bf16 forward puts 0.2 - 0.3 % of the pre-activations (those within bf16 rounding of zero) on the
other side of zero than the fp32 forward: the mask - the exact derivative of the network that was computed -
differs from the oracle's on those elements, and each such element is wrong by its whole gradient.  On the
elements with equal masks the same stages are as accurate as every other (~0.3 %).  The ~4 - 9 % per-parameter
gradient differences of the bf16 path are these flips, carried down the backbone; no stage has an indexing or
scaling error hidden under them.  The table is printed (pytest -s); every entry is bounded at measured + 50 %."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _oracle_with_taps(O, vcfg, hc, bb, hd, img, target, tw):
    """Forward + backward of the oracle with every block's intermediates kept (and their gradients retained)."""
    bbr = {k: v.clone().requires_grad_() for k, v in bb.items()}
    hdr = {k: v.clone().requires_grad_('running' not in k and 'num_batches' not in k) for k, v in hd.items()}
    x, (Hp, Wp) = O.patch_embed(bbr, img, vcfg)
    pe = bbr['pos_embed']
    x = x + pe[:, 1:] + pe[:, :1]
    taps, xin = [], []
    for i in range(vcfg['depth']):
        x.retain_grad()
        xin.append(x)
        t = {}
        x = O.block(bbr, i, x, vcfg, None, t)
        for v in t.values():
            v.retain_grad()
        taps.append(t)
    x.retain_grad()
    xin.append(x)
    tok = torch.nn.functional.layer_norm(x, (x.shape[-1],), bbr['last_norm.weight'], bbr['last_norm.bias'], 1e-6)
    tok.retain_grad()
    feat = tok.permute(0, 2, 1).reshape(img.shape[0], -1, Hp, Wp).contiguous()
    htaps = {}
    heat = O.head_forward(hdr, feat, hc, training=True, taps=htaps)
    for v in htaps.values():
        v.retain_grad()
    heat.retain_grad()
    loss = O.joints_mse_loss(heat, target, tw)
    loss.backward()
    taps.append(htaps)
    return bbr, hdr, taps, xin, tok, heat, loss


@pytest.mark.parametrize('E,heads', [(128, 2), (768, 12)])
def test_backward_stage_error_budget(E, heads):
    from oracle import vitpose_oracle as O
    from vitpose_amd.pose_engine import PoseEngine
    dev = torch.device('cuda:0')
    depth, B = 2, 2
    vcfg = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=E, depth=depth, num_heads=heads, mlp_ratio=4,
                qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0)
    hc = O.classic_head_cfg(E)
    bb, hd = O.synth_vit_weights(vcfg, seed=1), O.synth_head_weights(hc, seed=2)
    img, target, tw = O.synth_batch(B, seed=3)
    bbr, hdr, taps, xin, tok, heat_ref, loss_ref = _oracle_with_taps(O, vcfg, hc, bb, hd, img, target, tw)

    eng = PoseEngine(vcfg, hc, dev)
    sd = {'backbone.' + k: v for k, v in bb.items()}
    sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
    eng.load_state_dict(sd)
    loss, heat = eng.forward_backward(img.to(dev), target.to(dev), tw.to(dev), drop_path=False)
    torch.cuda.synchronize()
    plan, io = eng._train_plan(B)
    ws = eng.ws_train.bufs
    M, T, Hd = B * 192, 192, 4 * E

    def packed_qkv_grad(i):
        g = taps[i]['qkv'].grad                      # [3][B][h][N][d]
        return g.permute(1, 3, 0, 2, 4).reshape(M, 3 * E)

    def arg_tensor(idx, pos, dtype):
        """The tensor behind pointer argument `pos` of recorded call `idx`."""
        ptr = plan.calls[idx][2][pos]
        ptr = ptr.value if ptr is not None else None
        if ptr is None:
            return None
        for t in plan.keep:
            if torch.is_tensor(t) and t.data_ptr() == ptr and t.dtype == dtype and t.dim() == 2:
                return t
        raise AssertionError(f'argument {pos} of call {idx} not found')

    def cl(t):
        """NCHW oracle tensor -> channels-last rows [B*H*W][C]."""
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    ht = taps[depth]
    Kj = heat_ref.shape[1]
    # scan the backward half: (call index, label, output, oracle output grad, [(input buffer, oracle input grad)])
    stages = []
    first = eng.n_fwd_calls
    blk, n_ln = depth - 1, 0
    n_bn = 0
    dheat_buf = ws['dheat']
    dheat_ref = torch.zeros(dheat_buf.shape)
    dheat_ref[:, :Kj] = cl(heat_ref.grad)
    for idx in range(first, len(plan)):
        name = plan.calls[idx][0]
        tag = plan.tags.get(idx)
        if n_ln == 0 and name == 'vp_gemm_bf16' and tag[2] == 7:
            stages.append((idx, 'head g1  = (dheat Wf) * relu\'', ws['hg'], cl(ht['bn1'].grad), [(dheat_buf, dheat_ref)],
                           (ws['a.1'], cl(ht['bn1']))))
        elif n_ln == 0 and name == 'vp_bn_bwd':
            l = 1 - n_bn
            stages.append((idx, f'head dz{l} = BatchNorm backward', arg_tensor(idx, 7, torch.bfloat16),
                           cl(ht[f'deconv{l}'].grad), [(arg_tensor(idx, 0, torch.bfloat16), cl(ht[f'bn{l}'].grad))]))
            n_bn += 1
        elif n_ln == 0 and name == 'vp_deconv_dgrad_bf16':
            l = 2 - n_bn
            ref = cl(ht['bn0'].grad) if l == 1 else tok.grad.reshape(M, E)
            stages.append((idx, f'head g0  = (deconv1 data gradient) * relu\'' if l else 'dfeat    = deconv0 data gradient',
                           arg_tensor(idx, 2, torch.bfloat16), ref,
                           [(arg_tensor(idx, 0, torch.bfloat16), cl(ht[f'deconv{l}'].grad))],
                           (ws['a.0'], cl(ht['bn0'])) if l else None))
        if name == 'vp_layernorm_bwd':
            dy, dres, dx = arg_tensor(idx, 0, torch.bfloat16), arg_tensor(idx, 5, torch.float32), arg_tensor(idx, 6, torch.float32)
            if n_ln == 0:
                stages.append((idx, 'dx     = last_norm backward', dx, xin[depth].grad, [(dy, tok.grad)]))
            else:
                b_ = depth - 1 - (n_ln - 1) // 2
                if (n_ln - 1) % 2 == 0:
                    stages.append((idx, f'[{b_}] dx_mid = LN2 backward + residual', dx, taps[b_]['x_mid'].grad,
                                   [(dy, taps[b_]['ln2'].grad), (dres, xin[b_ + 1].grad)]))
                else:
                    stages.append((idx, f'[{b_}] dx_in  = LN1 backward + residual', dx, xin[b_].grad,
                                   [(dy, taps[b_]['ln1'].grad), (dres, taps[b_]['x_mid'].grad)]))
                    blk = b_ - 1
            n_ln += 1
            continue
        if n_ln == 0 or blk < 0:
            continue
        t, sfx = taps[blk], f'.{blk}'
        if name == 'vp_gemm_bf16' and tag[2] == 3:
            stages.append((idx, f"[{blk}] du     = (d_mlp W2) * gelu'(u)", ws['du' + sfx], t['fc1_pre'].grad,
                           [(ws[f'dxb.{blk}.mlp'], xin[blk + 1].grad)]))
        elif name == 'vp_gemm_bf16' and tag[1] == 1 and tag[2] == 0 and tag[5] == Hd:
            stages.append((idx, f'[{blk}] dxn2   = du W1', ws['dxn'], t['ln2'].grad, [(ws['du' + sfx], t['fc1_pre'].grad)]))
        elif name == 'vp_gemm_bf16' and tag[1] == 1 and tag[2] == 0 and tag[5] == E and tag[4] == E:
            stages.append((idx, f'[{blk}] dao    = d_attn Wproj', ws['dao'], t['attn_o'].grad,
                           [(ws[f'dxb.{blk}.attn'], t['x_mid'].grad)]))
        elif name == 'vp_attention_bwd':
            stages.append((idx, f'[{blk}] dqkv   = attention backward', ws['dqkv' + sfx], packed_qkv_grad(blk),
                           [(ws['dao'], t['attn_o'].grad)]))
        elif name == 'vp_gemm_bf16' and tag[1] == 1 and tag[2] == 0 and tag[5] == 3 * E:
            stages.append((idx, f'[{blk}] dxn1   = dqkv Wqkv', ws['dxn'], t['ln1'].grad, [(ws['dqkv' + sfx], packed_qkv_grad(blk))]))
    assert len(stages) == 5 + 1 + 7 * depth, [s_[1] for s_ in stages]
    by_idx = {s_[0]: s_ for s_ in stages}

    def replay(inject):
        """The backward half call by call; with `inject`, every stage's input gradients are first overwritten with
        the oracle's, so its output error is what the stage itself adds."""
        eng.store.grad.zero_()
        errs = []
        same.clear()
        for idx in range(first, len(plan)):
            st = by_idx.get(idx)
            if st is not None and inject:
                for buf, ref in st[4]:
                    if buf is not None:
                        buf.copy_(ref.reshape(buf.shape).to(buf.device))
            plan.run(first=idx, last=idx + 1)
            if st is not None:
                torch.cuda.synchronize()
                errs.append(rel(st[2].float(), st[3].reshape(st[2].shape)))
                if len(st) > 5 and st[5] is not None:
                    act, pre_ref = st[5]
                    agree = (act.float().cpu() > 0) == (pre_ref.reshape(act.shape) > 0)
                    got, want = st[2].float().cpu() * agree, st[3].reshape(st[2].shape) * agree
                    same[idx] = (rel(got, want), 1.0 - agree.float().mean().item())
        torch.cuda.synchronize()
        return errs

    same = {}

    acc = replay(False)
    gerr = {}
    for n in eng.store.names:
        key = n.split('.', 1)[1]
        src = bbr if n.startswith('backbone.') else hdr
        if key in src and src[key].grad is not None:
            gerr[n] = rel(eng.store.gview(n), src[key].grad)
    heat_err = rel(heat.cpu(), heat_ref.detach())
    dheat_err = rel(dheat_buf.float(), dheat_ref)
    own = replay(True)
    print(f'\nbackward stage errors, E = {E}, {depth} blocks, batch {B} (relative L2 vs fp32 autograd)')
    print(f'  {"stage":44s} {"accumulated":>12s} {"own":>10s} {"same-mask":>10s} {"masks differ":>13s}')
    print(f'  {"heat maps (train-mode forward)":44s} {heat_err:12.2e}')
    print(f'  {"dheat    = loss gradient":44s} {dheat_err:12.2e}')
    for st, a, o in zip(stages, acc, own):
        extra = f' {same[st[0]][0]:10.2e} {same[st[0]][1] * 100:12.2f}%' if st[0] in same else ''
        print(f'  {st[1]:44s} {a:12.2e} {o:10.2e}{extra}')
    worst = sorted(gerr.items(), key=lambda kv: -kv[1])[:6]
    print('  worst parameter gradients: ' + ', '.join(f'{k} {v:.2e}' for k, v in worst))
    # measured + 50 %: a smooth stage adds <= 5.1e-3 of its own; a stage behind a ReLU <= 5.5e-2, of which <= 4e-3 on the
    # elements whose mask agrees, with <= 0.35 % of the masks flipped; accumulated <= 9.1e-2; worst parameter 8.6e-2
    for st, o in zip(stages, own):
        if st[0] in same:
            assert o < 8.3e-2 and same[st[0]][0] < 6e-3 and same[st[0]][1] < 5.5e-3, (st[1], o, same[st[0]])
        else:
            assert o < 7.7e-3, (st[1], o)
    assert heat_err < 1e-2 and dheat_err < 8e-3
    assert max(acc) < 0.135 and max(gerr.values()) < 0.13, (acc, worst)
