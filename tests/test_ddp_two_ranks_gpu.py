"""Batch-sharded data parallelism of the HIP training plan with TWO ranks on one MI355X (gloo carries the
collectives between the two processes; more GPUs are not available to the tests): every rank runs the real
backward plan with its gradient buckets on its own half of the batch.  Checked: the reduced gradient buffer equals
the sum of the two ranks' stand-alone gradients (BatchNorm statistics per shard, as with broadcast_buffers=False in
the reference, mmpose/apis/train.py:129-133), both ranks hold bit-identical weights after the fused AdamW step
(1/world folded in), and that step equals a torch AdamW step on the averaged gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=64, depth=6, num_heads=2,
           mlp_ratio=4, qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.0)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, bf16=False):
    sys.path.insert(0, ROOT)
    if bf16:
        os.environ['VITPOSE_DDP_BF16'] = '1'
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    try:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from oracle import vitpose_oracle as O
        from vitpose_amd.pose_engine import PoseEngine
        dev = torch.device('cuda:0')
        hc = O.classic_head_cfg(64)
        bb, hd = O.synth_vit_weights(TOY, seed=5), O.synth_head_weights(hc, seed=6)
        sd = {'backbone.' + k: v for k, v in bb.items()}
        sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
        img, target, tw = O.synth_batch(4 * world, seed=7)
        lo, hi = 4 * rank, 4 * rank + 4
        x, t, w = img[lo:hi].to(dev), target[lo:hi].to(dev), tw[lo:hi].to(dev)

        def make():
            e = PoseEngine(TOY, hc, dev)
            e.load_state_dict(sd)
            return e
        # stand-alone gradients of this rank's shard
        solo = make()
        solo.forward_backward(x, t, w, drop_path=False)
        torch.cuda.synchronize()
        mine = solo.store.grad.clone()
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        want = sum(parts)
        # the bucketed plan
        eng = make()
        comm = eng.enable_ddp(blocks_per_bucket=2)
        assert comm.world == world and sorted(comm.triggers) == [2, 4] and comm.bf16 == bf16
        eng.setup_optimizer(lr=1e-3, weight_decay=0.1, layer_decay_rate=0.75, max_grad_norm=1.0)
        w0 = eng.store.data.clone()
        eng.forward_backward(x, t, w, drop_path=False)
        torch.cuda.synchronize()
        assert comm.launched == 0 and not comm.staged
        got = eng.store.grad.clone()
        for n in eng.store.names:
            a, b = eng.store._v(got, n), eng.store._v(want, n)
            rel = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
            assert rel < 2e-2, (n, rel)
        eng.optimizer_step()
        torch.cuda.synchronize()
        w1 = eng.store.data.clone()
        # every rank ends with the same weights, bit for bit
        allw = [torch.zeros_like(w1) for _ in range(world)]
        dist.all_gather(allw, w1)
        allg = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(allg, got)
        bad_g = [n for n in eng.store.names if not torch.equal(eng.store._v(allg[0], n), eng.store._v(allg[1], n))]
        bad_w = [n for n in eng.store.names if not torch.equal(eng.store._v(allw[0], n), eng.store._v(allw[1], n))]
        assert not bad_g, ('reduced gradients differ between the ranks', bad_g[:8], len(bad_g))
        assert not bad_w, ('weights differ between the ranks', bad_w[:8], len(bad_w))
        assert all(torch.equal(allw[0], a) for a in allw[1:])
        # ... and they are a torch AdamW step (clip at 1.0, layer decay) on the AVERAGED gradients
        from vitpose_amd.optim import layer_decay_param_groups
        groups = layer_decay_param_groups(eng.named_shapes(), TOY['depth'], 0.75, 1e-3, 0.1)
        gavg = (got / world).cpu()
        clip = min(1.0, 1.0 / (gavg.norm().item() + 1e-6))
        worst = 0.0
        for n in eng.store.names:
            p0 = eng.store._v(w0, n).cpu().clone().requires_grad_()
            p0.grad = eng.store._v(gavg, n).clone() * clip
            opt = torch.optim.AdamW([p0], lr=groups[n]['lr'], betas=(0.9, 0.999), eps=1e-8,
                                    weight_decay=groups[n]['weight_decay'])
            opt.step()
            d_ref = (p0.detach() - eng.store._v(w0, n).cpu())
            d_got = (eng.store._v(w1, n) - eng.store._v(w0, n)).cpu()
            err = (d_got - d_ref).abs().max().item() / max(d_ref.abs().max().item(), 1e-12)
            worst = max(worst, err)
            assert err < 2e-3, (n, err)
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-1500:]))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('bf16', [False, True])
def test_two_ranks_share_one_gpu(bf16):
    """bf16 = True: the opt-in VITPOSE_DDP_BF16 exchange (buckets travel as bf16 and are widened back)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, bf16)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res
