"""The N > 1 path on CPU: two gloo ranks drive the gradient-bucket schedule exactly as the backward
plan does (after_block triggers in reverse block order, then finish) over a CPU ParamStore, and the
result must equal the sum over ranks with every element reduced exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _HostPlan:
    """Stands in for a launch plan: runs the recorded host closures in order."""

    def __init__(self):
        self.calls = []

    def host(self, fn):
        self.calls.append(fn)

    def run(self):
        for fn in self.calls:
            fn()


def _worker(rank, world, port, depth, bpb, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vitpose_amd.ddp import GradBuckets, reduce_scalars, shard_batch
        from vitpose_amd.engine import ParamStore, vit_param_shapes
        from vitpose_amd.head_engine import head_param_shapes
        vcfg = dict(img_size=(256, 192), patch_size=16, embed_dim=64, depth=depth, num_heads=2, mlp_ratio=4, qkv_bias=True)
        hp, _ = head_param_shapes(dict(in_channels=64, out_channels=17, num_deconv_layers=2, num_deconv_filters=(256, 256),
                                       num_deconv_kernels=(4, 4), extra=dict(final_conv_kernel=1)))
        named = [('backbone.' + n, s) for n, s in vit_param_shapes(vcfg)] + [('keypoint_head.' + n, s) for n, s in hp]
        st = ParamStore(named, 'cpu')
        g = torch.Generator().manual_seed(100 + rank)
        st.grad.copy_(torch.randn(st.numel, generator=g))
        mine = st.grad.clone()
        comm = GradBuckets(st, depth=depth, blocks_per_bucket=bpb)
        plan = _HostPlan()
        for i in range(depth - 1, 0, -1):      # the order VitRunner.record_backward fires them in
            comm.after_block(plan, i)
        comm.finish(plan)
        # the bucket ranges tile [0, numel) exactly once
        spans = sorted(list(comm.triggers.values()) + [comm.tail])
        assert spans[0][0] == 0 and spans[-1][1] == st.numel
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        plan.run()
        others = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(others, mine)
        assert torch.allclose(st.grad, sum(others), atol=1e-6)
        assert comm.grad_scale == 1.0 / world
        v = reduce_scalars(torch.tensor([float(rank), 2.0]))
        assert torch.allclose(v, torch.tensor([(world - 1) / 2.0, 2.0]))
        assert shard_batch(8 * world, rank, world) == (8 * rank, 8 * rank + 8)
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('depth,bpb', [(12, 3), (5, 2), (2, 3)])
def test_gradient_buckets_two_gloo_ranks(depth, bpb):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, depth, bpb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res
