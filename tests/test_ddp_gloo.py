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
        # tile-schedule picks: every rank ends up with rank 0's, whatever it measured itself
        from vitpose_amd import plan as plan_mod
        from vitpose_amd.ddp import sync_tune_cache
        plan_mod._TUNE_CACHE.clear()
        plan_mod._TUNE_CACHE[('gemm', 12288, 768, 768, 0, 0, 2, True, False, 'cpu', 0, 0)] = 3 + rank
        plan_mod._TUNE_CACHE[('wgrad_group', ((768, 768), (3072, 768)), 12288, 'cpu')] = 7 - 4 * rank
        plan_mod._TUNE_CACHE[('gemm', 1, 2, 3, 0, 0, 0, False, False, 'cuda:7', 0, 0)] = 9      # another device: untouched
        changed = sync_tune_cache(torch.device('cpu'))
        assert changed == (rank != 0)
        assert plan_mod._TUNE_CACHE[('gemm', 12288, 768, 768, 0, 0, 2, True, False, 'cpu', 0, 0)] == 3
        assert plan_mod._TUNE_CACHE[('wgrad_group', ((768, 768), (3072, 768)), 12288, 'cpu')] == 7
        assert plan_mod._TUNE_CACHE[('gemm', 1, 2, 3, 0, 0, 0, False, False, 'cuda:7', 0, 0)] == 9
        assert sync_tune_cache(torch.device('cpu')) is False            # a second exchange changes nothing
        # the logged scalars of one iteration (detectors/base.py:38-76): averaged over ranks in ONE collective
        from vitpose_amd.models import TopDown
        calls, real = [], dist.all_reduce

        def counted(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        dist.all_reduce = counted
        try:
            loss, log_vars = TopDown._parse_losses(dict(heatmap_loss=torch.tensor(1.0 + rank, requires_grad=True),
                                                        aux_loss=[torch.tensor([2.0, 4.0]), torch.tensor(float(rank))],
                                                        acc_pose=0.25 + rank))
        finally:
            dist.all_reduce = real
        assert calls == [3], calls                               # heatmap_loss, aux_loss, loss: one vector, one call
        assert loss.requires_grad and abs(float(loss) - (1.0 + rank + 3.0 + rank)) < 1e-6     # the local loss, untouched
        mean_rank = (world - 1) / 2.0
        assert log_vars['acc_pose'] == 0.25 + rank              # a float is logged as it is (base.py:47-48)
        assert abs(log_vars['heatmap_loss'] - (1.0 + mean_rank)) < 1e-6
        assert abs(log_vars['aux_loss'] - (3.0 + mean_rank)) < 1e-6
        assert abs(log_vars['loss'] - (4.0 + 2 * mean_rank)) < 1e-6
        assert list(log_vars) == ['heatmap_loss', 'aux_loss', 'acc_pose', 'loss']
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


def test_parse_losses_without_process_group():
    """detectors/base.py:38-76 on one process: tensors are averaged, lists of tensors summed, every key containing
    'loss' enters the total, a float is logged unchanged, anything else is a TypeError."""
    from vitpose_amd.models import TopDown
    loss, log_vars = TopDown._parse_losses(dict(heatmap_loss=torch.tensor([1.0, 3.0]), acc_pose=0.5,
                                                other_loss=[torch.tensor(1.0), torch.tensor([2.0, 4.0])]))
    assert float(loss) == 6.0 and log_vars == dict(heatmap_loss=2.0, acc_pose=0.5, other_loss=4.0, loss=6.0)
    with pytest.raises(TypeError):
        TopDown._parse_losses(dict(heatmap_loss='nan'))


def test_rccl_branch_is_stream_ordered(monkeypatch):
    """The overlapped (RCCL) branch of GradBuckets with the device API replaced by recorders: every collective is
    enqueued under the communication stream, behind a wait for an event recorded on the compute stream at the
    bucket's trigger point; the compute stream waits exactly once, for an event recorded behind the LAST
    collective; the bf16 buckets are widened back only after that wait; nothing blocks the host."""
    from vitpose_amd import ddp
    log = []

    class Stream:
        def __init__(self, name):
            self.name = name

        def wait_event(self, ev):
            log.append(('wait', self.name, ev.id))

        def synchronize(self):
            log.append(('host_sync', self.name))

    class Event:
        n = 0

        def __init__(self, enable_timing=False):
            Event.n += 1
            self.id = Event.n

        def record(self, stream):
            log.append(('record', stream.name, self.id))

        def synchronize(self):
            log.append(('host_sync', 'event'))

    compute, comm_stream = Stream('compute'), Stream('comm')
    current = [compute]

    class StreamCtx:
        def __init__(self, st):
            self.st = st

        def __enter__(self):
            self.prev, current[0] = current[0], self.st

        def __exit__(self, *exc):
            current[0] = self.prev

    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: current[0])
    monkeypatch.setattr(torch.cuda, 'Event', Event)
    monkeypatch.setattr(torch.cuda, 'stream', StreamCtx)
    monkeypatch.setattr(ddp.dist, 'all_reduce',
                        lambda t, op=None, group=None: log.append(('all_reduce', current[0].name, t.numel(), t.dtype)))

    class Store:
        grad = torch.zeros(1000)
        numel = 1000

    for bf16 in (False, True):
        log.clear()
        gb = object.__new__(ddp.GradBuckets)
        gb.store, gb.group, gb.world, gb.on_gpu, gb.stream, gb.overlap = Store, None, 2, True, comm_stream, True
        gb.pending, gb.launched, gb.measure, gb.timings, gb._t_first = [], 0, False, [], None
        gb.bf16, gb.stage, gb.staged = bf16, torch.zeros(1000, dtype=torch.bfloat16), []
        gb._cast = lambda src, dst, lo, hi: log.append(('cast', current[0].name, str(src.dtype), lo, hi))
        gb.triggers, gb.tail = {2: (600, 1000), 1: (300, 600)}, (0, 300)
        plan = _HostPlan()
        gb.after_block(plan, 2)
        gb.after_block(plan, 1)
        gb.finish(plan)
        plan.run()
        assert not [e for e in log if e[0] == 'host_sync']
        ars = [i for i, e in enumerate(log) if e[0] == 'all_reduce']
        assert len(ars) == 3 and all(log[i][1] == 'comm' for i in ars)
        assert [log[i][2] for i in ars] == [400, 300, 300]
        assert all(log[i][3] == (torch.bfloat16 if bf16 else torch.float32) for i in ars)
        for i in ars:            # hand-over: record on compute, comm waits for that event, then the collective
            assert log[i - 2][:2] == ('record', 'compute') and log[i - 1] == ('wait', 'comm', log[i - 2][2])
            if bf16:             # narrowed on the compute stream BEFORE the hand-over event
                assert log[i - 3][:3] == ('cast', 'compute', 'torch.float32')
        tail = log[ars[-1] + 1:]
        assert tail[0][:2] == ('record', 'comm') and tail[1] == ('wait', 'compute', tail[0][2])
        assert len([e for e in log if e[0] == 'wait' and e[1] == 'compute']) == 1
        widen = [e for e in tail[2:] if e[0] == 'cast']
        assert len(widen) == (3 if bf16 else 0) and all(e[1] == 'compute' and e[2] == 'torch.bfloat16' for e in widen)
        assert gb.launched == 0 and not gb.staged
