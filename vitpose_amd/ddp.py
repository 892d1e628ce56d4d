"""Batch-sharded data parallelism: one process per GPU, parameters replicated, the flat
gradient buffer all-reduced over RCCL (torch.distributed backend "nccl") in a few large
contiguous buckets that are launched from inside the backward plan, in the order the
backward pass completes them (head first, then blocks depth-1 .. 0), on a side stream so the
exchange overlaps the remaining backward kernels.

This replaces the reference's MMDistributedDataParallel wrapper (mmpose/apis/train.py:116-133:
bucketed gradient all-reduce, averaged over ranks, `broadcast_buffers=False` so BatchNorm
statistics stay per shard).  The per-iteration all-reduce of the logged loss scalars
(mmpose/models/detectors/base.py:66-72) is available as `reduce_scalars`, to be called only on
logging iterations.
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, store, backbone_prefix='backbone.', head_prefix='keypoint_head.', depth=12,
                 blocks_per_bucket=4, process_group=None, split_tail=True):
        self.store = store
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.on_gpu = store.device.type == 'cuda'
        self.stream = torch.cuda.Stream(device=store.device) if self.on_gpu else None
        # asynchronous, stream-ordered collectives (overlapped with the rest of backward) exist with RCCL only
        self.overlap = self.on_gpu and dist.is_initialized() and dist.get_backend(process_group) == 'nccl'
        if self.overlap and os.environ.get('TORCH_NCCL_BLOCKING_WAIT', '0') == '1':
            import warnings
            warnings.warn('TORCH_NCCL_BLOCKING_WAIT=1 makes every all_reduce block the host until it has completed: the '
                          'gradient exchange is then serialised with the backward pass instead of overlapping it')
        self.pending = []          # (kept for callers that look at it: always empty, no Work handles are held)
        self.launched = 0          # collectives enqueued on the communication stream since the last wait()
        self.measure = False       # True: wait() brackets itself with timing events (bench.py --gpus N)
        self.timings = []
        self._t_first = None
        # VITPOSE_DDP_BF16=1 (off by default: the reference all-reduces f32): the buckets travel as bf16 -- half the
        # bytes over xGMI -- and are widened back into the f32 gradient buffer once the collective is done
        self.bf16 = os.environ.get('VITPOSE_DDP_BF16', '0') == '1' and self.on_gpu
        self.stage = torch.empty(store.numel, dtype=torch.bfloat16, device=store.device) if self.bf16 else None
        self.staged = []          # (lo, hi) of buckets that wait to be widened back
        # flat layout (forward order): pos/patch, block 0 .. depth-1, last_norm, head
        off = store.offsets
        end = store.numel

        def start_of(prefix):
            for n in store.names:
                if n.startswith(prefix):
                    return off[n]
            return end
        self.bounds = {}  # block index i -> offset where block i starts
        for i in range(depth):
            self.bounds[i] = start_of(f'{backbone_prefix}blocks.{i}.')
        # trigger points during backward: after block i is finished, everything from block i to the
        # end of the previously sent region is final
        self.triggers = {}
        hi = end
        i = depth - 1
        while i >= blocks_per_bucket:
            lo_blk = i - blocks_per_bucket + 1
            self.triggers[lo_blk] = (self.bounds[lo_blk], hi)
            hi = self.bounds[lo_blk]
            i = lo_blk - 1
        # the last bucket cannot overlap anything (backward is over when it is final), so the remaining
        # blocks are cut once more: only the lower half + patch/pos embedding is sent after backward ends
        if split_tail and i >= 3:
            lo_blk = (i + 1) // 2
            self.triggers[lo_blk] = (self.bounds[lo_blk], hi)
            hi = self.bounds[lo_blk]
        self.tail = (0, hi)

    # -- called from plan.host closures (i.e. in stream order of the compute stream) --
    # RCCL path: everything is STREAM-ordered and issued by the one thread that replays the plan, in plan order
    # (identical on every rank).  A bucket is handed to the communication stream by an event; the collective is
    # enqueued there (torch's NCCL backend makes its own stream wait for the stream that is current at the call and
    # makes that stream wait for the result: no Work handle is kept, nothing ever blocks the host); at the end of
    # the plan the compute stream waits for ONE event recorded behind the last collective.  There is no host-side
    # wait on communication and no second thread, so ranks cannot disagree on the order of collectives and a stalled
    # peer shows up as a stalled stream (RCCL's watchdog), not as two hosts waiting for each other.
    def _launch(self, lo, hi):
        if self.world == 1 or hi <= lo:
            return
        g = self.store.grad[lo:hi]
        if self.bf16:
            self._cast(self.store.grad, self.stage, lo, hi)      # on the compute stream, before the hand-over event
            g = self.stage[lo:hi]
            self.staged.append((lo, hi))
        if self.on_gpu and not self.overlap:
            # a backend other than RCCL (gloo, when several ranks rehearse on one GPU; tests only): reduced
            # synchronously at the trigger point, no overlap
            torch.cuda.current_stream().synchronize()
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            return
        if self.on_gpu:
            compute = torch.cuda.current_stream()
            ev = torch.cuda.Event(enable_timing=self.measure and not self.launched)
            ev.record(compute)
            if self.measure and not self.launched:
                self._t_first = ev
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            self.launched += 1
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)

    def after_block(self, plan, i):
        if i in self.triggers:
            lo, hi = self.triggers[i]
            plan.host(lambda lo=lo, hi=hi: self._launch(lo, hi))

    def finish(self, plan):
        lo, hi = self.tail
        plan.host(lambda: self._launch(lo, hi))
        plan.host(self.wait)

    def wait(self):
        """Make the compute stream wait for every outstanding bucket (no host wait)."""
        if self.launched:
            compute = torch.cuda.current_stream()
            done = torch.cuda.Event(enable_timing=self.measure)
            done.record(self.stream)                  # behind the last collective on the communication stream
            if self.measure:
                # exposed = what the compute stream spends in this wait; busy = first hand-over .. last collective done
                before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                before.record(compute)
                compute.wait_event(done)
                after.record(compute)
                self.timings.append((self._t_first, before, after, done))
            else:
                compute.wait_event(done)
            self.launched = 0
        for lo, hi in self.staged:
            self._cast(self.stage, self.store.grad, lo, hi)
        self.staged.clear()

    def exposed_ms(self):
        """[(exposed, busy)] in ms per recorded step (call after a synchronize; needs measure = True): `exposed` is
        the time the compute stream waited for the gradient exchange after backward had ended, `busy` the span from
        the first bucket's hand-over to the completion of the last collective."""
        out = [(b.elapsed_time(a), f.elapsed_time(d)) for f, b, a, d in self.timings]
        self.timings.clear()
        return out

    def _cast(self, src, dst, lo, hi):
        """dst[lo:hi] = (cast) src[lo:hi] on the current stream (f32 <-> bf16; flat buffers)."""
        import ctypes
        from . import _lib
        L = _lib.lib()
        n = hi - lo
        s_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        a = ctypes.c_void_p(src.data_ptr() + lo * src.element_size())
        b = ctypes.c_void_p(dst.data_ptr() + lo * dst.element_size())
        if src.dtype == torch.float32:
            rc = L.vp_cast_f32_bf16(a, b, n, s_)
        else:
            dims, strides = (ctypes.c_int * 4)(1, 1, 1, n), (ctypes.c_long * 4)(0, 0, 0, 1)
            rc = L.vp_strided_cast(a, _lib.VP_BF16, b, _lib.VP_F32, ctypes.byref(dims), ctypes.byref(strides), s_)
        if rc != 0:
            raise RuntimeError(f'gradient cast failed with status {rc}')

    @property
    def grad_scale(self):
        return 1.0 / self.world


def sync_tune_cache(device, group=None) -> bool:
    """Every rank times its GEMM tile schedules on its own GPU when it records a plan, and two GPUs can disagree on
    a close call; ranks that run different schedules finish their steps at different times and the fastest waits in
    every all-reduce.  Rank 0's picks are therefore broadcast and adopted by all (a collective: every rank calls it at
    the same point, right after recording a plan).  Returns True when this rank's picks changed (its plan has to be
    recorded again, now without timing anything)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return False
    from .plan import export_tune_cache, import_tune_cache
    box = [export_tune_cache(device) if dist.get_rank(group) == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return import_tune_cache(box[0], device)


def reduce_scalars(values: torch.Tensor, group=None) -> torch.Tensor:
    """Average a small tensor of logged scalars over ranks (one collective for all of them)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
        values /= dist.get_world_size(group)
    return values


def shard_batch(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous equal shards of the global batch (weak scaling: n_total = world * per_gpu).  Equal shards are what
    makes the averaged gradient the gradient of the global mean, so a remainder is an error, not a silent drop."""
    if n_total % world:
        raise ValueError(f'a global batch of {n_total} images does not split evenly over {world} ranks')
    per = n_total // world
    return rank * per, (rank + 1) * per
