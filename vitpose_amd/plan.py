"""Launch plans: a model pass is recorded ONCE as a flat list of C-ABI calls with
pre-bound device pointers, then replayed every step with no Python work per kernel
beyond the foreign call itself.  Tensors are only used for memory ownership
(torch allocates, the plan keeps them alive) and for the current HIP stream.
"""
import ctypes
import os
from ctypes import c_void_p

import torch

from . import _lib
from ._lib import GemmArgs


class KernelError(RuntimeError):
    pass


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, 'device tensor expected'
    return c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if t is None:
        return
    if t.dtype != dtype:
        raise TypeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name}: must be contiguous')


_TAIL_WS = {}
# Launch-shape choices measured once per problem signature on the device the plan is recorded on
# (tile schedule of vp_gemm_bf16, K split of the weight-gradient GEMMs); VITPOSE_AUTOTUNE=0 disables.
_TUNE_CACHE = {}
AUTOTUNE = os.environ.get('VITPOSE_AUTOTUNE', '1') != '0'
# VITPOSE_TUNE_OVERRIDE="<xk><yk>:<epilogue>:<N>:<K>=<tile_mode>,...": pins the tile schedule of a product class (whole-step
# A/B runs of one schedule against the record-time pick), e.g. "00:2:768:3072=2" = fc2 + residual on the 256 tiles
# VITPOSE_WGRAD_GROUP_TILE=2|3|7 pins the tile schedule of the grouped weight-gradient launch (256 / 192 / 192 with the
# K tile split between the wave groups) the same way
_GROUP_TILE_OVERRIDE = int(os.environ.get('VITPOSE_WGRAD_GROUP_TILE', '0'))
KSPLIT = os.environ.get('VITPOSE_KSPLIT', '1') == '1'      # tile_mode 7 among the autotune candidates
_TUNE_OVERRIDE = {k: int(v) for k, v in (kv.split('=') for kv in os.environ.get('VITPOSE_TUNE_OVERRIDE', '').split(',') if kv)}


def _time_launch(fn, reps=5):
    """`reps` back-to-back launches between two events (ms per launch)."""
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def _pick_fastest(cands, rounds=3):
    """cands: {key: launch function}.  Every candidate is warmed, then timed `rounds` times in round-robin
    order (clock ramps and neighbours' cache state hit all of them alike); the smallest time of each counts."""
    best = {}
    cands = {k: fn for k, fn in cands.items() if fn() in (None, 0)}   # a schedule may not cover the layout
    for fn in cands.values():
        fn()
    for _ in range(rounds):
        for k, fn in cands.items():
            t = _time_launch(fn)
            if k not in best or t < best[k]:
                best[k] = t
    return min(best, key=best.get)


def export_tune_cache(device):
    """The launch-shape picks recorded for `device`, keyed without the device (ranks of one job name their GPUs
    differently)."""
    d = str(device)
    return {tuple('<device>' if x == d else x for x in k): v for k, v in _TUNE_CACHE.items() if d in k}


def import_tune_cache(picks, device):
    """Adopt another process's picks (from export_tune_cache) for `device`; True if any pick of this process changed."""
    d, changed = str(device), False
    for k, v in picks.items():
        kk = tuple(d if x == '<device>' else x for x in k)
        if _TUNE_CACHE.get(kk) != v:
            _TUNE_CACHE[kk] = v
            changed = True
    return changed


_ZERO_PAGES = {}


def _zero_page(device):
    """256 zero bytes per device: the source of out-of-image taps of the implicit im2col."""
    key = str(device)
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros(128, dtype=torch.bfloat16, device=device)
    return _ZERO_PAGES[key]


def _tail_workspace(device):
    """One 32 MiB scratch per device for the GEMM tail-tile partial sums (launches are stream-ordered)."""
    key = (device.type, device.index)
    if key not in _TAIL_WS:
        _TAIL_WS[key] = torch.empty(8 << 20, dtype=torch.float32, device=device)
    return _TAIL_WS[key]


class Plan:
    def __init__(self):
        self.L = _lib.lib()
        self.calls = []     # (name, cfunc, args)
        self.keep = []      # tensors / ctypes objects that must outlive the plan
        self.tags = {}      # call index -> (x_kmajor, y_kmajor, epilogue, M, N, K) of GEMM launches
        self.use_tail_split = True
        self.timers = None  # a list: run() then brackets every GEMM launch with HIP events (see run_timed)
        self.timer_filter = None  # optional set of (x_kmajor, y_kmajor, epilogue): only these launches are bracketed
        # Side stream: calls recorded inside `with plan.on_side():` are launched on a second HIP stream
        # (weight-gradient GEMMs, off the backward pass's critical path); fork() / join() order the two.
        self.side_stream = None
        self._side = False
        self.side_calls = set()

    # -- replay --
    def run(self, first=0, last=None):
        """Replay calls [first, last) on torch's current HIP stream (set another with torch.cuda.stream(...): the
        host-side entries of a plan -- events, forks -- follow the current stream too)."""
        if self.timers is not None:
            return self.run_timed(self.timers, first, last)
        s = c_void_p(torch.cuda.current_stream().cuda_stream)
        if not self.side_calls:
            calls = self.calls if first == 0 and last is None else self.calls[first:last]
            for name, fn, args in calls:
                rc = fn(*args, s)
                if rc != 0:
                    raise KernelError(f'{name} failed with status {rc}')
            return
        s2 = c_void_p(self.side_stream.cuda_stream)
        side = self.side_calls
        for i in range(first, len(self.calls) if last is None else last):
            name, fn, args = self.calls[i]
            rc = fn(*args, s2 if i in side else s)
            if rc != 0:
                raise KernelError(f'{name} failed with status {rc}')

    def run_timed(self, timers, first=0, last=None):
        """Replay with a HIP event pair around every GEMM launch (events are recorded on the
        stream the kernels are launched on).  `timers` is a list that receives
        (gemm_tag, start_event, end_event); read elapsed times after a synchronize."""
        main = torch.cuda.current_stream()
        last = len(self.calls) if last is None else last
        for i in range(first, last):
            name, fn, args = self.calls[i]
            st = self.side_stream if i in self.side_calls else main
            s = c_void_p(st.cuda_stream)
            tag = self.tags.get(i)
            if tag is not None and self.timer_filter is not None and tag[:3] not in self.timer_filter:
                tag = None
            if tag is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                rc = fn(*args, s)
                e1.record(st)
                timers.append((tag, e0, e1))
            else:
                rc = fn(*args, s)
            if rc != 0:
                raise KernelError(f'{name} failed with status {rc}')

    def _add(self, name, *args, keep=()):
        if self._side:
            self.side_calls.add(len(self.calls))
        self.calls.append((name, getattr(self.L, name), args))
        self.keep.extend(keep)

    # -- second stream --
    def enable_side_stream(self, device):
        if self.side_stream is None:
            self.side_stream = torch.cuda.Stream(device=device)
        return self.side_stream

    def on_side(self):
        """Context manager: calls recorded inside go to the side stream (no-op without one)."""
        plan = self

        class _Ctx:
            def __enter__(self_):
                plan._side = plan.side_stream is not None

            def __exit__(self_, *exc):
                plan._side = False
        return _Ctx()

    def fork(self):
        """Side stream waits for everything recorded so far on the main stream."""
        if self.side_stream is None:
            return
        ev, side = torch.cuda.Event(), self.side_stream

        def f():
            ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
        self.calls.append(('fork', _Host(f), ()))

    def join(self):
        """Main stream waits for everything recorded so far on the side stream."""
        if self.side_stream is None:
            return
        ev, side = torch.cuda.Event(), self.side_stream

        def f():
            ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
        self.calls.append(('join', _Host(f), ()))

    def __len__(self):
        return len(self.calls)

    # -- recorded ops --
    def gemm(self, x, y, out, M, N, K, *, ldx, ldy, ldo, x_kmajor=False, y_kmajor=False,
             epilogue=_lib.VP_EPI_BF16, out2=None, bias=None, colscale=None, resid=None,
             rowscale=None, rows_per_scale=0, aux=None, ldaux=0, pos=None, pos_rows=0, split_k=1,
             row_map=None, split_stride=0, tail_force=False, tile_mode=0, route=None, gather16=None,
             split3=False, pair_act=0, ln_stats=None, ln_stats_out=None, ln_eps=1e-6):
        """gather16 = (H, W, C): y is a transposed conv's output-grid gradient [B*2H*2W][C] and the Y operand its
        reduction-major 16-tap gather (see vp_gemm_args in the header).
        route = (expert [images] int32, delta [experts] int64, bias_delta [experts] int64 or None, split):
        per-image routing of the Y operand (mixture-of-experts FFN; see vp_gemm_args in the header).
        split3: x and y are hi / lo pair arrays (rows of 2 K bf16) and the product is fp32-accurate (hi*hi + hi*lo +
        lo*hi per K step in one launch); epilogue VP_EPI_F32 / VP_EPI_RESID_F32 / VP_EPI_PAIR (out = pair array of
        act(acc * colscale + bias), pair_act 0 none / 1 exact GELU / 2 ReLU).
        LayerNorm fold (evaluation; vp_gemm_args.ln_stats in the header): ln_stats_out = f32 [M][slots >= ceil(N / 128)][2]
        makes a VP_EPI_RESID_F32 launch a PRODUCER (out2 = bf16 copy of the stream; returns the number of statistics
        slots per row its tile schedule writes); ln_stats = (that tensor, slots) makes a VP_EPI_BF16 / GELU launch the
        CONSUMER (x = the bf16 copy, y = folded weights, colscale = s, bias = c)."""
        a = GemmArgs()
        if ln_stats is not None:
            _chk(ln_stats[0], torch.float32, 'ln_stats')
            a.ln_stats, a.ln_slots, a.ln_eps = ln_stats[0].data_ptr(), int(ln_stats[1]), ln_eps
            self.keep.append(ln_stats[0])
        if ln_stats_out is not None:
            _chk(ln_stats_out, torch.float32, 'ln_stats_out')
            if out2 is None or ln_stats_out.numel() < 2 * M * ((N + 127) // 128):
                raise ValueError('a folded-LayerNorm producer needs out2 and statistics for ceil(N / 128) slots per row')
            a.ln_stats_out = ln_stats_out.data_ptr()
            self.keep.append(ln_stats_out)
        a.split3, a.pair_act = int(split3), int(pair_act)
        if gather16 is not None:
            zero = _zero_page(out.device)
            a.gather16_H, a.gather16_W, a.gather16_C = gather16
            a.gather16_zero = zero.data_ptr()
            self.keep.append(zero)
            tile_mode = 2
        if route is not None:
            r_exp, r_delta, r_bdelta, r_split = route
            _chk(r_exp, torch.int32, 'route expert'), _chk(r_delta, torch.int64, 'route delta')
            _chk(r_bdelta, torch.int64, 'route bias delta')
            a.route_expert, a.route_delta, a.route_split = r_exp.data_ptr(), r_delta.data_ptr(), int(r_split)
            a.route_bias_delta = r_bdelta.data_ptr() if r_bdelta is not None else None
            self.keep.extend(t for t in (r_exp, r_delta, r_bdelta) if t is not None)
            tile_mode = 0 if split3 else 3
        a.tail_force = int(tail_force)
        a.tile_mode = int(tile_mode)
        a.split_stride = split_stride
        a.M, a.N, a.K = M, N, K
        a.x, a.ldx, a.x_kmajor = x.data_ptr(), ldx, int(x_kmajor)
        a.y, a.ldy, a.y_kmajor = y.data_ptr(), ldy, int(y_kmajor)
        a.epilogue = epilogue
        a.out, a.ldo = out.data_ptr(), ldo
        for f, t in (('out2', out2), ('bias', bias), ('colscale', colscale), ('resid', resid),
                     ('rowscale', rowscale), ('aux', aux), ('pos', pos)):
            setattr(a, f, t.data_ptr() if t is not None else None)
        a.rows_per_scale, a.ldaux, a.pos_rows, a.split_k = rows_per_scale, ldaux, pos_rows, split_k
        if row_map is not None:
            a.out_row_div, a.out_row_hi, a.out_row_lo, a.out_row_off = row_map
        if self.use_tail_split and epilogue != _lib.VP_EPI_SPLITK_F32 and not split3 and not self._side:
            # (one scratch per device: safe for stream-ordered launches only, so side-stream calls go without)
            tw = _tail_workspace(out.device)
            a.tail_ws, a.tail_ws_bytes = tw.data_ptr(), tw.numel() * 4
            self.keep.append(tw)
        producer = ln_stats_out is not None
        if producer and tile_mode == 0:
            # A producer's tile schedule fixes the order in which a row's statistics are added up, so it is chosen by
            # the output width alone -- never by timing, never by the number of rows: an image's result must not depend
            # on the batch around it.  192-wide tiles where they divide the width (ViTPose-S / -B), else 128-wide
            # (ViTPose-L / -H).
            tile_mode = 3 if N % 192 == 0 else 6
            a.tile_mode = tile_mode
        if producer:
            a.ln_slots = self.L.vp_gemm_ln_slots(N, a.tile_mode)
        if (AUTOTUNE and tile_mode == 0 and epilogue != _lib.VP_EPI_SPLITK_F32 and M >= 512 and N >= 256 and K >= 256
                and route is None):
            key = ('gemm', M, N, K, int(x_kmajor), int(y_kmajor), epilogue, resid is not None, out2 is not None,
                   str(out.device), int(split3), int(pair_act), ln_stats is not None, producer)
            ov = _TUNE_OVERRIDE.get(f'{int(x_kmajor)}{int(y_kmajor)}:{epilogue}:{N}:{K}')
            if ov is not None and key not in _TUNE_CACHE:
                _TUNE_CACHE[key] = ov
            if key not in _TUNE_CACHE:
                fn = self.L.vp_gemm_bf16
                s_ = c_void_p(torch.cuda.current_stream().cuda_stream)

                def launch(mode):
                    def f():
                        a.tile_mode = mode
                        return fn(ctypes.byref(a), s_)
                    return f
                modes = [1, 2, 3] if x_kmajor else ([1, 2, 3, 4] if split3 else [1, 2, 3, 4, 6])
                if KSPLIT and not split3 and K >= 1536 and epilogue in (_lib.VP_EPI_BF16, _lib.VP_EPI_F32):
                    # K tile split between the wave groups: pays for long reductions only.  (Its residual epilogue wins the
                    # cache-hot timing done here -- fc2 + residual 55.3 against 58.9 us -- and loses inside the step, 50.5
                    # against 49.4 us, and 2 % of the ViTPose-H evaluation pass: not offered for VP_EPI_RESID_F32.)
                    modes.append(7)
                if (not x_kmajor and not split3 and K >= 640 and M % 192 == 0 and row_map is None and colscale is None
                        and ln_stats is None
                        and epilogue in (_lib.VP_EPI_BF16, _lib.VP_EPI_GELU, _lib.VP_EPI_DGELU)
                        and (M // 192) * ((N + 191) // 192) > 256):
                    modes.append(5)       # persistent 192-tile kernel: only where a workgroup gets several tiles
                _TUNE_CACHE[key] = _pick_fastest({m: launch(m) for m in modes})
            a.tile_mode = _TUNE_CACHE[key]
        if producer:
            a.ln_slots = self.L.vp_gemm_ln_slots(N, a.tile_mode)
        self.tags[len(self.calls)] = (int(x_kmajor), int(y_kmajor), epilogue + (200 if split3 else 0), M, N,
                                      3 * K if split3 else K)
        self._add('vp_gemm_bf16', ctypes.byref(a),
                  keep=(a, x, y, out, out2, bias, colscale, resid, rowscale, aux, pos))
        return a.ln_slots if producer else None

    def wgrad(self, x, y, out, M, N, K, *, ldx, ldy, split_k, slab, accumulate=False, tile_mode=0, gather16=None):
        """out[M][N] (f32) (+)= x^T y with x [K][M], y [K][N] (both reduction-major): the weight
        gradient of a Linear / conv-as-GEMM.  The reduction over K (= tokens) is cut into
        `split_k` slices that write f32 slabs, summed by vp_splitk_reduce (deterministic)."""
        if split_k <= 1 and not accumulate and gather16 is None:
            return self.gemm(x, y, out, M, N, K, ldx=ldx, ldy=ldy, ldo=N, x_kmajor=True, y_kmajor=True,
                             epilogue=_lib.VP_EPI_F32, tile_mode=tile_mode)
        split_k = max(1, min(split_k, K // 64))
        if gather16 is not None:
            # the gathered operand exists in the 256 x 256 kernel (one workgroup per CU): K slices for one round of the chip
            t256 = ((M + 255) // 256) * ((N + 255) // 256)
            tile_mode, split_k = 2, max(1, min(K // 64, 256 // t256, slab.numel() // (M * N)))
        elif AUTOTUNE and tile_mode == 0 and M >= 256 and N >= 256 and K >= 1024:
            key = ('wgrad', M, N, K, str(out.device))
            if key not in _TUNE_CACHE:
                nk = K // 64
                t256 = ((M + 255) // 256) * ((N + 255) // 256)
                t192 = ((M + 191) // 192) * ((N + 191) // 192)
                cands = {(1, split_k)}
                for units in (256, 512):
                    for mode, tiles in ((2, t256), (3, t192)):
                        sk = max(1, min(nk, units // tiles))
                        if sk * M * N <= slab.numel():
                            cands.add((mode, sk))
                probes = {}
                for mode, sk in sorted(cands):
                    probe = Plan()
                    probe.gemm(x, y, slab, M, N, K, ldx=ldx, ldy=ldy, ldo=N, x_kmajor=True, y_kmajor=True,
                               epilogue=_lib.VP_EPI_SPLITK_F32, split_k=sk, split_stride=M * N, tile_mode=mode)
                    probe._add('vp_splitk_reduce', _p(slab), sk, M * N, M * N, _p(out), 0, keep=(slab, out))
                    probes[(mode, sk)] = probe.run
                _TUNE_CACHE[key] = _pick_fastest(probes)
            tile_mode, split_k = _TUNE_CACHE[key]
        assert slab.numel() >= split_k * M * N
        self.gemm(x, y, slab, M, N, K, ldx=ldx, ldy=ldy, ldo=N, x_kmajor=True, y_kmajor=True,
                  epilogue=_lib.VP_EPI_SPLITK_F32, split_k=split_k, split_stride=M * N, tile_mode=tile_mode,
                  gather16=gather16)
        self._add('vp_splitk_reduce', _p(slab), split_k, M * N, M * N, _p(out), int(accumulate), keep=(slab, out))

    def wgrad_group(self, problems, K, tile_mode=0):
        """One launch for several independent weight gradients out_i [M_i][N_i] = x_i^T y_i over a common
        reduction extent K.  problems: list of (x [K][M] bf16, y [K][N] bf16, out f32, M, N, ldx, ldy[, rowsum
        [, image_mask, tiles_per_image]]); rowsum (f32 [M], optional) receives sum_k x[k][m], the bias gradient of
        the same Linear; image_mask (int64 device tensor of ceil(images / 64) words, optional): the K rows are images
        of tiles_per_image * 64 rows and the problem reduces over the images whose bit is set only (expert weight
        gradients: the tokens of the images routed to that expert)."""
        from ._lib import GroupEntry
        if len(problems) > 64:               # the kernel maps a workgroup to its problem with one 64-lane ballot
            self.wgrad_group(problems[:64], K, tile_mode)
            return self.wgrad_group(problems[64:], K, tile_mode)
        n = len(problems)
        dev = problems[0][2].device
        mapped = any(len(p) > 8 and p[8] is not None for p in problems)
        if mapped:
            tile_mode = 3

        def table(mode):
            arr = (GroupEntry * n)()
            for g, prob in zip(arr, problems):
                x, y, out, M, N, ldx, ldy = prob[:7]
                _chk(x, torch.bfloat16, 'x'), _chk(y, torch.bfloat16, 'y'), _chk(out, torch.float32, 'out')
                g.x, g.y, g.out, g.ldx, g.ldy, g.ldo, g.M, g.N = x.data_ptr(), y.data_ptr(), out.data_ptr(), ldx, ldy, N, M, N
                if len(prob) > 7 and prob[7] is not None:
                    _chk(prob[7], torch.float32, 'rowsum')
                    g.rowsum = prob[7].data_ptr()
                if len(prob) > 8 and prob[8] is not None:
                    _chk(prob[8], torch.int64, 'image_mask')
                    if K % (64 * prob[9]) or prob[8].numel() * 64 < K // (64 * prob[9]):
                        raise ValueError('image_mask: K must be whole images and the mask must cover them')
                    g.image_mask, g.tiles_per_image = prob[8].data_ptr(), int(prob[9])
            total = ctypes.c_int(0)
            rc = self.L.vp_wgrad_group_table(arr, n, mode, ctypes.byref(total))
            if rc != 0:
                raise KernelError(f'vp_wgrad_group_table -> {rc}')
            blob = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            return blob, total.value
        if tile_mode == 0 and _GROUP_TILE_OVERRIDE in (2, 3, 7):
            tile_mode = _GROUP_TILE_OVERRIDE
        if tile_mode == 0:
            key = ('wgrad_group', tuple((p[3], p[4]) for p in problems), K, str(dev))
            if AUTOTUNE and key not in _TUNE_CACHE:
                s_ = c_void_p(torch.cuda.current_stream().cuda_stream)
                gmodes = (2, 3, 7) if KSPLIT else (2, 3)
                tabs = {mode: table(mode) for mode in gmodes}
                _TUNE_CACHE[key] = _pick_fastest({
                    mode: (lambda mode=mode: self.L.vp_wgrad_grouped_bf16(_p(tabs[mode][0]), n, tabs[mode][1], K, mode, s_))
                    for mode in gmodes})
            tile_mode = _TUNE_CACHE.get(key, 3)
        blob, total = table(tile_mode)
        keep = [blob] + [t for p in problems for t in p[:3]] + [t for p in problems for t in p[7:9]]
        # timing tag in the (x_kmajor, y_kmajor, epilogue, M, N, K) form of gemm(): epilogue 100 = grouped,
        # M * N = summed output elements
        # (masked problems that share one x operand partition its K rows among them: counted once)
        seen, mn = set(), 0
        for p in problems:
            if len(p) > 8 and p[8] is not None:
                if p[0].data_ptr() in seen:
                    continue
                seen.add(p[0].data_ptr())
            mn += p[3] * p[4]
        self.tags[len(self.calls)] = (1, 1, 100, mn, 1, K)
        if mapped:
            self._add('vp_wgrad_grouped_kmap_bf16', _p(blob), n, total, K, keep=tuple(keep))
        else:
            self._add('vp_wgrad_grouped_bf16', _p(blob), n, total, K, tile_mode, keep=tuple(keep))

    def layernorm_fwd(self, x, gamma, beta, y_bf16, y_f32, mean, rstd, M, E, eps=1e-6):
        _chk(x, torch.float32, 'x')
        self._add('vp_layernorm_fwd', _p(x), _p(gamma), _p(beta), _p(y_bf16), _p(y_f32), _p(mean), _p(rstd),
                  M, E, eps, keep=(x, gamma, beta, y_bf16, y_f32, mean, rstd))

    def layernorm_bwd(self, dy, x, mean, rstd, gamma, dres, dx, dxb, rowscale, rows_per_scale,
                      dgamma, dbeta, dcolsum, M, E, workspace=None):
        if workspace is None:  # per-workgroup partial column sums
            workspace = torch.empty(3 * E * 512, dtype=torch.float32, device=x.device)
        self._add('vp_layernorm_bwd', _p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb),
                  _p(rowscale), rows_per_scale, _p(dgamma), _p(dbeta), _p(dcolsum), M, E, _p(workspace),
                  workspace.numel(),
                  keep=(dy, x, mean, rstd, gamma, dres, dx, dxb, rowscale, dgamma, dbeta, dcolsum, workspace))

    def ln_finalize(self, entries, M, E, overwrite=False):
        """One launch that adds the column sums several layernorm_bwd calls left in their workspaces (recorded with
        dgamma = dbeta = dcolsum = None) to their targets -- or, with `overwrite`, stores them there (every target then
        needs no zeroing).  entries: [(workspace, dgamma, dbeta, dcolsum)]."""
        from ._lib import LnPartial
        if not entries:
            return
        arr = (LnPartial * len(entries))()
        for t, (wsp, dg, db, dc) in zip(arr, entries):
            _chk(wsp, torch.float32, 'workspace')
            t.partial = wsp.data_ptr()
            t.blocks = self.L.vp_layernorm_bwd_blocks(M, E, wsp.numel())
            t.overwrite = int(overwrite)
            t.dgamma, t.dbeta, t.dcolsum = (x.data_ptr() if x is not None else None for x in (dg, db, dc))
        blob = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(entries[0][0].device)
        self._add('vp_layernorm_bwd_finalize', _p(blob), len(entries), E,
                  keep=(blob,) + tuple(x for e in entries for x in e))

    def ln_fold_weights(self, w, gamma, beta, bias, wg, s, c, N, K):
        """wg (bf16 [N][K]) = w * gamma, s [N] = row sums of wg, c [N] = w beta + bias (see vp_ln_fold_weights)."""
        _chk(w, torch.float32, 'w'), _chk(wg, torch.bfloat16, 'wg'), _chk(s, torch.float32, 's'), _chk(c, torch.float32, 'c')
        self._add('vp_ln_fold_weights', _p(w), _p(gamma), _p(beta), _p(bias), _p(wg), _p(s), _p(c), N, K,
                  keep=(w, gamma, beta, bias, wg, s, c))

    def attention_fwd(self, qkv, out, lse, B, H, d, tokens, scale):
        self._add('vp_attention_fwd', _p(qkv), _p(out), _p(lse), B, H, d, tokens, scale, keep=(qkv, out, lse))

    def attention_bwd(self, qkv, out_fwd, dout, lse, dqkv, B, H, d, tokens, scale):
        self._add('vp_attention_bwd', _p(qkv), _p(out_fwd), _p(dout), _p(lse), _p(dqkv), B, H, d, tokens, scale,
                  keep=(qkv, out_fwd, dout, lse, dqkv))

    def patch_im2col(self, img, col, B, C, H, W, patch, stride, pad):
        _chk(img, torch.float32, 'img')
        self._add('vp_patch_im2col', _p(img), _p(col), B, C, H, W, patch, stride, pad, keep=(img, col))

    # -- fp32-accurate path (hi / lo pair arrays; the products are gemm(..., split3=True)) --
    def split_pair(self, x, pair, n):
        _chk(x, torch.float32, 'x'), _chk(pair, torch.bfloat16, 'pair')
        assert pair.numel() >= 2 * n
        self._add('vp_split_pair_f32', _p(x), _p(pair), n, keep=(x, pair))

    def layernorm_fwd_pair(self, x, gamma, beta, y_pair, M, E, relu=False, y_f32=None, eps=1e-6):
        _chk(x, torch.float32, 'x'), _chk(y_pair, torch.bfloat16, 'y_pair')
        self._add('vp_layernorm_fwd_pair', _p(x), _p(gamma), _p(beta), _p(y_pair), _p(y_f32), int(relu), M, E, eps,
                  keep=(x, gamma, beta, y_pair, y_f32))

    def attention_split(self, qkv, out_pair, B, H, d, tokens, scale):
        _chk(qkv, torch.float32, 'qkv'), _chk(out_pair, torch.bfloat16, 'out_pair')
        self._add('vp_attention_split', _p(qkv), _p(out_pair), B, H, d, tokens, scale, keep=(qkv, out_pair))

    def patch_im2col_pair(self, img, col_pair, B, C, H, W, patch, stride, pad):
        _chk(img, torch.float32, 'img'), _chk(col_pair, torch.bfloat16, 'col_pair')
        self._add('vp_patch_im2col_pair', _p(img), _p(col_pair), B, C, H, W, patch, stride, pad, keep=(img, col_pair))

    def deconv_fwd_split(self, src_pair, wph_pair, out_pair, B, H, W, C, Cout, colscale, shift):
        zero = _zero_page(src_pair.device)
        self._add('vp_deconv_fwd_split', _p(src_pair), _p(wph_pair), _p(out_pair), B, H, W, C, Cout, _p(colscale),
                  _p(shift), _p(zero), keep=(src_pair, wph_pair, out_pair, colscale, shift, zero))

    def strided_cast(self, src, dst, dims, strides):
        dt = {torch.float32: _lib.VP_F32, torch.bfloat16: _lib.VP_BF16}
        d = (ctypes.c_int * 4)(*dims)
        s = (ctypes.c_long * 4)(*strides)
        self._add('vp_strided_cast', _p(src), dt[src.dtype], _p(dst), dt[dst.dtype], ctypes.byref(d),
                  ctypes.byref(s), keep=(src, dst, d, s))

    def colsum(self, a, out, M, N, lda):
        self._add('vp_colsum_bf16', _p(a), _p(out), M, N, lda, keep=(a, out))

    def batch_sum(self, x, out, B, n):
        self._add('vp_batch_sum_f32', _p(x), _p(out), B, n, keep=(x, out))

    def deconv_fwd(self, src, wph, out, B, H, W, C, Cout, epilogue=_lib.VP_EPI_BF16, colscale=None, shift=None):
        zero = _zero_page(src.device)
        self._add('vp_deconv_fwd_bf16', _p(src), _p(wph), _p(out), B, H, W, C, Cout, epilogue, _p(colscale), _p(shift),
                  _p(zero), keep=(src, wph, out, colscale, shift, zero))

    def deconv_dgrad(self, dout, wd, din, B, H, W, C, Cin, relu_aux=None):
        zero = _zero_page(dout.device)
        self._add('vp_deconv_dgrad_bf16', _p(dout), _p(wd), _p(din), B, H, W, C, Cin, _p(relu_aux), _p(zero),
                  keep=(dout, wd, din, relu_aux, zero))

    def deconv_weight_prep(self, w, wph, wd, Ci, Co):
        _chk(w, torch.float32, 'w')
        self._add('vp_deconv_weight_prep', _p(w), _p(wph), _p(wd), Ci, Co, keep=(w, wph, wd))

    def im2col3x3(self, src, col, B, H, W, C):
        _chk(src, torch.bfloat16, 'src'), _chk(col, torch.bfloat16, 'col')
        self._add('vp_im2col3x3_bf16', _p(src), _p(col), B, H, W, C, keep=(src, col))

    def col2im3x3(self, dcol, din, B, H, W, C, relu_aux=None):
        _chk(dcol, torch.bfloat16, 'dcol'), _chk(din, torch.bfloat16, 'din'), _chk(relu_aux, torch.bfloat16, 'relu_aux')
        self._add('vp_col2im3x3_bf16', _p(dcol), _p(din), _p(relu_aux), B, H, W, C, keep=(dcol, din, relu_aux))

    def resize_to_cl(self, src, dst, tabs, B, C, Hi, Wi, Ho, Wo, ld, coff):
        """tabs = (yi, yw, xi, xw) of head_engine.bilinear_tables, flattened, on the device."""
        _chk(src, torch.float32, 'src'), _chk(dst, torch.bfloat16, 'dst')
        self._add('vp_resize_to_cl_bf16', _p(src), _p(dst), *[_p(t) for t in tabs], B, C, Hi, Wi, Ho, Wo, ld, coff,
                  keep=(src, dst, tabs))

    def resize_to_cl_bwd(self, g, dx, ttabs, B, C, Hi, Wi, Ho, Wo, ld, coff):
        """ttabs = (tyi, tyw, txi, txw): the transposed lists [Hi][Ly], [Wi][Lx]."""
        _chk(g, torch.bfloat16, 'g'), _chk(dx, torch.float32, 'dx')
        self._add('vp_resize_to_cl_bwd', _p(g), _p(dx), *[_p(t) for t in ttabs], ttabs[0].shape[1], ttabs[2].shape[1],
                  B, C, Hi, Wi, Ho, Wo, ld, coff, keep=(g, dx, ttabs))

    def deconv_weight_pad4(self, w, w4, n, k):
        """w [n][k][k] (k 2 or 3) -> the equivalent 4 x 4 kernel w4 [n][4][4] (f32)."""
        _chk(w, torch.float32, 'w'), _chk(w4, torch.float32, 'w4')
        self._add('vp_deconv_weight_pad4', _p(w), _p(w4), n, k, keep=(w, w4))

    def deconv_phase_im2col(self, src, col, B, H, W, C, py, px):
        self._add('vp_deconv_phase_im2col', _p(src), _p(col), B, H, W, C, py, px, keep=(src, col))

    def deconv_out_gather16(self, dout, g, B, H, W, C):
        self._add('vp_deconv_out_gather16', _p(dout), _p(g), B, H, W, C, keep=(dout, g))

    def cast_f32_bf16(self, x, y, n):
        self._add('vp_cast_f32_bf16', _p(x), _p(y), n, keep=(x, y))

    def bn_stats(self, z, sums, M, C):
        self._add('vp_bn_stats', _p(z), _p(sums), M, C, keep=(z, sums))

    def bn_finalize(self, sums, M, C, gamma, beta, rmean, rvar, momentum, eps, scale, shift, smean, sinv):
        self._add('vp_bn_finalize', _p(sums), M, C, _p(gamma), _p(beta), _p(rmean), _p(rvar), momentum, eps,
                  _p(scale), _p(shift), _p(smean), _p(sinv),
                  keep=(sums, gamma, beta, rmean, rvar, scale, shift, smean, sinv))

    def bn_apply_relu(self, z, a, scale, shift, M, C):
        self._add('vp_bn_apply_relu', _p(z), _p(a), _p(scale), _p(shift), M, C, keep=(z, a, scale, shift))

    def bn_bwd(self, g, z, mean, invstd, gamma, dgamma, dbeta, dz, M, C, running_stats=False):
        """running_stats: mean / invstd are the running statistics (eval mode), constants of the backward pass."""
        self._add('vp_bn_bwd_eval' if running_stats else 'vp_bn_bwd', _p(g), _p(z), _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(dz), M, C,
                  keep=(g, z, mean, invstd, gamma, dgamma, dbeta, dz))

    def heatmap_loss(self, heat, ldh, heat_nchw, target, tw, loss, dheat, B, K, HW, loss_weight):
        self._add('vp_heatmap_loss', _p(heat), ldh, _p(heat_nchw), _p(target), _p(tw), _p(loss), _p(dheat),
                  B, K, HW, loss_weight, keep=(heat, heat_nchw, target, tw, loss, dheat))

    def mse_loss_nchw(self, pred, target, tw, loss, dpred, B, K, HW, loss_weight):
        self._add('vp_mse_loss_nchw', _p(pred), _p(target), _p(tw), _p(loss), _p(dpred), B, K, HW, loss_weight,
                  keep=(pred, target, tw, loss, dpred))

    def upconv_fwd(self, G, ldg, bias, out, tabs, B, K, kpad, Hg, Wg, Ho, Wo, ksize):
        yi, yw, xi, xw = tabs
        self._add('vp_upconv_fwd', _p(G), ldg, _p(bias), _p(out), _p(yi), _p(yw), _p(xi), _p(xw), B, K, kpad,
                  Hg, Wg, Ho, Wo, ksize, keep=(G, bias, out, yi, yw, xi, xw))

    def upconv_bwd(self, dout, dG, ldg, ttabs, B, K, kpad, Hg, Wg, Ho, Wo, ksize):
        tyi, tyw, txi, txw = ttabs
        self._add('vp_upconv_bwd', _p(dout), _p(dG), ldg, _p(tyi), _p(tyw), _p(txi), _p(txw), tyi.shape[1],
                  txi.shape[1], B, K, kpad, Hg, Wg, Ho, Wo, ksize, keep=(dout, dG, tyi, tyw, txi, txw))

    def plane_sum(self, x, out, B, K, HW):
        self._add('vp_plane_sum_f32', _p(x), _p(out), B, K, HW, keep=(x, out))

    def relu_bf16(self, x, y, n):
        self._add('vp_relu_bf16', _p(x), _p(y), n, keep=(x, y))

    def nchw_to_cl(self, src, dst, B, C, HW, ld):
        _chk(src, torch.float32, 'src')
        self._add('vp_nchw_to_cl_bf16', _p(src), _p(dst), B, C, HW, ld, keep=(src, dst))

    def argmax_rows(self, x, rows, L, idx, val):
        self._add('vp_argmax_rows', _p(x), rows, L, _p(idx), _p(val), keep=(x, idx, val))

    def sumsq(self, x, n, out):
        self._add('vp_sumsq_f32', _p(x), n, _p(out), keep=(x, out))

    def host(self, fn):
        """A host-side closure (tiny torch glue on the current stream) inside the plan order."""
        if self._side:
            side, inner = self.side_stream, fn

            def fn():
                with torch.cuda.stream(side):
                    inner()
        self.calls.append(('host', _Host(fn), ()))

    def memset0(self, t):
        """Zero a (contiguous) tensor on the stream the call is launched on."""
        assert t.is_contiguous()
        self._add('vp_zero', _p(t), t.numel() * t.element_size(), keep=(t,))

    def zero_small(self, tensors):
        """Zero several small f32 tensors with one launch (each a few KB at most)."""
        tensors = [t for t in tensors if t is not None]
        for i in range(0, len(tensors), 16):
            grp = tensors[i:i + 16]
            for t in grp:
                _chk(t, torch.float32, 'zero_small')
            ptrs = (c_void_p * len(grp))(*[t.data_ptr() for t in grp])
            sizes = (ctypes.c_int * len(grp))(*[t.numel() for t in grp])
            self._add('vp_zero_regions', ptrs, sizes, len(grp), keep=(ptrs, sizes) + tuple(grp))

    def slots_of(self, t):
        """The recorded pointer arguments that hold the address of tensor `t` -- the argument OBJECTS, to be taken
        right after recording, while every buffer of the plan still has an address of its own.  `rebind` then moves
        exactly these, whatever addresses later inputs happen to land on."""
        ptr = t.data_ptr()
        return [a for _, _, args in self.calls for a in args if isinstance(a, c_void_p) and a.value == ptr]

    @staticmethod
    def rebind(slots, new_ptr):
        """Point the argument slots of one input (from `slots_of`) at `new_ptr`: the caller's own tensor of the
        recorded shape is then read in place (saves the copy).  The caller keeps that tensor alive until the plan
        has run."""
        for a in slots:
            a.value = new_ptr

    def scale_bf16(self, x, scale):
        """x (bf16) *= scale[0] (a device scalar)."""
        _chk(x, torch.bfloat16, 'x'), _chk(scale, torch.float32, 'scale')
        self._add('vp_scale_bf16', _p(x), x.numel(), _p(scale), keep=(x, scale))

    def identity_norm_bwd(self, dy, rowscale, rows_per_scale, dx, dxb, M, E):
        """dx (f32) = dy (bf16); dxb (bf16) = dy * rowscale[row // rows_per_scale] (rowscale None: a plain copy)."""
        _chk(dy, torch.bfloat16, 'dy'), _chk(dx, torch.float32, 'dx'), _chk(dxb, torch.bfloat16, 'dxb')
        _chk(rowscale, torch.float32, 'rowscale')
        self._add('vp_identity_norm_bwd', _p(dy), _p(rowscale), rows_per_scale, _p(dx), _p(dxb), M, E,
                  keep=(dy, rowscale, dx, dxb))

    def sum_bf16(self, srcs, out):
        """out (bf16) = sum of the bf16 tensors `srcs` (at most 8, same size), accumulated in f32."""
        for t in srcs:
            _chk(t, torch.bfloat16, 'src')
        _chk(out, torch.bfloat16, 'out')
        arr = (c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
        self._add('vp_sum_bf16', arr, len(srcs), _p(out), out.numel(), keep=(arr, out) + tuple(srcs))

    def pos_combine(self, pos_embed, out, T, E):
        self._add('vp_pos_combine', _p(pos_embed), _p(out), T, E, keep=(pos_embed, out))

    def pos_embed_grad(self, dx, gpos, B, T, E):
        _chk(dx, torch.float32, 'dx'), _chk(gpos, torch.float32, 'gpos')
        self._add('vp_pos_embed_grad', _p(dx), _p(gpos), B, T, E, keep=(dx, gpos))


class _Host:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, _stream):
        self.fn()
        return 0
