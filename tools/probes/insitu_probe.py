#!/usr/bin/env python
"""Are the tile schedules that plan recording picks from back-to-back launches also the best ones INSIDE the step?

Records the ViTPose training plan (forward + backward) of one model / batch, groups its GEMM launches by product class
(operand majors, epilogue, M, N, K), and for every class replays the WHOLE plan with each alternative schedule bound
to all launches of the class, interleaved with replays of the recorded pick: ms per replay, lowest of each.  Ends with
all individually better alternatives bound together.

  python tools/probes/insitu_probe.py [--model B] [--batch 64] [--reps 15]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

from oracle import vitpose_oracle as O  # noqa: E402  (weights and batch generators only)
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.pose_engine import PoseEngine  # noqa: E402

EPI = {0: 'BF16', 1: 'GELU', 2: 'RESID', 3: 'DGELU', 5: 'F32', 6: 'SCALE_RELU', 7: 'DRELU'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='B')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=15)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    vcfg = O.vit_cfg(args.model)
    hc = O.classic_head_cfg(vcfg['embed_dim'])
    eng = PoseEngine(vcfg, hc, dev)
    sd = {'backbone.' + k: v for k, v in O.synth_vit_weights(vcfg, seed=0).items()}
    sd.update({'keypoint_head.' + k: v for k, v in O.synth_head_weights(hc, seed=1).items()})
    eng.load_state_dict(sd)
    B = args.batch
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, 256, 192, generator=g).to(dev)
    target = torch.rand(B, 17, 64, 48, generator=g).to(dev)
    tw = torch.ones(B, 17, 1).to(dev)
    for _ in range(3):
        eng.forward_backward(img, target, tw)
    torch.cuda.synchronize()
    plan, _ = eng._train_plan(B)
    stream = torch.cuda.current_stream()

    def replay_ms(n=args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        plan.run()
        e0.record(stream)
        for _ in range(n):
            plan.run()
        e1.record(stream)
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    groups = {}
    for i, (name, fn, cargs) in enumerate(plan.calls):
        if name != 'vp_gemm_bf16':
            continue
        a = cargs[0]._obj
        if (a.tile_mode == 0 or a.M < 512 or a.N < 256 or a.K < 256 or a.route_expert or a.gather16_C or a.split3
                or a.ln_stats or a.ln_stats_out or a.epilogue == _lib.VP_EPI_SPLITK_F32):
            continue
        key = (a.x_kmajor, a.y_kmajor, a.epilogue, a.M, a.N, a.K, bool(a.resid), bool(a.out2))
        groups.setdefault(key, []).append((a, fn, cargs))
    s_ = ctypes.c_void_p(stream.cuda_stream)
    print(f'ViTPose-{args.model} x {B}: {len(plan.calls)} calls, {len(groups)} tuned product classes; '
          f'plan replay {replay_ms():.3f} ms')
    better = {}
    for key, members in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        xk, yk, epi, M, N, K, resid, out2 = key
        cur = members[0][0].tile_mode
        assert all(m[0].tile_mode == cur for m in members)
        cands = [1, 2, 3] if xk else [1, 2, 3, 4, 6]
        if K >= 1536 and epi in (0, 5):
            cands.append(7)
        if (not xk and K >= 640 and M % 192 == 0 and epi in (0, 1, 3) and (M // 192) * ((N + 191) // 192) > 256
                and not members[0][0].colscale and not members[0][0].out_row_div):
            cands.append(5)
        line = [f'{xk}{yk} {EPI.get(epi, epi):6s} M{M} N{N} K{K} x{len(members)}: pick {cur}']
        base = []
        for alt in cands:
            if alt == cur:
                continue
            a0, fn0, _ = members[0]
            a0.tile_mode = alt
            rc = fn0(ctypes.byref(a0), s_)
            torch.cuda.synchronize()
            a0.tile_mode = cur
            if rc != 0:
                line.append(f'{alt}: n/a')
                continue
            t_alt = []
            for _ in range(2):
                base.append(replay_ms())
                for a, _, _ in members:
                    a.tile_mode = alt
                t_alt.append(replay_ms())
                for a, _, _ in members:
                    a.tile_mode = cur
            line.append(f'{alt}: {min(t_alt):.3f}')
            if min(t_alt) < min(base) - 0.015:
                if key not in better or min(t_alt) < better[key][1]:
                    better[key] = (alt, min(t_alt))
        line.insert(1, f'{min(base):.3f} ms' if base else '')
        print('  '.join(line), flush=True)
    if better:
        before = replay_ms()
        for key, (alt, _) in better.items():
            for a, _, _ in groups[key]:
                a.tile_mode = alt
        after = [replay_ms() for _ in range(3)]
        for key in better:
            cur_alt = better[key][0]
            print(f'  better in the step: {key} -> {cur_alt}')
        print(f'all of them together: {before:.3f} -> {min(after):.3f} ms')
    else:
        print('no alternative beats a recorded pick inside the step by more than 15 us')


if __name__ == '__main__':
    main()
