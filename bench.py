#!/usr/bin/env python
"""Headline benchmark: images/sec of the ViTPose-B (classic decoder) training step at
256x192 on N MI355X GPUs, one process per GPU, through the reference's own model API:
the model is built from a reference-shaped `model=dict(type='TopDown', ...)` config by the
registry (`build_posenet`), moved to the GPU and driven as the reference's runner drives it
(forward_train -> loss.backward() -> optimizer step).

A step = forward + backward (DropPath on, BatchNorm batch statistics) + gradient all-reduce
(N > 1) + global-norm clip + layer-decay AdamW, on one synthetic batch of 64 images per GPU
(weak scaling; global batch 512 at 8 GPUs; at N=1 this is the fwd+bwd metric workload on one
GPU).  Inputs are resident in HBM before the timed region.  The PCK accuracy that the reference
logs every 50 iterations is not part of the step (`train_cfg.skip_accuracy`).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
  python bench.py --mode infer --model B --batch 64      (the inference configuration of the headline benchmark: one eval forward)

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.0  # 256 CU x 4096 FLOP/clk x 2.4 GHz, dense bf16 MFMA
MODELS = {
    'S': dict(embed_dim=384, depth=12, num_heads=12, drop_path_rate=0.1, layer_decay=0.8, train_gflop=33.45,
              fwd_gflop=11.19),
    'B': dict(embed_dim=768, depth=12, num_heads=12, drop_path_rate=0.3, layer_decay=0.75, train_gflop=110.91,
              fwd_gflop=37.05),
    'L': dict(embed_dim=1024, depth=24, num_heads=16, drop_path_rate=0.5, layer_decay=0.8, train_gflop=369.11,
              fwd_gflop=123.14),
    'H': dict(embed_dim=1280, depth=32, num_heads=16, drop_path_rate=0.55, layer_decay=0.85, train_gflop=754.60,
              fwd_gflop=251.66),
}
EPI_NAMES = {0: 'VP_EPI_BF16', 1: 'VP_EPI_GELU', 2: 'VP_EPI_RESID_F32', 3: 'VP_EPI_DGELU', 4: 'VP_EPI_SPLITK_F32',
             5: 'VP_EPI_F32', 6: 'VP_EPI_SCALE_RELU', 7: 'VP_EPI_DRELU', 100: 'grouped weight gradients, VP_EPI_F32',
             202: 'split3 (hi/lo pair operands, 3 MFMAs per fragment pair), VP_EPI_RESID_F32',
             205: 'split3 (hi/lo pair operands, 3 MFMAs per fragment pair), VP_EPI_F32',
             208: 'split3 (hi/lo pair operands, 3 MFMAs per fragment pair), VP_EPI_PAIR'}


def model_cfg(m):
    """The `model` block of configs/body/2d_kpt_sview_rgb_img/topdown_heatmap/coco/ViTPose_base_coco_256x192.py
    (and its small / large siblings): same keys, same values."""
    return dict(
        type='TopDown', pretrained=None,
        backbone=dict(type='ViT', img_size=(256, 192), patch_size=16, embed_dim=m['embed_dim'], depth=m['depth'],
                      num_heads=m['num_heads'], ratio=1, use_checkpoint=False, mlp_ratio=4, qkv_bias=True,
                      drop_path_rate=m['drop_path_rate']),
        keypoint_head=dict(type='TopdownHeatmapSimpleHead', in_channels=m['embed_dim'], num_deconv_layers=2,
                           num_deconv_filters=(256, 256), num_deconv_kernels=(4, 4),
                           extra=dict(final_conv_kernel=1), out_channels=17,
                           loss_keypoint=dict(type='JointsMSELoss', use_target_weight=True)),
        train_cfg=dict(skip_accuracy=True),
        test_cfg=dict(flip_test=False, post_process='default', shift_heatmap=False, target_type='GaussianHeatmap',
                      modulate_kernel=11, use_udp=False))


def plus_model_cfg(m, heads=6):
    """The `model` block of configs/.../coco/vitPose+_base_coco+aic+mpii+ap10k+apt36k+wholebody_256x192_udp.py:
    TopDownMoE, a ViTMoE backbone with 6 experts on the last quarter of the FFN output channels, the main head and
    five associate heads.  Every head emits 17 heat maps here: the detector's forward_train feeds ONE target tensor
    to all heads (top_down_moe.py:181-197), so a trainable batch needs equal keypoint counts."""
    cfg = model_cfg(m)
    cfg['type'] = 'TopDownMoE'
    cfg['backbone'].update(type='ViTMoE', num_expert=6, part_features=m['embed_dim'] // 4)
    cfg['associate_keypoint_head'] = [dict(cfg['keypoint_head'], loss_keypoint=dict(cfg['keypoint_head']['loss_keypoint']))
                                      for _ in range(heads - 1)]
    return cfg


def host_cores():
    """Threads for the CPU leg: the job's CPU share.  The GPU box gives a one-GPU job 16 cores of a machine whose
    os.cpu_count() is many times that (it exports OMP_NUM_THREADS = its share); elsewhere every visible core."""
    visible = os.cpu_count() or 1
    share = int(os.environ.get('OMP_NUM_THREADS', '0') or 0)
    return (min(share, visible) if share > 0 else visible), visible


def parity_vs_oracle(eng, model_key, batch, dev):
    """Heat-map max-abs difference GPU vs CPU oracle on one identical batch, in this run: non-degenerate synthetic
    weights (the reference's own init gives heat maps ~1e-4, where any tolerance is vacuous) loaded into the
    engine that was just timed; eval forward, bf16 path and fp32-accurate path."""
    from oracle import vitpose_oracle as O
    m = MODELS[model_key]
    vcfg, hc = O.vit_cfg(model_key), O.classic_head_cfg(m['embed_dim'])
    bb, hd = O.synth_vit_weights(vcfg, seed=0), O.synth_head_weights(hc, seed=1)
    img, _, _ = O.synth_batch(batch, seed=2)
    with torch.no_grad():
        ref = O.head_forward(hd, O.vit_forward(bb, img, vcfg), hc, training=False)
    sd = {'backbone.' + k: v for k, v in bb.items()}
    sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
    eng.load_state_dict(sd)
    x = img.to(dev)
    bf = (eng.infer(x).cpu() - ref).abs().max().item()
    ex = (eng.infer(x, exact=True).cpu() - ref).abs().max().item()
    return dict(parity_max_abs=round(bf, 6), parity_max_abs_fp32_exact=round(ex, 7),
                parity_ref_max=round(ref.abs().max().item(), 3),
                parity_note=f'GPU eval heat maps vs oracle/vitpose_oracle.py (fp32 CPU) on the same {batch} synthetic images, '
                            'synthetic non-degenerate weights; parity_max_abs = the bf16 path that is timed in '
                            'training / default inference, parity_max_abs_fp32_exact = the split3 (hi/lo pair) path '
                            '(bench.py --mode infer --exact, test_cfg fp32_exact); north_star tolerance 1e-3')


def cpu_baseline(model_key, batch, mode, iters=None):
    """The CPU oracle (torch fp32 restatement of the reference ops) timed on this box's host cores
    on a bounded sample of the same workload."""
    from oracle import vitpose_oracle as O
    m = MODELS[model_key]
    cores, visible = host_cores()
    if iters is None:
        iters = 5
    torch.set_num_threads(cores)
    vcfg = O.vit_cfg(model_key) if model_key in O.VIT_CFGS else None
    hc = O.classic_head_cfg(m['embed_dim'])
    bb = O.synth_vit_weights(vcfg, seed=0)
    hd = O.synth_head_weights(hc, seed=1)
    img, target, tw = O.synth_batch(batch, seed=2)
    if mode == 'train':
        for v in bb.values():
            v.requires_grad_()
        for k, v in hd.items():
            if 'running' not in k:
                v.requires_grad_()
        params = [v for v in list(bb.values()) + list(hd.values()) if v.requires_grad]
        opt = torch.optim.AdamW(params, lr=5e-4, betas=(0.9, 0.999), weight_decay=0.1)

        def step():
            heat = O.head_forward(hd, O.vit_forward(bb, img, vcfg), hc, training=True)
            O.joints_mse_loss(heat, target, tw).backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            opt.zero_grad()
        what = 'forward+backward+grad-clip+AdamW (torch.optim, one lr group)'
    else:
        def step():
            with torch.no_grad():
                O.head_forward(hd, O.vit_forward(bb, img, vcfg), hc, training=False)
        what = 'eval forward'
    for _ in range(2):
        step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = time.perf_counter() - t0
    return dict(value=round(batch * iters / dt, 3), unit='images/sec', cores=cores, kind='port',
                sample=f'oracle/vitpose_oracle.py fp32 {what}, ViTPose-{model_key} classic, batch {batch}, '
                       f'{iters} timed iterations after 2 warm-ups, torch {torch.__version__} CPU, {cores} threads = '
                       f'the CPU share of this job ({visible} logical CPUs visible on the machine)')


def pmc_traffic(workload, kernel_key, flop_per_launch):
    """(HBM bytes per launch, where the figure comes from) for the dominant kernel.  The bytes are counters of an
    earlier profiling run of this command (rocprofv3 --pmc passes cannot run inside the bench), committed as
    profiles/pmc_traffic.json by tools/pmc_traffic.py under "<mode>_<model><batch>:<kernel>" together with the tree
    state and the launch size they were taken on; a record of another workload or of launches of another size does
    not describe this run and gives (None, reason)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None, 'profiles/pmc_traffic.json not found'
    try:
        d = json.load(open(path))
    except (OSError, ValueError) as e:
        return None, f'profiles/pmc_traffic.json unreadable: {e}'
    rec = d.get(f'{workload}:{kernel_key}')
    if not rec:
        return None, f'profiles/pmc_traffic.json has no record for {workload}:{kernel_key}'
    src = f"profiles/pmc_traffic.json (state {d.get('_state', '?')}; {d.get('_source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')})"
    want = rec.get('gflop_per_launch')
    if want is not None and abs(want - flop_per_launch / 1e9) > 0.01 * want:
        return None, src + f": taken on launches of {want} GFLOP, this run's are {flop_per_launch / 1e9:.1f}"
    return rec.get('hbm_bytes_per_launch'), src


def launch_ranks(n):
    """Start `n` ranks of this script (one process per GPU) the way the reference's tools/dist_train.sh:16-21 does,
    with torch.distributed.run on 127.0.0.1 and a free port; returns the launcher's exit status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs between processes on these hosts
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='B', choices=list(MODELS))
    ap.add_argument('--mode', default='train', choices=['train', 'infer'])
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--plus', action='store_true',
                    help='ViTPose+ (ViTMoE backbone, TopDownMoE detector with 6 heads) instead of the headline model')
    ap.add_argument('--cpu-batch', type=int, default=8)
    ap.add_argument('--no-default-iter', action='store_true',
                    help='skip the second figure (the reference-default iteration through model.train_step): profiling runs')
    ap.add_argument('--exact', action='store_true',
                    help='--mode infer only: the fp32-accurate evaluation path (test_cfg fp32_exact: split3 GEMMs on '
                         'hi/lo bf16 pairs, everything else f32) instead of the bf16 path')
    ap.add_argument('--fold-ln', action='store_true',
                    help='--mode infer only: test_cfg fold_layernorm (norm1 / norm2 of every Block folded into qkv / fc1)')
    args = ap.parse_args()

    if os.environ.get('VITPOSE_BENCH_WATCHDOG'):        # debugging aid: dump every thread's stack and exit when stuck
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['VITPOSE_BENCH_WATCHDOG']), exit=True)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process has not touched a GPU yet (importing torch does not) and never
        # will -- it starts the N ranks as fresh child processes through torch.distributed.run, lets their output
        # through (rank 0 prints the JSON line) and exits with their status
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)')
    # Rehearsal of the N > 1 path on a box with fewer GPUs than ranks: VITPOSE_BENCH_SHARE_GPU=1 lets the ranks share
    # the GPUs that exist and VITPOSE_BENCH_BACKEND=gloo carries the collectives between them (RCCL refuses two ranks
    # on one device).  The numbers of such a run mean nothing; the driver never sets either variable.
    dev_index = local_rank
    if os.environ.get('VITPOSE_BENCH_SHARE_GPU', '0') == '1':
        dev_index = local_rank % max(1, torch.cuda.device_count())
    backend = os.environ.get('VITPOSE_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    import torch.distributed as dist
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import vitpose_amd.models  # noqa: F401  (registers ViT / TopdownHeatmapSimpleHead / JointsMSELoss / TopDown)
    from vitpose_amd.builder import build_posenet
    m = MODELS[args.model]
    torch.manual_seed(0)                                    # identical random-init replicas on every rank
    # reference init: trunc_normal .02 / head normal .001
    cfg = plus_model_cfg(m) if args.plus else model_cfg(m)
    if args.exact:
        if args.mode != 'infer':
            raise SystemExit('--exact is an evaluation path: use it with --mode infer')
        cfg['test_cfg']['fp32_exact'] = True
    if args.fold_ln:
        if args.mode != 'infer' or args.exact:
            raise SystemExit('--fold-ln is an option of the bf16 evaluation path: use it with --mode infer')
        cfg['test_cfg']['fold_layernorm'] = True
    model = build_posenet(cfg).to(dev)
    metas = [dict(dataset_idx=i % 6) for i in range(args.batch)] if args.plus else None
    loss_keys = ['main_stream_loss'] + [f'{i}_loss' for i in range(1, 6)] if args.plus else ['heatmap_loss']
    eng = model._engine
    assert eng is not None, 'the fused HIP plan must be active'
    B = args.batch
    g = torch.Generator().manual_seed(1000 + rank)          # each rank its own shard of the global batch
    img = torch.randn(B, 3, 256, 192, generator=g).to(dev)
    target = torch.rand(B, 17, 64, 48, generator=g).to(dev)
    tw = torch.ones(B, 17, 1).to(dev)

    if args.mode == 'train':
        model.train()
        opt = model.build_optimizer(
            dict(type='AdamW', lr=5e-4, betas=(0.9, 0.999), weight_decay=0.1,
                 constructor='LayerDecayOptimizerConstructor',
                 paramwise_cfg=dict(num_layers=m['depth'], layer_decay_rate=m['layer_decay'])),
            grad_clip=dict(max_norm=1., norm_type=2))
        comm = None
        # VITPOSE_BENCH_FORCE_DDP=1 (rehearsal on one GPU, never set by the driver): a one-rank RCCL group with the bucketed
        # plan of N > 1 and every collective launched (a sum over one rank) -- what the bucket structure itself costs
        force_ddp = world == 1 and os.environ.get('VITPOSE_BENCH_FORCE_DDP', '0') == '1'
        if force_ddp:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        if world > 1 or force_ddp:
            comm = model.enable_ddp()
            comm.measure = True        # event pairs around the end-of-backward wait: exposed vs hidden all-reduce time
            if force_ddp:
                comm.world = 2         # launches the all-reduces (and halves the gradients: a rehearsal of the schedule only)

        def step():
            losses = model.forward_train(img, target, tw, metas)
            loss = losses[loss_keys[0]] if len(loss_keys) == 1 else sum(losses[k] for k in loss_keys)
            loss.backward()
            opt.step()
            opt.zero_grad()
            return loss
        plan_of = lambda: eng.plans[('train', B)][0]  # noqa: E731
        gflop = m['train_gflop']
        what = (f'ViTPose-{args.model} 256x192 classic-decoder training step (fwd+bwd+grad-clip+AdamW, DropPath on, '
                f'BN batch stats) via TopDown.forward_train, {B} img/GPU, global batch {world * B}, random-init weights')
        if args.plus:
            E = m['embed_dim']
            head_gmac = (192 * E * 256 * 16 + 768 * 256 * 256 * 16 + 3072 * 256 * 17) / 1e9
            gflop += 5 * 6 * head_gmac      # five more heads, forward + both backward products
            what = (f'ViTPose+-{args.model} (ViTMoE: 6 experts on {E // 4} FFN channels; TopDownMoE: 6 classic heads, '
                    f'images spread evenly over 6 datasets) training step via TopDownMoE.forward_train, {B} img/GPU, '
                    f'global batch {world * B}, random-init weights')
    else:
        model.eval()
        comm = None

        def step():
            with torch.no_grad():
                return model.forward_dummy(img)
        plan_of = lambda: eng.plans[('exact', B) if args.exact else ('infer', B, True) if args.fold_ln else ('infer', B)][0]  # noqa: E731
        gflop = m['fwd_gflop']
        what = (f'ViTPose-{args.model} 256x192 classic-decoder inference (one eval forward, flip_test off as in '
                f'tools/analysis/speed_test.py) via TopDown.forward_dummy, batch {B}/GPU, random-init weights')
        if args.exact:
            what += ', fp32-accurate path (test_cfg fp32_exact)'
        if args.fold_ln:
            what += ', LayerNorms folded into qkv / fc1 (test_cfg fold_layernorm)'

    def group(timers):
        groups = {}
        for (xk, yk, epi, M, N, K), e0, e1 in timers:
            d = groups.setdefault((xk, yk, epi), [0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += 2.0 * M * N * K
            d[2] += 1
        return groups

    last = None
    for _ in range(max(1, args.warmup)):
        last = step()
    if args.mode == 'train' and comm is not None:
        torch.cuda.synchronize()
        comm.exposed_ms()              # drop the warm-up records
    # one more untimed step with a HIP-event pair around EVERY GEMM launch: finds the dominant kernel
    survey = []
    plan_of().timers = survey
    last = step()
    torch.cuda.synchronize()
    plan_of().timers = None
    all_groups = group(survey)
    dom_key = max(all_groups.items(), key=lambda kv: kv[1][0])[0]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timers = []
    plan_of().timers = timers          # timed region: events (on the launch stream) around the dominant kernel only
    plan_of().timer_filter = {dom_key}
    # one event per step boundary on the compute stream (no host sync inside the timed region): per-step GPU times
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        last = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    comm_ms = comm.exposed_ms() if args.mode == 'train' and comm is not None else []
    plan_of().timers = None
    plan_of().timer_filter = None
    rank_info = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        # what the collective library itself reports: a sum of ones over the group = the number of ranks that took
        # part; every rank's median step time (ranks tune their tile schedules independently: skew shows here)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        mine = torch.tensor([step_ms[len(step_ms) // 2]], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(float(e.item()), 3) for e in every]
        rank_info = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                     'allreduce_of_ones': int(ones.item()), 'step_ms_median_per_rank': per_rank,
                     'step_ms_rank_min': min(per_rank), 'step_ms_rank_max': max(per_rank),
                     'env': {k: os.environ.get(k) for k in ('NCCL_MAX_NCHANNELS', 'NCCL_MIN_NCHANNELS', 'NCCL_ALGO',
                                                            'NCCL_PROTO', 'VITPOSE_DDP_BF16', 'VITPOSE_WGRAD_GROUP',
                                                            'TORCH_NCCL_BLOCKING_WAIT')}}

    # The reference's default iteration (second figure): BasePose.train_step (detectors/base.py:78-113) with the PCK
    # accuracy of every iteration (topdown_heatmap_simple_head.py:170-195) and the logged scalars reduced over ranks
    # and read on the host (_parse_losses, base.py:38-76) -- what a user gets without `train_cfg.skip_accuracy`.
    default_iter = None
    if args.mode == 'train' and not args.plus and not args.no_default_iter:
        model.train_cfg['skip_accuracy'] = False
        batch = dict(img=img, target=target, target_weight=tw, img_metas=metas)

        def ref_step():
            out = model.train_step(batch, None)
            out['loss'].backward()
            opt.step()
            opt.zero_grad()
            return out
        # The headline figures above are complete at this point: an error in this second leg (the same on every rank:
        # all ranks run the same code on the same shapes) is reported in the line instead of costing the line.
        try:
            for _ in range(2):
                ref_out = ref_step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            r0 = time.perf_counter()
            for _ in range(args.steps):
                ref_out = ref_step()
            torch.cuda.synchronize()
            rdt = time.perf_counter() - r0
            if world > 1:
                t = torch.tensor([rdt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rdt = t.item()
        except Exception as e:  # noqa: BLE001
            rdt = None
            default_iter = {'error': f'{type(e).__name__}: {e}'}
        model.train_cfg['skip_accuracy'] = True
        default_iter = default_iter or {
                        'images_per_sec': round(world * B * args.steps / rdt, 2),
                        'ms_per_step': round(rdt / args.steps * 1e3, 3),
                        'log_vars': {k: round(float(v), 6) for k, v in ref_out['log_vars'].items()},
                        'note': 'model.train_step per iteration: forward_train WITH get_accuracy (device arg-max, '
                                '[B,17] indices to the host) + _parse_losses (one collective for all logged scalars, one '
                                'host read) + backward + clip + AdamW; the headline value is the same step without the '
                                'per-iteration accuracy and host read (train_cfg.skip_accuracy)'}

    groups = all_groups
    tot_t = sum(d[0] for d in groups.values())
    tot_f = sum(d[1] for d in groups.values())
    (xk, yk, epi), (dsec, dflop, dn) = dom_key, group(timers)[dom_key]

    if rank == 0:
        value = world * B * args.steps / dt
        # one launcher (vp_gemm_bf16) serves three tile schedules (gemm_kernel 128x128, gemm192_kernel,
        # gemm256_kernel), chosen per shape by timing at plan-record time; the instantiation is <x_kmajor, y_kmajor, epilogue>
        kname = f'vp_gemm_bf16<{"true" if xk else "false"}, {"true" if yk else "false"}, {epi % 100}>'
        if epi == 100:   # the grouped launcher runs gemm192_kernel / gemm256_kernel<true, true, VP_EPI_F32>
            kname = 'vp_wgrad_grouped_bf16'
        traffic, traffic_src = pmc_traffic(f'{args.mode}_{args.model}{B}' + ('_exact' if args.exact else '') + ('_fold' if args.fold_ln else ''), kname, dflop / dn)
        out = {
            'metric': 'images/sec (256x192) ViTPose-B fwd+bwd at 1/2/4/8 GPUs; % bf16 MFMA roofline',
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'step_ms_min': round(step_ms[0], 3), 'step_ms_median': round(step_ms[len(step_ms) // 2], 3),
            'step_ms_max': round(step_ms[-1], 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16x3 (hi/lo bf16 pairs, f32 accumulate)' if args.exact else 'bf16',
            'data': 'synthetic',
            'config': {'workload': what, 'parallelism': f'dp{world}', 'global_batch': world * B},
            'step_tflops_per_gpu': round(value * gflop / world / 1e3, 2),
            'frac_of_bf16_mfma_peak': round(value * gflop / world / 1e3 / PEAK_BF16_TFLOPS, 4),
            'roofline': {
                'bound': 'mfma', 'kernel': f'{kname} ({EPI_NAMES[epi]})',
                'achieved': round(dflop / dsec / 1e12, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(dflop / dsec / 1e12 / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                'traffic_source': traffic_src,
                'launches': dn, 'avg_launch_us': round(dsec / dn * 1e6, 2),
                'algorithmic_gflop_per_launch': round(dflop / dn / 1e9, 3),
                'share_of_step_time': round(dsec / dt, 3),
            },
            'gemm_kernels': {f'{xk_}{yk_}/{EPI_NAMES[e_]}': {'launches': d[2], 'avg_us': round(d[0] / d[2] * 1e6, 2),
                                                           'tflops': round(d[1] / d[0] / 1e12, 2)}
                             for (xk_, yk_, e_), d in sorted(groups.items())},
            'all_gemms': {'tflops': round(tot_f / tot_t / 1e12, 2),
                          'share_of_step_time': round(tot_t / (dt / args.steps), 3),
                          'note': 'gemm_kernels / all_gemms: one untimed survey step with events around every GEMM'},
        }
        if args.mode == 'train':
            out['loss'] = float(last.item())
        if default_iter is not None:
            out['reference_default_iteration'] = default_iter
        if rank_info is not None:
            out['ranks'] = rank_info
        if comm_ms:
            ex, busy = sorted(e for e, _ in comm_ms), sorted(b for _, b in comm_ms)
            out['allreduce'] = {
                'exposed_ms_per_step_median': round(ex[len(ex) // 2], 3), 'exposed_ms_per_step_max': round(ex[-1], 3),
                'busy_ms_per_step_median': round(busy[len(busy) // 2], 3),
                'hidden_fraction': round(1.0 - ex[len(ex) // 2] / max(busy[len(busy) // 2], 1e-9), 3),
                'gradient_bytes': int(eng.store.numel * (2 if comm.bf16 else 4)),
                'note': 'rank 0; exposed = compute stream waiting for the gradient exchange after backward ended, busy = '
                        'first bucket hand-over to last collective done (HIP events on the two streams)'}
        if world == 1 and not args.no_cpu_baseline and not args.plus:
            out['cpu_baseline'] = cpu_baseline(args.model, args.cpu_batch, args.mode)
            out.update(parity_vs_oracle(eng, args.model, args.cpu_batch, dev))
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
