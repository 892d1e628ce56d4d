#!/usr/bin/env python
"""Headline benchmark: images/sec of the ViTPose-B (classic decoder) training step at
256x192 on N MI355X GPUs, one process per GPU.

A step = forward + backward (DropPath on, BatchNorm batch statistics) + gradient all-reduce
(N > 1) + global-norm clip + layer-decay AdamW, on one synthetic batch of 64 images per GPU
(weak scaling; global batch 512 at 8 GPUs = BASELINE.json config 3; at N=1 this is the
fwd+bwd metric workload on one GPU).  Inputs are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (schema in the repo's DESIGN.md).
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
}
EPI_NAMES = {0: 'VP_EPI_BF16', 1: 'VP_EPI_GELU', 2: 'VP_EPI_RESID_F32', 3: 'VP_EPI_DGELU', 4: 'VP_EPI_SPLITK_F32',
             5: 'VP_EPI_F32', 6: 'VP_EPI_SCALE_RELU', 7: 'VP_EPI_DRELU'}


def vit_cfg(m):
    return dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=m['embed_dim'], depth=m['depth'],
                num_heads=m['num_heads'], mlp_ratio=4, qkv_bias=True, ratio=1, last_norm=True,
                drop_path_rate=m['drop_path_rate'])


def head_cfg(E):
    return dict(in_channels=E, out_channels=17, num_deconv_layers=2, num_deconv_filters=(256, 256),
                num_deconv_kernels=(4, 4), extra=dict(final_conv_kernel=1), upsample=0)


def random_init_state(eng, seed=0):
    """Random-init weights of the reference's recipe (trunc_normal .02 Linear / pos_embed, LN (1,0),
    head normal .001): values do not affect speed, shapes are ViTPose-B's."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in eng.named_shapes():
        if 'norm' in n and n.endswith('.weight') or (n.startswith('keypoint_head.deconv') and len(s) == 1 and n.endswith('weight')):
            sd[n] = torch.ones(s)
        elif n.endswith('.bias') or len(s) == 1:
            sd[n] = torch.zeros(s)
        elif n.startswith('keypoint_head.'):
            sd[n] = torch.randn(s, generator=g) * 0.001
        else:
            sd[n] = torch.nn.init.trunc_normal_(torch.empty(s), std=0.02, generator=g)
    return sd


def cpu_baseline(model_key, batch, iters=3):
    """The CPU oracle (torch fp32 restatement of the reference ops) timed on this box's host cores:
    forward + backward of the same model on a bounded sample."""
    from oracle import vitpose_oracle as O
    m = MODELS[model_key]
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    vcfg, hc = vit_cfg(m), head_cfg(m['embed_dim'])
    bb = {k: v.requires_grad_() for k, v in O.synth_vit_weights(vcfg, seed=0).items()}
    hd = O.synth_head_weights(hc, seed=1)
    for k, v in hd.items():
        if 'running' not in k:
            v.requires_grad_()
    img, target, tw = O.synth_batch(batch, seed=2)

    def step():
        heat = O.head_forward(hd, O.vit_forward(bb, img, vcfg), hc, training=True)
        O.joints_mse_loss(heat, target, tw).backward()
    step()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = time.perf_counter() - t0
    return dict(value=round(batch * iters / dt, 3), unit='images/sec', cores=cores, kind='port',
                sample=f'oracle/vitpose_oracle.py fp32 forward+backward (no optimizer), ViTPose-{model_key} classic, '
                       f'batch {batch}, {iters} timed iterations after 1 warm-up, torch {torch.__version__} CPU, '
                       f'{cores} threads')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='B', choices=list(MODELS))
    ap.add_argument('--batch', type=int, default=64, help='images per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)

    from vitpose_amd.pose_engine import PoseEngine
    m = MODELS[args.model]
    eng = PoseEngine(vit_cfg(m), head_cfg(m['embed_dim']), dev)
    eng.load_state_dict(random_init_state(eng, seed=0))   # identical replicas on every rank
    eng.setup_optimizer(lr=5e-4, weight_decay=0.1, layer_decay_rate=m['layer_decay'], max_grad_norm=1.0)
    if world > 1:
        eng.enable_ddp()
    B = args.batch
    g = torch.Generator().manual_seed(1000 + rank)          # each rank its own shard of the global batch
    img = torch.randn(B, 3, 256, 192, generator=g).to(dev)
    target = torch.rand(B, 17, 64, 48, generator=g).to(dev)
    tw = torch.ones(B, 17, 1).to(dev)

    def step(timers=None):
        eng.forward_backward(img, target, tw, drop_path=True, timers=timers)
        eng.optimizer_step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timers = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timers)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = float(eng.plans[('train', B)][1]['loss'].item())

    # per-kernel (template instantiation) GEMM timing from the HIP events of the timed region
    groups = {}
    for (xk, yk, epi, M, N, K), e0, e1 in timers:
        d = groups.setdefault((xk, yk, epi), [0.0, 0.0, 0])
        d[0] += e0.elapsed_time(e1) * 1e-3
        d[1] += 2.0 * M * N * K
        d[2] += 1
    tot_t = sum(d[0] for d in groups.values())
    tot_f = sum(d[1] for d in groups.values())
    dom = max(groups.items(), key=lambda kv: kv[1][0])
    (xk, yk, epi), (dsec, dflop, dn) = dom

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            'metric': 'images/sec (256x192) ViTPose-B fwd+bwd at 1/2/4/8 GPUs; % bf16 MFMA roofline',
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'ViTPose-{args.model} 256x192 classic-decoder training step '
                                   f'(fwd+bwd+grad-clip+AdamW, DropPath on, BN batch stats), {B} img/GPU, '
                                   f'global batch {world * B}, random-init weights',
                       'parallelism': f'dp{world}', 'global_batch': world * B},
            'step_tflops_per_gpu': round(value * m['train_gflop'] / world / 1e3, 2),
            'frac_of_bf16_mfma_peak': round(value * m['train_gflop'] / world / 1e3 / PEAK_BF16_TFLOPS, 4),
            'loss': loss,
            'roofline': {
                'bound': 'mfma',
                'kernel': f'gemm_kernel<{"true" if xk else "false"}, {"true" if yk else "false"}, {epi}>'
                          f' ({EPI_NAMES[epi]})',
                'achieved': round(dflop / dsec / 1e12, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(dflop / dsec / 1e12 / PEAK_BF16_TFLOPS, 4), 'traffic': None,
                'launches': dn, 'avg_launch_us': round(dsec / dn * 1e6, 2),
                'algorithmic_gflop_per_launch': round(dflop / dn / 1e9, 3),
                'share_of_step_time': round(dsec / dt, 3),
            },
            'gemm_kernels': {f'{xk_}{yk_}/{EPI_NAMES[e_]}': {'launches': d[2], 'avg_us': round(d[0] / d[2] * 1e6, 2),
                                                           'tflops': round(d[1] / d[0] / 1e12, 2)}
                             for (xk_, yk_, e_), d in sorted(groups.items())},
            'all_gemms': {'tflops': round(tot_f / tot_t / 1e12, 2), 'share_of_step_time': round(tot_t / dt, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.model, args.cpu_batch)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
