#!/usr/bin/env python
"""Reference point for the hand-written GEMM kernels: the vendor library (hipBLASLt through torch.matmul, bf16
operands, f32 accumulation, plain bf16 output, no fused epilogue) on the ViTPose-B / -L / -H products.  Measurement
only; nothing in the product calls it."""
import torch

SHAPES = [  # (M, N, K, what)
    (12288, 2304, 768, 'B qkv fwd'), (12288, 768, 768, 'B proj fwd'), (12288, 3072, 768, 'B fc1 fwd'),
    (12288, 768, 3072, 'B fc2 fwd'), (3072, 768, 12288, 'B fc1 wgrad (K = tokens)'),
    (2304, 768, 12288, 'B qkv wgrad'), (6144, 4096, 1024, 'L fc1 fwd (32 img)'), (24576, 5120, 1280, 'H fc1 fwd (128 img)'),
]


def main():
    dev = torch.device('cuda:0')
    for M, N, K, what in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        for nt, fn in (('x w^T', lambda: a @ b.t()),):
            for _ in range(5):
                fn()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
            print(f'{what:28s} {M:6d} x {N:5d} x {K:5d}  {best * 1e3:8.1f} us  {2 * M * N * K / best / 1e9:8.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
