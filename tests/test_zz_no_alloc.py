"""vp_layernorm_bwd_static (the entry point that allocates nothing: its partial sums live in a __device__
array of the library) against vp_layernorm_bwd with a caller-owned workspace of the same size: bit-identical
outputs, on two calls in a row, the second with a smaller input (the array keeps the first call's contents)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

MAX_E, MAX_BLOCKS = 1280, 512


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _case(L, M, E, seed, with_res):
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(M, E, generator=g).to(dev)
    dy = torch.randn(M, E, generator=g).to(dev).bfloat16()
    gamma = (1 + 0.1 * torch.randn(E, generator=g)).to(dev)
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt().contiguous()
    dres = torch.randn(M, E, generator=g).to(dev) if with_res else None
    rows = 192 if M % 192 == 0 else M
    scale = (torch.rand(M // rows, generator=g) + 0.5).to(dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for static in (False, True):
        dx = torch.full((M, E), float('nan'), device=dev)
        dxb = torch.zeros(M, E, device=dev, dtype=torch.bfloat16)
        dg, db, dc = (torch.full((E,), 0.25, device=dev) for _ in range(3))     # the sums accumulate
        args = [_p(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxb), _p(scale), rows,
                _p(dg), _p(db), _p(dc), M, E]
        if static:
            rc = L.vp_layernorm_bwd_static(*args, s)
        else:
            ws = torch.empty(MAX_BLOCKS * 3 * MAX_E, device=dev)
            rc = L.vp_layernorm_bwd(*args, _p(ws), ws.numel(), s)
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append((dx, dxb, dg, db, dc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # and the numbers are a LayerNorm backward (f64 autograd of the same function)
    xd = x.double().requires_grad_()
    gd = gamma.double().requires_grad_()
    y = torch.nn.functional.layer_norm(xd, (E,), gd, torch.zeros(E, device=dev, dtype=torch.float64), 1e-6)
    y.backward(dy.double())
    ref_dx = xd.grad + (dres.double() if with_res else 0)
    assert (outs[1][0].double() - ref_dx).abs().max().item() < 2e-4 * ref_dx.abs().max().item()
    assert (outs[1][2].double() - 0.25 - gd.grad).abs().max().item() < 2e-4 * gd.grad.abs().max().item()


def test_static_scratch_layernorm_backward_matches_workspace_version():
    from vitpose_amd import _lib
    L = _lib.lib()
    _case(L, 64 * 192, 1280, 0, True)        # the largest supported row (ViTPose-H), 512 workgroups of partials
    _case(L, 2 * 192, 384, 1, False)         # then a smaller call: fewer workgroups, shorter rows, stale scratch behind
    _case(L, 5, 64, 2, True)


def test_static_scratch_rejects_rows_beyond_its_size():
    from vitpose_amd import _lib
    L = _lib.lib()
    dev = torch.device('cuda:0')
    M, E = 8, MAX_E + 4
    x = torch.zeros(M, E, device=dev)
    dy = torch.zeros(M, E, device=dev, dtype=torch.bfloat16)
    v = torch.zeros(E, device=dev)
    r = torch.zeros(M, device=dev)
    rc = L.vp_layernorm_bwd_static(_p(dy), _p(x), _p(r), _p(r), _p(v), None, _p(x.clone()), None, None, 0,
                                   _p(v.clone()), None, None, M, E, None)
    assert rc == -2
    # no target for the sums: rejected (they cannot be deferred in a shared scratch)
    rc = L.vp_layernorm_bwd_static(_p(dy), _p(x), _p(r), _p(r), _p(v), None, _p(x.clone()), None, None, 0,
                                   None, None, None, M, MAX_E, None)
    assert rc == -1
