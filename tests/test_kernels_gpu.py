"""Per-kernel parity of the HIP kernels (through the C ABI) against plain torch fp32
references on the same (bf16-rounded) inputs.  Integer-valued data makes the MFMA
GEMM checks exact (products and f32 sums of small integers are exact), which catches
any fragment-layout / transpose / swizzle mistake; random data checks rounding."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from vitpose_amd import _lib
    _lib.lib()
    return torch.device('cuda:0')


def run(build):
    from vitpose_amd.plan import Plan
    p = Plan()
    build(p)
    p.run()
    torch.cuda.synchronize()


def ints(shape, dev, lo=-3, hi=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float().to(dev)


def rnd(shape, dev, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize('xk,yk', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (192, 384, 128), (200, 136, 192), (384, 64, 256), (8, 8, 64)])
def test_gemm_layouts_exact(dev, xk, yk, M, N, K):
    from vitpose_amd import _lib
    X = ints((M, K), dev, seed=1)          # asymmetric operands
    Y = ints((N, K), dev, seed=2)
    ref = X @ Y.t()
    xs = (X.t() if xk else X).contiguous().bfloat16()
    ys = (Y.t() if yk else Y).contiguous().bfloat16()
    out = torch.full((M, N), float('nan'), device=dev)
    run(lambda p: p.gemm(xs, ys, out, M, N, K, ldx=xs.shape[1], ldy=ys.shape[1], ldo=N, x_kmajor=xk, y_kmajor=yk,
                         epilogue=_lib.VP_EPI_F32))
    assert torch.equal(out, ref)


def test_gemm_epilogues(dev):
    from vitpose_amd import _lib
    M, N, K, T = 384, 256, 128, 192
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.1).bfloat16()
    bias, cs = rnd((N,), dev, 5), rnd((N,), dev, 6)
    acc = X.float() @ Y.float().t()
    # BF16 with colscale + bias
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, colscale=cs))
    torch.testing.assert_close(out.float(), (acc * cs + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)
    # SCALE_RELU
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, colscale=cs,
                         epilogue=_lib.VP_EPI_SCALE_RELU))
    torch.testing.assert_close(out.float(), torch.relu(acc * cs + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)
    # GELU (+ pre-activation copy)
    u = torch.empty_like(out)
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, out2=u, epilogue=_lib.VP_EPI_GELU))
    torch.testing.assert_close(u.float(), (acc + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(acc + bias), rtol=2e-2, atol=2e-2)
    # RESID_F32 with per-sample row scale and position rows
    resid, pos = rnd((M, N), dev, 7), rnd((T, N), dev, 8)
    rs = torch.tensor([0.0, 1.0 / 0.7], device=dev)
    o32 = torch.empty(M, N, device=dev)
    run(lambda p: p.gemm(X, Y, o32, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, resid=resid, rowscale=rs,
                         rows_per_scale=T, epilogue=_lib.VP_EPI_RESID_F32))
    ref = resid + rs.repeat_interleave(T)[:, None] * (acc + bias)
    torch.testing.assert_close(o32, ref, rtol=1e-4, atol=1e-4)
    # the addend is the residual row OR the position row (row % pos_rows), never both
    run(lambda p: p.gemm(X, Y, o32, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, pos=pos, pos_rows=T,
                         epilogue=_lib.VP_EPI_RESID_F32))
    torch.testing.assert_close(o32, acc + bias + pos.repeat(2, 1), rtol=1e-4, atol=1e-4)
    # DGELU / DRELU
    aux = rnd((M, N), dev, 9).bfloat16()
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, aux=aux, ldaux=N, epilogue=_lib.VP_EPI_DGELU))
    a = aux.float().requires_grad_()
    torch.nn.functional.gelu(a).sum().backward()
    torch.testing.assert_close(out.float(), acc * a.grad, rtol=2e-2, atol=2e-2)
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, aux=aux, ldaux=N, epilogue=_lib.VP_EPI_DRELU))
    torch.testing.assert_close(out.float(), (acc * (aux.float() > 0)).bfloat16().float(), rtol=1e-2, atol=1e-2)
    # split-K slabs + reduce, accumulating on top of existing values (integer data: exact);
    # 5 slices over 8 K steps leaves the last slice empty (it must still write zeros)
    Xi, Yi = ints((M, 512), dev, seed=10).bfloat16(), ints((N, 512), dev, seed=11).bfloat16()
    o32.fill_(1.0)
    slab = torch.full((5 * M * N,), float('nan'), device=dev)

    def splitk(p):
        p.gemm(Xi, Yi, slab, M, N, 512, ldx=512, ldy=512, ldo=N, epilogue=_lib.VP_EPI_SPLITK_F32, split_k=5,
               split_stride=M * N)
        p._add('vp_splitk_reduce', slab.data_ptr(), 5, M * N, M * N, o32.data_ptr(), 1)
    run(splitk)
    assert torch.equal(o32, Xi.float() @ Yi.float().t() + 1.0)
    # output row remap (sub-pixel phase placement): row i -> (i / W) * 4W + (i % W) * 2 + off
    W = 12
    big = torch.zeros(4 * M, N, device=dev)
    run(lambda p: p.gemm(Xi, Yi, big, M, N, 512, ldx=512, ldy=512, ldo=N, epilogue=_lib.VP_EPI_F32,
                         row_map=(W, 4 * W, 2, 2 * W + 1)))
    i = torch.arange(M, device=dev)
    rows = (i // W) * 4 * W + (i % W) * 2 + 2 * W + 1
    assert torch.equal(big[rows], Xi.float() @ Yi.float().t())
    assert big.abs().sum() == big[rows].abs().sum()


def test_gemm_rejects_bad_args(dev):
    from vitpose_amd.plan import KernelError
    X = torch.zeros(64, 100, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(64, 64, device=dev, dtype=torch.bfloat16)
    with pytest.raises(KernelError):
        run(lambda p: p.gemm(X, X, out, 64, 64, 100, ldx=100, ldy=100, ldo=64))  # K % 64 != 0


@pytest.mark.parametrize('E', [64, 384, 768, 1280])
def test_layernorm_fwd_bwd(dev, E):
    M, T = 4 * 192 + 5, 192
    x = rnd((M, E), dev, 1) * 2 + 0.5
    g, b = 1 + 0.2 * rnd((E,), dev, 2), 0.1 * rnd((E,), dev, 3)
    y, y32 = torch.empty(M, E, device=dev, dtype=torch.bfloat16), torch.empty(M, E, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    run(lambda p: p.layernorm_fwd(x, g, b, y, y32, mean, rstd, M, E))
    xr = x.clone().requires_grad_()
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    ref = torch.nn.functional.layer_norm(xr, (E,), gr, br, 1e-6)
    torch.testing.assert_close(y32, ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y.float(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean, x.mean(1), rtol=1e-5, atol=1e-5)
    dy = rnd((M, E), dev, 4).bfloat16()
    dres = rnd((M, E), dev, 5)
    nb = (M + T - 1) // T
    rs = torch.rand(nb, device=dev) + 0.5
    dx, dxb = torch.empty(M, E, device=dev), torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    dg, db, dc = torch.ones(E, device=dev), torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    run(lambda p: p.layernorm_bwd(dy, x, mean, rstd, g, dres, dx, dxb, rs, T, dg, db, dc, M, E))
    ref.backward(dy.float())
    dx_ref = xr.grad + dres
    torch.testing.assert_close(dx, dx_ref, rtol=1e-4, atol=1e-4)
    scaled = dx_ref * rs.repeat_interleave(T)[:M, None]
    torch.testing.assert_close(dxb.float(), scaled.bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dg - 1.0, gr.grad, rtol=1e-3, atol=2e-3)  # accumulates onto the 1s
    torch.testing.assert_close(db, br.grad, rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(dc, scaled.sum(0), rtol=1e-3, atol=5e-3)
    # deferred column sums: two backward passes leave their partial sums in their own workspaces, one launch
    # adds them to the targets (NULL targets are skipped) -- the same numbers as the direct path, bit for bit
    w1, w2 = torch.empty(3 * E * 512, device=dev), torch.empty(3 * E * 40, device=dev)
    dg2, db2, dc2 = torch.ones(E, device=dev), torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    dg3 = torch.zeros(E, device=dev)
    dx2 = torch.empty_like(dx)

    def deferred(p):
        p.layernorm_bwd(dy, x, mean, rstd, g, dres, dx2, dxb, rs, T, None, None, None, M, E, workspace=w1)
        p.layernorm_bwd(dy, x, mean, rstd, g, dres, dx2, dxb, rs, T, None, None, None, M, E, workspace=w2)
        p.ln_finalize([(w1, dg2, db2, dc2), (w2, dg3, None, None)], M, E)
    run(deferred)
    assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(dc2, dc)
    torch.testing.assert_close(dg3, gr.grad, rtol=1e-3, atol=2e-3)   # a smaller workspace: fewer, longer partial sums


def _attn_ref(qkv, B, H, d, scale):
    T = 192
    q, k, v = qkv.float().reshape(B, T, 3, H, d).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-2, -1)
    p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * T, H * d), torch.logsumexp(s, -1).reshape(B * H, T)


@pytest.mark.parametrize('H,d', [(2, 32), (3, 64), (12, 64), (2, 80), (16, 80)])
def test_attention_fwd_bwd(dev, H, d):
    B, T, E = 2, 192, H * d
    qkv = (rnd((B * T, 3 * E), dev, 1) * 1.5).bfloat16()
    scale = d ** -0.5
    out = torch.empty(B * T, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H, T, device=dev)
    run(lambda p: p.attention_fwd(qkv, out, lse, B, H, d, T, scale))
    qr = qkv.float().requires_grad_()
    ref, lse_ref = _attn_ref(qr, B, H, d, scale)
    torch.testing.assert_close(lse, lse_ref, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)
    dout = rnd((B * T, E), dev, 2).bfloat16()
    dqkv = torch.full((B * T, 3 * E), float('nan'), device=dev, dtype=torch.bfloat16)
    run(lambda p: p.attention_bwd(qkv, out, dout, lse, dqkv, B, H, d, T, scale))
    ref.backward(dout.float())
    err = (dqkv.float() - qr.grad).abs().max().item()
    assert err < 3e-2 * max(1.0, qr.grad.abs().max().item()), err
    rel = (dqkv.float() - qr.grad).norm() / qr.grad.norm()
    assert rel < 2e-2, rel


def test_attention_integer_exact_layout(dev):
    """Uniform scores (q = 0) make P = 1/192 exactly: out = column means of V, which pins
    the V transposed-read / P*V operand pairing independent of softmax rounding."""
    B, H, d, T = 1, 2, 64, 192
    E = H * d
    qkv = torch.zeros(B * T, 3 * E, device=dev)
    qkv[:, 2 * E:] = ints((B * T, E), dev, -8, 9, seed=3)
    qkv[:, E:2 * E] = ints((B * T, E), dev, seed=4)
    out = torch.empty(B * T, E, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.attention_fwd(qkv.bfloat16(), out, None, B, H, d, T, d ** -0.5))
    ref = qkv[:, 2 * E:].mean(0, keepdim=True).expand(T, E)
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=1e-2)


def test_patch_im2col_and_casts(dev):
    B = 2
    img = rnd((B, 3, 256, 192), dev, 1)
    col = torch.empty(B * 192, 768, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.patch_im2col(img, col, B, 3, 256, 192, 16, 16, 2))
    ref = torch.nn.functional.unfold(img, 16, padding=2, stride=16).transpose(1, 2).reshape(B * 192, 768)
    assert torch.equal(col.float(), ref.bfloat16().float())
    for pad, pch in ((0, 16), (4, 16), (1, 16), (2, 8)):       # even pads: the strip kernel; odd pad / patch 8: the general one
        Hp, Wp = (256 + 2 * pad - pch) // pch + 1, (192 + 2 * pad - pch) // pch + 1
        c2 = torch.full((B * Hp * Wp, 3 * pch * pch), float('nan'), device=dev, dtype=torch.bfloat16)
        run(lambda p: p.patch_im2col(img, c2, B, 3, 256, 192, pch, pch, pad))
        r2 = torch.nn.functional.unfold(img, pch, padding=pad, stride=pch).transpose(1, 2).reshape(B * Hp * Wp, -1)
        assert torch.equal(c2.float(), r2.bfloat16().float()), (pad, pch)
    # strided cast: NCHW <-> token-major permute
    x = rnd((B, 192, 64), dev, 2)
    o = torch.empty(B, 64, 192, device=dev)
    run(lambda p: p.strided_cast(x, o, (1, B, 64, 192), (0, 192 * 64, 1, 64)))
    assert torch.equal(o, x.permute(0, 2, 1).contiguous())
    a = rnd((777, 264), dev, 3).bfloat16()
    cs = torch.ones(264, device=dev)
    run(lambda p: p.colsum(a, cs, 777, 264, 264))
    torch.testing.assert_close(cs, a.float().sum(0) + 1, rtol=1e-4, atol=1e-3)
    for Mn, Nn in ((12288, 32), (1001, 8), (4097, 64), (333, 16)):     # packed narrow matrices (the heat-map gradient)
        an = rnd((Mn, Nn), dev, 30 + Nn).bfloat16()
        cn = torch.full((Nn,), 2.0, device=dev)
        run(lambda p: p.colsum(an, cn, Mn, Nn, Nn))
        torch.testing.assert_close(cn, an.float().sum(0) + 2, rtol=1e-4, atol=2e-3)
    gp = rnd((193, 768), dev, 41)                                       # position-embedding gradient: row 0 = sum of rows 1..T
    dxs = rnd((3, 192 * 768), dev, 42)
    run(lambda p: p.pos_embed_grad(dxs, gp, 3, 192, 768))
    torch.testing.assert_close(gp[1:], dxs.sum(0).view(192, 768), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gp[0], dxs.sum(0).view(192, 768).sum(0), rtol=1e-5, atol=1e-4)
    xs = rnd((3, 192 * 64), dev, 4)
    bs = torch.empty(192 * 64, device=dev)
    run(lambda p: p.batch_sum(xs, bs, 3, 192 * 64))
    torch.testing.assert_close(bs, xs.sum(0), rtol=1e-5, atol=1e-5)


def test_deconv_as_subpixel_gemms(dev):
    """ConvTranspose2d(k4,s2,p1) forward, data gradient and weight gradient via the im2col views."""
    from vitpose_amd import _lib
    B, H, W, Ci, Co = 2, 16, 12, 64, 64
    x = ints((B, Ci, H, W), dev, -2, 3, seed=1)
    wt = ints((Ci, Co, 4, 4), dev, -2, 3, seed=2)
    ref = torch.nn.functional.conv_transpose2d(x, wt, stride=2, padding=1)
    xcl = x.permute(0, 2, 3, 1).contiguous().bfloat16().view(B * H * W, Ci)
    out = torch.full((B * 4 * H * W, Co), float('nan'), device=dev)
    col = torch.empty(B * H * W, 4 * Ci, device=dev, dtype=torch.bfloat16)
    wph = torch.empty(4, Co, 4 * Ci, device=dev, dtype=torch.bfloat16)

    def fwd(p):
        for py in range(2):
            for px in range(2):
                p.strided_cast(wt.view(-1)[(1 - py) * 4 + (1 - px):], wph[py * 2 + px], (Co, 2, 2, Ci),
                               (16, 8, 2, Co * 16))
                p.deconv_phase_im2col(xcl, col, B, H, W, Ci, py, px)
                p.gemm(col, wph[py * 2 + px], out, B * H * W, Co, 4 * Ci, ldx=4 * Ci, ldy=4 * Ci, ldo=Co,
                       epilogue=_lib.VP_EPI_F32, row_map=(W, 4 * W, 2, py * 2 * W + px))
    run(fwd)
    assert torch.equal(out.view(B, 2 * H, 2 * W, Co).permute(0, 3, 1, 2), ref)
    # backward
    dout = ints((B, Co, 2 * H, 2 * W), dev, -2, 3, seed=3)
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    torch.nn.functional.conv_transpose2d(xr, wr, stride=2, padding=1).backward(dout)
    dcl = dout.permute(0, 2, 3, 1).contiguous().bfloat16().view(-1, Co)
    G = torch.empty(B * H * W, 16 * Co, device=dev, dtype=torch.bfloat16)
    wd = torch.empty(Ci, 16 * Co, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(B * H * W, Ci, device=dev)
    dwd = torch.zeros(Ci, 16 * Co, device=dev)

    def bwd(p):
        p.strided_cast(wt, wd, (Ci, 16, Co, 1), (Co * 16, 1, 16, 0))
        p.deconv_out_gather16(dcl, G, B, H, W, Co)
        p.gemm(G, wd, dx, B * H * W, Ci, 16 * Co, ldx=16 * Co, ldy=16 * Co, ldo=Ci, epilogue=_lib.VP_EPI_F32)
        p.wgrad(xcl, G, dwd, Ci, 16 * Co, B * H * W, ldx=Ci, ldy=16 * Co, split_k=3,
                slab=torch.empty(3 * Ci * 16 * Co, device=dev))
    run(bwd)
    assert torch.equal(dx.view(B, H, W, Ci).permute(0, 3, 1, 2), xr.grad)
    assert torch.equal(dwd.view(Ci, 16, Co).permute(0, 2, 1).reshape(Ci, Co, 4, 4), wr.grad)


@pytest.mark.parametrize('B,H,W,Ci,Co', [(2, 16, 12, 64, 64), (3, 16, 12, 256, 128), (1, 5, 7, 40, 64), (2, 32, 24, 768, 256)])
def test_deconv_dgrad_implicit_16_taps(dev, B, H, W, Ci, Co):
    """Data gradient of ConvTranspose2d(k4, s2, p1) in one launch (16-tap rows gathered by the loader): exact on
    integer data against torch autograd, with and without the ReLU mask; partial tiles in rows and columns."""
    x = ints((B, Ci, H, W), dev, -2, 3, seed=1)
    wt = ints((Ci, Co, 4, 4), dev, -1, 2, seed=2)
    dout = ints((B, Co, 2 * H, 2 * W), dev, -1, 2, seed=3)
    xr = x.clone().requires_grad_()
    torch.nn.functional.conv_transpose2d(xr, wt, stride=2, padding=1).backward(dout)
    ref = xr.grad.permute(0, 2, 3, 1).reshape(B * H * W, Ci)
    dcl = dout.permute(0, 2, 3, 1).contiguous().bfloat16().view(-1, Co)
    wd = torch.empty(Ci, 16 * Co, device=dev, dtype=torch.bfloat16)
    aux = ints((B * H * W, Ci), dev, -2, 3, seed=4).bfloat16()
    dx = torch.full((B * H * W, Ci), float('nan'), device=dev, dtype=torch.bfloat16)
    dxm = torch.full_like(dx, float('nan'))

    def bwd(p):
        p.strided_cast(wt, wd, (Ci, 16, Co, 1), (Co * 16, 1, 16, 0))
        p.deconv_dgrad(dcl, wd, dx, B, H, W, Co, Ci)
        p.deconv_dgrad(dcl, wd, dxm, B, H, W, Co, Ci, relu_aux=aux)
    run(bwd)
    assert ref.abs().max() < 256                       # integers that bf16 holds exactly
    assert torch.equal(dx.float(), ref)
    assert torch.equal(dxm.float(), ref * (aux.float() > 0))


@pytest.mark.parametrize('B,H,W,Ci,Co,sk', [(2, 16, 12, 64, 128, 1), (3, 16, 12, 256, 128, 4), (2, 32, 24, 768, 256, 5),
                                             (1, 32, 24, 72, 128, 12), (5, 16, 12, 264, 256, 7)])
def test_deconv_wgrad_implicit_16_taps(dev, B, H, W, Ci, Co, sk):
    """Weight gradient of ConvTranspose2d(k4, s2, p1) with the reduction-major 16-tap operand gathered by the
    loader (never materialised): exact on integer data against torch autograd; several K slices (each starts its
    pixel walk in the middle of an image), partial row tiles."""
    from vitpose_amd import _lib
    x = ints((B, Ci, H, W), dev, -2, 3, seed=1)
    wt = ints((Ci, Co, 4, 4), dev, -1, 2, seed=2)
    dout = ints((B, Co, 2 * H, 2 * W), dev, -1, 2, seed=3)
    wr = wt.clone().requires_grad_()
    torch.nn.functional.conv_transpose2d(x, wr, stride=2, padding=1).backward(dout)
    xcl = x.permute(0, 2, 3, 1).contiguous().bfloat16().view(B * H * W, Ci)
    dcl = dout.permute(0, 2, 3, 1).contiguous().bfloat16().view(-1, Co)
    K = B * H * W
    slab = torch.full((sk * Ci * 16 * Co,), float('nan'), device=dev)
    dwd = torch.full((Ci, 16 * Co), float('nan'), device=dev)

    def bwd(p):
        p.gemm(xcl, dcl, slab, Ci, 16 * Co, K, ldx=Ci, ldy=Co, ldo=16 * Co, x_kmajor=True, y_kmajor=True,
               epilogue=_lib.VP_EPI_SPLITK_F32, split_k=sk, split_stride=Ci * 16 * Co, gather16=(H, W, Co))
        p._add('vp_splitk_reduce', slab.data_ptr(), sk, Ci * 16 * Co, Ci * 16 * Co, dwd.data_ptr(), 0)
    run(bwd)
    assert torch.equal(dwd.view(Ci, 16, Co).permute(0, 2, 1).reshape(Ci, Co, 4, 4), wr.grad)


@pytest.mark.parametrize('B,H,W,Ci,Co', [(2, 16, 12, 64, 64), (3, 16, 12, 128, 256), (1, 5, 7, 64, 40)])
def test_deconv_fwd_fused_implicit_im2col(dev, B, H, W, Ci, Co):
    """ConvTranspose2d(k4, s2, p1) forward in one launch (four parities, gathered X operand): exact on
    integer data against torch, both epilogues; partial tiles in rows and columns."""
    from vitpose_amd import _lib
    x = ints((B, Ci, H, W), dev, -2, 3, seed=1)
    wt = ints((Ci, Co, 4, 4), dev, -2, 3, seed=2)
    ref = torch.nn.functional.conv_transpose2d(x, wt, stride=2, padding=1)
    xcl = x.permute(0, 2, 3, 1).contiguous().bfloat16().view(B * H * W, Ci)
    wph = torch.empty(4, Co, 4 * Ci, device=dev, dtype=torch.bfloat16)
    out = torch.full((B * 4 * H * W, Co), float('nan'), device=dev, dtype=torch.bfloat16)
    out2 = torch.full_like(out, float('nan'))
    cs, sh = rnd((Co,), dev, 3).abs() / 16, rnd((Co,), dev, 4)

    def fwd(p):
        for py in range(2):
            for px in range(2):
                p.strided_cast(wt.view(-1)[(1 - py) * 4 + (1 - px):], wph[py * 2 + px], (Co, 2, 2, Ci),
                               (16, 8, 2, Co * 16))
        p.deconv_fwd(xcl, wph, out, B, H, W, Ci, Co)
        p.deconv_fwd(xcl, wph, out2, B, H, W, Ci, Co, epilogue=_lib.VP_EPI_SCALE_RELU, colscale=cs, shift=sh)
    run(fwd)
    got = out.float().view(B, 2 * H, 2 * W, Co).permute(0, 3, 1, 2)
    assert torch.equal(got, ref.bfloat16().float())
    want2 = torch.relu(ref * cs.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    torch.testing.assert_close(out2.float().view(B, 2 * H, 2 * W, Co).permute(0, 3, 1, 2), want2, rtol=1e-2, atol=1e-2)


def test_batchnorm_relu_fwd_bwd(dev):
    M, C = 2 * 32 * 24, 256
    z = (rnd((M, C), dev, 1) * 1.5 + 0.3).bfloat16()
    gamma, beta = 1 + 0.2 * rnd((C,), dev, 2), 0.1 * rnd((C,), dev, 3)
    rm, rv = rnd((C,), dev, 4) * 0.1, torch.rand(C, device=dev) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    sums = torch.zeros(2 * C, device=dev)
    scale, shift, sm, si = (torch.empty(C, device=dev) for _ in range(4))
    a = torch.empty_like(z)

    def fwd(p):
        p.bn_stats(z, sums, M, C)
        p.bn_finalize(sums, M, C, gamma, beta, rm, rv, 0.1, 1e-5, scale, shift, sm, si)
        p.bn_apply_relu(z, a, scale, shift, M, C)
    run(fwd)
    zr = z.float().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    ref = torch.relu(torch.nn.functional.batch_norm(zr, rm0, rv0, gr, br, True, 0.1, 1e-5))
    torch.testing.assert_close(a.float(), ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(rm, rm0, rtol=1e-4, atol=1e-5)   # batch_norm updated rm0 in place
    torch.testing.assert_close(rv, rv0, rtol=1e-4, atol=1e-5)
    da = rnd((M, C), dev, 5)
    ref.backward(da)
    gmask = (da * (a.float() > 0)).bfloat16()
    dg, db, dz = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.empty_like(z)
    run(lambda p: p.bn_bwd(gmask, z, sm, si, gamma, dg, db, dz, M, C))
    torch.testing.assert_close(dg, gr.grad, rtol=2e-2, atol=0.4)  # bf16-rounded g summed over 1536 rows
    torch.testing.assert_close(db, br.grad, rtol=2e-2, atol=0.4)
    torch.testing.assert_close(dz.float(), zr.grad, rtol=3e-2, atol=3e-2)
    # eval mode: running statistics
    run(lambda p: p.bn_finalize(None, M, C, gamma, beta, rm, rv, 0.1, 1e-5, scale, shift, None, None))
    torch.testing.assert_close(scale, gamma / torch.sqrt(rv + 1e-5), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('B,K,H,W,LD', [(3, 17, 64, 48, 64), (2, 17, 10, 13, 32), (1, 32, 9, 7, 32), (2, 40, 8, 8, 64),
                                        (1, 133, 16, 12, 192)])
def test_heatmap_loss_and_argmax(dev, B, K, H, W, LD):
    """(K <= 32 in rows of <= 64 columns: the tiled kernel; wider heads: one pixel per thread)"""
    heat_nchw = rnd((B, K, H, W), dev, 1)
    heat = torch.zeros(B * H * W, LD, device=dev)
    heat[:, :K] = heat_nchw.permute(0, 2, 3, 1).reshape(-1, K)
    target = torch.rand(B, K, H, W, device=dev)
    tw = (torch.rand(B, K, device=dev) > 0.3).float()
    out = torch.empty(B, K, H, W, device=dev)
    loss = torch.zeros(1, device=dev)
    dheat = torch.full((B * H * W, LD), float('nan'), device=dev, dtype=torch.bfloat16)
    run(lambda p: p.heatmap_loss(heat, LD, out, target, tw, loss, dheat, B, K, H * W, 1.0))
    assert torch.equal(out, heat_nchw)
    pr = heat_nchw.clone().requires_grad_()
    w = tw.view(B, K, 1, 1)
    ref = sum(torch.nn.functional.mse_loss(pr[:, k] * w[:, k], target[:, k] * w[:, k]) for k in range(K)) / K
    ref.backward()
    torch.testing.assert_close(loss[0], ref, rtol=1e-5, atol=1e-7)
    dref = pr.grad.permute(0, 2, 3, 1).reshape(-1, K)
    torch.testing.assert_close(dheat[:, :K].float(), dref.bfloat16().float(), rtol=1e-2, atol=1e-9)
    assert K == LD or dheat[:, K:].float().abs().max() == 0
    idx = torch.empty(B * K, device=dev, dtype=torch.int32)
    val = torch.empty(B * K, device=dev)
    run(lambda p: p.argmax_rows(heat_nchw, B * K, H * W, idx, val))
    flat = heat_nchw.view(B * K, -1)
    assert torch.equal(idx.long(), flat.argmax(1))
    assert torch.equal(val, flat.max(1).values)


def test_adamw_matches_torch(dev):
    from vitpose_amd.engine import ParamStore
    from vitpose_amd.optim import FlatAdamW
    shapes = [('backbone.blocks.0.attn.qkv.weight', (96, 40)), ('backbone.blocks.0.attn.qkv.bias', (97,)),
              ('keypoint_head.final_layer.weight', (17, 33, 1, 1))]
    st = ParamStore(shapes, dev)
    ref = []
    for i, (n, s) in enumerate(shapes):
        st.view(n).copy_(rnd(s, dev, i))
        ref.append(st.view(n).clone().requires_grad_())
    groups = {shapes[0][0]: dict(lr_scale=0.5, weight_decay=0.1), shapes[1][0]: dict(lr_scale=0.5, weight_decay=0.0),
              shapes[2][0]: dict(lr_scale=1.0, weight_decay=0.1)}
    opt = FlatAdamW(st, groups, lr=1e-2, max_grad_norm=1.0)
    topt = torch.optim.AdamW([dict(params=[ref[0]], lr=5e-3, weight_decay=0.1),
                              dict(params=[ref[1]], lr=5e-3, weight_decay=0.0),
                              dict(params=[ref[2]], lr=1e-2, weight_decay=0.1)], lr=1e-2)
    for step in range(3):
        for i, (n, s) in enumerate(shapes):
            gr = rnd(s, dev, 10 * step + i) * 3
            st.gview(n).copy_(gr)
            ref[i].grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        topt.step()
        opt.step()
    for i, (n, s) in enumerate(shapes):
        torch.testing.assert_close(st.view(n), ref[i].detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(st.hview(n).float(), ref[i].detach().bfloat16().float(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize('xk,yk', [(False, False), (False, True)])
def test_gemm_tail_tiles_split_along_k(dev, xk, yk):
    """More tiles than the chip holds at once: the last partial round is cut along K and finished
    by the fix-up kernel (integer data: bit-exact, with a fused epilogue on top)."""
    from vitpose_amd import _lib
    M, N, K = 66 * 128 + 40, 1024, 512          # 67 x 8 = 536 tiles > 512 resident workgroups
    X, Y = ints((M, K), dev, seed=21), ints((N, K), dev, seed=22)
    bias = ints((N,), dev, seed=23)
    ys = (Y.t() if yk else Y).contiguous().bfloat16()
    out = torch.full((M, N), float('nan'), device=dev)
    run(lambda p: p.gemm(X.bfloat16(), ys, out, M, N, K, ldx=K, ldy=ys.shape[1], ldo=N, y_kmajor=yk,
                         epilogue=_lib.VP_EPI_F32, bias=bias, tail_force=True))
    assert torch.equal(out, X @ Y.t() + bias)


@pytest.mark.parametrize('upsample,fk,align', [(4, 3, False), (4, 1, False), (0, 3, False), (2, 3, True)])
def test_simple_decoder_fwd_bwd(dev, upsample, fk, align):
    """ReLU -> bilinear resize -> Conv2d (topdown_heatmap_simple_head.py:278-287,132-139) as
    GEMM-on-the-low-res-grid + interpolate-shift-sum, against torch fp32 autograd on the same
    bf16-rounded feature / weights.  Tolerance: bf16 rounding of G's operands only (f32 accumulation):
    2e-2 relative to the tensor's largest magnitude."""
    import torch.nn.functional as F
    from vitpose_amd.engine import ParamStore, Workspace
    from vitpose_amd.head_engine import HeadRunner, head_param_shapes
    from vitpose_amd.plan import Plan
    B, C, K, Hg, Wg = 2, 64, 17, 16, 12
    cfg = dict(in_channels=C, out_channels=K, num_deconv_layers=0, num_deconv_filters=[], num_deconv_kernels=[],
               upsample=upsample, align_corners=align, extra=dict(final_conv_kernel=fk))
    params, _ = head_param_shapes(cfg)
    st = ParamStore(params, dev)
    Wt = (rnd((K, C, fk, fk), dev, 1, 0.2)).bfloat16().float()
    bias = rnd((K,), dev, 2)
    st.load({'final_layer.weight': Wt, 'final_layer.bias': bias})
    hr = HeadRunner(cfg, st, {}, (Hg, Wg))
    prep = Plan()
    hr.record_weight_prep(prep)
    prep.run()
    feat = rnd((B, C, Hg, Wg), dev, 3).bfloat16().float()
    feat_cl = feat.permute(0, 2, 3, 1).reshape(B * Hg * Wg, C).contiguous().bfloat16()
    Ho, Wo = hr.out_grid
    heat = torch.empty(B, K, Ho, Wo, device=dev)
    target = rnd((B, K, Ho, Wo), dev, 4)
    tw = (rnd((B, K), dev, 5) > -0.5).float()
    loss = torch.zeros(1, device=dev)
    ws = Workspace(dev)
    plan = Plan()
    plan.memset0(st.grad)
    dpred = hr.record_forward(plan, ws, feat_cl, B, True, heat, target, tw, loss, want_grad=True)
    dfeat = hr.record_backward(plan, ws, feat_cl, dpred, B)
    plan.run()
    torch.cuda.synchronize()
    # torch reference
    f = feat.clone().requires_grad_()
    Wr, br = Wt.clone().requires_grad_(), bias.clone().requires_grad_()
    x = f
    if upsample > 0:
        x = F.interpolate(F.relu(x), scale_factor=upsample, mode='bilinear', align_corners=align)
    ref = F.conv2d(x, Wr, br, padding=1 if fk == 3 else 0)
    w = tw.view(B, K, 1, 1)
    rl = ((w * (ref - target)) ** 2).mean()
    rl.backward()

    def chk(got, want, tol=2e-2):
        err = (got.float() - want).abs().max().item()
        assert err <= tol * want.abs().max().item() + 1e-6, (err, want.abs().max().item())
    chk(heat, ref.detach())
    assert abs(loss.item() - rl.item()) < 2e-2 * rl.item()
    chk(dfeat.view(B, Hg, Wg, C).permute(0, 3, 1, 2), f.grad, 3e-2)
    chk(st.gview('final_layer.weight'), Wr.grad, 3e-2)
    chk(st.gview('final_layer.bias'), br.grad, 3e-2)


def test_nchw_bridges_and_relu(dev):
    from vitpose_amd import _lib
    B, C, HW, ld = 3, 17, 48, 64
    src = rnd((B, C, HW), dev, 1)
    dst = torch.full((B * HW, ld), float('nan'), device=dev, dtype=torch.bfloat16)
    run(lambda p: p.nchw_to_cl(src, dst, B, C, HW, ld))
    assert torch.equal(dst.view(B, HW, ld)[:, :, :C].float(), src.bfloat16().float().transpose(1, 2))
    assert torch.equal(dst.view(B, HW, ld)[:, :, C:].float(), torch.zeros(B, HW, ld - C, device=dev))
    x = rnd((4096,), dev, 2).bfloat16()
    y = torch.empty_like(x)
    run(lambda p: p.relu_bf16(x, y, x.numel()))
    assert torch.equal(y, torch.relu(x))
    s = torch.zeros(C, device=dev)
    run(lambda p: p.plane_sum(src, s, B, C, HW))
    torch.testing.assert_close(s, src.sum((0, 2)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('xk,yk', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(512, 256, 256), (520, 264, 320), (256, 256, 64), (1024, 768, 768), (768, 512, 128),
                                   (384, 192, 192), (576, 392, 448)])
@pytest.mark.parametrize('mode', [2, 3])
def test_gemm256_layouts_exact(dev, xk, yk, M, N, K, mode):
    """The 256 x 256 ping-pong kernel (tile_mode=2): integer operands make every product and f32 sum
    exact, so any fragment / swizzle / pipeline-slot mistake (including a read of a half tile that
    has not landed) shows as a mismatch.  Partial tiles, a single K step and many K steps."""
    from vitpose_amd import _lib
    X = ints((M, K), dev, seed=1)
    Y = ints((N, K), dev, seed=2)
    ref = X @ Y.t()
    xs = (X.t() if xk else X).contiguous().bfloat16()
    ys = (Y.t() if yk else Y).contiguous().bfloat16()
    for rep in range(3):
        out = torch.full((M, N), float('nan'), device=dev)
        run(lambda p: p.gemm(xs, ys, out, M, N, K, ldx=xs.shape[1], ldy=ys.shape[1], ldo=N, x_kmajor=xk, y_kmajor=yk,
                             epilogue=_lib.VP_EPI_F32, tile_mode=mode))
        assert torch.equal(out, ref), rep


@pytest.mark.parametrize('mode', [2, 3])
def test_gemm256_epilogues_and_splitk(dev, mode):
    from vitpose_amd import _lib
    M, N, K, T = 768, 512, 256, 192
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.1).bfloat16()
    bias = rnd((N,), dev, 5)
    acc = X.float() @ Y.float().t()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    u = torch.empty_like(out)
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, out2=u, epilogue=_lib.VP_EPI_GELU,
                         tile_mode=mode))
    torch.testing.assert_close(u.float(), (acc + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(acc + bias), rtol=2e-2, atol=2e-2)
    resid = rnd((M, N), dev, 7)
    rs = torch.tensor([0.0, 1.0 / 0.7, 1.0, 2.0], device=dev)
    o32 = torch.empty(M, N, device=dev)
    run(lambda p: p.gemm(X, Y, o32, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, resid=resid, rowscale=rs,
                         rows_per_scale=T, epilogue=_lib.VP_EPI_RESID_F32, tile_mode=mode))
    torch.testing.assert_close(o32, resid + rs.repeat_interleave(T)[:, None] * (acc + bias), rtol=1e-4, atol=1e-4)
    # weight gradient: both operands reduction-major, K split into slabs + deterministic reduce
    Mt = 2048
    A, Bm = ints((Mt, 512), dev, seed=8).bfloat16(), ints((Mt, 256), dev, seed=9).bfloat16()
    dw = torch.empty(512, 256, device=dev)
    slab = torch.empty(4 * 512 * 256, device=dev)
    run(lambda p: p.wgrad(A, Bm, dw, 512, 256, Mt, ldx=512, ldy=256, split_k=4, slab=slab, tile_mode=mode))
    assert torch.equal(dw, A.float().t() @ Bm.float())


# ------------------------------------------------------------------ 192 x 192 tiles, K tile split between the wave groups
@pytest.mark.parametrize('xk,yk', [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize('M,N,K', [(512, 256, 256), (520, 264, 320), (192, 192, 128), (1024, 768, 768), (384, 192, 192),
                                   (576, 392, 448), (8, 8, 128)])
def test_gemm_ksplit_layouts_exact(dev, xk, yk, M, N, K):
    """tile_mode 7 (a wave owns a 96 x 96 block for one 32-wide half of every K tile; the halves meet through LDS):
    integer operands, so any fragment / sub-step / exchange mistake shows as a mismatch.  Partial tiles, 2 .. 12 K tiles."""
    from vitpose_amd import _lib
    X = ints((M, K), dev, seed=1)
    Y = ints((N, K), dev, seed=2)
    ref = X @ Y.t()
    xs = (X.t() if xk else X).contiguous().bfloat16()
    ys = (Y.t() if yk else Y).contiguous().bfloat16()
    for rep in range(3):
        out = torch.full((M, N), float('nan'), device=dev)
        run(lambda p: p.gemm(xs, ys, out, M, N, K, ldx=xs.shape[1], ldy=ys.shape[1], ldo=N, x_kmajor=xk, y_kmajor=yk,
                             epilogue=_lib.VP_EPI_F32, tile_mode=7))
        assert torch.equal(out, ref), rep


def test_gemm_ksplit_epilogues(dev):
    from vitpose_amd import _lib
    M, N, K, T = 768, 512, 256, 192
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.1).bfloat16()
    bias = rnd((N,), dev, 5)
    acc = X.float() @ Y.float().t()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, epilogue=_lib.VP_EPI_BF16, tile_mode=7))
    torch.testing.assert_close(out.float(), (acc + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)
    Wt = rnd((N, K), dev, 8, 0.1).bfloat16()        # dgrad layout: [reduction = N][outputs = K]
    g = rnd((M, N), dev, 6).bfloat16()
    dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.gemm(g, Wt, dx, M, K, N, ldx=N, ldy=K, ldo=K, y_kmajor=True, epilogue=_lib.VP_EPI_BF16, tile_mode=7))
    torch.testing.assert_close(dx.float(), g.float() @ Wt.float(), rtol=2e-2, atol=2e-2)
    resid = rnd((M, N), dev, 7)
    rs = torch.tensor([0.0, 1.0 / 0.7, 1.0, 2.0], device=dev)
    o32 = torch.empty(M, N, device=dev)
    run(lambda p: p.gemm(X, Y, o32, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, resid=resid, rowscale=rs,
                         rows_per_scale=T, epilogue=_lib.VP_EPI_RESID_F32, tile_mode=7))
    torch.testing.assert_close(o32, resid + rs.repeat_interleave(T)[:, None] * (acc + bias), rtol=1e-4, atol=1e-4)
    with pytest.raises(Exception):                 # a single K tile has no second half to split off
        run(lambda p: p.gemm(X[:, :64].contiguous(), Y[:, :64].contiguous(), o32, M, N, 64, ldx=64, ldy=64, ldo=N,
                             epilogue=_lib.VP_EPI_F32, tile_mode=7))


@pytest.mark.parametrize('mode,M,N,K,yk', [(6, 6144, 1024, 4096, False), (6, 6144, 1024, 3072, True),
                                          (7, 12288, 768, 3072, False), (7, 12288, 768, 2304, True)])
def test_gemm_new_schedules_at_model_sizes(dev, mode, M, N, K, yk):
    """tile_mode 6 / 7 on the products they are chosen for (ViTPose-L N = 1024 at 6144 rows; ViTPose-B long reductions):
    random data against an f32 matmul of the same bf16 operands, and bit-identical to the 192 x 192 schedule's output
    wherever the summation order per output element is the same (mode 6; mode 7 adds its two halves last)."""
    from vitpose_amd import _lib
    X, W = rnd((M, K), dev, 21).bfloat16(), rnd((K, N) if yk else (N, K), dev, 22, 0.05).bfloat16()
    ref = X.float() @ (W.float() if yk else W.float().t())
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    base = torch.empty_like(out)
    run(lambda p: p.gemm(X, W, out, M, N, K, ldx=K, ldy=W.shape[1], ldo=N, y_kmajor=yk, epilogue=_lib.VP_EPI_BF16,
                         tile_mode=mode))
    run(lambda p: p.gemm(X, W, base, M, N, K, ldx=K, ldy=W.shape[1], ldo=N, y_kmajor=yk, epilogue=_lib.VP_EPI_BF16,
                         tile_mode=3))
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=2e-2)
    if mode == 6:
        assert torch.equal(out, base)
    else:
        torch.testing.assert_close(out.float(), base.float(), rtol=2 ** -6, atol=1e-2)   # a different f32 summation order: at most two bf16 steps


def test_wgrad_grouped_ksplit_at_model_size(dev):
    """The four Linear weight gradients of one ViTPose-B Block (64 images) on the K-split schedule, with the row sums of
    qkv and fc1, against f32 matmuls."""
    Kt, E = 12288, 768
    shapes = [(3 * E, E, True), (E, E, False), (4 * E, E, True), (E, 4 * E, False)]
    probs, refs = [], []
    for i, (M, N, with_rs) in enumerate(shapes):
        x, y = rnd((Kt, M), dev, 30 + 2 * i, 0.1).bfloat16(), rnd((Kt, N), dev, 31 + 2 * i, 0.1).bfloat16()
        out = torch.full((M, N), float('nan'), device=dev)
        rs = torch.full((M,), float('nan'), device=dev) if with_rs else None
        probs.append((x, y, out, M, N, M, N, rs))
        refs.append((x.float().t() @ y.float(), x.float().sum(0)))
    run(lambda p: p.wgrad_group(probs, Kt, tile_mode=7))
    for (x, y, out, *rest), (ref, rsum) in zip(probs, refs):
        torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-2)
        if rest[-1] is not None:
            torch.testing.assert_close(rest[-1], rsum, rtol=1e-3, atol=1e-2)


# ------------------------------------------------------------------ fp32-accurate path
def to_pair(x):
    """Reference of the hi / lo pair layout: [R][K] f32 -> [R][2K] bf16, groups of 32 hi then their 32 lo."""
    R, K = x.shape
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return torch.stack([hi.view(R, K // 32, 32), lo.view(R, K // 32, 32)], 2).reshape(R, 2 * K).contiguous()


def from_pair(p):
    R, K2 = p.shape
    v = p.float().view(R, K2 // 64, 2, 32)
    return (v[:, :, 0] + v[:, :, 1]).reshape(R, K2 // 2)


def test_split_pair_layout(dev):
    x = rnd((96, 160), dev, 1)
    pair = torch.zeros(96, 320, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.split_pair(x, pair, x.numel()))
    assert torch.equal(pair, to_pair(x))
    assert (from_pair(pair) - x).abs().max() <= x.abs().max() * 2 ** -16


@pytest.mark.parametrize('mode', [1, 2, 3, 4])
@pytest.mark.parametrize('M,N,K', [(384, 256, 768), (200, 136, 96), (192, 192, 32), (768, 512, 1024)])
def test_split3_gemm_reaches_fp32_accuracy(dev, mode, M, N, K):
    """One split3 launch on pair operands is ~1e-5 from an fp64 product (a single bf16 GEMM is ~4e-3), for every
    tile schedule; integer operands (lo = 0) come out exactly."""
    from vitpose_amd import _lib
    x, w, bias = rnd((M, K), dev, 1), rnd((N, K), dev, 2, 0.05), rnd((N,), dev, 3)
    xp, wp = to_pair(x), to_pair(w)
    acc = torch.full((M, N), float('nan'), device=dev)
    run(lambda p: p.gemm(xp, wp, acc, M, N, K, ldx=2 * K, ldy=2 * K, ldo=N, epilogue=_lib.VP_EPI_F32, bias=bias,
                         split3=True, tile_mode=mode))
    ref = (x.double() @ w.double().t() + bias.double()).float()
    err = (acc - ref).abs().max().item()
    one = (x.bfloat16().float() @ w.bfloat16().float().t() + bias - ref).abs().max().item()
    assert err < 2e-4 * ref.abs().max().item() and err < one / 30, (err, one)
    xi, wi = ints((M, K), dev, seed=4), ints((N, K), dev, seed=5)
    run(lambda p: p.gemm(to_pair(xi), to_pair(wi), acc, M, N, K, ldx=2 * K, ldy=2 * K, ldo=N, epilogue=_lib.VP_EPI_F32,
                         split3=True, tile_mode=mode))
    assert torch.equal(acc, xi @ wi.t())


@pytest.mark.parametrize('mode', [1, 2, 3, 4])
def test_split3_gemm_epilogues(dev, mode):
    from vitpose_amd import _lib
    M, N, K, T = 384, 256, 128, 192
    x, w = rnd((M, K), dev, 3), rnd((N, K), dev, 4, 0.1)
    bias, cs = rnd((N,), dev, 5), rnd((N,), dev, 6)
    xp, wp = to_pair(x), to_pair(w)
    acc = (x.double() @ w.double().t()).float()
    # residual + per-sample row scale; position rows
    resid, pos = rnd((M, N), dev, 7), rnd((T, N), dev, 8)
    rs = torch.tensor([0.5, 1.0 / 0.7], device=dev)
    o32 = torch.empty(M, N, device=dev)
    run(lambda p: p.gemm(xp, wp, o32, M, N, K, ldx=2 * K, ldy=2 * K, ldo=N, bias=bias, resid=resid, rowscale=rs,
                         rows_per_scale=T, epilogue=_lib.VP_EPI_RESID_F32, split3=True, tile_mode=mode))
    torch.testing.assert_close(o32, resid + rs.repeat_interleave(T)[:, None] * (acc + bias), rtol=1e-4, atol=1e-4)
    run(lambda p: p.gemm(xp, wp, o32, M, N, K, ldx=2 * K, ldy=2 * K, ldo=N, bias=bias, pos=pos, pos_rows=T,
                         epilogue=_lib.VP_EPI_RESID_F32, split3=True, tile_mode=mode))
    torch.testing.assert_close(o32, acc + bias + pos.repeat(2, 1), rtol=1e-4, atol=1e-4)
    # pair outputs: none / exact GELU / scale + shift + ReLU
    out = torch.zeros(M, 2 * N, device=dev, dtype=torch.bfloat16)
    for act, fn in ((0, lambda v: v), (1, torch.nn.functional.gelu), (2, torch.relu)):
        run(lambda p: p.gemm(xp, wp, out, M, N, K, ldx=2 * K, ldy=2 * K, ldo=2 * N, bias=bias, colscale=cs,
                             epilogue=_lib.VP_EPI_PAIR, pair_act=act, split3=True, tile_mode=mode))
        want = fn(acc * cs + bias)
        got = from_pair(out)
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)
        hi = out.view(M, N // 32, 2, 32)[:, :, 0].reshape(M, N)
        assert (hi.float() - got).abs().max() <= got.abs().max() * 2 ** -8      # the hi half is bf16(value)


def test_layernorm_fwd_pair(dev):
    M, E = 400, 768
    x, g, b = rnd((M, E), dev, 1) * 2 + 0.3, rnd((E,), dev, 2) + 1, rnd((E,), dev, 3)
    ref = torch.nn.functional.layer_norm(x.double(), (E,), g.double(), b.double(), 1e-6).float()
    for relu in (False, True):
        y = torch.zeros(M, 2 * E, device=dev, dtype=torch.bfloat16)
        run(lambda p: p.layernorm_fwd_pair(x, g, b, y, M, E, relu=relu))
        want = ref.relu() if relu else ref
        torch.testing.assert_close(from_pair(y), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('H,d', [(2, 32), (6, 64), (2, 80)])
def test_attention_split(dev, H, d):
    B, T, E = 2, 192, H * d
    qkv = rnd((B * T, 3 * E), dev, 1) * 1.5
    out = torch.zeros(B * T, 2 * E, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.attention_split(qkv, out, B, H, d, T, d ** -0.5))
    ref, _ = _attn_ref(qkv.double(), B, H, d, d ** -0.5)
    torch.testing.assert_close(from_pair(out), ref.float(), rtol=1e-4, atol=1e-4)   # scores carry ~2^-16 of |q||k|


def test_patch_im2col_pair(dev):
    B = 2
    img = rnd((B, 3, 256, 192), dev, 1)
    col = torch.zeros(B * 192, 2 * 768, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.patch_im2col_pair(img, col, B, 3, 256, 192, 16, 16, 2))
    ref = torch.nn.functional.unfold(img, 16, padding=2, stride=16).transpose(1, 2).reshape(B * 192, 768)
    assert torch.equal(col, to_pair(ref))


def test_deconv_fwd_split(dev):
    """ConvTranspose2d(k4, s2, p1) + folded BatchNorm + ReLU on pair arrays against torch fp64."""
    B, H, W, C, Co = 2, 8, 6, 64, 64
    x = rnd((B, C, H, W), dev, 1)
    wt = rnd((C, Co, 4, 4), dev, 2, 0.05)
    cs, sh = rnd((Co,), dev, 3) * 0.2 + 1, rnd((Co,), dev, 4) * 0.1
    ref = torch.relu(torch.nn.functional.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1)
                     * cs.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float()
    xin = to_pair(x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous())
    # wph[phase][co][(ty, tx, ci)] = W[ci][co][(1 - py) + 2 ty][(1 - px) + 2 tx]
    wph = torch.empty(4, Co, 4 * C, device=dev)
    for py in range(2):
        for px in range(2):
            sub = wt[:, :, (1 - py)::2, (1 - px)::2]                       # [ci][co][ty][tx]
            wph[py * 2 + px] = sub.permute(1, 2, 3, 0).reshape(Co, 4 * C)
    wpp = to_pair(wph.view(4 * Co, 4 * C)).view(4, Co, 8 * C)
    out = torch.zeros(B * 4 * H * W, 2 * Co, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.deconv_fwd_split(xin, wpp, out, B, H, W, C, Co, cs, sh))
    got = from_pair(out).view(B, 2 * H, 2 * W, Co).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('mode', [2, 3, 7])
def test_wgrad_grouped_exact(dev, mode):
    """Several weight-gradient problems of different shapes in one launch: every tile finds its own
    problem (integer data: exact), including edge tiles and outputs with a wider leading dimension."""
    K = 640
    shapes = [(768, 384), (200, 136), (64, 1000), (384, 192), (8, 8), (520, 264)]
    probs, refs = [], []
    for i, (M, N) in enumerate(shapes):
        x, y = ints((K, M), dev, seed=2 * i), ints((K, N), dev, seed=2 * i + 1)
        out = torch.full((M, N), float('nan'), device=dev)
        rs = torch.full((M,), float('nan'), device=dev) if i % 2 == 0 else None   # every other problem: row sums too
        probs.append((x.bfloat16(), y.bfloat16(), out, M, N, M, N, rs))
        refs.append(x.t() @ y)
    run(lambda p: p.wgrad_group(probs, K, tile_mode=mode))
    for (x, y, out, *rest), ref in zip(probs, refs):
        assert torch.equal(out, ref)
        if rest[-1] is not None:
            assert torch.equal(rest[-1], x.float().sum(0))


# ------------------------------------------------------------------ 192 x 128 tiles: 2 workgroups / CU (tile_mode 4), and the
# three-stage one-workgroup-per-CU kernel on the same tile (tile_mode 6)
@pytest.mark.parametrize('mode', [4, 6])
@pytest.mark.parametrize('yk', [False, True])
@pytest.mark.parametrize('M,N,K', [(768, 512, 256), (520, 264, 320), (192, 128, 64), (1152, 768, 768), (200, 136, 192),
                                   (8, 8, 64), (1536, 1024, 1024)])
def test_gemm_w2_layouts_exact(dev, mode, yk, M, N, K):
    from vitpose_amd import _lib
    X, Y = ints((M, K), dev, seed=1), ints((N, K), dev, seed=2)
    ys = (Y.t() if yk else Y).contiguous().bfloat16()
    for rep in range(2):
        out = torch.full((M, N), float('nan'), device=dev)
        run(lambda p: p.gemm(X.bfloat16(), ys, out, M, N, K, ldx=K, ldy=ys.shape[1], ldo=N, y_kmajor=yk,
                             epilogue=_lib.VP_EPI_F32, tile_mode=mode))
        assert torch.equal(out, X @ Y.t()), rep


@pytest.mark.parametrize('mode', [4, 6])
def test_gemm_w2_epilogues(dev, mode):
    from vitpose_amd import _lib
    M, N, K, T = 768, 512, 256, 192
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.1).bfloat16()
    bias = rnd((N,), dev, 5)
    acc = X.float() @ Y.float().t()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    u = torch.empty_like(out)
    run(lambda p: p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, out2=u, epilogue=_lib.VP_EPI_GELU,
                         tile_mode=mode))
    torch.testing.assert_close(u.float(), (acc + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(acc + bias), rtol=2e-2, atol=2e-2)
    g = rnd((M, N), dev, 6).bfloat16()              # DGELU: dgrad product (y k-major) masked by gelu'(aux)
    Wt = rnd((N, K), dev, 8, 0.1).bfloat16()        # [reduction = N][outputs = K]
    du = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    aux = rnd((M, K), dev, 9).bfloat16()
    run(lambda p: p.gemm(g, Wt, du, M, K, N, ldx=N, ldy=K, ldo=K, y_kmajor=True, epilogue=_lib.VP_EPI_DGELU, aux=aux,
                         ldaux=K, tile_mode=mode))
    a32 = aux.float().requires_grad_()
    torch.nn.functional.gelu(a32).sum().backward()
    torch.testing.assert_close(du.float(), (g.float() @ Wt.float()) * a32.grad, rtol=2e-2, atol=2e-2)
    resid = rnd((M, N), dev, 7)
    rs = torch.tensor([0.0, 1.0 / 0.7, 1.0, 2.0], device=dev)
    o32 = torch.empty(M, N, device=dev)
    run(lambda p: p.gemm(X, Y, o32, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, resid=resid, rowscale=rs,
                         rows_per_scale=T, epilogue=_lib.VP_EPI_RESID_F32, tile_mode=mode))
    torch.testing.assert_close(o32, resid + rs.repeat_interleave(T)[:, None] * (acc + bias), rtol=1e-4, atol=1e-4)
    bad = torch.empty(K, M, device=dev, dtype=torch.bfloat16)
    with pytest.raises(Exception):                 # k-major x is outside this schedule
        run(lambda p: p.gemm(bad, Y, o32, M, N, K, ldx=M, ldy=K, ldo=N, x_kmajor=True, epilogue=_lib.VP_EPI_F32,
                             tile_mode=mode))


# ------------------------------------------------------------------ persistent 192 x 192 tiles (tile_mode 5)
@pytest.mark.parametrize('M,N,K', [(7680, 1536, 768), (12288, 3072, 768), (1920, 1000, 640), (3840, 4096, 1024),
                                   (192, 192, 640)])
def test_gemm_persistent_matches_one_tile_per_workgroup(dev, M, N, K):
    """The persistent kernel (a workgroup walks several output tiles and writes tile i under the K loop of tile
    i + 1) accumulates and rounds exactly as the one-tile-per-workgroup kernel of the same tile shape does: the
    outputs are bit-identical for every epilogue it supports, with one round of tiles or several, ragged N included."""
    from vitpose_amd import _lib
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.1).bfloat16()
    bias = rnd((N,), dev, 5)
    outs = {}
    for mode in (3, 5):
        o = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
        run(lambda p: p.gemm(X, Y, o, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, epilogue=_lib.VP_EPI_BF16, tile_mode=mode))
        g, u = torch.full_like(o, float('nan')), torch.full_like(o, float('nan'))
        run(lambda p: p.gemm(X, Y, g, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, out2=u, epilogue=_lib.VP_EPI_GELU,
                             tile_mode=mode))
        g1 = torch.full_like(o, float('nan'))      # inference: no pre-activation copy
        run(lambda p: p.gemm(X, Y, g1, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, epilogue=_lib.VP_EPI_GELU, tile_mode=mode))
        Yt = Y.t().contiguous()                     # [K][N]: the data-gradient layout
        aux = rnd((M, N), dev, 9).bfloat16()
        d = torch.full_like(o, float('nan'))
        run(lambda p: p.gemm(X, Yt, d, M, N, K, ldx=K, ldy=N, ldo=N, y_kmajor=True, epilogue=_lib.VP_EPI_DGELU, aux=aux,
                             ldaux=N, tile_mode=mode))
        outs[mode] = (o, g, u, g1, d)
    for a, b, name in zip(outs[3], outs[5], ('bias', 'gelu', 'pre-activation', 'gelu without copy', 'dgelu')):
        assert not torch.isnan(b.float()).any(), name
        assert torch.equal(a, b), name
    acc = X.float() @ Y.float().t() + bias
    torch.testing.assert_close(outs[5][0].float(), acc, rtol=1e-2, atol=1e-2)


def test_gemm_persistent_refuses_what_it_does_not_cover(dev):
    from vitpose_amd import _lib
    M, N, K = 384, 384, 256       # fewer than 9 K tiles
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.1).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    with pytest.raises(Exception):
        run(lambda p: p.gemm(X, Y, o, M, N, K, ldx=K, ldy=K, ldo=N, epilogue=_lib.VP_EPI_BF16, tile_mode=5))
    o32 = torch.empty(M, N, device=dev)
    X2, Y2 = rnd((M, 768), dev, 3).bfloat16(), rnd((N, 768), dev, 4, 0.1).bfloat16()
    with pytest.raises(Exception):                 # f32 outputs are outside this schedule
        run(lambda p: p.gemm(X2, Y2, o32, M, N, 768, ldx=768, ldy=768, ldo=N, epilogue=_lib.VP_EPI_F32, tile_mode=5))


@pytest.mark.parametrize('Ci,Co', [(64, 64), (768, 256), (40, 72)])
def test_deconv_weight_prep_layouts(dev, Ci, Co):
    wt = rnd((Ci, Co, 4, 4), dev, 1)
    wph = torch.full((4, Co, 4 * Ci), float('nan'), device=dev, dtype=torch.bfloat16)
    wd = torch.full((Ci, 16 * Co), float('nan'), device=dev, dtype=torch.bfloat16)
    run(lambda p: p.deconv_weight_prep(wt, wph, wd, Ci, Co))
    wb = wt.bfloat16()
    assert torch.equal(wd.view(Ci, 16, Co), wb.view(Ci, Co, 16).permute(0, 2, 1))
    for py in range(2):
        for px in range(2):
            want = wb[:, :, (1 - py)::2, (1 - px)::2]                  # [ci][co][ty][tx]
            assert torch.equal(wph[py * 2 + px].view(Co, 2, 2, Ci), want.permute(1, 2, 3, 0))


@pytest.mark.parametrize('B,H,W,C', [(2, 16, 12, 64), (1, 5, 7, 8), (3, 64, 48, 256)])
def test_im2col3x3_and_col2im3x3_bit_exact(dev, B, H, W, C):
    """3 x 3 / stride 1 / padding 1 conv as a GEMM: the column matrix [(b, y, x)][(ky, kx, c)] against F.unfold (bit-exact
    copy) and its adjoint against F.fold on small integers (exact in bf16), with and without the ReLU mask."""
    x = ints((B, C, H, W), dev, -3, 4, seed=1)
    xcl = x.permute(0, 2, 3, 1).contiguous().bfloat16().view(B * H * W, C)
    col = torch.full((B * H * W, 9 * C), float('nan'), device=dev, dtype=torch.bfloat16)
    run(lambda p: p.im2col3x3(xcl, col, B, H, W, C))
    # unfold orders a column (c, ky, kx); the kernel's is (ky, kx, c)
    ref = torch.nn.functional.unfold(x, 3, padding=1).view(B, C, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * C)
    assert torch.equal(col.float(), ref)
    d = ints((B * H * W, 9 * C), dev, -2, 3, seed=2).bfloat16()
    din = torch.full((B * H * W, C), float('nan'), device=dev, dtype=torch.bfloat16)
    dm = torch.full_like(din, float('nan'))
    run(lambda p: (p.col2im3x3(d, din, B, H, W, C), p.col2im3x3(d, dm, B, H, W, C, relu_aux=xcl)))
    fold_in = d.float().view(B, H * W, 9, C).permute(0, 3, 2, 1).reshape(B, C * 9, H * W)
    want = torch.nn.functional.fold(fold_in, (H, W), 3, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, C)
    assert torch.equal(din.float(), want)
    assert torch.equal(dm.float(), want * (xcl.float() > 0))


@pytest.mark.parametrize('E', [384, 768, 1024, 1280])
def test_layernorm_fwd_row_result_is_independent_of_the_call_size(dev, E):
    """A row's normalised values, mean and rstd are the same bits whatever the number of rows of the call, i.e.
    whichever wave and register buffer of the pipelined kernel the row lands in (an image's heat maps must not depend on
    the batch around it: tests/test_fullsize_gpu.py compares batch 64 with batch 2 bit for bit)."""
    M = 12288
    x = rnd((M, E), dev, 11, 3.0) + 0.5
    gamma, beta = rnd((E,), dev, 12).abs() + 0.5, rnd((E,), dev, 13)
    outs = []
    for rows in (M, 1000, 384, 7, 2, 1):
        y = torch.full((rows, E), float('nan'), device=dev, dtype=torch.bfloat16)
        y32 = torch.full((rows, E), float('nan'), device=dev)
        mean, rstd = torch.full((rows,), float('nan'), device=dev), torch.full((rows,), float('nan'), device=dev)
        xr = x[:rows].contiguous()
        run(lambda p: p.layernorm_fwd(xr, gamma, beta, y, y32, mean, rstd, rows, E))
        outs.append((rows, y, y32, mean, rstd))
    _, y0, y320, mean0, rstd0 = outs[0]
    ref = torch.nn.functional.layer_norm(x, (E,), gamma, beta, 1e-6)
    torch.testing.assert_close(y320, ref, rtol=1e-5, atol=1e-5)
    for rows, y, y32, mean, rstd in outs[1:]:
        assert torch.equal(y, y0[:rows]) and torch.equal(y32, y320[:rows]), rows
        assert torch.equal(mean, mean0[:rows]) and torch.equal(rstd, rstd0[:rows]), rows


# ------------------------------------------------------------------ LayerNorm folded into the next Linear (evaluation)
@pytest.mark.parametrize('mode', [2, 3, 6])
@pytest.mark.parametrize('M,N,K', [(768, 768, 768), (1000, 1280, 512), (384, 1024, 3072)])
def test_gemm_residual_epilogue_with_layernorm_statistics(dev, mode, M, N, K):
    """The producer side of the fold: a VP_EPI_RESID_F32 launch that also writes the bf16 copy of the stream and,
    per row and column tile, the sum and the sum of squares -- the f32 stream itself is bit-identical to the plain
    launch, the copy is its rounding, the statistics add up to the row's (ragged rows and column tiles included)."""
    from vitpose_amd import _lib
    X, Y = rnd((M, K), dev, 3).bfloat16(), rnd((N, K), dev, 4, 0.05).bfloat16()
    bias, resid = rnd((N,), dev, 5), rnd((M, N), dev, 6, 2.0)
    plain = torch.full((M, N), float('nan'), device=dev)
    run(lambda p: p.gemm(X, Y, plain, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, resid=resid,
                         epilogue=_lib.VP_EPI_RESID_F32, tile_mode=mode))
    out = torch.full((M, N), float('nan'), device=dev)
    xb = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
    stats = torch.full((M, (N + 127) // 128, 2), float('nan'), device=dev)
    got = {}
    run(lambda p: got.update(slots=p.gemm(X, Y, out, M, N, K, ldx=K, ldy=K, ldo=N, bias=bias, resid=resid, out2=xb,
                                         epilogue=_lib.VP_EPI_RESID_F32, tile_mode=mode, ln_stats_out=stats)))
    slots = got['slots']
    assert slots == -(-N // {2: 256, 3: 192, 6: 128}[mode])
    assert torch.equal(out, plain) and torch.equal(xb, out.bfloat16())
    st = stats.view(-1)[:M * slots * 2].view(M, slots, 2).double()      # [M][slots][2], compact in the allocation
    assert torch.isfinite(st).all()
    torch.testing.assert_close(st[..., 0].sum(1), out.double().sum(1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(st[..., 1].sum(1), (out.double() ** 2).sum(1), rtol=1e-5, atol=1e-3)


def _folded_case(dev, kind, M=960, E=768, N=1536, seed=20):
    """A residual stream [M][E] with the statistics a trained ViT shows (`outliers`: a few channels carry values
    30 x the rest) or with a common offset many times its spread (`offset`: the case in which E[x^2] - mean^2 and
    the bf16 rounding of the raw stream lose what the plain path keeps), LayerNorm parameters and a Linear."""
    x = rnd((M, E), dev, seed)
    if kind == 'outliers':
        x[:, ::97] = x[:, ::97] * 3 + 30.0
        x += rnd((M, 1), dev, seed + 1, 0.5)
    elif kind == 'offset':
        x += 10.0
    gamma, beta = rnd((E,), dev, seed + 2).abs() + 0.5, rnd((E,), dev, seed + 3, 0.3)
    W, b = rnd((N, E), dev, seed + 4, 0.04), rnd((N,), dev, seed + 5, 0.1)
    ref = torch.nn.functional.layer_norm(x.double(), (E,), gamma.double(), beta.double(), 1e-6) @ W.double().t() + b.double()
    return x, gamma, beta, W, b, ref


@pytest.mark.parametrize('mode', [1, 2, 3, 4, 6])
@pytest.mark.parametrize('kind', ['plain', 'outliers'])
def test_gemm_consumes_folded_layernorm(dev, mode, kind):
    """The consumer side: x W'^T on the bf16 copy of the RAW stream with W' = W diag(gamma), finished in the epilogue
    as rstd (acc - mean s) + c, against LayerNorm (f64) followed by the Linear -- as accurate as the plain bf16 path
    (LayerNorm launch, then the GEMM on its bf16 output), also when a few channels carry values 30 x the others (the
    massive activations of trained transformers), on every tile schedule, with and without the GELU."""
    from vitpose_amd import _lib
    M, E, N = 960, 768, 1536
    x, gamma, beta, W, b, ref = _folded_case(dev, kind, M, E, N)
    slots = 4
    stats = torch.stack([x.view(M, slots, E // slots).sum(2), (x.view(M, slots, E // slots) ** 2).sum(2)], dim=2).contiguous()
    wg = torch.empty(N, E, device=dev, dtype=torch.bfloat16)
    s, c = torch.empty(N, device=dev), torch.empty(N, device=dev)
    run(lambda p: p.ln_fold_weights(W, gamma, beta, b, wg, s, c, N, E))
    assert torch.equal(wg, (W * gamma).bfloat16())
    torch.testing.assert_close(s, wg.float().sum(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(c, (W.double() @ beta.double() + b.double()).float(), rtol=1e-5, atol=1e-5)
    xb = x.bfloat16()
    xn = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.layernorm_fwd(x, gamma, beta, xn, None, None, None, M, E))

    def rel(a, r):
        return ((a.double() - r).norm() / r.norm()).item()
    for epi, fin in ((_lib.VP_EPI_BF16, lambda t: t), (_lib.VP_EPI_GELU, torch.nn.functional.gelu)):
        folded = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
        run(lambda p: p.gemm(xb, wg, folded, M, N, E, ldx=E, ldy=E, ldo=N, colscale=s, bias=c, epilogue=epi,
                             tile_mode=mode, ln_stats=(stats, slots)))
        plain = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
        run(lambda p: p.gemm(xn, W.bfloat16(), plain, M, N, E, ldx=E, ldy=E, ldo=N, bias=b, epilogue=epi, tile_mode=mode))
        want = fin(ref)
        ef, ep = rel(folded, want), rel(plain, want)
        print(f'folded LayerNorm, {kind}, tile_mode {mode}, epilogue {epi}: folded {ef:.2e}, plain {ep:.2e}')
        assert not torch.isnan(folded.float()).any()
        assert ef < 1.3 * ep + 1e-4, (ef, ep)


def test_folded_layernorm_loses_precision_under_a_common_offset(dev):
    """Where the fold is WORSE, measured so that the limit is on record: a stream whose rows sit at 10 x their spread
    (mean / std = 10).  The raw stream rounded to bf16 keeps 8 bits of a number ten times larger than what the
    LayerNorm keeps, so the folded product is ~10 x less accurate than the plain path there.  Trained ViT streams are
    not of this kind (their large values sit in a few channels: the `outliers` case above, where both paths agree);
    `VITPOSE_LN_FOLD=0` selects the plain path."""
    from vitpose_amd import _lib
    M, E, N = 960, 768, 1536
    x, gamma, beta, W, b, ref = _folded_case(dev, 'offset', M, E, N)
    stats = torch.stack([x.view(M, 4, E // 4).sum(2), (x.view(M, 4, E // 4) ** 2).sum(2)], dim=2).contiguous()
    wg = torch.empty(N, E, device=dev, dtype=torch.bfloat16)
    s, c = torch.empty(N, device=dev), torch.empty(N, device=dev)
    run(lambda p: p.ln_fold_weights(W, gamma, beta, b, wg, s, c, N, E))
    folded = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.gemm(x.bfloat16(), wg, folded, M, N, E, ldx=E, ldy=E, ldo=N, colscale=s, bias=c, tile_mode=3,
                         ln_stats=(stats, 4)))
    xn = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.layernorm_fwd(x, gamma, beta, xn, None, None, None, M, E))
    plain = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run(lambda p: p.gemm(xn, W.bfloat16(), plain, M, N, E, ldx=E, ldy=E, ldo=N, bias=b, tile_mode=3))
    ef = ((folded.double() - ref).norm() / ref.norm()).item()
    ep = ((plain.double() - ref).norm() / ref.norm()).item()
    print(f'common offset 10 sigma: folded {ef:.2e}, plain {ep:.2e}')
    assert ep < 5e-3 and ef < 12 * ep and ef < 5e-2
