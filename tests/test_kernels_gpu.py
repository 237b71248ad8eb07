"""GPU numerics of the individual forward/backward kernels behind the C ABI against plain torch fp32
(autograd) on the CPU, on data kept away from the ReLU / max kinks so that a comparison at ~1e-5 is
meaningful (the whole-net gradient test can only be as tight as the reference's own reproducibility)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from ieee_amd import _lib
    return _lib, _lib.require_gpu()


def away_from_zero(t, margin=0.05):
    return torch.where(t.abs() < margin, torch.full_like(t, margin) * torch.sign(t + 1e-9), t)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("G,M,C,residual,relu", [(3, 1000, 64, False, True), (1, 515, 256, True, True),
                                                 (3, 384, 2048, False, False), (2, 4096, 512, True, True)])
def test_bn2d_fwd_bwd(dtype, G, M, C, residual, relu):
    L, lib = _lib()
    g = torch.Generator().manual_seed(C + M)
    rt = (lambda t: t.to(torch.bfloat16).float()) if dtype == torch.bfloat16 else (lambda t: t)
    y = rt(torch.randn(G, M, C, generator=g) * 2 + 0.5)
    res = rt(torch.randn(G, M, C, generator=g)) if residual else None
    gamma = torch.rand(G, C, generator=g) + 0.5
    beta = torch.randn(G, C, generator=g) * 0.3
    rm, rv = torch.randn(G, C, generator=g) * 0.1, torch.rand(G, C, generator=g) + 0.5
    dout = rt(torch.randn(G, M, C, generator=g))
    # torch reference (per group)
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rmr, rvr = rm.clone(), rv.clone()
    outs = []
    for i in range(G):
        o = F.batch_norm(yr[i], rmr[i], rvr[i], gr[i], br[i], True, 0.1, 1e-5)
        if residual:
            o = o + res[i]
        outs.append(F.relu(o) if relu else o)
    out_ref = torch.stack(outs)
    # keep the test away from the ReLU kink: drop elements whose pre-activation is within 0.02 of 0
    keep = (out_ref.detach().abs() > 0.02) if relu else torch.ones_like(out_ref, dtype=torch.bool)
    out_ref.backward(dout * keep)
    # native
    dev = "cuda"
    yd, resd = y.to(dev, dtype), (res.to(dev, dtype) if residual else None)
    out = torch.empty_like(yd)
    gd, bd = gamma.to(dev), beta.to(dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    stats = torch.empty(G, 4, C, device=dev)
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    npart = lib.ieee_bn_partial_floats(dt, M, C)
    part = torch.empty(G * npart + 64, device=dev)
    bits = torch.zeros(G * M * C // 8, device=dev, dtype=torch.uint8) if dtype == torch.bfloat16 else None
    L.check(lib.ieee_bn2d_fwd(L.ptr(yd), L.ptr(resd), L.ptr(out), dt, G, M, C, M * C, L.ptr(gd), L.ptr(bd), C,
                              L.ptr(rmd), L.ptr(rvd), C, L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, int(relu), 0, L.ptr(bits),
                              L.stream()))
    if bits is not None:   # relu_bits: bit e of byte k = [stored out[8k + e] > 0]
        want_bits = ((out.view(-1, 8) > 0).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(1)
        assert torch.equal(bits.to(torch.int32), want_bits)
    tol = dict(rtol=2e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.float().cpu(), out_ref.detach(), **tol)
    torch.testing.assert_close(rmd.cpu(), rmr, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rvd.cpu(), rvr, rtol=1e-4, atol=1e-5)
    # backward (mask from the reference output so both sides use the same ReLU decisions)
    mask = (out_ref.detach() > 0).to(dev, dtype) if relu else None
    dd = (dout * keep).to(dev, dtype)
    dy = torch.empty_like(yd)
    gout = torch.empty_like(yd)
    dgam, dbet = torch.zeros(G, C, device=dev), torch.zeros(G, C, device=dev)
    coef = torch.empty(G, 3, C, device=dev)
    L.check(lib.ieee_bn2d_bwd(L.ptr(dd), L.ptr(mask), L.ptr(yd), L.ptr(dy), L.ptr(gout), dt, G, M, C, M * C,
                              L.ptr(gd), C, L.ptr(stats), L.ptr(dgam), L.ptr(dbet), C, L.ptr(part), L.ptr(coef), 0,
                              0, 0, L.stream()))
    btol = dict(rtol=3e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(dy.float().cpu(), yr.grad, **btol)
    gtol = dict(rtol=3e-2, atol=0.5) if dtype == torch.bfloat16 else dict(rtol=1e-3, atol=5e-3)
    torch.testing.assert_close(dgam.cpu(), gr.grad, **gtol)
    torch.testing.assert_close(dbet.cpu(), br.grad, **gtol)
    g_ref = dout * keep * (out_ref.detach() > 0) if relu else dout
    torch.testing.assert_close(gout.float().cpu(), g_ref, **(dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16
                                                              else dict(rtol=1e-6, atol=1e-6)))


# general index math / the all-powers-of-two shift path / the LDS-tiled form (even extents, >= 64 columns), also ragged in h
@pytest.mark.parametrize("hw", [(14, 10), (16, 8), (8, 64), (6, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stem_pooled_bn_backward(dtype, hw):
    """ieee_bn2d_bwd_pooled (the stem: max-pool backward gathered inside the two passes of the BatchNorm backward) against
    the chain it replaces, maxpool_bwd -> bn2d_bwd(mask from y).  fp32: same numbers up to the contraction of one
    multiply-add; bf16: the chain rounds the un-pooled gradient to bf16 in between, the fused form does not"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    G, B, (H, W), C = 3, 3 if hw[0] == 14 else 4, hw, 64
    dev = "cuda"
    y = (torch.randn(G, B, H, W, C, generator=g) * 1.5 + 0.2).to(dev, dtype)
    gam, bet = (torch.rand(G, C, generator=g) + 0.5).to(dev), (torch.randn(G, C, generator=g) * 0.3).to(dev)
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    M = B * H * W
    stats = torch.empty(G, 4, C, device=dev)
    part = torch.empty(G * lib.ieee_bn_partial_floats(dt, M, C) + 64, device=dev)
    a = torch.empty_like(y)
    L.check(lib.ieee_bn2d_fwd(L.ptr(y), None, L.ptr(a), dt, G, M, C, M * C, L.ptr(gam), L.ptr(bet), C, None, None, 0,
                              L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, 1, 0, None, L.stream()))
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    out = torch.empty(G, B, Ho, Wo, C, device=dev, dtype=dtype)
    arg = torch.empty(G, B, Ho, Wo, C, device=dev, dtype=torch.uint8)
    L.check(lib.ieee_maxpool3x3s2_fwd(L.ptr(a), L.ptr(out), L.ptr(arg), dt, G, B, H, W, C, L.stream()))
    dpool = torch.randn(G, B, Ho, Wo, C, generator=g).to(dev, dtype)
    da = torch.empty_like(y)
    L.check(lib.ieee_maxpool3x3s2_bwd(L.ptr(dpool), L.ptr(arg), L.ptr(da), dt, G, B, H, W, C, L.stream()))
    res = []
    for fused in (False, True):
        dy = torch.empty_like(y)
        dg, db = torch.zeros(G, C, device=dev), torch.zeros(G, C, device=dev)
        coef = torch.empty(G, 3, C, device=dev)
        if fused:
            L.check(lib.ieee_bn2d_bwd_pooled(L.ptr(dpool), L.ptr(arg), L.ptr(y), L.ptr(dy), dt, G, B, H, W, C, L.ptr(gam), C,
                                             L.ptr(stats), L.ptr(dg), L.ptr(db), C, L.ptr(part), L.ptr(coef), 0, L.stream()))
        else:
            L.check(lib.ieee_bn2d_bwd(L.ptr(da), None, L.ptr(y), L.ptr(dy), None, dt, G, M, C, M * C, L.ptr(gam), C,
                                      L.ptr(stats), L.ptr(dg), L.ptr(db), C, L.ptr(part), L.ptr(coef), 0, 1, 0, L.stream()))
        res.append((dy, dg, db))
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(res[1][0].float(), res[0][0].float(), **tol)
    gt = dict(rtol=1e-5, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-2, atol=0.3)
    torch.testing.assert_close(res[1][1], res[0][1], **gt)
    torch.testing.assert_close(res[1][2], res[0][2], **gt)


@pytest.mark.parametrize("hw", [(12, 10), (16, 8), (7, 9), (9, 33), (8, 64)])    # >= 32 columns: the LDS-tiled form
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_relu_maxpool_in_one_pass_equals_the_two_kernel_form(dtype, hw):
    """ieee_bn_relu_maxpool3x3s2_fwd (the stem's training forward: the full-resolution activation is never written) against
    ieee_bn2d_fwd(out = a, relu) -> ieee_maxpool3x3s2_fwd(a): pooled values and argmax bytes must be the same bits."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(9)
    G, B, (H, W), C = 3, 3, hw, 64
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    y = (torch.randn(G, B, H, W, C, generator=g) * 1.3 - 0.1).cuda().to(dtype)
    gam, bet = (torch.rand(G, C, generator=g) + 0.5).cuda(), (torch.randn(G, C, generator=g) * 0.3).cuda()
    M = B * H * W
    stats = torch.empty(G, 4, C, device="cuda")
    part = torch.empty(G * lib.ieee_bn_partial_floats(dt, M, C) + 64, device="cuda")
    a = torch.empty_like(y)
    L.check(lib.ieee_bn2d_fwd(L.ptr(y), None, L.ptr(a), dt, G, M, C, M * C, L.ptr(gam), L.ptr(bet), C, None, None, 0,
                              L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, 1, 0, None, L.stream()))
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    outs = []
    for fused in (False, True):
        out = torch.full((G, B, Ho, Wo, C), -7.0, device="cuda", dtype=dtype)
        arg = torch.full((G, B, Ho, Wo, C), 77, device="cuda", dtype=torch.uint8)
        if fused:
            L.check(lib.ieee_bn_relu_maxpool3x3s2_fwd(L.ptr(y), L.ptr(stats), L.ptr(out), L.ptr(arg), dt, G, B, H, W, C, L.stream()))
        else:
            L.check(lib.ieee_maxpool3x3s2_fwd(L.ptr(a), L.ptr(out), L.ptr(arg), dt, G, B, H, W, C, L.stream()))
        outs.append((out, arg))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[1][0].float().max()) > 0


@pytest.mark.parametrize("hw", [(12, 10), (16, 8)])    # general index math / the all-powers-of-two shift path
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_fwd_bwd(dtype, hw):
    L, lib = _lib()
    g = torch.Generator().manual_seed(0)
    G, B, (H, W), C = 3, 2, hw, 64
    x = torch.randn(G, B, C, H, W, generator=g)
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()
    xr = x.clone().requires_grad_(True)
    o_ref = F.max_pool2d(xr.view(G * B, C, H, W), 3, 2, 1)
    Ho, Wo = o_ref.shape[-2:]
    dout = torch.randn(G * B, C, Ho, Wo, generator=g)
    if dtype == torch.bfloat16:
        dout = dout.to(dtype).float()
    o_ref.backward(dout)
    xd = x.permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
    out = torch.empty(G, B, Ho, Wo, C, device="cuda", dtype=dtype)
    arg = torch.empty(G, B, Ho, Wo, C, device="cuda", dtype=torch.uint8)
    dt = L.IEEE_BF16 if dtype == torch.bfloat16 else L.IEEE_F32
    L.check(lib.ieee_maxpool3x3s2_fwd(L.ptr(xd), L.ptr(out), L.ptr(arg), dt, G, B, H, W, C, L.stream()))
    assert torch.equal(out.float().cpu().permute(0, 1, 4, 2, 3).reshape(G * B, C, Ho, Wo), o_ref.detach())
    dd = dout.view(G, B, C, Ho, Wo).permute(0, 1, 3, 4, 2).contiguous().cuda().to(dtype)
    dx = torch.empty_like(xd)
    L.check(lib.ieee_maxpool3x3s2_bwd(L.ptr(dd), L.ptr(arg), L.ptr(dx), dt, G, B, H, W, C, L.stream()))
    tol = dict(rtol=1e-2, atol=1e-2) if dtype == torch.bfloat16 else dict(rtol=0, atol=0)
    torch.testing.assert_close(dx.float().cpu().permute(0, 1, 4, 2, 3), xr.grad, **tol)


def _cim_torch(F3, w, flags):
    """CIM tail + part pooling for 3 modalities in torch (same structure as the oracle), from the conv
    outputs y1,y2 and their BN parameters; returns pooled parts [3][B][6][C]"""
    outs = []
    for m in range(3):
        one = F.relu(F.batch_norm(w["y1"][m], None, None, w["g1"][m], w["b1"][m], True, 0.1, 1e-5))
        rest = F.relu(F.batch_norm(w["y2"][m], None, None, w["g2"][m], w["b2"][m], True, 0.1, 1e-5))
        if flags == 0:
            avg, mx = F.adaptive_avg_pool2d(rest, 1), F.adaptive_max_pool2d(rest, 1)
            mlp = lambda v: F.conv2d(F.relu(F.conv2d(v, w["w1"][m])), w["w2"][m])
            att = torch.sigmoid(mlp(avg) + mlp(mx))
            rest = att * rest + rest
        out = one + rest
        outs.append(F.adaptive_avg_pool2d(out, (6, 1))[..., 0].permute(0, 2, 1))   # [B,6,C]
    return torch.stack(outs)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1])
def test_cim_tail_chain_fwd_bwd(mode, dtype):
    """ca_pool -> CA MLP (grouped GEMMs) -> cim_tail forward, and the full backward chain (cim_tail_bwd_datt,
    cim_tail_bwd_g with its fused BN-backward sums, bn2d_bwd), vs autograd.  bf16: the conv outputs and the two
    gradient maps are bf16 tensors (the storage of the speed mode), everything pooled / reduced stays fp32; the
    reference is torch fp32 on the SAME bf16-rounded inputs, so what is measured is the kernels' own arithmetic and the
    one bf16 rounding of each stored gradient element."""
    L, lib = _lib()
    from ieee_amd import _ops
    g = torch.Generator().manual_seed(mode)
    B, H, W, C, hid = 3, 16, 8, 256, 16
    P = H * W
    bf = dtype == torch.bfloat16
    dt = L.IEEE_BF16 if bf else L.IEEE_F32
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf else (lambda t: t)
    w = {"y1": rnd(torch.randn(3, B, C, H, W, generator=g)), "y2": rnd(torch.randn(3, B, C, H, W, generator=g)),
         "g1": torch.rand(3, C, generator=g) + 0.5, "b1": torch.randn(3, C, generator=g) * 0.2,
         "g2": torch.rand(3, C, generator=g) + 0.5, "b2": torch.randn(3, C, generator=g) * 0.2,
         "w1": torch.randn(3, hid, C, 1, 1, generator=g) * 0.1, "w2": torch.randn(3, C, hid, 1, 1, generator=g) * 0.1}
    for k in w:
        w[k].requires_grad_(True)
    parts_ref = _cim_torch(None, w, mode)
    dparts = torch.randn(3, B, 6, C, generator=g)
    parts_ref.backward(dparts)

    dev = "cuda"
    nhwc = lambda t: t.detach().permute(0, 1, 3, 4, 2).contiguous().to(dev).to(dtype)
    y1, y2 = nhwc(w["y1"]), nhwc(w["y2"])
    st1, st2 = torch.empty(3, 4, C, device=dev), torch.empty(3, 4, C, device=dev)
    npart = lib.ieee_bn_partial_floats(dt, B * P, C)
    part = torch.empty(3 * npart + 64, device=dev)
    for y, ga, be, st in ((y1, w["g1"], w["b1"], st1), (y2, w["g2"], w["b2"], st2)):
        gd, bd = ga.detach().to(dev), be.detach().to(dev)
        L.check(lib.ieee_bn2d_fwd(L.ptr(y), None, None, dt, 3, B * P, C, B * P * C, L.ptr(gd), L.ptr(bd), C, None, None, 0,
                                  L.ptr(st), L.ptr(part), 0.1, 1e-5, 1, 1, 0, None, L.stream()))
    avgmax = torch.empty(3, 2 * B, C, device=dev)
    amax = torch.empty(3, B, C, device=dev, dtype=torch.int32)
    att = torch.zeros(3, B, C, device=dev)
    w1d, w2d = w["w1"].detach().view(3, hid, C).to(dev), w["w2"].detach().view(3, C, hid).to(dev)
    Hh = torch.empty(3, 2 * B, hid, device=dev)
    Hs = torch.empty(3, B, hid, device=dev)

    def tab(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    if mode == 0:
        L.check(lib.ieee_ca_pool(L.ptr(y2), L.ptr(st2), L.ptr(avgmax), ctypes.c_void_p(avgmax.data_ptr() + B * C * 4),
                                 2 * B * C, L.ptr(amax), dt, B, H, W, C, L.stream()))
        L.check(lib.ieee_sgemm_grouped(3, tab(list(avgmax)), tab(list(w1d)), tab(list(Hh)), None, 2 * B, hid, C, C, 1, C, 1,
                                       hid, 1.0, 1, 0, L.stream()))
        L.check(lib.ieee_ca_mix_fwd(L.ptr(Hh), L.ptr(Hs), B, hid, L.stream()))
        L.check(lib.ieee_sgemm_grouped(3, tab(list(Hs)), tab(list(w2d)), tab(list(att)), None, B, C, hid, hid, 1, hid, 1, C,
                                       1.0, 0, 0, L.stream()))
        L.check(lib.ieee_sigmoid_fwd(L.ptr(att), att.numel(), L.stream()))
    Pp = torch.empty(3, B, 6, C, device=dev)
    L.check(lib.ieee_cim_tail_fwd(L.ptr(y1), L.ptr(y2), L.ptr(st1), L.ptr(st2), L.ptr(att), L.ptr(Pp), dt, B, H, W, C, 6,
                                  mode, L.stream()))
    torch.testing.assert_close(Pp.cpu(), parts_ref.detach(), rtol=1e-4, atol=2e-5)        # inputs exact in both dtypes
    # ---- backward chain
    dP = dparts.to(dev)
    datt = torch.zeros(3, B, C, device=dev)
    davgmax = torch.zeros(3, 2 * B, C, device=dev)
    dw1, dw2 = torch.zeros_like(w1d), torch.zeros_like(w2d)
    if mode == 0:
        L.check(lib.ieee_cim_tail_bwd_datt(L.ptr(dP), L.ptr(y2), L.ptr(st2), L.ptr(datt), dt, B, H, W, C, 6, L.stream()))
        L.check(lib.ieee_sigmoid_bwd(L.ptr(datt), L.ptr(att), L.ptr(datt), datt.numel(), L.stream()))
        dHs, dH = torch.empty_like(Hs), torch.empty_like(Hh)
        L.check(lib.ieee_sgemm_grouped(3, tab(list(datt)), tab(list(Hs)), tab(list(dw2)), None, C, hid, B, 1, C, 1, hid, hid,
                                       1.0, 0, 0, L.stream()))
        L.check(lib.ieee_sgemm_grouped(3, tab(list(datt)), tab(list(w2d)), tab(list(dHs)), None, B, hid, C, C, 1, 1, hid, hid,
                                       1.0, 0, 0, L.stream()))
        L.check(lib.ieee_ca_mix_bwd(L.ptr(dHs), L.ptr(Hh), L.ptr(dH), B, hid, L.stream()))
        L.check(lib.ieee_sgemm_grouped(3, tab(list(dH)), tab(list(avgmax)), tab(list(dw1)), None, hid, C, 2 * B, 1, hid, 1, C, C,
                                       1.0, 0, 0, L.stream()))
        L.check(lib.ieee_sgemm_grouped(3, tab(list(dH)), tab(list(w1d)), tab(list(davgmax)), None, 2 * B, C, hid, hid, 1, 1, C,
                                       C, 1.0, 0, 0, L.stream()))
    g1, g2 = torch.empty_like(y1), torch.empty_like(y2)
    # the kernel also emits the BN-backward sums per sample ([3][2][C][B]) -> bn2d_bwd(stats_rblocks = B)
    bp1, bp2 = torch.zeros(3, 2, C, B, device=dev), torch.zeros(3, 2, C, B, device=dev)
    L.check(lib.ieee_cim_tail_bwd_g(L.ptr(dP), L.ptr(y1), L.ptr(y2), L.ptr(st1), L.ptr(st2), L.ptr(att), L.ptr(davgmax),
                                    ctypes.c_void_p(davgmax.data_ptr() + B * C * 4), 2 * B * C, L.ptr(amax), L.ptr(g1),
                                    L.ptr(g2), dt, B, H, W, C, 6, mode, L.ptr(bp1), L.ptr(bp2), L.stream()))
    for bp, gq, y in ((bp1, g1, y1), (bp2, g2, y2)):     # the fused sums are sums of the STORED (rounded) gradient
        torch.testing.assert_close(bp[:, 0].sum(-1), gq.float().view(3, -1, C).sum(1), rtol=1e-4, atol=2e-4 if bf else 1e-4)
        torch.testing.assert_close(bp[:, 1].sum(-1), (gq.float() * y.float()).view(3, -1, C).sum(1), rtol=1e-4,
                                   atol=2e-4 if bf else 1e-4)
    coef = torch.empty(3, 3, C, device=dev)
    res = {}
    for name, gq, y, ga, st, bp in (("1", g1, y1, w["g1"], st1, bp1), ("2", g2, y2, w["g2"], st2, bp2)):
        gd = ga.detach().to(dev)
        dg, db = torch.zeros(3, C, device=dev), torch.zeros(3, C, device=dev)
        L.check(lib.ieee_bn2d_bwd(L.ptr(gq), None, L.ptr(y), L.ptr(gq), None, dt, 3, B * P, C, B * P * C, L.ptr(gd), C,
                                  L.ptr(st), L.ptr(dg), L.ptr(db), C, L.ptr(bp), L.ptr(coef), 0, 0, B, L.stream()))
        res["dy" + name], res["dg" + name], res["db" + name] = gq, dg, db
    back = lambda t: t.float().cpu().permute(0, 1, 4, 2, 3)
    # bf16: g is rounded once when stored (2^-9 relative), dy once more, and dy = k1*g + k2*y + k3 cancels partly
    tol_dy = dict(rtol=2e-2, atol=2e-3) if bf else dict(rtol=2e-3, atol=2e-5)
    tol_p = dict(rtol=1e-2, atol=2e-2) if bf else dict(rtol=1e-3, atol=1e-4)      # sums over 384 rounded terms
    torch.testing.assert_close(back(res["dy1"]), w["y1"].grad, **tol_dy)
    torch.testing.assert_close(back(res["dy2"]), w["y2"].grad, **tol_dy)
    torch.testing.assert_close(res["dg1"].cpu(), w["g1"].grad, **tol_p)
    torch.testing.assert_close(res["db2"].cpu(), w["b2"].grad, **tol_p)
    if mode == 0:       # the attention branch only touches fp32 tensors
        torch.testing.assert_close(dw1.cpu().view_as(w["w1"]), w["w1"].grad, rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(dw2.cpu().view_as(w["w2"]), w["w2"].grad, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gpool_sum_others_and_combine(dtype):
    """global average pool + sum of the two other modalities (forward) and the three-way gradient combine (backward);
    bf16: bf16 maps in / out, fp32 pooled vectors, checked against fp32 torch on the same rounded inputs (each stored
    element rounded once)"""
    L, lib = _lib()
    g = torch.Generator().manual_seed(3)
    B, H, W, C = 2, 16, 8, 128
    bf = dtype == torch.bfloat16
    dt = L.IEEE_BF16 if bf else L.IEEE_F32
    Fm = torch.randn(3, B, H * W, C, generator=g).cuda().to(dtype)
    S = torch.empty_like(Fm)
    Gp = torch.empty(3, B, C, device="cuda")
    L.check(lib.ieee_gpool_sum_others(L.ptr(Fm), L.ptr(S), L.ptr(Gp), dt, B, H, W, C, L.stream()))
    torch.testing.assert_close(Gp, Fm.float().mean(2), rtol=1e-5, atol=1e-6)
    for m in range(3):
        a, b = [k for k in range(3) if k != m]
        assert torch.equal(S[m], (Fm[a].float() + Fm[b].float()).to(dtype))           # one exact add, one rounding
    D1, DS = torch.randn_like(Fm), torch.randn_like(Fm)
    dG = torch.randn(3, B, C, device="cuda")
    dF = torch.empty_like(Fm)
    L.check(lib.ieee_cim_bwd_combine(L.ptr(D1), L.ptr(DS), L.ptr(dG), L.ptr(dF), dt, B, H, W, C, 0, L.stream()))
    for m in range(3):
        a, b = [k for k in range(3) if k != m]
        ref = D1[m].float() + DS[a].float() + DS[b].float() + dG[m][:, None, :] / (H * W)
        torch.testing.assert_close(dF[m].float(), ref, rtol=2 ** -8 if bf else 1e-5, atol=1e-6)


def test_rowbn_rem_l2norm_fwd_bwd():
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    R, C = 48, 200
    x = torch.randn(R, C, generator=g, requires_grad=True)
    ga = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    be = (torch.randn(C, generator=g) * 0.3).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    pre = F.batch_norm(x, rm, rv, ga, be, True, 0.1, 1e-5)
    out_ref = F.relu(pre)
    dout = torch.randn(R, C, generator=g) * (pre.detach().abs() > 0.02)
    out_ref.backward(dout)

    def tab(*ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else 0 for t in ts])
    xd, gad, bed = x.detach().cuda(), ga.detach().cuda(), be.detach().cuda()
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    out, sv = torch.empty(R, C, device="cuda"), torch.empty(2, C, device="cuda")
    L.check(lib.ieee_rowbn_fwd(1, tab(xd), tab(out), tab(gad), tab(bed), tab(rmd), tab(rvd), tab(sv), R, C, C, C, 0.1, 1e-5,
                               1, 1, L.stream()))
    torch.testing.assert_close(out.cpu(), out_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rmd.cpu(), rm, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rvd.cpu(), rv, rtol=1e-4, atol=1e-6)
    dx, dg, db = torch.empty_like(xd), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    omask = out_ref.detach().cuda()
    L.check(lib.ieee_rowbn_bwd(1, tab(dout.cuda()), tab(omask), tab(xd), tab(gad), tab(sv), tab(dx), tab(dg), tab(db), R, C,
                               C, C, C, C, 1, 0, L.stream()))
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(dg.cpu(), ga.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(db.cpu(), be.grad, rtol=1e-3, atol=1e-5)
    # REM closed form + backward
    B, parts, D = 5, 6, 96
    part = torch.randn(3, B, parts, D, generator=g, requires_grad=True)
    r = torch.randn(3, B, D, generator=g, requires_grad=True)
    param = torch.tensor([[0.1], [-0.3], [0.7]], requires_grad=True)
    out_ref = part + 2 * param.view(3, 1, 1, 1) * r.unsqueeze(2)
    do = torch.randn(3, B, parts, D, generator=g)
    out_ref.backward(do)
    pd, rd, pad = part.detach().cuda(), r.detach().cuda(), param.detach().cuda().contiguous()
    o = torch.empty_like(pd)
    L.check(lib.ieee_rem_fwd(L.ptr(pd), L.ptr(rd), L.ptr(pad), 1, L.ptr(o), B, parts, D, L.stream()))
    torch.testing.assert_close(o.cpu(), out_ref.detach(), rtol=1e-6, atol=1e-6)
    dr, dparam, work = torch.empty_like(rd), torch.zeros(3, device="cuda"), torch.empty(3 * B + 64, device="cuda")
    L.check(lib.ieee_rem_bwd(L.ptr(do.cuda()), L.ptr(rd), L.ptr(pad), 1, L.ptr(dr), L.ptr(dparam), 1, L.ptr(work), B, parts, D,
                             0, L.stream()))
    torch.testing.assert_close(dr.cpu(), r.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dparam.cpu(), param.grad.flatten(), rtol=1e-4, atol=1e-4)
    # l2norm
    x = torch.randn(10, 768, generator=g, requires_grad=True)
    y_ref = F.normalize(x, p=2, dim=1)
    dy = torch.randn(10, 768, generator=g)
    y_ref.backward(dy)
    xd = x.detach().cuda()
    y, nrm, dxx = torch.empty_like(xd), torch.empty(10, device="cuda"), torch.empty_like(xd)
    L.check(lib.ieee_l2norm_fwd(L.ptr(xd), L.ptr(y), L.ptr(nrm), 10, 768, L.stream()))
    L.check(lib.ieee_l2norm_bwd(L.ptr(dy.cuda()), L.ptr(y), L.ptr(nrm), L.ptr(dxx), 10, 768, 0, L.stream()))
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dxx.cpu(), x.grad, rtol=1e-4, atol=1e-6)


def test_losses_match_oracle_and_autograd():
    from ieee_amd.losses import CrossEntropyLoss, multiModalMarginLossNew
    from oracle import model as om
    g = torch.Generator().manual_seed(9)
    B, C = 16, 171
    logits = torch.randn(B, C, generator=g, requires_grad=True)
    pids = torch.arange(B) // 4
    ref = om.cross_entropy_ls(logits, pids, C)
    ref.backward()
    lg = logits.detach().cuda().requires_grad_(True)
    loss = CrossEntropyLoss(C, use_gpu=True)(lg, pids.cuda())
    loss.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    torch.testing.assert_close(lg.grad.cpu(), logits.grad, rtol=1e-4, atol=1e-7)
    for B2, ids in ((16, torch.arange(16) // 4), (8, torch.arange(8) // 4), (4, torch.zeros(4, dtype=torch.long)),
                    (10, torch.tensor([0, 0, 0, 1, 1, 1, 2, 2, 2, 3]))):
        f = [F.normalize(torch.randn(B2, 768, generator=g), dim=1).requires_grad_(True) for _ in range(3)]
        ref = om.margin3m(f[0], f[1], f[2], ids, 1.0)
        ref.backward()
        fd = [t.detach().cuda().requires_grad_(True) for t in f]
        loss = multiModalMarginLossNew(margin=1)(fd[0], fd[1], fd[2], ids.cuda())
        loss.backward()
        assert abs(float(loss) - float(ref)) < 1e-5
        for a, b in zip(fd, f):
            if b.grad is None:      # the modality outside the selected pair is not in autograd's graph at all
                assert float(a.grad.abs().max()) == 0.0
            else:
                torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-4, atol=1e-7)


def test_fused_sgd_matches_torch_sgd():
    L, lib = _lib()
    g = torch.Generator().manual_seed(11)
    n = 100003
    p0, g1, g2 = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([p], lr=1e-2, momentum=0.9, weight_decay=5e-4, dampening=0, nesterov=True)
    pd, buf = p0.clone().cuda(), torch.zeros(n, device="cuda")
    for gr in (g1, g2):
        p.grad = gr.clone()
        opt.step()
        L.check(lib.ieee_sgd_nesterov_step(L.ptr(pd), L.ptr(gr.cuda()), L.ptr(buf), n, 1e-2, 0.9, 5e-4, 1, L.stream()))
    torch.testing.assert_close(pd.cpu(), p.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_sgemm_pair_launch_has_the_bits_of_the_two_separate_launches():
    """ieee_sgemm_grouped_pair_ws: dW and dX of a Linear (different shapes, strides, group counts, one with split-K, one
    accumulating) as ONE launch against the two ieee_sgemm_grouped_ws calls planned against half of the workspace"""
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(8)

    def tab(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    cases = []
    # (groups, M, N, K): a long-K few-tile set (split-K) paired with a short-K many-tile one; 18-group sets; a ragged one
    for (G0, M0, N0, K0), (G1, M1, N1, K1) in [((3, 512, 2048, 384), (3, 384, 2048, 512)), ((18, 171, 128, 64), (18, 64, 128, 171)),
                                               ((3, 70, 130, 2084), (5, 33, 65, 31))]:
        A0, B0 = torch.randn(G0, M0, K0, generator=g).cuda(), torch.randn(G0, N0, K0, generator=g).cuda()
        A1, B1 = torch.randn(G1, K1, M1, generator=g).cuda(), torch.randn(G1, N1, K1, generator=g).cuda()     # A1 stored [K][M]
        C0i, C1i = torch.randn(G0, M0, N0, generator=g).cuda(), torch.randn(G1, M1, N1, generator=g).cuda()
        bias1 = torch.randn(G1, N1, generator=g).cuda()
        work = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        half = (work.numel() // 2) & ~255
        sep0, sep1 = C0i.clone(), C1i.clone()
        L.check(lib.ieee_sgemm_grouped_ws(G0, tab(list(A0)), tab(list(B0)), tab(list(sep0)), None, M0, N0, K0, K0, 1, K0, 1, N0, 1.0, 0, 1,
                                          L.ptr(work), half, L.stream()))
        L.check(lib.ieee_sgemm_grouped_ws(G1, tab(list(A1)), tab(list(B1)), tab(list(sep1)), tab(list(bias1)), M1, N1, K1, 1, M1, K1, 1, N1,
                                          0.5, 1, 0, L.ptr(work), half, L.stream()))
        torch.cuda.synchronize()
        p0, p1 = C0i.clone(), C1i.clone()
        keep = [tab(list(A0)), tab(list(B0)), tab(list(p0)), tab(list(A1)), tab(list(B1)), tab(list(p1)), tab(list(bias1))]
        s0 = L.SgemmSet(G0, ctypes.addressof(keep[0]), ctypes.addressof(keep[1]), ctypes.addressof(keep[2]), None, M0, N0, K0, K0, 1, K0, 1,
                        N0, 1.0, 0, 1)
        s1 = L.SgemmSet(G1, ctypes.addressof(keep[3]), ctypes.addressof(keep[4]), ctypes.addressof(keep[5]), ctypes.addressof(keep[6]),
                        M1, N1, K1, 1, M1, K1, 1, N1, 0.5, 1, 0)
        L.check(lib.ieee_sgemm_grouped_pair_ws(ctypes.addressof(s0), ctypes.addressof(s1), L.ptr(work), work.numel(), L.stream()))
        torch.cuda.synchronize()
        assert torch.equal(p0, sep0) and torch.equal(p1, sep1)
        ref0 = (A0.double() @ B0.double().transpose(1, 2)) + C0i.double()
        ref1 = torch.relu(0.5 * (A1.double().transpose(1, 2) @ B1.double().transpose(1, 2)) + bias1.double()[:, None, :])
        torch.testing.assert_close(p0.double(), ref0, rtol=1e-4, atol=5e-3)
        torch.testing.assert_close(p1.double(), ref1, rtol=1e-4, atol=5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nt", "tn", "nn_strided"])
def test_sgemm_grouped_splitk_matches_matmul(layout):
    """grouped fp32 GEMM (ieee_sgemm_grouped_ws): 16-byte operand loads along either axis, the scalar fallback,
    and the deterministic split-K path (long K, few tiles) against torch.matmul in fp64"""
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(5)
    G, M, N, K = 3, 70, 130, 2048 + 36      # ragged in every dimension
    A = torch.randn(G, M, K, generator=g).cuda()
    Bm = torch.randn(G, N, K, generator=g).cuda()
    bias = torch.randn(G, N, generator=g).cuda()
    C0 = torch.randn(G, M, N, generator=g).cuda()
    ref = torch.relu((0.5 * (A.double() @ Bm.double().transpose(1, 2)) + bias.double()[:, None, :]) + C0.double())

    def tab(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    if layout == "nt":            # both operands k-contiguous
        a, b, sa, sb = A, Bm, (K, 1), (K, 1)
    elif layout == "tn":          # both operands stored [K][rows] (row-contiguous)
        a, b = A.transpose(1, 2).contiguous(), Bm.transpose(1, 2).contiguous()
        sa, sb = (1, M), (1, N)
        # M = 70, N = 130 are not multiples of 4 -> scalar fallback for the strides; exercise the vector path too
    else:                         # odd pointer offset: unaligned -> scalar path
        pa = torch.zeros(G, M * K + 1, device="cuda"); pa[:, 1:] = A.reshape(G, -1)
        a, b, sa, sb = pa[:, 1:], Bm, (K, 1), (K, 1)
    work = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    for w in (work, None):
        C = C0.clone()
        L.check(lib.ieee_sgemm_grouped_ws(G, tab([a[i] for i in range(G)]), tab([b[i] for i in range(G)]), tab(list(C)),
                                          tab(list(bias)), M, N, K, sa[0], sa[1], sb[0], sb[1], N, 0.5, 1, 1,
                                          L.ptr(w) if w is not None else None, work.numel() if w is not None else 0,
                                          L.stream()))
        torch.testing.assert_close(C.double(), ref, rtol=1e-4, atol=2e-3)
        outs.append(C)
    # split-K is deterministic: same bits on a second run
    C = C0.clone()
    L.check(lib.ieee_sgemm_grouped_ws(G, tab([a[i] for i in range(G)]), tab([b[i] for i in range(G)]), tab(list(C)),
                                      tab(list(bias)), M, N, K, sa[0], sa[1], sb[0], sb[1], N, 0.5, 1, 1, L.ptr(work),
                                      work.numel(), L.stream()))
    assert torch.equal(C, outs[0])
    if layout == "tn":            # aligned row-contiguous operands (vector path along rows)
        M2, N2 = 64, 128
        a2, b2 = a[:, :, :M2].contiguous(), b[:, :, :N2].contiguous()
        C2 = torch.zeros(G, M2, N2, device="cuda")
        L.check(lib.ieee_sgemm_grouped_ws(G, tab(list(a2)), tab(list(b2)), tab(list(C2)), None, M2, N2, K, 1, M2, 1, N2, N2,
                                          1.0, 0, 0, L.ptr(work), work.numel(), L.stream()))
        torch.testing.assert_close(C2.double(), (A.double() @ Bm.double().transpose(1, 2))[:, :M2, :N2], rtol=1e-4, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("amsgrad", [False, True])
def test_adam_step_matches_torch_adam(amsgrad):
    """ieee_adam_step against torch.optim.Adam (what the reference builds for optim='adam'/'amsgrad',
    optim/optimizer.py:113-128) over three steps with changing gradients"""
    from ieee_amd import _lib as L
    lib = L.require_gpu()
    g = torch.Generator().manual_seed(9)
    n = 100003
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref_p], lr=3e-4, betas=(0.9, 0.99), weight_decay=5e-4, amsgrad=amsgrad)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    vmax = torch.zeros(n, device="cuda") if amsgrad else None
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 2))
        ref_p.grad = grad.clone()
        opt.step()
        L.check(lib.ieee_adam_step(L.ptr(p), L.ptr(grad.cuda()), L.ptr(m), L.ptr(v), L.ptr(vmax) if amsgrad else None, n,
                                   3e-4, 0.9, 0.99, 1e-8, 5e-4, step, L.stream()))
        torch.testing.assert_close(p.cpu(), ref_p.detach(), rtol=1e-6, atol=1e-7)
    st = opt.state[ref_p]
    torch.testing.assert_close(m.cpu(), st["exp_avg"], rtol=1e-5, atol=1e-6)        # sums with cancellation near 0
    torch.testing.assert_close(v.cpu(), st["exp_avg_sq"], rtol=1e-5, atol=1e-8)
