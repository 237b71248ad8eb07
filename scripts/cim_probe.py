"""time the CIM-tail kernels alone at the step's shape (B=64, 16x8 positions, 2048 channels, bf16) against their bytes:
   python scripts/cim_probe.py            (GPU box)"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from ieee_amd import _lib as L  # noqa: E402

lib = L.require_gpu()
dev = "cuda"
B, H, W, C, PARTS = 64, 16, 8, 2048, 6
dt = L.IEEE_BF16
g = torch.Generator().manual_seed(0)
y1 = torch.randn(3, B, H * W, C, generator=g).to(dev, torch.bfloat16)
y2 = torch.randn(3, B, H * W, C, generator=g).to(dev, torch.bfloat16)
st1, st2 = torch.randn(3, 4, C, generator=g).to(dev), torch.randn(3, 4, C, generator=g).to(dev)
att = torch.rand(3, B, C, generator=g).to(dev)
Pp = torch.empty(3, B, PARTS, C, device=dev)
dP = torch.randn(3, B, PARTS, C, generator=g).to(dev)
datt = torch.zeros(3, B, C, device=dev)
avgmax = torch.empty(3, 2 * B, C, device=dev)
davgmax = torch.randn(3, 2 * B, C, generator=g).to(dev)
amax = torch.randint(0, H * W, (3, B, C), generator=g).to(dev, torch.int32)
g1, g2 = torch.empty_like(y1), torch.empty_like(y2)
bp1, bp2 = torch.zeros(3, 2, C, B, device=dev), torch.zeros(3, 2, C, B, device=dev)
S = torch.empty_like(y1)
Gp = torch.empty(3, B, C, device=dev)
dF = torch.empty_like(y1)
st = L.stream()
MB = y1.numel() * 2 / 1e6


def timed(name, fn, mbytes, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-16s %7.1f us   %6.0f MB  -> %5.2f TB/s" % (name, us, mbytes, mbytes / us))


timed("gpool_sum_others", lambda: L.check(lib.ieee_gpool_sum_others(L.ptr(y1), L.ptr(S), L.ptr(Gp), dt, B, H, W, C, st)), 2 * MB)
timed("ca_pool", lambda: L.check(lib.ieee_ca_pool(L.ptr(y2), L.ptr(st2), L.ptr(avgmax), ctypes.c_void_p(avgmax.data_ptr() + B * C * 4),
                                                  2 * B * C, L.ptr(amax), dt, B, H, W, C, st)), MB)
timed("cim_tail_fwd", lambda: L.check(lib.ieee_cim_tail_fwd(L.ptr(y1), L.ptr(y2), L.ptr(st1), L.ptr(st2), L.ptr(att), L.ptr(Pp), dt,
                                                            B, H, W, C, PARTS, 0, st)), 2 * MB)
timed("cim_bwd_datt", lambda: L.check(lib.ieee_cim_tail_bwd_datt(L.ptr(dP), L.ptr(y2), L.ptr(st2), L.ptr(datt), dt, B, H, W, C, PARTS,
                                                                 st)), MB)
timed("cim_bwd_g", lambda: L.check(lib.ieee_cim_tail_bwd_g(
    L.ptr(dP), L.ptr(y1), L.ptr(y2), L.ptr(st1), L.ptr(st2), L.ptr(att), L.ptr(davgmax),
    ctypes.c_void_p(davgmax.data_ptr() + B * C * 4), 2 * B * C, L.ptr(amax), L.ptr(g1), L.ptr(g2), dt, B, H, W, C, PARTS, 0,
    L.ptr(bp1), L.ptr(bp2), st)), 4 * MB)
timed("cim_bwd_combine", lambda: L.check(lib.ieee_cim_bwd_combine(L.ptr(g1), L.ptr(g2), L.ptr(Gp), L.ptr(dF), dt, B, H, W, C, 0, st)),
      3 * MB)

# ---- the stem's max-pool at its shape (B=64, 128x64 -> 64x32, 64 channels)
Hs, Ws, Cs = 128, 64, 64
xs = torch.randn(3, B, Hs, Ws, Cs, generator=g).to(dev, torch.bfloat16)
Ho, Wo = (Hs + 2 - 3) // 2 + 1, (Ws + 2 - 3) // 2 + 1
po = torch.empty(3, B, Ho, Wo, Cs, device=dev, dtype=torch.bfloat16)
pa = torch.empty(3, B, Ho, Wo, Cs, device=dev, dtype=torch.uint8)
dpo = torch.randn(3, B, Ho, Wo, Cs, generator=g).to(dev, torch.bfloat16)
dxs = torch.empty_like(xs)
MBs = xs.numel() * 2 / 1e6
timed("maxpool_fwd", lambda: L.check(lib.ieee_maxpool3x3s2_fwd(L.ptr(xs), L.ptr(po), L.ptr(pa), dt, 3, B, Hs, Ws, Cs, st)), MBs * 1.375)
timed("maxpool_bwd", lambda: L.check(lib.ieee_maxpool3x3s2_bwd(L.ptr(dpo), L.ptr(pa), L.ptr(dxs), dt, 3, B, Hs, Ws, Cs, st)), MBs * 1.375)
