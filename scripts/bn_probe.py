"""isolated bandwidth of the BN apply / backward-apply kernels at the ResNet-50 stage shapes of the B=64 step
(3 modalities as groups); partial sums are pretended to be emitted by the conv so that only finalize + apply run"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ieee_amd import _lib as L

lib = L.load()
dev = "cuda"
G, B = 3, 64
shapes = [(B * 128 * 64, 64), (B * 64 * 32, 64), (B * 64 * 32, 256), (B * 32 * 16, 128), (B * 32 * 16, 512),
          (B * 16 * 8, 256), (B * 16 * 8, 1024), (B * 16 * 8, 512), (B * 16 * 8, 2048)]
RB = 8


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, C in shapes:
    y = torch.randn(G, M, C, device=dev).bfloat16()
    res = torch.randn(G, M, C, device=dev).bfloat16()
    out = torch.empty_like(y)
    dd = torch.randn(G, M, C, device=dev).bfloat16()
    dy = torch.empty_like(y)
    gout = torch.empty_like(y)
    gam, bet = torch.ones(G, C, device=dev), torch.zeros(G, C, device=dev)
    rm, rv = torch.zeros(G, C, device=dev), torch.ones(G, C, device=dev)
    stats = torch.empty(G, 4, C, device=dev)
    part = torch.rand(G * max(lib.ieee_bn_partial_floats(1, M, C), 2 * C * RB) + 64, device=dev)
    coef = torch.empty(G, 3, C, device=dev)
    dg, db = torch.zeros(G, C, device=dev), torch.zeros(G, C, device=dev)
    st = L.stream()
    fwd = lambda r: L.check(lib.ieee_bn2d_fwd(L.ptr(y), L.ptr(res) if r else None, L.ptr(out), 1, G, M, C, M * C, L.ptr(gam),
                                              L.ptr(bet), C, L.ptr(rm), L.ptr(rv), C, L.ptr(stats), L.ptr(part), 0.1, 1e-5, 1, 1,
                                              RB, None, st))
    bwd = lambda kind, go: L.check(lib.ieee_bn2d_bwd(L.ptr(dd), L.ptr(out) if kind == 1 else None, L.ptr(y), L.ptr(dy),
                                                     L.ptr(gout) if go else None, 1, G, M, C, M * C, L.ptr(gam), C, L.ptr(stats),
                                                     L.ptr(dg), L.ptr(db), C, L.ptr(part), L.ptr(coef), 0, 1 if kind == 2 else 0, RB, st))
    n = G * M * C * 2
    t1 = timeit(lambda: fwd(False)); t2 = timeit(lambda: fwd(True))
    t0 = timeit(lambda: bwd(0, False)); t3 = timeit(lambda: bwd(2, False)); t4 = timeit(lambda: bwd(1, True))
    print("M=%7d C=%4d  %5.1f MB | fwd %6.1f us %4.2f TB/s | fwd+res %6.1f us %4.2f | bwd(g masked) %6.1f us %4.2f | "
          "bwd(mask from y) %6.1f us %4.2f | bwd(mask,gout) %6.1f us %4.2f" % (
              M, C, n / 1e6, t1, 2 * n / t1 / 1e6, t2, 3 * n / t2 / 1e6, t0, 3 * n / t0 / 1e6, t3, 3 * n / t3 / 1e6,
              t4, 5 * n / t4 / 1e6))
