"""time the distmat variants on the 10k x 100k x 768 problem (for rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ieee_amd.metrics import compute_distance_matrix
g = torch.Generator(device="cpu").manual_seed(1)
qf = torch.randn(10000, 768, generator=g).abs().cuda()
gf = torch.randn(100000, 768, generator=g).abs().cuda()
for prec in sys.argv[1:] or ["fp32", "bf16x3", "f16x2"]:
    compute_distance_matrix(qf, gf, precision=prec); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        compute_distance_matrix(qf, gf, precision=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-7s %.3f ms  %.1f TFLOP/s of the fp32 problem" % (prec, ms, 2 * 10000 * 100000 * 768 / ms / 1e9))
if not sys.argv[1:]:
    for dd in (64, 768, 1536, 2304, 4608):
        a = torch.randn(10000, dd, generator=g).abs().cuda().bfloat16()
        b = torch.randn(100000, dd, generator=g).abs().cuda().bfloat16()
        compute_distance_matrix(a, b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            compute_distance_matrix(a, b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("bf16 d=%4d  %.3f ms  %.0f TFLOP/s" % (dd, ms, 2 * 10000 * 100000 * dd / ms / 1e9))
        del a, b
