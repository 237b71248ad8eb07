"""stride-2 dgrads of the step alone (B=64, 3 modalities), with the parity-class row order on / off:
   IEEE_DGRAD_PERM=0|1 python scripts/dgrad_s2_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from ieee_amd import _ops  # noqa: E402

G, B = 3, 64
for Ci, Co, R, H, W in ((128, 128, 3, 64, 32), (256, 256, 3, 32, 16), (256, 512, 1, 64, 32), (512, 1024, 1, 32, 16)):
    pad = R // 2
    Ho, Wo = (H + 2 * pad - R) // 2 + 1, (W + 2 * pad - R) // 2 + 1
    g = torch.Generator().manual_seed(0)
    dy = torch.randn(G, B, Ho, Wo, Co, generator=g).cuda().bfloat16()
    w = (torch.randn(G, Co, Ci, R, R, generator=g) * 0.05).cuda()
    wpd = _ops.pack_conv_weight(w, torch.bfloat16, 1)
    for _ in range(3):
        dx = _ops.conv2d_dgrad(dy, wpd, (H, W), Ci, R, R, 2, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dx = _ops.conv2d_dgrad(dy, wpd, (H, W), Ci, R, R, 2, pad)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    useful = 2.0 * G * B * Ho * Wo * Co * Ci * R * R
    print("%4d->%4d k%d s2 %dx%d  %7.1f us  %6.1f TFLOP/s (useful)  out %.0f MB" % (Ci, Co, R, H, W, us, useful / us / 1e6, dx.numel() * 2 / 1e6))
