"""eps scan behind tests/test_backward_units_gpu.py::test_b64_bf16_descent_along_the_native_gradient: loss decrease along -g
and along a random direction of the same length, bf16 and fp32 at B = 64, on the generated (chaotic) state and on the tamed
one (tests/util_model.py: tame_), plus the cosine between the native bf16 and fp32 gradients"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests.util_model import C, generated_state, tame_
from ieee_amd.engine import Image3MEngine
from ieee_amd.models import build_model
from ieee_amd.optim import build_optimizer


class DM(object):
    num_train_pids = C; train_loader = []; test_loader = {}; sources = ["s"]


B = 64
g = torch.Generator().manual_seed(9)
data = {"img": [torch.randn(B, 3, 256, 128, generator=g) for _ in range(3)], "pid": torch.arange(B) // 4,
        "camid": torch.zeros(B), "impath": "", "timeid": torch.zeros(B)}
for tame in (True, False):
    grads = {}
    for dt in (torch.bfloat16, torch.float32):
        m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=dt)
        st = generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 11)
        if tame:
            tame_(st)
        m.load_state_dict(st)
        m.train()
        with torch.no_grad():      # a generic (tie-free) starting point
            u = torch.rand(m._flat_params.shape, generator=torch.Generator(device="cuda").manual_seed(7), device="cuda") * 2 - 1
            m._flat_params.mul_(1 + u * 2.0 ** -10)
        eng = Image3MEngine(DM(), m, build_optimizer(m, optim="sgd", lr=0.0, weight_decay=0.0, momentum=0.0), margin=1, use_gpu=True)
        l0 = float(eng.forward_backward(data)["loss"])
        grad = m._flat_grads.clone()
        grads[dt] = (grad, {n: (m._offsets[n], p.numel()) for n, p in m.named_parameters()})
        runs = m.trainable_runs()
        g2 = sum(float((grad[a:b].double() ** 2).sum()) for a, b in runs)
        w0 = m._flat_params.clone()
        rnd = torch.randn(w0.shape, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda")
        rn2 = sum(float((rnd[a:b].double() ** 2).sum()) for a, b in runs)
        print("tame" if tame else "chaotic", dt, "L0 %.6f |g|^2 %.4e" % (l0, g2), flush=True)

        def loss_at(direction, step):
            with torch.no_grad():
                m._flat_params.copy_(w0)
                for a, b in runs:
                    m._flat_params[a:b] = w0[a:b] + step * direction[a:b]
            return float(eng.forward_backward(data)["loss"])
        for eps in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
            pred = eps * g2
            lm, lp = loss_at(grad, -eps), loss_at(grad, eps)
            s = eps * (g2 / rn2) ** 0.5
            rm, rp = loss_at(rnd, -s), loss_at(rnd, s)
            print("  eps %.0e predicted %.6f  along -g %.6f (ratio %.3f) | symmetric ratio %.3f | random: %+.6f / %+.6f"
                  % (eps, pred, l0 - lm, (l0 - lm) / pred, (lp - lm) / (2 * pred), rm - l0, rp - l0), flush=True)
        del eng, m
        torch.cuda.empty_cache()
    (g16, names), (g32, _) = grads[torch.bfloat16], grads[torch.float32]
    cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-300))
    per = [cos(g16[o:o + n], g32[o:o + n]) for o, n in names.values() if float(g32[o:o + n].abs().max()) > 0]
    print("  cosine(native bf16 gradient, native fp32 gradient): global %.4f, per-tensor median %.4f, min %.4f, 10th pct %.4f"
          % (cos(g16, g32), float(np.median(per)), min(per), float(np.percentile(per, 10))), flush=True)
