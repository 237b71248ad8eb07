"""eps scan behind tests/test_backward_units_gpu.py::test_b64_bf16_descent_along_the_native_gradient: loss decrease along -g
and along a random direction of the same length, bf16 (and fp32 for comparison) at B = 64"""
import sys
import torch
sys.path.insert(0, '.')
from tests.util_model import C, generated_state
from ieee_amd.engine import Image3MEngine
from ieee_amd.models import build_model
from ieee_amd.optim import build_optimizer


class DM(object):
    num_train_pids = C; train_loader = []; test_loader = {}; sources = ["s"]


B = 64
g = torch.Generator().manual_seed(9)
data = {"img": [torch.randn(B, 3, 256, 128, generator=g) for _ in range(3)], "pid": torch.arange(B) // 4,
        "camid": torch.zeros(B), "impath": "", "timeid": torch.zeros(B)}
for dt in (torch.bfloat16, torch.float32):
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=dt)
    m.load_state_dict(generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 11))
    m.train()
    # the generated weights sit on a coarse binary grid: many are exact ties of the bf16 rounding, and ANY perturbation moves
    # all of those by a whole bf16 ulp at once (a jump of ~0.2 in the loss, found in the first scan) -- start from a generic point
    with torch.no_grad():
        u = torch.rand(m._flat_params.shape, generator=torch.Generator(device="cuda").manual_seed(7), device="cuda") * 2 - 1
        m._flat_params.mul_(1 + u * 2.0 ** -10)
    eng = Image3MEngine(DM(), m, build_optimizer(m, optim="sgd", lr=0.0, weight_decay=0.0, momentum=0.0), margin=1, use_gpu=True)
    l0 = float(eng.forward_backward(data)["loss"])
    grad = m._flat_grads.clone()
    runs = m.trainable_runs()
    g2 = sum(float((grad[a:b].double() ** 2).sum()) for a, b in runs)
    w0 = m._flat_params.clone()
    rnd = torch.randn(w0.shape, generator=torch.Generator(device="cuda").manual_seed(1), device="cuda")
    rn2 = sum(float((rnd[a:b].double() ** 2).sum()) for a, b in runs)
    print(dt, "L0 %.6f |g|^2 %.4e" % (l0, g2))
    for eps in (5e-8, 1e-7, 2e-7, 4e-7, 1e-6):
        with torch.no_grad():
            m._flat_params.copy_(w0)
            for a, b in runs:
                m._flat_params[a:b] = w0[a:b] - eps * grad[a:b]
        lg = float(eng.forward_backward(data)["loss"])
        with torch.no_grad():
            m._flat_params.copy_(w0)
            for a, b in runs:
                m._flat_params[a:b] = w0[a:b] + eps * (g2 / rn2) ** 0.5 * rnd[a:b]
        lr = float(eng.forward_backward(data)["loss"])
        with torch.no_grad():
            m._flat_params.copy_(w0)
            for a, b in runs:
                m._flat_params[a:b] = w0[a:b] + eps * grad[a:b]
        lp = float(eng.forward_backward(data)["loss"])
        with torch.no_grad():
            m._flat_params.copy_(w0)
            for a, b in runs:
                m._flat_params[a:b] = w0[a:b] - eps * (g2 / rn2) ** 0.5 * rnd[a:b]
        lr2 = float(eng.forward_backward(data)["loss"])
        print("  eps %.0e predicted %.6f  along -g %.6f (ratio %.3f)  random %+.6f | symmetric: (L+ - L-)/(2 eps g2) = %.3f, random (L+ - L-) %+.6f"
              % (eps, eps * g2, l0 - lg, (l0 - lg) / (eps * g2), lr - l0, (lp - lg) / (2 * eps * g2), lr - lr2))
    del eng, m
    torch.cuda.empty_cache()
