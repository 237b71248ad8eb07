import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_shapes_gpu import _model, _imgs, C
from ieee_amd.engine import Image3MEngine
from ieee_amd.optim import build_optimizer
class DM(object):
    num_train_pids = C; train_loader = []; test_loader = {}; sources = ["s"]
B = 32
data = {"img": _imgs(B, 256, 128, 9), "pid": torch.arange(B) // 4, "camid": torch.zeros(B), "impath": "", "timeid": torch.zeros(B)}
for dt in (torch.float32, torch.bfloat16):
    for eps in (2e-9, 2e-8, 2e-7):
        m, _ = _model(dt)
        opt = build_optimizer(m, optim="sgd", lr=eps, weight_decay=0.0, momentum=0.0)
        eng = Image3MEngine(DM(), m, opt, margin=1, use_gpu=True); m.train()
        l0 = eng.forward_backward(data)["loss"]
        g2 = sum(float((m._flat_grads[a:b].double() ** 2).sum()) for a, b in m.trainable_runs())
        opt.param_groups[0]["lr"] = 0.0
        l1 = eng.forward_backward(data)["loss"]
        l2 = eng.forward_backward(data)["loss"]
        print(dt, eps, "L0 %.5f L1 %.5f (repeat %.5f) predicted %.5f ratio %.3f" % (l0, l1, l2, eps*g2, (l0-l1)/(eps*g2)))
