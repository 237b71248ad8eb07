import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests.test_shapes_gpu import _model, _imgs, C
from ieee_amd.engine import Image3MEngine
from ieee_amd.optim import build_optimizer
class DM(object):
    num_train_pids = C; train_loader = []; test_loader = {}; sources = ["s"]
for dt in (torch.float32, torch.bfloat16):
    for lr in (1e-5, 1e-4):
        m, _ = _model(dt)
        eng = Image3MEngine(DM(), m, build_optimizer(m, optim="sgd", lr=lr), margin=1, use_gpu=True)
        m.train()
        B = 32
        data = {"img": _imgs(B, 256, 128, 9), "pid": torch.arange(B) // 4, "camid": torch.zeros(B), "impath": "", "timeid": torch.zeros(B)}
        losses = [eng.forward_backward(data)["loss"] for _ in range(8)]
        print(dt, lr, ['%.3f' % l for l in losses])
# gradient agreement bf16 vs fp32 on the same weights (cosine per tensor)
ms = {}
for dt in (torch.float32, torch.bfloat16):
    m, _ = _model(dt); m.train()
    from oracle import model as om
    xs = [x.cuda() for x in _imgs(16, 256, 128, 5)]
    out = m(xs); loss, _ = om.losses(out, (torch.arange(16)//4).cuda(), C); loss.backward()
    ms[dt] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
cos = []
for k in ms[torch.float32]:
    a, b = ms[torch.float32][k].flatten().double(), ms[torch.bfloat16][k].flatten().double()
    if a.norm() < 1e-6: continue
    cos.append((float((a*b).sum()/(a.norm()*b.norm())), k))
cos.sort()
print('worst cos', cos[:6]); print('median cos', np.median([c[0] for c in cos]))
