"""diagnostic: cosine between fp32 and bf16-autocast gradients of the STOCK torch oracle on the GPU, next to the
native fp32-vs-bf16 cosine, on the same synthetic net (is the bf16 gradient decorrelation inherent?)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import model as om
from tests.util_model import C, generated_state, images
from ieee_amd._spec import state_spec
B, seed = 16, 5
shapes = {k: s for k, s, _ in state_spec(C)}
pids = (torch.arange(B) // 4).cuda()
xs = [x.cuda() for x in images(B, seed)]
def torch_grads(autocast):
    sd = {k: v.cuda() for k, v in generated_state(shapes, 11).items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if k.rsplit('.', 1)[-1] in om.PARAM_LEAVES}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = om.forward(sd, xs, True)
    out = tuple([o.float() for o in oo] if isinstance(oo, list) else oo.float() for oo in out)
    loss, _ = om.losses(out, pids, C)
    g = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    return dict(zip(params.keys(), g)), float(loss)
g32, l32 = torch_grads(False)
g16, l16 = torch_grads(True)
cos = []
for k in g32:
    if g32[k] is None or g16[k] is None: continue
    a, b = g32[k].flatten().double(), g16[k].flatten().double()
    if a.norm() < 1e-6: continue
    cos.append(float((a * b).sum() / (a.norm() * b.norm())))
print("stock torch: loss fp32 %.4f bf16 %.4f; per-tensor cosine fp32 vs bf16-autocast: median %.3f, 10%%-quantile %.3f" % (l32, l16, np.median(cos), np.quantile(cos, 0.1)))
