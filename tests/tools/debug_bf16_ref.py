"""diagnostic (not a test): how far does STOCK torch bf16 autocast (MIOpen/rocBLAS on the GPU) drift
from fp32 on the same synthetic net?  Puts the native bf16 mode's drift in context."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import model as om  # noqa: E402
from tests.util_model import C, generated_state, images  # noqa: E402

B = 16
shapes_model = None
from ieee_amd._spec import state_spec  # noqa: E402
shapes = {k: s for k, s, _ in state_spec(C)}
sd = {k: v.cuda() for k, v in generated_state(shapes, 2).items()}
xs = [x.cuda() for x in images(B, 2)]
t32, t16 = {}, {}
with torch.no_grad():
    o32 = om.forward({k: v.clone() for k, v in sd.items()}, xs, True, taps=t32)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o16 = om.forward({k: v.clone() for k, v in sd.items()}, xs, True, taps=t16)
for k in t32:
    if k.startswith("backbone.0") or k in ("glob", "parts_pre_rem"):
        a, b = t16[k].float(), t32[k].float()
        print("%-24s torch-bf16 vs torch-fp32 rel-L2 %.3e" % (k, ((a - b).norm() / b.norm()).item()))
f16, f32 = torch.stack(list(o16[3:])).float(), torch.stack(list(o32[3:]))
print("feats rel-L2 %.3e" % ((f16 - f32).norm() / f32.norm()).item())
