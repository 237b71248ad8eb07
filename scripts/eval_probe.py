"""inference (feature-extraction) throughput of the model: 3-modal images/s of model.eval() forward, B = 64"""
import sys
import time

import torch

sys.path.insert(0, ".")
from bench import make_batch  # noqa: E402
from ieee_amd.models import build_model  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = build_model("ieee3modalPart", num_classes=171, loss="margin", pretrained=False, use_gpu=True,
                compute_dtype=torch.bfloat16, device=dev)
m.eval()
batch = make_batch(64, seed=0, device=dev)
with torch.no_grad():
    for _ in range(5):
        f = m(batch["img"])
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(30):
        f = m(batch["img"])
    torch.cuda.synchronize()
dt = (time.time() - t0) / 30
print("eval forward: %.3f ms per 64 triples, %.0f 3-modal images/s, features %s" % (dt * 1e3, 64 / dt, tuple(f.shape)))
