"""inference (feature-extraction) throughput of the model: 3-modal images/s of model.eval() forward; batch sizes as arguments (default 64)"""
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
for B in [int(a) for a in sys.argv[1:]] or [64]:
    batch = make_batch(B, seed=0, device=dev)
    with torch.no_grad():
        for _ in range(5):
            f = m(batch["img"])
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(30):
            f = m(batch["img"])
        torch.cuda.synchronize()
    dt = (time.time() - t0) / 30
    print("eval forward: %.3f ms per %d triples, %.0f 3-modal images/s (%.0f TFLOP/s), features %s"
          % (dt * 1e3, B, B / dt, 30.902e9 * B / dt / 1e12, tuple(f.shape)))
