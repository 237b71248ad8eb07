"""diagnostic (not a test): per-parameter gradient error of the native backward against the golden
gradient statistics captured from the reference.  python tests/tools/debug_grads.py [train8|train16]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import model as om  # noqa: E402
from tests.util_model import C, generated_state, images, stats  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "train8"
    B, seed = {"train8": (8, 3), "train16": (16, 2)}[tag]
    G = np.load("tests/golden/model_golden.npz")
    from ieee_amd.models import build_model
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.float32)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(generated_state(shapes, seed))
    m.train()
    xs = [x.cuda() for x in images(B, seed)]
    pids = (torch.arange(B) // 4).cuda()
    out = m(xs)
    loss, _ = om.losses(out, pids, C)
    loss.backward()
    names = [str(n) for n in G[tag + "/param_names"]]
    ref = G[tag + "/grad_stats"]
    params = dict(m.named_parameters())
    shown = 0
    for i, n in enumerate(names):
        g = params[n].grad
        if g is None:
            continue
        s = stats(g)
        nref = ref[i][2]
        e_norm = abs(s[2] - nref) / max(abs(nref), 1e-12)
        scale = max(np.abs(ref[i][3:]).max(), 1e-12)
        e_samp = np.abs(s[3:] - ref[i][3:]).max() / scale
        if (e_norm > 1e-3 or e_samp > 3e-3) and abs(nref) > 1e-6:
            print("%-48s |g| %.3e  e_norm %.2e  e_samp %.2e" % (n, nref, e_norm, e_samp))
            shown += 1
    print("tensors over threshold:", shown, "of", len(names))


if __name__ == "__main__":
    main()
