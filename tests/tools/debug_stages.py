"""diagnostic (not a test): per-stage max|diff| between the native executor's intermediates and the
oracle on the CPU, to localise a parity failure.  python tests/tools/debug_stages.py [B] [train|eval]"""
import sys

import torch

sys.path.insert(0, ".")
from oracle import model as om  # noqa: E402
from tests.util_model import C, generated_state, images  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    training = (sys.argv[2] if len(sys.argv) > 2 else "eval") == "train"
    cdt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float32
    from ieee_amd.models import build_model
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=cdt)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = generated_state(shapes, 1)
    m.load_state_dict(sd)
    m.train(training)
    xs = images(B, 1)
    taps = {}
    with torch.no_grad():
        ref = om.forward({k: v.clone() for k, v in sd.items()}, xs, training, taps=taps)
        out = m([x.cuda() for x in xs])
    net = list(m._nets.values())[0]

    def nhwc(name, shape):
        return net.tensor(name).float().view(shape).cpu()

    def show(name, a, b):
        print("%-28s max|diff| %.3e   max|ref| %.3e   rel-L2 %.3e" % (name, (a - b).abs().max().item(), b.abs().max().item(),
                                                                  ((a - b).norm() / b.norm()).item()))
    stem = torch.stack([taps["backbone.%d.stem" % i] for i in range(3)]).permute(0, 1, 3, 4, 2)
    show("stem", nhwc("backbone.{m}.conv1.a", stem.shape), stem)
    for li, last in ((1, 2), (2, 3), (3, 5), (4, 2)):
        t = torch.stack([taps["backbone.%d.layer%d" % (i, li)] for i in range(3)]).permute(0, 1, 3, 4, 2)
        show("layer%d" % li, nhwc("backbone.{m}.layer%d.%d.conv3.a" % (li, last), t.shape), t)
    show("glob", nhwc("glob", taps["glob"].shape), taps["glob"])
    show("parts(pre REM)", nhwc("part", taps["parts_pre_rem"].shape), taps["parts_pre_rem"])
    if training:
        lg = torch.stack([torch.stack(list(o)) for o in ref[:3]]).reshape(18, B, C)
        my = torch.stack([torch.stack(list(o)) for o in out[:3]]).reshape(18, B, C).cpu()
        show("logits", my, lg)
        show("feats", torch.stack(list(out[3:])).cpu(), torch.stack(list(ref[3:])))
    else:
        show("fc_all", out.cpu(), ref)


if __name__ == "__main__":
    main()
