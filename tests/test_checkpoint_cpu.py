"""CPU: checkpoint / pretrained-weight interop of the flat-buffer model against the behaviour of
the reference's torchreid/utils/torchtools.py and resnet.py:1075-1089 — same file format, same key set, "module."
prefix handling, unmatched layers ignored, three backbones initialised from one ImageNet ResNet-50 dict."""
import os
from collections import OrderedDict

import pytest
import torch

from ieee_amd import checkpoint as ckpt
from ieee_amd.models import build_model

C = 171


def _model(num_classes=C, seed=0):
    torch.manual_seed(seed)
    return build_model("ieee3modalPart", num_classes=num_classes, loss="margin", pretrained=False, device="cpu")


def _fake_resnet50(model):
    """a torchvision-style ResNet-50 state_dict with distinctive values (names = the reference backbone's own names)"""
    sd = OrderedDict()
    for i, (k, v) in enumerate(model.state_dict().items()):
        if k.startswith("backbone.0."):
            sd[k[len("backbone.0."):]] = torch.full_like(v, (i % 97) + 1) if v.dtype.is_floating_point else v.clone() + 7
    sd["fc.weight"] = torch.ones(1000, 2048)
    sd["fc.bias"] = torch.ones(1000)
    return sd


def test_checkpoint_round_trip_and_module_prefix(tmp_path):
    a = _model(seed=1)
    with torch.no_grad():
        a._flat_params.normal_()
        a._flat_buffers.uniform_(0.5, 1.5)
    ckpt.save_checkpoint({"state_dict": a.state_dict(), "epoch": 3, "rank1": 0.5}, str(tmp_path))
    f = str(tmp_path / "model.pth.tar-3")
    assert os.path.exists(f)
    b = _model(seed=2)
    assert ckpt.resume_from_checkpoint(f, b) == 3
    for (k1, v1), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    # parameters are still views of the flat buffer (load_state_dict copied in place)
    assert b.backbone[0].conv1.weight.data_ptr() == b._flat_params.data_ptr()
    # DataParallel-style keys + a classifier of another size: matched layers load, the rest is reported
    pref = OrderedDict(("module." + k, v) for k, v in a.state_dict().items())
    torch.save({"state_dict": pref}, str(tmp_path / "dp.pth"))
    c = _model(num_classes=C + 5, seed=3)
    before = c.classifier_R[0].weight.clone()
    matched, discarded = ckpt.load_pretrained_weights(c, str(tmp_path / "dp.pth"))
    assert len(matched) + len(discarded) == len(pref)
    assert sorted(discarded) == sorted(k for k in a.state_dict() if k.startswith("classifier_"))
    assert torch.equal(c.classifier_R[0].weight, before)
    assert torch.equal(c.backbone[2].layer4[2].conv3.weight, a.backbone[2].layer4[2].conv3.weight)
    with pytest.raises(FileNotFoundError):
        ckpt.load_checkpoint(str(tmp_path / "nope"))
    with pytest.raises(ValueError):
        ckpt.load_checkpoint(None)


def test_imagenet_resnet50_initialises_the_three_backbones(tmp_path, monkeypatch):
    m = _model(seed=4)
    fake = _fake_resnet50(m)
    head_before = m.reduce_layer[0].layers[0].weight.clone()
    n = ckpt.init_pretrained_backbones(m, fake)
    assert n == 3 * (len(fake) - 2)                     # fc.weight / fc.bias have no counterpart
    sd = m.state_dict()
    for k, v in fake.items():
        if k.startswith("fc."):
            continue
        for mod in range(3):
            assert torch.equal(sd["backbone.%d.%s" % (mod, k)], v)
    assert torch.equal(m.reduce_layer[0].layers[0].weight, head_before)
    # pretrained=True: uses the file when it is on disk, explains itself when it is not
    path = str(tmp_path / ckpt.RESNET50_FILE)
    torch.save(fake, path)
    monkeypatch.setenv("IEEE_RESNET50_PTH", path)
    p = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=True, device="cpu")
    assert torch.equal(p.state_dict()["backbone.1.layer3.5.bn2.weight"], fake["layer3.5.bn2.weight"])
    monkeypatch.delenv("IEEE_RESNET50_PTH")
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "empty"))
    with pytest.raises(RuntimeError, match="no network"):
        build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=True, device="cpu")


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchreid"), reason="reference tree not present")
def test_checkpoints_interoperate_with_the_reference_model(tmp_path):
    """a checkpoint written here loads into the reference's IEEE3modalPart with its own loader (strict), and one
    written by the reference loads here: same keys, same shapes, same order"""
    from oracle.ref_import import import_reference
    torchreid = import_reference()
    ours = _model(seed=5)
    with torch.no_grad():
        ours._flat_params.normal_(0, 0.05)
    ckpt.save_checkpoint({"state_dict": ours.state_dict(), "epoch": 1}, str(tmp_path))
    rm = torchreid.models.build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, use_gpu=False)
    state = torch.load(str(tmp_path / "model.pth.tar-1"), weights_only=False)["state_dict"]
    rm.load_state_dict(state)                                           # strict: key sets identical
    assert list(rm.state_dict().keys()) == list(ours.state_dict().keys())
    with torch.no_grad():
        for p in rm.parameters():
            p.mul_(0.5)
    torch.save({"state_dict": rm.state_dict(), "epoch": 9}, str(tmp_path / "ref.pth"))
    back = _model(seed=6)
    assert ckpt.resume_from_checkpoint(str(tmp_path / "ref.pth"), back) == 9
    assert torch.equal(back.backbone[1].layer2[0].conv2.weight, ours.backbone[1].layer2[0].conv2.weight * 0.5)


def test_fused_optimizer_state_interops_with_torch_optim(tmp_path):
    """the fused optimizers keep momentum / moments in flat buffers, but their state_dict() is what torch.optim.SGD / Adam
    (the reference's optimizers, optim/optimizer.py:113-138) write and read: per-parameter momentum_buffer / exp_avg /
    exp_avg_sq / step.  A checkpoint of either implementation resumes in the other without losing the state."""
    from ieee_amd.optim import FusedAdam, FusedSGD
    m = _model(seed=3)
    params = list(m.parameters())
    # fused SGD -> torch SGD (through a file, as resume_from_checkpoint does)
    fs = FusedSGD(m, lr=1e-3, momentum=0.9, weight_decay=5e-4)
    with torch.no_grad():
        fs.momentum_buffer().copy_(torch.arange(m._flat_params.numel(), dtype=torch.float32) * 1e-6)
    torch.save({"optimizer": fs.state_dict()}, str(tmp_path / "o.pt"))
    sd = torch.load(str(tmp_path / "o.pt"), weights_only=False)["optimizer"]
    ts = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True)
    ts.load_state_dict(sd)
    for name, p in m._param_items[:5] + m._param_items[-5:]:
        off = m._offsets[name]
        assert torch.equal(ts.state[p]["momentum_buffer"].flatten(), fs.momentum_buffer()[off:off + p.numel()])
    # torch SGD -> fused SGD
    for p in params:
        ts.state[p]["momentum_buffer"] = torch.full_like(p, float(p.numel() % 13))
    fs2 = FusedSGD(m, lr=1e-3)
    fs2.load_state_dict(ts.state_dict())
    for name, p in m._param_items:
        off = m._offsets[name]
        assert bool((fs2.momentum_buffer()[off:off + p.numel()] == float(p.numel() % 13)).all()), name
    assert fs2.param_groups[0]["momentum"] == 0.9
    # fused Adam <-> torch Adam (amsgrad: all three moment buffers and the step count)
    fa = FusedAdam(m, lr=3e-4, amsgrad=True)
    bufs = fa._buffers()
    with torch.no_grad():
        for k, b in enumerate(bufs):
            b.fill_(0.25 * (k + 1))
    fa._step = 7
    ta = torch.optim.Adam(params, lr=3e-4, betas=(0.9, 0.99), weight_decay=5e-4, amsgrad=True)
    ta.load_state_dict(fa.state_dict())
    st = ta.state[params[10]]
    assert float(st["step"]) == 7 and bool((st["exp_avg"] == 0.25).all()) and bool((st["exp_avg_sq"] == 0.5).all()) \
        and bool((st["max_exp_avg_sq"] == 0.75).all())
    for p in params:
        ta.state[p]["step"] = torch.tensor(11.0)
        ta.state[p]["exp_avg"] = torch.full_like(p, 2.0)
    fa2 = FusedAdam(m, lr=3e-4, amsgrad=True)
    fa2.load_state_dict(ta.state_dict())
    assert fa2._step == 11 and bool((fa2._buffers()[0] == 2.0).all()) and bool((fa2._buffers()[1] == 0.5).all())


def test_part_runs_follow_frozen_parameters():
    """the staged optimizer update (FusedSGD.step_part) must see requires_grad changes made between epochs
    (Engine.two_stepped_transfer_learning / open_specified_layers): the five parts together are always step()'s runs"""
    m = _model(seed=4)

    def flat(runs):
        return sorted(runs)

    def union(parts):
        return flat([r for part in parts for r in part])

    def merged(runs):          # adjacent runs cut at part boundaries are one run for step()
        out = []
        for a, b in flat(runs):
            if out and out[-1][1] == a:
                out[-1] = (out[-1][0], b)
            else:
                out.append((a, b))
        return out
    assert merged(union(m.part_runs())) == merged(m.trainable_runs())
    for name, child in m.named_children():
        for p in child.parameters():
            p.requires_grad = name in ("classifier_R", "fc_T")
    frozen = merged(union(m.part_runs()))
    assert frozen == merged(m.trainable_runs()) and len(frozen) < 10
    assert all(not r for r in m.part_runs()[1:])            # nothing trainable in the trunk parts
    for p in m.parameters():
        p.requires_grad = True
    assert merged(union(m.part_runs())) == merged(m.trainable_runs()) and len(merged(m.trainable_runs())) >= 1


def test_default_arithmetic_is_announced_once_per_construction(monkeypatch):
    """a caller who only swaps the import gets the bf16 speed mode: build_model says so (RuntimeWarning naming
    compute_dtype=torch.float32 as the parity mode) unless the mode was chosen by keyword or by IEEE_COMPUTE_DTYPE"""
    import warnings
    monkeypatch.delenv("IEEE_COMPUTE_DTYPE", raising=False)
    with pytest.warns(RuntimeWarning, match=r"bf16 SPEED mode.*compute_dtype=torch.float32"):
        m = build_model("ieee3modalPart", num_classes=5, loss="margin", pretrained=False, device="cpu")
    assert m.compute_dtype == torch.bfloat16
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert build_model("ieee3modalPart", num_classes=5, loss="margin", pretrained=False, device="cpu",
                           compute_dtype=torch.bfloat16).compute_dtype == torch.bfloat16
        assert build_model("ieee3modalPart", num_classes=5, loss="margin", pretrained=False, device="cpu",
                           compute_dtype=torch.float32).compute_dtype == torch.float32
        monkeypatch.setenv("IEEE_COMPUTE_DTYPE", "bf16")
        assert build_model("ieee3modalPart", num_classes=5, loss="margin", pretrained=False, device="cpu").compute_dtype == torch.bfloat16
        monkeypatch.setenv("IEEE_COMPUTE_DTYPE", "fp32")
        assert build_model("ieee3modalPart", num_classes=5, loss="margin", pretrained=False, device="cpu").compute_dtype == torch.float32
